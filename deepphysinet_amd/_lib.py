"""ctypes binding of libdpn_hip.so (include/dpn_hip.h).  There is no CPU fallback: if the
library is missing or no MI355X is visible, every point-path call raises."""
import ctypes
import os
from ctypes import POINTER, Structure, c_double, c_float, c_int, c_int32, c_int64, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DPN_LIB', os.path.join(_HERE, 'libdpn_hip.so'))    # DPN_LIB: experiment builds only

NETS = 6
PREC_BF16 = 1       # bf16 MFMA operands, fp32 accumulate
PREC_BF16X2 = 2     # bf16 hi+lo split operands (3 MFMAs per product), fp32-class accuracy
PREC_NAMES = {'bf16': PREC_BF16, 'bf16x2': PREC_BF16X2, 1: PREC_BF16, 2: PREC_BF16X2}

_NET_FIELDS = ('w1b1', 'w2b2', 'evec', 'Wd', 'bd', 'W1', 'bf1', 'W2', 'bf2', 'wo', 'bo')


class DpnNetPtrs(Structure):
    _fields_ = [(n, c_void_p) for n in _NET_FIELDS] + [('ld_w1b1', c_int64), ('ld_w2b2', c_int64)]


class DpnNetGradPtrs(Structure):
    _fields_ = [(n, c_void_p) for n in _NET_FIELDS] + [('ld_w1b1', c_int64), ('ld_w2b2', c_int64)]


class DpnGeometry(Structure):
    _fields_ = [('dx', c_float), ('dy', c_float), ('lon_m1', c_float), ('lat_m1', c_float), ('pred_t_span', c_float)]


class DpnPhysics(Structure):
    _fields_ = [('mean', c_float * NETS), ('std', c_float * NETS), ('clip_lo', c_float * NETS), ('clip_hi', c_float * NETS),
                ('clip_on', c_int * NETS), ('factor', c_float * NETS), ('criterion', c_int), ('beta', c_float), ('sq_on', c_int * NETS),
                ('sq_add', c_float * NETS), ('reduce_sum', c_int)]


CRIT_MSE, CRIT_L1, CRIT_SMOOTH_L1 = 0, 1, 2          # DpnPhysics.criterion (include/dpn_hip.h)


class DpnGemmProblem(Structure):
    _fields_ = [('A', c_void_p * 12), ('B', c_void_p * 12), ('lda', c_int32 * 12), ('ldb', c_int32 * 12), ('k_term', c_int32 * 12),
                ('bias', c_void_p), ('C', c_void_p),
                ('asum', c_void_p), ('M', c_int32), ('N', c_int32), ('K', c_int32), ('ldc', c_int32), ('ta', c_int32), ('tb', c_int32),
                ('nterms', c_int32), ('aux', c_void_p), ('aux_out', c_void_p), ('epi', c_int32)]


EPI_NONE, EPI_GELU, EPI_MUL_GELU_GRAD, EPI_ADD = 0, 1, 2, 3


class DpnLnGemm(Structure):
    _fields_ = [(n, c_int32) for n in ('mode', 'M', 'N', 'tb', 'ldb', 'ldc', 'epi')] + \
               [(n, c_void_p) for n in ('x', 'r', 'gamma', 'beta', 'rstd_in', 'y_out', 'xhat_out', 'rstd_out', 'partial', 'B', 'bias', 'C', 'aux', 'aux_out')]


class DpnColsumJob(Structure):
    _fields_ = [('partial', c_void_p), ('out_a', c_void_p), ('out_b', c_void_p), ('n_blocks', c_int32)]


class DpnEncFwd(Structure):
    """include/dpn_hip.h DpnEncFwd: one row-local forward launch of the encoder (csrc/dpn_encoder_chain.hip)."""
    _fields_ = [('wpack', c_void_p)] + [(n, c_int32) for n in ('n_mats', 'rows', 'row_tiles', 'tail', 'next', 'm_o', 'm_c1', 'm_c2', 'm_n0', 'm_n1', 'm_n2')] + \
               [(n, c_void_p) for n in ('o', 'x', 'xin', 'bo', 'g1', 'be1', 'bc1', 'bc2', 'g2', 'be2', 'gf', 'bef', 'bn0', 'bn1', 'bn2',
                                        'x1', 'xhat1', 'rstd1', 'pre', 'act', 'x2', 'xhat2', 'rstd2', 'xf', 'xhatf', 'rstdf', 'y0', 'y1', 'y2',
                                        'emb_parts', 'emb_bias', 'emb_pos', 'emb_te', 'emb_token', 'emb_out')] + \
               [('emb_part_stride', c_int64), ('emb_n_parts', c_int32), ('emb_n_tok', c_int32)]


class DpnEncBwd(Structure):
    _fields_ = [('wpack', c_void_p)] + [(n, c_int32) for n in ('n_mats', 'rows', 'row_tiles', 'head', 'body', 'm_h0', 'm_h1', 'm_h2', 'm_c2', 'm_c1', 'm_o')] + \
               [(n, c_void_p) for n in ('res', 'dq', 'dk', 'dv', 'dmeta', 'xhatf', 'rstdf', 'gin', 'xhat2', 'rstd2', 'pre', 'xhat1', 'rstd1', 'g2', 'g1', 'gf',
                                        'gs2', 'dpre', 'gs1', 'dout', 'gx', 'partial_f', 'partial2', 'partial1', 'gx_head')] + [('gx_head_rows', c_int32)]


class DpnGemm16Problem(Structure):
    _fields_ = [(n, c_void_p) for n in ('A', 'B', 'C', 'asum', 'bias')] + [(n, c_int32) for n in ('M', 'N', 'K', 'ldc')] + \
               [(n, c_int64) for n in ('a_sm', 'a_sk', 'b_sn', 'b_sk')]


class DpnEncPrep(Structure):
    _fields_ = [('n_mats', c_int32), ('weights', c_void_p), ('packed', c_void_p), ('status_dev', c_void_p),
                ('x', c_void_p), ('T', c_int32), ('C', c_int32), ('batch', c_int32), ('xu', c_void_p),
                ('h', c_void_p), ('freqs_a', c_void_p), ('n_a', c_int32), ('out_a', c_void_p), ('freqs_b', c_void_p), ('n_b', c_int32), ('out_b', c_void_p)]


class DpnWgradProblem(Structure):
    _fields_ = [('G', c_void_p), ('X', c_void_p), ('dW', c_void_p), ('db', c_void_p)] + [(n, c_int32) for n in ('M', 'N', 'rows', 'ldg', 'ldx', 'ldw')]


ENC_MAX_MATS = 32
WGRAD_MAX_PROBLEMS = 32
GEMM_MAX_PROBLEMS, GEMM_MAX_JOBS = 26, 10


class DpnSampler(Structure):
    _fields_ = [('lon', c_int32), ('lat', c_int32), ('lon_in', c_int32), ('lat_in', c_int32), ('t_in', c_int32), ('t_hours', c_int32),
                ('cells_x', c_double), ('cells_y', c_double), ('t_step_hours', c_double), ('begin_lat', c_double), ('dlat', c_double),
                ('dx', c_float), ('dy', c_float)]


SAMPLE_INTERIOR, SAMPLE_MARGIN, SAMPLE_EXPLICIT = 0, 1, 2


class DpnSizes(Structure):
    _fields_ = [('n_pad', c_int64), ('packed', c_int64), ('saved', c_int64), ('operands', c_int64), ('partials', c_int64),
                ('k_splits', c_int32)]


EXPORTS = {
    'dpn_version': (c_int, []),
    'dpn_sizes': (c_int, [c_int64, c_int, POINTER(DpnSizes)]),
    'dpn_clock_stamp': (c_int, [c_void_p, c_void_p, c_uint32, c_void_p]),
    'dpn_clock_rate_khz': (c_int, [POINTER(c_int)]),
    'dpn_pack_weights': (c_int, [POINTER(DpnNetPtrs), c_int, c_void_p, c_void_p]),
    'dpn_fwd_form': (c_int, [c_int, c_int]),
    'dpn_pack_weights_form': (c_int, [POINTER(DpnNetPtrs), c_int, c_int, c_void_p, c_void_p]),
    'dpn_pack_weights_batch': (c_int, [POINTER(DpnNetPtrs), c_int, c_int64, c_int64, c_int, c_int, c_void_p, c_int64, c_void_p]),
    'dpn_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, POINTER(DpnGeometry), c_void_p, c_int,
                        c_void_p, c_void_p, c_void_p, c_void_p]),
    'dpn_fwd_ref': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, POINTER(DpnGeometry), c_void_p, c_int,
                            c_void_p, c_void_p, c_void_p, c_void_p]),
    'dpn_fwd_ref_nets': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, POINTER(DpnGeometry), c_void_p, c_int,
                                 c_int, c_void_p, c_void_p, c_void_p]),
    'dpn_contract_gpe': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    'dpn_residual': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, POINTER(DpnGeometry), POINTER(DpnPhysics), c_void_p, c_void_p, c_void_p,
                             c_void_p, c_void_p, c_void_p]),
    'dpn_residual_finish': (c_int, [c_void_p, c_int64, POINTER(DpnPhysics), c_void_p, c_void_p]),
    'dpn_residual_finish_batch': (c_int, [c_void_p, c_int64, c_int, POINTER(DpnPhysics), c_void_p, c_void_p]),
    'dpn_bwd_points': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, POINTER(DpnGeometry), c_void_p, c_int,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'dpn_bwd_points_scaled': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, POINTER(DpnGeometry), c_void_p, c_int,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'dpn_wgrad': (c_int, [c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'dpn_wgrad_finish': (c_int, [POINTER(DpnNetPtrs), c_void_p, c_int64, c_int, c_void_p, POINTER(DpnNetGradPtrs), c_void_p]),
    'dpn_wgrad_finish_parts': (c_int, [POINTER(DpnNetPtrs), c_void_p, c_int64, c_int, c_void_p, POINTER(DpnNetGradPtrs), c_int, c_void_p]),
    'dpn_smooth_l1': (c_int, [c_void_p, c_void_p, c_int64, c_float, c_float, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    'dpn_sgemm': (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int,
                          c_void_p, c_int64, c_void_p]),
    'dpn_sgemm_ln': (c_int, [POINTER(DpnLnGemm), c_void_p]),
    'dpn_sgemm_batch_jobs': (c_int, [c_int, c_void_p, c_int, c_void_p, c_void_p]),
    'dpn_sgemm_batch': (c_int, [c_int, POINTER(DpnGemmProblem), c_void_p]),
    'dpn_attn_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'dpn_attn16_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'dpn_attn_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'dpn_enc_pack_bytes': (c_int64, [c_int]),
    'dpn_enc_pack': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'dpn_wgrad16_partial_floats': (c_int64, [c_int, c_void_p, c_int]),
    'dpn_wgrad16': (c_int, [c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    'dpn_enc_prep': (c_int, [POINTER(DpnEncPrep), c_void_p]),
    'dpn_enc_fwd': (c_int, [POINTER(DpnEncFwd), c_void_p]),
    'dpn_enc_bwd': (c_int, [POINTER(DpnEncBwd), c_void_p]),
    'dpn_add_ln_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'dpn_add_ln_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'dpn_clip_adam_scratch_doubles': (c_int64, [c_int, c_void_p]),
    'dpn_clip_adam': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float,
                              c_float, c_float, c_void_p, c_void_p]),
    'dpn_sum_parts': (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p]),
    'dpn_lead_pe': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    'dpn_im2col_circ3': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'dpn_embed_assemble': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'dpn_sample_points': (c_int, [POINTER(DpnSampler), c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_uint64, c_uint64,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'dpn_sample_points_replay': (c_int, [POINTER(DpnSampler), c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_uint64, c_uint64,
                                         c_void_p, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'dpn_grid_maps': (c_int, [c_void_p, c_int, c_int, POINTER(DpnPhysics), c_int, c_void_p, c_void_p]),
    'dpn_clip_adam_flat_floats': (c_int64, [c_int, c_void_p]),
    'dpn_clip_adam_flat': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float,
                                   c_float, c_float, c_void_p, c_void_p]),
    'dpn_clip_adam_flat_dev': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'dpn_gemm_fp8_mx': (c_int, [c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'dpn_selftest': (c_int, [c_void_p, c_void_p]),
}

# Shelved experiments (include/dpn_hip_experiments.h): compiled only into libdpn_hip_exp.so (`python -m deepphysinet_amd.build --experiments`), which also
# holds every product symbol; reached through load_experiments() by the code paths behind the matching frozen switches (config.py)
EXPERIMENT_EXPORTS = {
    'dpn_gemm16_partial_floats': (c_int64, [c_int, c_void_p, c_int]),
    'dpn_gemm16': (c_int, [c_int, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    'dpn_conv16_kp': (c_int64, [c_int]),
    'dpn_conv16_split': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'dpn_conv16': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'dpn_gemm_fp8': (c_int, [c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
}
EXP_LIB_PATH = os.path.join(_HERE, 'libdpn_hip_exp.so')

_lib = None
_exp = None

# The per-FIELD math (encoder GEMMs, attention, LayerNorm, hyper-network heads, SmoothL1) has plain-torch expressions for host tensors.
# They exist for the CPU-side tests of the module tree / state_dict contract / encoder math against the reference's golden vectors
# (the build container has no GPU); the product never takes them: a host tensor reaching one of these ops raises unless reference math
# has been switched on explicitly (tests do, DPN_CPU_REFERENCE_MATH=1 does).  The per-POINT path has no host form at all.
_cpu_reference_math = [os.environ.get('DPN_CPU_REFERENCE_MATH') == '1']


def enable_cpu_reference_math(on=True):
    _cpu_reference_math[0] = bool(on)


def host_math_or_raise(t, what):
    """Called where an op is about to evaluate its torch expression instead of a HIP kernel: allowed for device tensors of a shape the
    kernels do not cover, and for host tensors only in reference-math mode."""
    if not t.is_cuda and not _cpu_reference_math[0]:
        raise RuntimeError('deepphysinet_amd: %s got a tensor on %s; the HIP kernels are the only product path (no CPU fallback). '
                           'deepphysinet_amd._lib.enable_cpu_reference_math() switches the torch expressions on for CPU-side tests.' % (what, t.device))


def load_experiments():
    """dlopen the experiment library (product symbols + the shelved kernels' entry points); raises if it has not been built."""
    global _exp
    if _exp is not None:
        return _exp
    if not os.path.exists(EXP_LIB_PATH):
        raise RuntimeError('deepphysinet_amd: a shelved experiment was switched on (deepphysinet_amd/config.py) but %s is missing; build it with '
                           '`python -m deepphysinet_amd.build --experiments`' % EXP_LIB_PATH)
    lib = ctypes.CDLL(EXP_LIB_PATH)
    for name, (res, args) in list(EXPORTS.items()) + list(EXPERIMENT_EXPORTS.items()):
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _exp = lib
    return lib


def load():
    """dlopen libdpn_hip.so and declare every prototype; raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('deepphysinet_amd: %s is missing. Build it with `python -m deepphysinet_amd.build` '
                           '(hipcc --offload-arch=gfx950). There is no CPU fallback for the point path.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        raise RuntimeError('libdpn_hip: %s failed with code %d' % (what, code))
