"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/dpn_hip.h declares; the layout
algebra the kernels rely on (dpn_layout.h) is bijective.  No compute calls (no GPU here)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_build_and_load():
    import __graft_entry__ as g
    g.build()
    from deepphysinet_amd import _lib
    assert os.path.exists(_lib.LIB_PATH)
    assert _lib.load().dpn_version() >= 1


def test_every_declared_symbol_is_exported():
    from deepphysinet_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, 'include', 'dpn_hip.h')).read()
    declared = set(re.findall(r'^\s*(?:int|int64_t)\s+(dpn_\w+)\s*\(', header, flags=re.M))
    assert len(declared) >= 12
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    # shelved experiments: declared in their own header, bound by their own table, and compiled OUT of the product library
    exp_header = open(os.path.join(ROOT, 'include', 'dpn_hip_experiments.h')).read()
    exp_declared = set(re.findall(r'^\s*(?:int|int64_t)\s+(dpn_\w+)\s*\(', exp_header, flags=re.M))
    assert exp_declared == set(_lib.EXPERIMENT_EXPORTS), exp_declared ^ set(_lib.EXPERIMENT_EXPORTS)
    for name in exp_declared:
        assert not hasattr(lib, name), 'the product library exports the shelved experiment %s' % name
    from deepphysinet_amd.build import build_experiments
    build_experiments()
    exp = _lib.load_experiments()
    for name in exp_declared | declared:
        assert hasattr(exp, name), name


def test_sizes_host_function():
    from deepphysinet_amd import _lib
    lib = _lib.load()
    sz = _lib.DpnSizes()
    assert lib.dpn_sizes(37265, 1, ctypes.byref(sz)) == 0
    assert sz.n_pad == 37376 and sz.n_pad % 128 == 0
    blocks = 6 * (800 * 1024 + 6 * 1024 + 16)                   # six nets' fragment blocks + vectors (either packed form, csrc/dpn_layout.h)
    assert sz.packed == (blocks + 255) // 256 * 256
    assert lib.dpn_fwd_form(2, 0) == 1 and lib.dpn_fwd_form(2, 1) == 0 and lib.dpn_fwd_form(1, 0) == 0
    sz2 = _lib.DpnSizes()
    assert lib.dpn_sizes(37265, 2, ctypes.byref(sz2)) == 0
    assert sz2.saved > sz.saved and sz2.operands > sz.operands
    # point ranges of the weight-gradient kernel: a per-product plan that fills one round of the chip at full size (42 workgroups per
    # net over the three products M2^T Z1, M2^T G6, T1^T Z0: 13,13,16 ranges in single bf16, 13,14,15 in the hi+lo mode), one range per 16
    # tiles below that; k_splits (the most ranges one product is cut into) dimensions the partial-sum buffer
    assert sz.k_splits == 16 and sz2.k_splits == 15
    small = _lib.DpnSizes()
    assert lib.dpn_sizes(1037, 2, ctypes.byref(small)) == 0 and small.k_splits == 2
    assert lib.dpn_sizes(256, 1, ctypes.byref(small)) == 0 and small.k_splits == 1
    assert lib.dpn_sizes(0, 1, ctypes.byref(sz)) != 0          # bad arguments are rejected, not ignored
    assert lib.dpn_sizes(16, 3, ctypes.byref(sz)) != 0


def test_struct_layouts_match_header():
    from deepphysinet_amd import _lib
    assert ctypes.sizeof(_lib.DpnNetPtrs) == 13 * 8
    assert ctypes.sizeof(_lib.DpnNetGradPtrs) == 13 * 8
    assert ctypes.sizeof(_lib.DpnGeometry) == 5 * 4
    assert ctypes.sizeof(_lib.DpnPhysics) == 6 * 4 * 6 + 2 * 4 + 2 * 6 * 4 + 4      # + criterion, beta, sq_on, sq_add, reduce_sum (round 5)
    assert ctypes.sizeof(_lib.DpnSampler) == 6 * 4 + 5 * 8 + 2 * 4


def test_ctypes_structs_have_the_c_compilers_layout(tmp_path):
    """Every struct of include/dpn_hip.h the binding mirrors: size and the offset of every field as gcc lays them out (the round-4 entry
    points take structs of ~30 pointers and a dozen ints: one missing pad would shift every pointer behind it)."""
    import subprocess
    from deepphysinet_amd import _lib
    names = ['DpnNetPtrs', 'DpnNetGradPtrs', 'DpnGeometry', 'DpnPhysics', 'DpnSizes', 'DpnGemmProblem', 'DpnColsumJob', 'DpnLnGemm', 'DpnSampler',
             'DpnEncPrep', 'DpnEncFwd', 'DpnEncBwd', 'DpnWgradProblem', 'DpnGemm16Problem']
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "dpn_hip_experiments.h"', 'int main(void) {']      # (includes dpn_hip.h; DpnGemm16Problem lives there)
    want = {}
    for n in names:
        st = getattr(_lib, n)
        lines.append('  printf("%s %%zu\\n", sizeof(%s));' % (n, n))
        want[n] = ctypes.sizeof(st)
        for f in st._fields_:
            fname = {'n_blocks': 'n_blocks'}.get(f[0], f[0])
            lines.append('  printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (n, f[0], n, fname))
            want['%s.%s' % (n, f[0])] = getattr(st, f[0]).offset
    lines += ['  return 0;', '}']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(['gcc', '-I' + os.path.join(root, 'include'), str(src), '-o', str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    got = {ln.split()[0]: int(ln.split()[1]) for ln in out.splitlines()}
    assert got == want, {k: (got.get(k), want.get(k)) for k in set(got) | set(want) if got.get(k) != want.get(k)}


LAYOUT_TEST = r'''
#include <cstdio>
#include <set>
#include "dpn_layout.h"
using namespace dpn;
int slot_of_ch(int ch) { return (ch & ~15) + 8 * ((ch >> 2) & 1) + 4 * ((ch >> 3) & 1) + (ch & 3); }
int main() {
    // chained k-slots cover every channel once, and agree with the accumulator rows they are built from
    std::set<int> seen;
    for (int ks = 0; ks < 16; ++ks) for (int h = 0; h < 2; ++h) for (int e = 0; e < 8; ++e) {
        const int ch = chain_ch(ks, h, e);
        if (!seen.insert(ch).second) return 1;
        const int T = ks / 2, r = 8 * (ks & 1) + e;
        if (ch != 32 * T + drow32(r, h)) return 2;
        if (slot_of_ch(ch) != 16 * ks + 8 * h + e) return 3;
    }
    if (seen.size() != 256) return 4;
    // coordinate / data PE slots are permutations of the reference's 192 channels
    std::set<int> p3, p6, g;
    for (int ks = 0; ks < 12; ++ks) for (int h = 0; h < 2; ++h) for (int e = 0; e < 8; ++e) {
        p3.insert(pe3_ch(ks, h, e)); p6.insert(pe6_ch(ks, h, e));
        // sin/cos of one angle sit in adjacent slots; the coordinate index is constant over a k-step quad
        if ((e & 1) == 0 && pe3_ch(ks, h, e + 1) != pe3_ch(ks, h, e) + 3) return 5;
        if (pe3_ch(ks, h, e) % 3 != ks / 4) return 6;
        if (pe6_ch(ks, h, e) % 6 != ks / 2) return 7;
    }
    if (p3.size() != 192 || p6.size() != 192 || *p3.rbegin() != 191 || *p6.rbegin() != 191) return 8;
    // the Jacobian GEMM's row order hands every lane the cotangent of its own q-th feature
    for (int rho = 0; rho < 192; ++rho) g.insert(gpe_row_to_pe3_ch(rho));
    if (g.size() != 192) return 9;
    for (int T = 0; T < 6; ++T) for (int h = 0; h < 2; ++h) for (int r = 0; r < 16; ++r) {
        const int rho = 32 * T + drow32(r, h);
        const int q = 16 * T + r;
        if (gpe_row_to_pe3_ch(rho) != pe3_ch(q / 8, h, q % 8)) return 10;
    }
    if (kPackKB != 800) return 11;
    std::puts("layout ok");
    return 0;
}
'''


def test_layout_algebra(tmp_path):
    src = tmp_path / 'layout_test.cpp'
    src.write_text(LAYOUT_TEST)
    exe = tmp_path / 'layout_test'
    subprocess.run(['g++', '-std=c++17', '-O1', '-I', os.path.join(ROOT, 'deepphysinet_amd', 'csrc'), str(src), '-o', str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, 'layout check %d failed' % out.returncode
    assert 'layout ok' in out.stdout


def test_point_path_refuses_cpu_tensors():
    import torch
    from deepphysinet_amd.configs import ncep_config
    from deepphysinet_amd.interface import builder_models
    m = builder_models(**ncep_config())
    n = 8
    z = torch.zeros(n, 1)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m.place_one_batch(z, z, z, z, torch.zeros(1, 159, 2405), torch.zeros(n, 6), torch.zeros(1, 1, 1), torch.nn.MSELoss(),
                          m.train_cfg['losses']['loss_factor'], 0, 0, 'cpu')


def test_inline_asm_lds_reads_are_waited_for_before_any_use(tmp_path):
    """The point kernels read LDS with inline-asm ds_read_b128 + counted lgkmcnt waits the compiler knows nothing about:
    scan the generated gfx950 assembly for any instruction that touches a destination register before its wait
    (tools/lds_hazard_check.py).  Build-time check, no GPU needed."""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('hipcc not available')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from deepphysinet_amd.build import COMMON, UNITS
    src = os.path.join(root, 'deepphysinet_amd', 'csrc', 'dpn_kernels.hip')
    checked = 0
    for (usrc, flags, _), kernels in zip(UNITS[:2], (['dpn_fwd_kernel', 'dpn_bwd_kernel'], ['dpn_wgrad_kernel'])):   # as the library builds them
        assert os.path.samefile(usrc, src)
        asm = str(tmp_path / ('unit%d.s' % checked))
        subprocess.run([hipcc, *[f for f in COMMON if f != '-fPIC'], *flags, '--cuda-device-only', '-S', '-I' + os.path.join(root, 'include'),
                        src, '-o', asm], check=True, capture_output=True)
        r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'lds_hazard_check.py'), asm, *kernels], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:]
        assert r.stdout.count(' 0 hazards') == 2 * len(kernels), r.stdout
        # ... and no inline-asm instruction is the first toucher of a fresh MFMA result (the hazard recogniser cannot see inside inline asm)
        r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'mfma_hazard_check.py'), asm, *kernels], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:]
        assert r.stdout.count(' 0 of them touch') == 2 * len(kernels), r.stdout
        text = open(asm).read()
        assert 's_swappc_b64' not in text, 'a helper was not inlined: the kernels must not make calls'
        if checked == 0:
            _check_tile_split_schedule(text)
        checked += 1


def _check_tile_split_schedule(text):
    """The tile-split kernels' k-step loops are unrolled straight-line code whose shape hipcc decides: no spilled vector registers (a spill in
    the first layer's loop once cost 150 us: every fragment load followed by s_waitcnt vmcnt(0) + scratch store) and weight loads
    interleaved with the MFMAs (never more than two k-steps' worth -- 2 k-steps x 2 tiles x 2 planes -- of buffer loads in a row)."""
    import re
    for name in ('_Z20dpn_fwd_tiles_kernelILi2EEv7FwdArgs', '_Z20dpn_bwd_tiles_kernelILi2EEv7BwdArgs'):
        meta = text[text.index('.name:           ' + name):]
        spills = int(re.search(r'\.vgpr_spill_count:\s+(\d+)', meta).group(1))
        assert spills == 0, '%s spills %d vector registers' % (name, spills)
        body = text[text.index(name + ':'):]
        body = body[:body.index('s_endpgm')]
        run = worst = 0
        for line in body.splitlines():
            ins = line.strip().split(' ')[0]
            if ins.startswith('buffer_load_dwordx4'):
                run += 1
                worst = max(worst, run)
            elif ins.startswith('v_mfma'):
                run = 0
        assert worst <= 12, '%s: %d weight-fragment loads in a row without an MFMA between them' % (name, worst)


def test_mfma_hazard_checker_flags_inline_asm_on_a_fresh_accumulator(tmp_path):
    """The second checker: an inline-asm instruction naming a register an MFMA wrote fewer than 12 wait states earlier is reported
    (the inline-asm v_max_f32 ReLU of round 1 read accumulators before the matrix core had written them whenever the scheduler put it
    first); the same instruction behind an s_nop 11, or a compiler-visible instruction in its place, passes."""
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'mfma_hazard_check.py')
    body = ('_Z3badv:\n\tv_mfma_f32_32x32x16_bf16 v[2:17], v[34:37], v[182:185], v[2:17]\n%s'
            '\t;;#ASMSTART\n\tv_max_f32 v0, 0, v2\n\t;;#ASMEND\n\ts_endpgm\n.Lfunc_end0:\n')
    for filler, want in (('', 1), ('\ts_nop 11\n', 0), ('\ts_nop 3\n\tv_mov_b32_e32 v40, v41\n', 1)):
        f = tmp_path / 'k.s'
        f.write_text(body % filler)
        r = subprocess.run([sys.executable, tool, str(f), 'bad'], capture_output=True, text=True)
        assert r.returncode == want and (' %d of them touch' % want) in r.stdout, r.stdout
    f.write_text('_Z3badv:\n\tv_mfma_f32_32x32x16_bf16 v[2:17], v[34:37], v[182:185], v[2:17]\n\tv_max_f32 v0, 0, v2\n\ts_endpgm\n.Lfunc_end0:\n')
    r = subprocess.run([sys.executable, tool, str(f), 'bad'], capture_output=True, text=True)
    assert r.returncode == 0                   # a visible instruction is the compiler's to protect


def test_hazard_checker_flags_a_read_before_its_wait(tmp_path):
    """The checker itself: a register touched between its ds_read and the covering lgkmcnt wait is reported, in-order retirement
    (lgkmcnt(N) leaves the N youngest outstanding) is modelled, and a clean stream passes."""
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'lds_hazard_check.py')
    bad = tmp_path / 'bad.s'
    # (the compiler's own LDS operations -- the ds_bpermute outside the asm markers -- take a counter slot but are not tracked by register)
    bad.write_text('_Z3badv:\n\t;;#ASMSTART\n\tds_read_b128 v[0:3], v9\n\tds_read_b128 v[4:7], v9 offset:16\n\t;;#ASMEND\n'
                   '\tds_bpermute_b32 v12, v10, v11\n\tv_mov_b32_e32 v12, 0\n\ts_waitcnt lgkmcnt(2)\n'
                   '\tv_add_f32_e32 v8, v0, v1\n\tv_accvgpr_write_b32 a0, v5\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32_e32 v8, v6\n\ts_endpgm\n.Lfunc_end0:\n')
    r = subprocess.run([sys.executable, tool, str(bad), 'bad'], capture_output=True, text=True)
    assert r.returncode == 1 and ' 1 hazards' in r.stdout and 'v_accvgpr_write_b32 a0, v5' in r.stdout, r.stdout
    good = tmp_path / 'good.s'
    good.write_text(bad.read_text().replace('\tv_accvgpr_write_b32 a0, v5\n', ''))
    r = subprocess.run([sys.executable, tool, str(good), 'bad'], capture_output=True, text=True)
    assert r.returncode == 0 and ' 0 hazards' in r.stdout, r.stdout
