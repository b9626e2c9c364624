"""Builds libdpn_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the
.so travels to the GPU box with the snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRCS = [os.path.join(HERE, 'csrc', 'dpn_kernels.hip'), os.path.join(HERE, 'csrc', 'dpn_encoder.hip'),
        os.path.join(HERE, 'csrc', 'dpn_sampler.hip'), os.path.join(HERE, 'csrc', 'dpn_fp8.hip'),
        os.path.join(HERE, 'csrc', 'dpn_encoder_chain.hip')]
DEPS = SRCS + [os.path.join(HERE, 'csrc', 'dpn_layout.h'), os.path.join(HERE, 'csrc', 'dpn_fwd_tiles.h'), os.path.join(HERE, 'csrc', 'dpn_fwd_pp.h'), os.path.join(HERE, 'csrc', 'dpn_fwd_tiles_persist.h'), os.path.join(os.path.dirname(HERE), 'include', 'dpn_hip.h'),
               os.path.join(os.path.dirname(HERE), 'include', 'dpn_hip_experiments.h')]
LIB = os.path.join(HERE, 'libdpn_hip.so')


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


# (source, extra flags, object name).  dpn_kernels.hip is two translation units: the point forward / backward kernels are compiled with
# MFMA results in VGPRs (no v_accvgpr_read per epilogue element: forward kernel -10 %), the rest (weight-gradient kernel, GEMMs,
# optimiser) measures better with hipcc's default AGPR accumulators -- see the DPN_TU comment in the source.
UNITS = [(SRCS[0], ['-DDPN_TU=1', '-mllvm', '-amdgpu-mfma-vgpr-form'], 'dpn_point.o'),
         (SRCS[0], ['-DDPN_TU=2'], 'dpn_rest.o'),
         (SRCS[1], [], 'dpn_encoder.o'),
         (SRCS[2], [], 'dpn_sampler.o'),
         (SRCS[3], [], 'dpn_fp8.o'),
         (SRCS[4], ['-mllvm', '-amdgpu-mfma-vgpr-form'], 'dpn_encoder_chain.o')]
COMMON = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC']


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        raise RuntimeError('hipcc not found: libdpn_hip.so cannot be built (no CPU fallback exists)')
    obj_dir = os.path.join(HERE, 'csrc', '_obj')
    os.makedirs(obj_dir, exist_ok=True)
    procs = []
    for src, flags, obj in UNITS:                                   # the units compile side by side
        cmd = [hipcc, *COMMON, *flags, '-I' + os.path.join(os.path.dirname(HERE), 'include'), '-c', src, '-o', os.path.join(obj_dir, obj)]
        if verbose:
            print(' '.join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            if '-mllvm' in cmd:                                     # a toolchain without that (internal) LLVM option: the unit is
                i = cmd.index('-mllvm')                             # correct without it, only ~2 % slower
                retry = cmd[:i] + cmd[i + 2:]
                print('[build] retrying without %s' % ' '.join(cmd[i:i + 2]))
                subprocess.run(retry, check=True)
                continue
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', *[os.path.join(obj_dir, u[2]) for u in UNITS], '-o', LIB + '.tmp']
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(LIB + '.tmp', LIB)
    return LIB


EXP_LIB = os.path.join(HERE, 'libdpn_hip_exp.so')
EXP_UNITS = (4, 5)            # dpn_fp8.hip (the non-scaled fp8 GEMM), dpn_encoder_chain.hip (dpn_conv16*, dpn_gemm16): include/dpn_hip_experiments.h


def build_experiments(force: bool = False) -> str:
    """libdpn_hip_exp.so: the product objects with the units that hold shelved kernels recompiled -DDPN_EXPERIMENTS (their entry points are
    compiled out of the product library).  Tests of those kernels and the tools under tools/ that measure them load it."""
    build_library()
    deps = DEPS + [os.path.join(os.path.dirname(HERE), 'include', 'dpn_hip_experiments.h')]
    if not force and os.path.exists(EXP_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(EXP_LIB) for d in deps):
        return EXP_LIB
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    obj_dir = os.path.join(HERE, 'csrc', '_obj')
    objs, procs = [], []
    for i, (src, flags, obj) in enumerate(UNITS):
        if i in EXP_UNITS:
            o = os.path.join(obj_dir, 'exp_' + obj)
            procs.append(subprocess.Popen([hipcc, *COMMON, *flags, '-DDPN_EXPERIMENTS', '-I' + os.path.join(os.path.dirname(HERE), 'include'), '-c', src, '-o', o]))
            objs.append(o)
        else:
            objs.append(os.path.join(obj_dir, obj))
    for pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, 'hipcc -DDPN_EXPERIMENTS')
    subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', EXP_LIB + '.tmp'], check=True)
    os.replace(EXP_LIB + '.tmp', EXP_LIB)
    return EXP_LIB


if __name__ == '__main__':
    import sys
    print(build_library(force=True, verbose=True))
    if '--experiments' in sys.argv:
        print(build_experiments(force=True))
