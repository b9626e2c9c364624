"""Builds libdpn_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the
.so travels to the GPU box with the snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRCS = [os.path.join(HERE, 'csrc', 'dpn_kernels.hip'), os.path.join(HERE, 'csrc', 'dpn_encoder.hip'),
        os.path.join(HERE, 'csrc', 'dpn_sampler.hip')]
DEPS = SRCS + [os.path.join(HERE, 'csrc', 'dpn_layout.h'), os.path.join(os.path.dirname(HERE), 'include', 'dpn_hip.h')]
LIB = os.path.join(HERE, 'libdpn_hip.so')


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        raise RuntimeError('hipcc not found: libdpn_hip.so cannot be built (no CPU fallback exists)')
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', *SRCS, '-o', LIB + '.tmp']
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(LIB + '.tmp', LIB)
    return LIB


if __name__ == '__main__':
    print(build_library(force=True, verbose=True))
