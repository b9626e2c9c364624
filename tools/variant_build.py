"""Experiment builds of the library: only the point unit is recompiled with extra -D flags, the other objects come from the product build.

    python tools/variant_build.py NAME [-DFLAG ...]      ->  deepphysinet_amd/libdpn_hip_NAME.so   (select it with DPN_LIB=<path>)
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepphysinet_amd import build as B


def build(name, extra):
    obj = os.path.join(B.HERE, 'csrc', '_obj')
    # the product library is NOT rebuilt here: an experiment that edits a header must not leak into libdpn_hip.so (it did once: a timing
    # ablation with wrong arithmetic sat in the product library until the next build).  Only the objects of the other units are needed.
    if not all(os.path.exists(os.path.join(obj, u[2])) for u in B.UNITS[1:]):
        B.build_library()
    src, flags, base = B.UNITS[0]
    o = os.path.join(obj, 'var_%s_%s' % (name, base))
    subprocess.run(['hipcc', *B.COMMON, *flags, *extra, '-I' + os.path.join(ROOT, 'include'), '-c', src, '-o', o], check=True)
    lib = os.path.join(B.HERE, 'libdpn_hip_%s.so' % name)
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', o, *[os.path.join(obj, u[2]) for u in B.UNITS[1:]], '-o', lib], check=True)
    return lib


if __name__ == '__main__':
    print(build(sys.argv[1], sys.argv[2:]))
