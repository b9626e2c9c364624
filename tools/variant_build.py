"""Experiment builds of the library: only ONE unit (default: the point unit) is recompiled with extra -D flags, the other objects come from
the product build.

    python tools/variant_build.py NAME [--unit=I] [-DFLAG ...]      ->  tools/_variants/libdpn_hip_NAME.so   (select it with DPN_LIB=<path>)
    (--unit=I[,J]: indices into deepphysinet_amd.build.UNITS; 5 = the row-local encoder nodes, csrc/dpn_encoder_chain.hip)
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepphysinet_amd import build as B


def build(name, extra, unit=0):
    """unit: an index into deepphysinet_amd.build.UNITS or a list of them (every listed unit is recompiled with the extra flags)"""
    units = [unit] if isinstance(unit, int) else list(unit)
    obj = os.path.join(B.HERE, 'csrc', '_obj')
    # the product library is NOT rebuilt here: an experiment that edits a header must not leak into libdpn_hip.so (it did once: a timing
    # ablation with wrong arithmetic sat in the product library until the next build).  Only the objects of the other units are needed.
    others = [u for i, u in enumerate(B.UNITS) if i not in units]
    if not all(os.path.exists(os.path.join(obj, u[2])) for u in others):
        B.build_library()
    objs = []
    for k in units:
        src, flags, base = B.UNITS[k]
        o = os.path.join(obj, 'var_%s_%s' % (name, base))
        subprocess.run(['hipcc', *B.COMMON, *flags, *extra, '-I' + os.path.join(ROOT, 'include'), '-c', src, '-o', o], check=True)
        objs.append(o)
    os.makedirs(os.path.join(ROOT, 'tools', '_variants'), exist_ok=True)          # experiment libraries stay OUT of the package directory
    lib = os.path.join(ROOT, 'tools', '_variants', 'libdpn_hip_%s.so' % name)
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', *objs, *[os.path.join(obj, u[2]) for u in others], '-o', lib], check=True)
    return lib


if __name__ == '__main__':
    args = sys.argv[2:]
    unit = 0
    for a in list(args):
        if a.startswith('--unit='):
            unit = [int(v) for v in a.split('=')[1].split(',')]
            unit = unit[0] if len(unit) == 1 else unit
            args.remove(a)
    print(build(sys.argv[1], args, unit))
