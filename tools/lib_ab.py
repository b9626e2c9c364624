"""Interleaved timing of the forward + Jacobian launch (training shape: saved state + Jacobian) of several library builds on the same box: each build runs in
its own process, `rounds` times in turn.  usage: lib_ab.py rounds lib1 lib2 ...   ("product" = deepphysinet_amd/libdpn_hip.so; extra env as NAME=VALUE:lib)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds = int(sys.argv[1])
res = {}
for r in range(rounds):
    for spec in sys.argv[2:]:
        env = dict(os.environ)
        lib = spec
        while '=' in lib.split(':')[0] and ':' in lib:
            kv, lib = lib.split(':', 1)
            k, v = kv.split('=', 1)
            env[k] = v
        if lib != 'product':
            env['DPN_LIB'] = os.path.join(ROOT, lib)
        else:
            env.pop('DPN_LIB', None)
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'fwd_time.py'), 'bf16x2', '37265', '6', 'tiles', 'full'], env=env, capture_output=True, text=True).stdout
        for ln in out.splitlines():
            if 'median' in ln:
                res.setdefault(spec, []).append((float(ln.split('min')[1].split('us')[0]), float(ln.split('median')[1].split('us')[0])))
for spec, v in res.items():
    print('%-60s min %s   median %s' % (spec, ' '.join('%.1f' % a for a, _ in v), ' '.join('%.1f' % b for _, b in v)))
