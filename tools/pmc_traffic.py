"""HBM bytes per launch of the two roofline kernels from rocprofv3 PMC passes -> profiles/pmc_traffic.json (read by bench.py).

    python tools/pmc_traffic.py <prec> <points> <FETCH_SIZE results.db> <WRITE_SIZE results.db> [out.json]

bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: the counters are in KiB and, on gfx950, FETCH_SIZE reports half of the bytes of wide
coalesced streaming reads (MI355X_MICROARCH.md, HBM section) -- the two counters do not fit one pass, hence two databases.
Values are those of the LAST dispatch of each kernel in the profiled run (tools/pmc_run.py: two eager steps of the bench workload)."""
import collections
import json
import os
import sqlite3
import sys


def last_dispatch(db, counter):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    ix = {n: i for i, n in enumerate(cols)}
    kn = next(k for k in ('kernel_name', 'name') if k in ix)
    cn = next(k for k in ('counter_name', 'pmc_name', 'counter') if k in ix)
    vn = next(k for k in ('value', 'counter_value') if k in ix)
    did = next((k for k in ('dispatch_id', 'id') if k in ix), None)
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in c.execute("select * from counters_collection"):
        if r[ix[cn]] != counter:
            continue
        name = r[ix[kn]].split('(')[0].replace('void ', '').strip()
        per[name][r[ix[did]] if did else 0] += float(r[ix[vn]])
    return {k: v[max(v)] for k, v in per.items()}


def main():
    prec, points, fdb, wdb = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
    out = sys.argv[5] if len(sys.argv) > 5 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'pmc_traffic.json')
    fetch, write = last_dispatch(fdb, 'FETCH_SIZE'), last_dispatch(wdb, 'WRITE_SIZE')
    tab = json.load(open(out)) if os.path.exists(out) else {}
    for kernel in ('dpn_fwd_kernel', 'dpn_fwd_tiles_kernel', 'dpn_features_kernel', 'dpn_wgrad_kernel', 'dpn_bwd_kernel', 'dpn_bwd_tiles_kernel'):
        f = next((v for k, v in fetch.items() if k.startswith(kernel)), None)
        w = next((v for k, v in write.items() if k.startswith(kernel)), None)
        if f is None or w is None:
            continue
        tab['%s|%s|%d' % (kernel, prec, points)] = {
            'bytes': (2.0 * f + w) * 1024.0, 'fetch_size_kib': f, 'write_size_kib': w,
            'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python3 tools/pmc_run.py %s; 2 x FETCH_SIZE + WRITE_SIZE' % prec}
    json.dump(tab, open(out, 'w'), indent=1, sort_keys=True)
    print(json.dumps({k: v['bytes'] for k, v in tab.items()}, indent=1))


if __name__ == '__main__':
    main()
