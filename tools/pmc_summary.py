"""Per-kernel PMC values from a rocprofv3 results DB (last dispatch of every dpn_* kernel).  usage: pmc_summary.py <db>"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
rows = c.execute("select * from counters_collection").fetchall()
ix = {n: i for i, n in enumerate(cols)}
kn = next(k for k in ('kernel_name', 'name') if k in ix)
cn = next(k for k in ('counter_name', 'pmc_name', 'counter') if k in ix)
vn = next(k for k in ('value', 'counter_value') if k in ix)
did = next((k for k in ('dispatch_id', 'id') if k in ix), None)
agg = collections.OrderedDict()
for r in rows:
    name = r[ix[kn]]
    if 'dpn_' not in name:
        continue
    key = (name.split('(')[0][:48], r[ix[cn]])
    agg.setdefault(key, []).append((r[ix[did]] if did else 0, float(r[ix[vn]])))
print('%-50s %-28s %10s %16s' % ('kernel', 'counter', 'dispatches', 'value/dispatch(last)'))
for (k, cnt), vals in agg.items():
    per = collections.defaultdict(float)
    for d, v in vals:
        per[d] += v
    last = per[max(per)]
    print('%-50s %-28s %10d %16.1f' % (k, cnt, len(per), last))
