"""Timeline of ONE step out of a rocprofv3 --kernel-trace results DB: kernels in start order with duration, gap to the previous
end, and overlap; then totals.  usage: timeline.py <results.db> [step_index_from_end=2]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in c.execute('pragma table_info(%s)' % kd)]
rows = c.execute('select k.start, k.end, s.kernel_name from %s k join %s s on k.kernel_id = s.id order by k.start' % (kd, sym)).fetchall()
# a step starts at each dpn_sgemm(_splitk) of the token embedding... simpler: split at dpn_pack_matrices and walk back to the previous adam
idx = [i for i, r in enumerate(rows) if 'dpn_adam_kernel' in r[2]]
# steps end at the LAST adam kernel of a group of consecutive adam launches
ends = [i for j, i in enumerate(idx) if j + 1 == len(idx) or idx[j + 1] - i > 3]
# steps of the PRODUCT graph only: bench.py replays an instrumented copy (dpn_clock_stamp nodes around the point kernels) behind the timed region
steps_ = [(ends[j] + 1, ends[j + 1] + 1) for j in range(len(ends) - 1)]
steps_ = [(a, b) for a, b in steps_ if not any('dpn_clock_stamp' in r[2] for r in rows[a:b])]
# ... and of the commonest kernel count (the eager warm-up steps, the roofline harness and the power soak have other shapes)
from collections import Counter
common = Counter(b - a for a, b in steps_).most_common(1)[0][0]
steps_ = [(a, b) for a, b in steps_ if b - a == common]
# ... and not one that straddles a block boundary of the bench (barrier + synchronize: milliseconds of idle host time inside the "step")
spans = [rows[b - 1][1] - rows[a][0] for a, b in steps_]
steps_ = [ab for ab, sp in zip(steps_, spans) if sp <= 1.1 * min(spans)]
a_, b_ = steps_[-back]
step = rows[a_:b_]
t0 = step[0][0]
prev_end = t0
busy = 0
tot_gap = 0
print('%9s %8s %7s  %s' % ('start_us', 'dur_us', 'gap_us', 'kernel'))
cur_end = t0
for s, e, n in step:
    gap = (s - cur_end) / 1e3
    print('%9.1f %8.2f %7.2f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, n[:70]))
    if s > cur_end:
        tot_gap += s - cur_end
    if e > cur_end:
        busy += e - max(s, cur_end)
        cur_end = e
print('kernels %d  span %.1f us  union-busy %.1f us  idle gaps %.1f us  sum of durations %.1f us' %
      (len(step), (cur_end - t0) / 1e3, busy / 1e3, tot_gap / 1e3, sum(e - s for s, e, _ in step) / 1e3))
# period: start of one step to the start of the next, over the steps that directly follow each other (what a block time divided by its steps measures;
# period - span = the idle time BETWEEN two replays)
pairs = [(rows[a2][0] - rows[a1][0]) / 1e3 for (a1, b1), (a2, b2) in zip(steps_, steps_[1:]) if a2 == b1]
if pairs:
    pairs.sort()
    print('period between consecutive steps: median %.1f us (min %.1f, %d pairs)' % (pairs[len(pairs) // 2], pairs[0], len(pairs)))
