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
e1 = ends[-back]
e0 = ends[-back - 1]
step = rows[e0 + 1:e1 + 1]
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
