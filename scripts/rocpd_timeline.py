"""GPU busy-union / per-queue occupancy from a rocprofv3 rocpd kernel trace."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = rows[int(len(rows) * 0.3):]  # drop warm-up
t0, t1 = rows[0][1], max(r[2] for r in rows)
span = t1 - t0
busy, cur_s, cur_e = 0, None, None
for _, s, e, *_ in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
ksum = sum(e - s for _, s, e, *_ in rows)
print('dispatches %d  span %.2f ms  busy-union %.2f ms (%.0f%%)  kernel-sum %.2f ms  overlap factor %.2f' % (
    len(rows), span / 1e6, busy / 1e6, 100 * busy / span, ksum / 1e6, ksum / busy))
perq = collections.defaultdict(lambda: [0, 0])
for _, s, e, q, st in rows:
    perq[(q, st)][0] += 1; perq[(q, st)][1] += e - s
for k, (n, t) in perq.items():
    print('  queue/stream', k, 'dispatches', n, 'busy %.2f ms (%.0f%% of span)' % (t / 1e6, 100 * t / span))
# gap histogram within the union
gaps = []
cur_e = None
for _, s, e, *_ in rows:
    if cur_e is not None and s > cur_e: gaps.append(s - cur_e)
    cur_e = e if cur_e is None else max(cur_e, e)
gaps.sort()
if gaps:
    print('idle gaps: n=%d total %.2f ms  median %.1f us  p90 %.1f us  max %.1f us' % (len(gaps), sum(gaps) / 1e6, gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * .9)] / 1e3, gaps[-1] / 1e3))
