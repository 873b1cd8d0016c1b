"""GPU busy fraction of a rocprofv3 --kernel-trace run: the union of all kernel intervals over the span of the densest window of
the trace (the timed passes of bench.py), and how many kernels run concurrently on average.  usage: python tools/gpu_busy.py <rocpd .db>"""
import sqlite3, sys
rows = sqlite3.connect(sys.argv[1]).cursor().execute("select name, start, end from kernels order by start").fetchall()
rows = [(n, s, e) for n, s, e in rows if not n.startswith("__amd")]
# every stretch without a gap above 2 ms = one stream of passes (warm-up, timed passes, the serial replays of bench.py)
windows = []; i0 = 0; last_end = rows[0][2]
for i, (n, s, e) in enumerate(rows):
    if s - last_end > 2_000_000:
        windows.append((i0, i - 1)); i0 = i
    last_end = max(last_end, e)
windows.append((i0, len(rows) - 1))
for a, b in windows:
  window = rows[a:b + 1]
  if len(window) < 20: continue
  t0, t1 = window[0][1], max(e for _, _, e in window)
  events = sorted([(s, 1) for _, s, e in window] + [(e, -1) for _, s, e in window])
  busy = 0; depth = 0; prev = t0; weighted = 0
  for t, d in events:
    if depth > 0: busy += t - prev; weighted += (t - prev) * depth
    depth += d; prev = t
  print("window %.2f ms, %d kernels; at least one kernel running %.1f %% of it; average number of concurrent kernels while busy %.2f; sum of kernel durations %.2f ms" % (
      (t1 - t0) / 1e6, len(window), 100.0 * busy / (t1 - t0), weighted / max(1, busy), sum(e - s for _, s, e in window) / 1e6))
