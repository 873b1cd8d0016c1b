#!/usr/bin/env python3
"""Concurrency of a rocprofv3 kernel trace (rocpd SQLite): how the batch lanes' kernels overlap in the LAST `--window-ms` of the trace's kernel activity
(bench.py's timed region ends with the run).  Prints the union busy time, the time at each level of concurrency, per-stream chains and gaps.

  python tools/concurrency.py results.db [--tail-kernels N] [--after-first k_accumulate_home]
"""
import sqlite3
import sys


def main(path, tail, after):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
    stream_col = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
    q = "select name, start, end%s from kernels order by start" % (", " + stream_col if stream_col else "")
    rows = db.execute(q).fetchall()
    rows = [r for r in rows if not r[0].startswith("__amd_rocclr_fill")]
    if tail:
        rows = rows[-tail:]
    if after:
        ends = [r[2] for r in rows if after in r[0]]
        if ends:
            rows = [r for r in rows if r[1] >= ends[0]]
    t0 = rows[0][1]; t1 = max(r[2] for r in rows)
    ev = []
    for r in rows:
        ev.append((r[1], 1, r[0])); ev.append((r[2], -1, r[0]))
    ev.sort()
    level = 0; last = t0; hist = {}
    wide = 0; lastw = t0; histw = {}
    for t, d, name in ev:
        hist[level] = hist.get(level, 0) + (t - last); last = t
        histw[wide] = histw.get(wide, 0) + (t - lastw); lastw = t
        level += d
        if "k_trace_wide" in name: wide += d
    span = t1 - t0
    print("# %s: %d kernels, span %.2f ms" % (path, len(rows), span / 1e6))
    print("kernels in flight -> share of the span:", {k: round(v / span, 3) for k, v in sorted(hist.items())})
    print("k_trace_wide in flight -> share of the span:", {k: round(v / span, 3) for k, v in sorted(histw.items())})
    tot = {}
    for r in rows:
        n = r[0].split("(")[0].replace("void ", "")[:32]
        tot[n] = tot.get(n, 0) + (r[2] - r[1])
    print("sum of durations (ms):", {k: round(v / 1e6, 2) for k, v in sorted(tot.items(), key=lambda x: -x[1])}, "total", round(sum(tot.values()) / 1e6, 2))
    if stream_col:
        streams = {}
        for r in rows:
            streams.setdefault(r[3], []).append(r)
        for s, rs in streams.items():
            busy = sum(r[2] - r[1] for r in rs); gaps = sum(max(0, rs[i + 1][1] - rs[i][2]) for i in range(len(rs) - 1))
            print("stream %s: %d kernels, first %.2f ms, last end %.2f ms, busy %.2f ms, gaps %.2f ms" %
                  (s, len(rs), (rs[0][1] - t0) / 1e6, (rs[-1][2] - t0) / 1e6, busy / 1e6, gaps / 1e6))


if __name__ == "__main__":
    tail = int(sys.argv[sys.argv.index("--tail-kernels") + 1]) if "--tail-kernels" in sys.argv else 0
    after = sys.argv[sys.argv.index("--after-first") + 1] if "--after-first" in sys.argv else ""
    main(sys.argv[1], tail, after)
