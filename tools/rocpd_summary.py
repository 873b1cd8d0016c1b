#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (ROCm 7.x default output): per-kernel call count, total / average /
min / max duration, share of GPU time, and (when the run collected PMC counters) per-kernel counter sums.

  python tools/rocpd_summary.py gpurun_out/prof/stats/r01_results.db > profiles/r01_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
                       "max(accum_vgpr_count), max(sgpr_count), max(scratch_size), max(lds_size), max(grid_x), max(workgroup_x) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("# %s" % path)
    print("%-44s %7s %12s %11s %11s %11s %6s  %s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "%", "vgpr/agpr/sgpr scratch lds grid block"))
    for r in rows:
        name = r[0].split("(")[0][:44]
        print("%-44s %7d %12.3f %11.2f %11.2f %11.2f %6.2f  %s/%s/%s %s %s %s %s" %
              (name, r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total, r[6], r[7], r[8], r[9], r[10], r[11], r[12]))
    try:
        pmc = cur.execute("select k.name, p.counter_name, count(*), sum(p.value) from pmc_events p join kernels k on p.event_id = k.id "
                          "group by k.name, p.counter_name order by k.name").fetchall()
    except Exception:
        try:
            pmc = cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
        except Exception as e:   # no counters in this run
            pmc = []
    if pmc:
        print("\n%-44s %-16s %7s %18s %18s" % ("kernel", "counter", "calls", "sum", "per_call"))
        for name, counter, n, s in pmc:
            print("%-44s %-16s %7d %18.1f %18.1f" % (name.split("(")[0][:44], counter, n, s, s / max(1, n)))


def timeline(path):
    """Every kernel launch in start order: offset from the first launch, duration, gap to the previous kernel's end."""
    db = sqlite3.connect(path)
    rows = db.cursor().execute("select name, start, end from kernels order by start").fetchall()
    if not rows:
        return
    t0, prev_end = rows[0][1], rows[0][1]
    print("# %s\n%-40s %12s %12s %10s" % (path, "kernel", "start_ms", "duration_us", "gap_us"))
    for name, start, end in rows:
        print("%-40s %12.3f %12.1f %10.1f" % (name.split("(")[0][:40], (start - t0) / 1e6, (end - start) / 1e3, (start - prev_end) / 1e3))
        prev_end = end


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--timeline":
        timeline(sys.argv[2])
    else:
        main(sys.argv[1])
