#!/usr/bin/env python
"""Summarise a rocprofv3 results database (rocpd sqlite, `rocprofv3 --kernel-trace --stats -d DIR -o NAME`)
into the plain-text per-kernel table that gets committed under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof/x_results.db > profiles/r01_kernel_stats.txt
"""
import sqlite3
import sys


def main(path, note=""):
    c = sqlite3.connect(path)
    rows = list(c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc"))
    tot = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    if note:
        print(f"# {note}")
    print(f"# durations in microseconds (rocpd stores ns); {sum(r[1] for r in rows)} dispatches, {tot / 1e3:.1f} us of kernel time")
    print("%-64s %8s %12s %10s %10s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for name, calls, total, avg, mn, mx in rows:
        short = name.replace("(anonymous namespace)::", "").split("(")[0][-64:]
        print("%-64s %8d %12.1f %10.2f %10.2f %10.2f %7.2f" % (short, calls, total / 1e3, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * total / tot))


if __name__ == "__main__":
    main(sys.argv[1], " ".join(sys.argv[2:]))
