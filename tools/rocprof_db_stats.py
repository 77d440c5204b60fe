#!/usr/bin/env python3
"""Turn a rocprofv3 (ROCm 7.2, default rocpd sqlite output) results .db into the
per-kernel stats table (`--kernel-trace --stats` summary) as text.
usage: tools/rocprof_db_stats.py gpurun_out/prof_X/X_results.db > profiles/X_kernel_stats.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
print("%-58s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for name, calls, tot, avg, pct in rows:
    if len(name) > 56:
        name = name[:53] + "..."
    print("%-58s %8d %14.1f %12.2f %8.2f" % (name, calls, tot, avg, pct))
