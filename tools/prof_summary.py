#!/usr/bin/env python3
"""Dump the per-kernel summary of a rocprofv3 --kernel-trace --stats run (rocpd sqlite DB) as CSV.
usage: tools/prof_summary.py gpurun_out/prof_x/fe_results.db > profiles/rNN_name.csv"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
print("kernel,calls,total_us,avg_us,percent")
for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    short = name.split("(")[0].replace("void ", "")
    print(f"{short},{calls},{total:.2f},{avg:.3f},{pct:.2f}")
