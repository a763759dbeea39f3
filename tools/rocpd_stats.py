#!/usr/bin/env python3
"""Dump the per-kernel summary (name, calls, total us, average us, %) of a rocprofv3 rocpd .db file."""
import glob
import sqlite3
import sys

path = sys.argv[1]
dbs = glob.glob(path + "/**/*.db", recursive=True) if not path.endswith(".db") else [path]
for db in dbs:
    c = sqlite3.connect(db)
    print(f"# {db}")
    print(f"{'calls':>7} {'total_us':>14} {'avg_us':>12} {'pct':>7}  kernel")
    for name, calls, tot, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print(f"{calls:7d} {tot:14.1f} {avg:12.1f} {pct:7.2f}  {name}")
