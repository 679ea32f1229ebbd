#!/usr/bin/env python
"""Summarise a rocprofv3 run (rocpd sqlite .db written by `rocprofv3 --kernel-trace --stats`)
as a small CSV: kernel, calls, total_us, avg_us, pct.  Usage: rocprof_summary.py run.db [out.csv]"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    if len(name) > 140:
        name = name[:137] + "..."
    return name


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    out = open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout
    w = csv.writer(out)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "pct"])
    for name, calls, tot, avg, pct in rows:
        w.writerow([short(name), calls, f"{tot / 1.0:.1f}", f"{avg:.2f}", f"{pct:.2f}"])


if __name__ == "__main__":
    main()
