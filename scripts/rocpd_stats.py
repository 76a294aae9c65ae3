#!/usr/bin/env python3
"""Per-kernel statistics of a rocprofv3 run stored in its default rocpd (SQLite) output, as the CSV `--stats` used to print:

    rocprofv3 --kernel-trace --stats -d out -o run -- python3 bench.py …        (writes out/run_results.db)
    python scripts/rocpd_stats.py out/run_results.db > profiles/rNN_kernel_stats_….csv

Columns: Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs."""
import csv
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    rows = con.execute('select name, count(*), sum("end" - start), avg("end" - start), min("end" - start), max("end" - start) '
                       "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, calls, tot, avg, lo, hi in rows:
        w.writerow([name, calls, tot, round(avg, 1), round(100.0 * tot / total, 3), lo, hi])


if __name__ == "__main__":
    main()
