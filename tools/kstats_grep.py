#!/usr/bin/env python3
"""Rows of a rocprofv3 kernel_stats.csv whose kernel name contains a pattern: calls, average / min / max duration.
usage: kstats_grep.py <kernel_stats.csv> <pattern>"""
import csv, sys
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    if pat in r["Name"]:
        print("   ", r["Name"].replace("(anonymous namespace)::", "")[:34].ljust(34), r["Calls"].rjust(5), "avg", str(round(float(r["AverageNs"]))).rjust(7),
              "min", r["MinNs"].rjust(6), "max", r["MaxNs"].rjust(7), "ns")
