#!/usr/bin/env python3
"""print a rocprofv3 *_kernel_stats.csv as a short table:  python tools/kstats.py gpurun_out/prof_x/x_kernel_stats.csv [rows]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print(f"{r['Name'][:48]:48s} calls {r['Calls']:>6s}  total {float(r['TotalDurationNs']) / 1e6:9.2f} ms  avg {float(r['AverageNs']) / 1e3:9.1f} us  {float(r['Percentage']):5.1f} %")
