"""Median / mean duration per kernel from a rocprofv3 --kernel-trace CSV, leaving out launches shorter than `floor` us
(the guarded PCG launches that return at once pull a plain average down).  usage: trace_medians.py <dir> [floor_us] [n]"""
import csv
import glob
import statistics
import sys
from collections import defaultdict

d, floor = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
dur = defaultdict(list)
for r in csv.DictReader(open(f)):
    dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
rows = []
for k, v in dur.items():
    w = [x for x in v if x >= floor]
    if w:
        rows.append((sum(w), k, len(v), len(w), statistics.median(w), sum(w) / len(w)))
for tot, k, nall, nw, med, mean in sorted(rows, reverse=True)[:n]:
    print(f"{k[:64]:64s} calls {nall:4d} counted {nw:4d} median {med:8.1f} mean {mean:8.1f} us")
