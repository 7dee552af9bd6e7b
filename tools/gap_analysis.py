#!/usr/bin/env python3
"""Where the device sits idle: gaps between consecutive kernel dispatches of a rocprofv3 kernel trace.

  python tools/gap_analysis.py <stats_dir> [--from KERNEL] [--top N] [--md OUT.md] [--tag TAG]

<stats_dir>: `rocprofv3 --kernel-trace --output-format csv -d <stats_dir> -- python bench.py ...`
Dispatches are ordered by start time; the gap after dispatch i is start[i+1] - end[i] (negative = overlap, counted
as 0).  Gaps are summed per (previous kernel -> next kernel) pair.  With --from, only the part of the trace after the
FIRST dispatch of that kernel name fragment is looked at (e.g. the first product of the timed region), and gaps longer
than 2 ms (the host between solves) are listed apart.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("tmi::", "")
    if "rocprim" in n:
        n = "rocprim::" + n.split("wrapped_")[-1].split("<")[0]
    return n


def main():
    args = sys.argv[1:]
    frm, top, md, tag = None, 25, None, "trace"
    for flag in ("--from", "--top", "--md", "--tag"):
        if flag in args:
            i = args.index(flag)
            val = args[i + 1]
            del args[i:i + 2]
            if flag == "--from":
                frm = val
            elif flag == "--top":
                top = int(val)
            elif flag == "--md":
                md = val
            else:
                tag = val
    d = args[0]
    f = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    if frm:
        for i, r in enumerate(rows):
            if frm in r[2]:
                rows = rows[i:]
                break
    busy = sum(e - s for s, e, _ in rows)
    span = rows[-1][1] - rows[0][0]
    pair = defaultdict(lambda: [0, 0.0])
    long_gaps = []
    idle = 0.0
    for a, b in zip(rows, rows[1:]):
        g = b[0] - a[1]
        if g <= 0:
            continue
        if g > 2e6:
            long_gaps.append((g, a[2], b[2]))
            continue
        idle += g
        p = pair[(a[2], b[2])]
        p[0] += 1
        p[1] += g
    out = []
    out.append(f"# device idle time between dispatches `{tag}`\n")
    out.append(f"{len(rows)} dispatches, span {span / 1e6:.2f} ms, kernels busy {busy / 1e6:.2f} ms, "
               f"idle in gaps <= 2 ms: {idle / 1e6:.2f} ms ({100.0 * idle / max(busy + idle, 1):.1f} % of busy + idle), "
               f"{len(long_gaps)} longer gaps (host between solves: {sum(g for g, _, _ in long_gaps) / 1e6:.1f} ms)\n")
    out.append("| previous kernel -> next kernel | gaps | total us | mean us |")
    out.append("|---|---|---|---|")
    for (a, b), (n, t) in sorted(pair.items(), key=lambda kv: -kv[1][1])[:top]:
        out.append(f"| {a} -> {b} | {n} | {t / 1e3:.1f} | {t / 1e3 / n:.2f} |")
    text = "\n".join(out) + "\n"
    print(text)
    if md:
        open(md, "w").write(text)


if __name__ == "__main__":
    main()
