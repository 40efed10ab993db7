#!/usr/bin/env python
"""Print the kernel timeline of ONE graph-replayed learner step from a rocprofv3 kernel trace:
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o apex -- python bench.py --steps 100 --warmup 10 --no-extra
   python scripts/timeline.py gpurun_out/tl/apex_kernel_trace.csv [marker-kernel-prefix]
Columns: start us, end us, duration us, kernel, queue. Shows which branches really overlap and where the gaps are."""
import csv
import sys


def main():
    path = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "k_sample"
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    a, b = idx[-3], idx[-2]
    t0 = int(rows[a]["Start_Timestamp"])
    tot = 0
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        tot += e - s
        print("%8.1f %8.1f %6.1f  %-72s q=%s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r["Kernel_Name"][:72], r.get("Queue_Id", "")))
    print("sum of kernel durations %.1f us, wall %.1f us" % (tot / 1e3, (int(rows[b]["Start_Timestamp"]) - t0) / 1e3))


if __name__ == "__main__":
    main()
