#!/usr/bin/env python
"""Turn a rocprofv3 ``*_kernel_stats.csv`` into the markdown table committed under profiles/.

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o apex -- python bench.py ...
    python scripts/prof_summary.py gpurun_out/prof/apex_kernel_stats.csv "title" > profiles/<name>.md
"""
import csv
import sys


def main():
    path = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else path
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    rows = list(csv.DictReader(open(path)))
    print("# %s\n" % title)
    hdr = "| kernel | calls | total us | avg us | % |"
    if steps:
        hdr = "| kernel | calls | total us | avg us | us/step | % |"
    print(hdr)
    print("|" + "---|" * (hdr.count("|") - 1))
    for r in rows:
        name = r["Name"][:120].replace("|", "/")
        calls = int(r["Calls"])
        tot = float(r["TotalDurationNs"]) / 1e3
        avg = float(r["AverageNs"]) / 1e3
        pct = r["Percentage"]
        if steps:
            print("| `%s` | %d | %.1f | %.2f | %.2f | %s |" % (name, calls, tot, avg, tot / steps, pct))
        else:
            print("| `%s` | %d | %.1f | %.2f | %s |" % (name, calls, tot, avg, pct))


if __name__ == "__main__":
    main()
