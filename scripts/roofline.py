#!/usr/bin/env python
"""Per-kernel roofline table of a learner step from the files scripts/roofline.sh wrote (VERDICT r1 #3 / Weak #4).

  time        un-profiled average duration of the kernel (rocprofv3 --kernel-trace --stats)
  TFLOP/s     EXECUTED matrix FLOPs / time: SQ_INSTS_VALU_MFMA_MOPS_{BF16,F16} x 512 FLOP (one 16x16x32 MFMA = 32 MOPS =
              16 384 FLOP), i.e. including tile padding - the useful share is the analytic FLOP count of the layer
  MFMA %      SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8): the share of all SIMD-cycles of the
              kernel's own duration in which a matrix pipe was busy. (GRBM_GUI_ACTIVE is summed over the 8 XCDs; the busy
              counter advances 16 cycles per 16x16x32 MFMA. The r1 table divided by wave-resident cycles instead, which
              is neither a chip-level utilisation nor comparable between kernels.)
  % of peak   TFLOP/s / 2 500 (dense bf16 peak of an MI355X; the clock the chip holds under MFMA load is lower than the
              2.4 GHz this assumes, so ~45 % is what hipBLASLt reaches on its best shapes here)
  HBM GB/s    (FETCH_SIZE + WRITE_SIZE) x 1 KiB / time: what the L2s exchanged with memory (Infinity Cache hits included)
  AI          executed FLOP per byte of that traffic; ridge point of the chip: 2 500 TFLOP/s / 8 TB/s = 312 FLOP/B

    python scripts/roofline.py gpurun_out/roofline apex "Ape-X learner step" 47 > profiles/roofline_apex_r2.md
"""
import csv
import sys


def load(path, key="kernel"):
    try:
        return {r[key]: r for r in csv.DictReader(open(path))}
    except FileNotFoundError:
        return {}


def short(name):
    name = name.replace("void ", "")
    for a, b in (("GemmBody", "Gemm"), ("GldsBody", "Glds"), ("ColsumBody", "Colsum"), ("TreeBody", "Tree"), ("SampleBody", "Sample"),
                 ("DenseOp", "Dense"), ("HeadDwBody", "HeadDw"), ("NoBody", "-")):
        name = name.replace(a, b)
    return name[:110].replace("|", "/")


def main():
    d, algo, title, steps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    stats = load("%s/%s_stats.csv" % (d, algo), "Name")
    A, B, C = (load("%s/%s_pmc%s.csv" % (d, algo, p)) for p in "ABC")
    print("# %s: per-kernel roofline (MI355X, in situ)\n" % title)
    print(__doc__.split("\n\n")[1].replace("\n  ", "\n* ").replace("  time", "* time", 1))
    print()
    print("| kernel | launches/step | us/launch | us/step | TFLOP/s executed | % of 2.5 PF | MFMA busy % | HBM GB/s | % of 8 TB/s | FLOP/B | bound |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    tot_us = tot_flop = 0.0
    rows = sorted(stats.values(), key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows:
        name = r["Name"]
        if name.startswith("void at::") or "rocclr" in name or "at::native" in name:
            continue                                   # synthetic-data generation / memsets of the benchmark program
        calls, avg_us = int(r["Calls"]), float(r["AverageNs"]) / 1e3
        a, b, c = A.get(name), B.get(name), C.get(name)
        flop = mf = hbm = None
        if a:
            n = float(a["dispatches"])
            flop = (float(a.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0)) + float(a.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0))) * 512 / n
            gui = float(a["GRBM_GUI_ACTIVE"]) / n / 8.0
            mf = float(a["SQ_VALU_MFMA_BUSY_CYCLES"]) / n / (1024.0 * gui) if gui else 0.0
        if b and c:
            hbm = (float(b["FETCH_SIZE"]) / float(b["dispatches"]) + float(c["WRITE_SIZE"]) / float(c["dispatches"])) * 1024.0
        tf = flop / avg_us / 1e6 if flop is not None else None
        gbs = hbm / avg_us / 1e3 if hbm is not None else None
        ai = flop / hbm if flop and hbm else None
        if tf is None or gbs is None:
            bound = ""
        elif flop < 1e6:
            bound = "memory" if gbs > 2000 else "latency"
        else:
            bound = "matrix" if tf / 2500 > 0.25 else ("memory" if gbs / 8000 > 0.4 else "latency / occupancy")
        fmt = lambda v, f: (f % v) if v is not None else "-"   # noqa: E731
        print("| `%s` | %.2f | %.1f | %.1f | %s | %s | %s | %s | %s | %s | %s |" % (
            short(name), calls / steps, avg_us, avg_us * calls / steps, fmt(tf, "%.0f"), fmt(tf / 25 if tf is not None else None, "%.1f"),
            fmt(mf * 100 if mf is not None else None, "%.1f"), fmt(gbs, "%.0f"), fmt(gbs / 80 if gbs is not None else None, "%.1f"),
            fmt(ai, "%.0f"), bound))
        tot_us += avg_us * calls / steps
        tot_flop += (flop or 0.0) * calls / steps
    print("\nsum of kernel time %.1f us/step (streams overlap: the step is shorter), executed matrix work %.1f GFLOP/step "
          "= %.0f TFLOP/s over that sum" % (tot_us, tot_flop / 1e9, tot_flop / tot_us / 1e6 if tot_us else 0))


if __name__ == "__main__":
    main()
