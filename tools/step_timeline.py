#!/usr/bin/env python3
"""One training step on the device timeline, from a rocprofv3 --kernel-trace run (SQLite): the dispatches between two k_adam
launches (the step's last kernel), with start offset, duration and the idle gap in front of each -- where the time outside the
MFMA kernels goes (small kernels, launch gaps).

  tools/step_timeline.py <kernel-trace dir> [step index from the end, default 3]
"""
import glob
import os
import sqlite3
import sys


def main():
    d = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    f = glob.glob(os.path.join(d, "*_results.db")) + glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
    con = sqlite3.connect(f[0])
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    ends = [i for i, r in enumerate(rows) if "k_adam" in r[0]]
    if len(ends) < back + 2:
        print("not enough steps:", len(ends))
        return
    lo, hi = ends[-back - 1] + 1, ends[-back]
    t0 = rows[lo][1]
    prev_end = rows[lo - 1][2]
    tot_gap = tot_small = tot_big = 0.0
    print(f"# step of {hi - lo + 1} dispatches, {(rows[hi][2] - rows[lo - 1][2]) / 1e3:.1f} us from the previous step's last kernel to this step's last")
    print(f"{'start_us':>9} {'gap_us':>7} {'dur_us':>8}  kernel")
    for n, s, e in rows[lo:hi + 1]:
        gap = (s - prev_end) / 1e3
        dur = (e - s) / 1e3
        k = n.split("(")[0].replace("void ", "").replace("anerf::", "")[:60]
        print(f"{(s - t0) / 1e3:9.1f} {gap:7.1f} {dur:8.1f}  {k}")
        tot_gap += max(gap, 0.0)
        if any(x in n for x in ("k_mlp_", "k_gemm_tn")):
            tot_big += dur
        else:
            tot_small += dur
        prev_end = max(prev_end, e)
    print(f"# MFMA kernels {tot_big:.1f} us, other kernels {tot_small:.1f} us, idle gaps {tot_gap:.1f} us")


if __name__ == "__main__":
    main()
