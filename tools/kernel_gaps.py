#!/usr/bin/env python3
"""Timeline view of a rocprofv3 --kernel-trace run (SQLite output): per kernel name the busy time, and the idle gaps between
consecutive dispatches on the device -- what the --stats view cannot show (launch gaps, tails between chunk calls).

  tools/kernel_gaps.py <kernel-trace dir> [--after-ms T]    (T: ignore the first T ms of the trace = set-up / warm-up)
"""
import glob
import os
import sqlite3
import sys


def main():
    d = sys.argv[1]
    after = float(sys.argv[sys.argv.index("--after-ms") + 1]) if "--after-ms" in sys.argv else 0.0
    f = glob.glob(os.path.join(d, "*_results.db")) + glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
    con = sqlite3.connect(f[0])
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    if "start" not in cols or "end" not in cols:
        print("kernels view columns:", cols)
        return
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    if not rows:
        print("no dispatches")
        return
    t0 = rows[0][1]
    rows = [(n, s, e) for n, s, e in rows if (s - t0) / 1e6 >= after]
    span = (rows[-1][2] - rows[0][1]) / 1e6
    busy, gaps, per = 0.0, [], {}
    last_end = rows[0][1]
    for n, s, e in rows:
        busy += (e - max(s, last_end)) / 1e6 if e > last_end else 0.0
        if s > last_end:
            gaps.append(((s - last_end) / 1e3, n))
        last_end = max(last_end, e)
        k = n.split("(")[0].replace("void ", "")[:70]
        p = per.setdefault(k, [0, 0.0])
        p[0] += 1
        p[1] += (e - s) / 1e6
    print(f"# {len(rows)} dispatches over {span:.2f} ms; device busy {busy:.2f} ms ({100 * busy / span:.1f} %), idle {span - busy:.2f} ms in {len(gaps)} gaps")
    for k, (c, ms) in sorted(per.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"{c:7d} {ms:12.3f} ms {ms / c * 1e3:10.1f} us/call  {k}")
    gaps.sort(reverse=True)
    print("# largest gaps (us, kernel that followed):")
    for g, n in gaps[:8]:
        print(f"{g:10.1f}  {n.split('(')[0][:70]}")
    import statistics
    if gaps:
        print(f"# gap median {statistics.median(g for g, _ in gaps):.1f} us, total {sum(g for g, _ in gaps) / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
