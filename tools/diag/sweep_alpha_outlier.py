"""Diagnostic for tests/test_hip_sweep.py: a draw whose fine-pass alpha differs from the oracle's in a handful of elements.
Prints the offending (ray, sample) pairs, the merged depths of the two evaluations around them and the inverse-CDF quantities
(bin, cdf step, u) of the importance sample involved.  Run on the GPU box:  python tools/diag/sweep_alpha_outlier.py seed [precision]"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
T = importlib.import_module("test_hip_sweep")
oracle = importlib.import_module("anerf_oracle")

seed = int(sys.argv[1])
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
d = T.draw(seed)
if d["mv"] == 0:
    d["code"] = 0
R = T.run_case(oracle, seed, prec, d)
out, o = R["out"], R["o"]
ex = o["_extras"]
print("draw", {k: v for k, v in d.items() if k not in ("rng", "cut_v", "cut_d")})
print("HIP output keys", sorted(out.keys()))
S, Ni = d["S"], d["Ni"]
da = (out["alpha"].cpu() - o["alpha"].detach()).abs()
bad = torch.nonzero(da > 5e-5)
print("alpha elements off by > 5e-5:", bad.tolist(), [f"{float(da[i, j]):.2e}" for i, j in bad.tolist()])
print("alpha0 max diff", float((out["alpha0"].cpu() - o["alpha0"].detach()).abs().max()), " rgb_map", float((out["rgb_map"].cpu() - o["rgb_map"].detach()).abs().max()))
w = ex["weights"].detach().double()
pw = w[:, 1:-1] + 1e-5
pdf = pw / pw.sum(-1, keepdim=True)
cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)
for r in sorted(set(i for i, _ in bad.tolist())):
    zf = ex["z_fine"][r].detach()
    idx = ex["sorted_idx"][r]
    print(f"ray {r}: sum(pw) - 1 = {float(pw[r].sum() - 1):.3e}, w_first {float(w[r, 0]):.3e}, w_last {float(w[r, -1]):.3e}")
    for key in ("z_vals", "z_fine", "z_merged"):
        if key in out:
            dz = (out[key][r].cpu() - zf).abs() if out[key].shape[-1] == zf.shape[-1] else None
            if dz is not None:
                print(f"   HIP {key} vs oracle z_fine: max diff {float(dz.max()):.3e} at {int(dz.argmax())}")
    cols = sorted(set(j for i, j in bad.tolist() if i == r))
    for j in cols:
        src = int(idx[j])
        what = f"coarse sample {src}" if src < S else f"importance sample {src - S}"
        line = f"   element {j}: {what}, z {float(zf[j]):.6f}, alpha HIP {float(out['alpha'][r, j]):.6f} oracle {float(o['alpha'][r, j]):.6f}"
        print(line)
    # every importance sample of the ray: its u, the cdf step it falls in
    u = torch.tensor(np.asarray(R.get("u_imp")))[r].double() if R.get("u_imp") is not None else None
    if u is not None:
        k = torch.searchsorted(cdf[r], u.contiguous(), right=True)
        den = cdf[r][k.clamp(max=cdf.shape[-1] - 1)] - cdf[r][(k - 1).clamp(min=0)]
        order = torch.argsort(den)[:4]
        print("   smallest cdf steps hit by this ray's samples:", [(int(q), f"{float(den[q]):.4e}", f"u - c_lo {float(u[q] - cdf[r][(k[q] - 1).clamp(min=0)]):.3e}") for q in order])
        # distance of each u to the nearest cdf entry (a sample on a bin edge picks either bin)
        gap = (u[:, None] - cdf[r][None]).abs().min(-1).values
        q = int(gap.argmin())
        print(f"   closest (u, cdf entry) pair: sample {q}, |u - cdf| = {float(gap[q]):.3e}")
