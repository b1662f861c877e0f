"""Diagnostic for tests/test_hip_sweep.py::test_seeded_render_configuration_vs_oracle: where do the fine-pass alphas of a drawn
render configuration leave the oracle's?  Re-runs the draw through the STAGED HIP calls (extras) and prints, for the offending
(ray, sample) pairs, depths, raw densities and the inverse-CDF step of the importance sample next to them.
Run on the GPU box:  python tools/diag/sweep_render_outlier.py seed [precision]"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
T = importlib.import_module("test_hip_sweep")
oracle = importlib.import_module("anerf_oracle")
ops, pipeline, synth, dev, t = T.ops, T.pipeline, T.synth, T.dev, T.t

seed = int(sys.argv[1])
precision = sys.argv[2] if len(sys.argv) > 2 else "fp32"
d = T.draw_render(seed)
n, S, Ni, mv, code, r = d["n"], d["S"], d["Ni"], d["mv"], d["code"], d["rng"]
b3 = precision == "bf16x3"
print("draw", {k: v for k, v in d.items() if k != "rng"})
ck = dict(multires_views=mv, framecode_ch=code, density_scale=d["density_scale"], softplus_shift=1.0 if d["softplus"] else None)
cfg, ocfg = ops.PathConfig(cutoff_bones=d["gate_bones"], **ck), oracle.OracleConfig(**ck)
mk = dict(multires_views=mv, **(dict(framecode_ch=code, n_codes=5) if code else {}))
Pc_np = synth.make_net_params(500 + seed, **mk)
Pf_np = Pc_np if d["single_net"] else synth.make_net_params(600 + seed, **mk)
ro, rd, kp, skts, bones, cyls, pidx = synth.scene_batch(n, list(range(30, 30 + d["n_poses"])), ray_seed=80 + seed, per_ray_pose=d["per_ray"])
rnd = {}
if d["perturb"]:
    rnd = {"t_rand": r.rand(n, S).astype(np.float32), "noise": r.randn(n, S).astype(np.float32)}
    if Ni:
        rnd.update(u_imp=r.rand(n, Ni).astype(np.float32), noise_fine=r.randn(n, S + Ni).astype(np.float32))
cam = r.randint(0, 5, n).astype(np.float32)
Pc, Pf = {k: dev(v) for k, v in Pc_np.items()}, {k: dev(v) for k, v in Pf_np.items()}
which = 3 if b3 else 0
net_c = ops.pack_params(cfg, Pc, which)
net_f = net_c if d["single_net"] else ops.pack_params(cfg, Pf, which)
codes_c, codes_f, cam_d = Pc.get("framecodes.codes.weight"), Pf.get("framecodes.codes.weight"), dev(cam) if code else None
if d["mean_code"]:
    codes_c, codes_f, cam_d = codes_c.mean(0, keepdim=True), codes_f.mean(0, keepdim=True), torch.zeros(n, device="cuda")
kw = dict(tau_v=d["tau_v"], tau_d=d["tau_d"], cam_idx=cam_d, codes_c=codes_c, codes_f=codes_f, lindisp=d["lindisp"], single_net=d["single_net"],
          precision=precision, **{k: dev(v) for k, v in rnd.items()})
rb = pipeline.make_ray_batch(dev(ro), dev(rd))
out = pipeline.render_rays_forward(cfg, net_c, net_f, rb, dev(skts), dev(cyls), S, Ni, extras=True, **kw)
with torch.no_grad():
    P1 = oracle.params_from_numpy(Pc_np)
    o = oracle.render_rays(ocfg, P1, P1 if d["single_net"] else oracle.params_from_numpy(Pf_np), oracle.make_ray_batch(t(ro), t(rd)), t(skts), t(cyls),
                           S, Ni, tau_v=d["tau_v"], tau_d=d["tau_d"], cam_idx=(-torch.ones(n) if d["mean_code"] else t(cam)) if code else None,
                           gate_r=d["gate_bones"], lindisp=d["lindisp"], single_net=d["single_net"], eval_mean_code=d["mean_code"],
                           return_extras=True, **{k: t(v) for k, v in rnd.items()})
ex, oe = out["_extras"], o["_extras"]
da = (out["alpha"].cpu() - o["alpha"]).abs()
bad = torch.nonzero(da > 1e-4).tolist()
print("alpha elements off by > 1e-4:", bad, [f"{float(da[i, j]):.2e}" for i, j in bad])
for key, okey in (("z_vals", "z_vals"), ("raw", "raw"), ("weights", "weights"), ("z_samples", "z_samples"), ("z_fine", "z_fine"), ("raw_fine", "raw_fine")):
    if key in ex and okey in oe:
        dd = (ex[key].cpu() - oe[okey]).abs()
        print(f"   {key:10s} max |HIP - oracle| {float(dd.max()):.3e}   (largest |oracle| {float(oe[okey].abs().max()):.3e})")
w = oe["weights"].double()
pw = w[:, 1:-1] + 1e-5
pdf = pw / pw.sum(-1, keepdim=True)
cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)
u_all = t(rnd["u_imp"]).double() if d["perturb"] else torch.linspace(0.0, 1.0, Ni)[None].expand(n, -1).double()
for ray in sorted(set(i for i, _ in bad)):
    idx = oe["sorted_idx"][ray]
    u = u_all[ray].contiguous()
    k = torch.searchsorted(cdf[ray], u, right=True)
    den = cdf[ray][k.clamp(max=cdf.shape[-1] - 1)] - cdf[ray][(k - 1).clamp(min=0)]
    print(f"ray {ray}: cdf steps under its importance samples {[f'{float(x):.3e}' for x in den]}")
    for j in sorted(set(j for i, j in bad if i == ray)):
        src = int(idx[j])
        what = f"coarse sample {src}" if src < S else f"importance sample {src - S} (u = {float(u[src - S]):.4f}, cdf step {float(den[src - S]):.3e})"
        print(f"   element {j}: {what}; z HIP {float(ex['z_fine'][ray, j]):.6f} oracle {float(oe['z_fine'][ray, j]):.6f}; sigma HIP {float(ex['raw_fine'][ray, j, 3]):.5f} "
              f"oracle {float(oe['raw_fine'][ray, j, 3]):.5f}; alpha HIP {float(out['alpha'][ray, j]):.6f} oracle {float(o['alpha'][ray, j]):.6f}")
