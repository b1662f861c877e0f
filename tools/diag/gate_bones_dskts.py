#!/usr/bin/env python3
"""Where does the cutoff_bones pose gradient differ from the oracle's?  (tests/test_hip_backward.py::test_fused_input_gradient_kernel_variants_vs_oracle[4-0-True])"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import anerf_oracle as oracle
ops = importlib.import_module("a-nerf_amd.ops"); synth = importlib.import_module("a-nerf_amd.synth")
pipeline = importlib.import_module("a-nerf_amd.pipeline"); ap = importlib.import_module("a-nerf_amd.autograd_path")
render_mod = importlib.import_module("a-nerf_amd.render")
dev = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32, device="cuda"); t = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32)
n, S, Ni = 96, 24, 8
cfg = ops.PathConfig(cutoff_bones=True); ocfg = oracle.OracleConfig()
Pc_np, Pf_np = synth.make_net_params(51), synth.make_net_params(52)
ro, rd, kp, skts, bones, cyls, pidx = synth.scene_batch(n, [3, 4, 5, 6], ray_seed=21, per_ray_pose=True)
rng = np.random.RandomState(8)
rnd = {"t_rand": rng.rand(n, S).astype(np.float32), "u_imp": rng.rand(n, Ni).astype(np.float32), "noise": rng.randn(n, S).astype(np.float32), "noise_fine": rng.randn(n, S + Ni).astype(np.float32)}
target = np.random.default_rng(9).random((n, 3)).astype(np.float32)
Pc, Pf = {k: dev(v) for k, v in Pc_np.items()}, {k: dev(v) for k, v in Pf_np.items()}
pk = lambda P, w: ops.pack_params(cfg, P, w)
shapes = [tuple(Pc[nm + sfx].shape) for nm in ops.PARAM_ORDER for sfx in (".weight", ".bias")]
out, state = ops.train_forward(cfg, pk(Pc, 0), pk(Pf, 0), pipeline.make_ray_batch(dev(ro), dev(rd)), dev(skts), dev(cyls), S, Ni, **{k: dev(v) for k, v in rnd.items()})
leaf = {k: out[k].detach().clone().requires_grad_(True) for k in ("rgb_map", "acc_map", "rgb0", "acc0")}
loss, _ = render_mod.nerf_loss(leaf, dev(target), bgs=1.0, loss_fn="MSE")
g = dict(zip(leaf, torch.autograd.grad(loss, list(leaf.values()))))
_, _, g_skts, _, _ = ops.backward(state, g, pk(Pc, 1)[0], pk(Pf, 1)[0], ap.perm_tables(cfg, torch.device("cuda")), shapes, shapes, pk(Pc, 2)[0], pk(Pf, 2)[0], want_skts=True)
got = g_skts.cpu()
def orc(gate_r, tau_scale=1.0):
    oc, of = oracle.params_from_numpy(Pc_np, True), oracle.params_from_numpy(Pf_np, True)
    sk = t(skts).requires_grad_(True)
    o = oracle.render_rays(ocfg, oc, of, oracle.make_ray_batch(t(ro), t(rd)), sk, t(cyls), S, Ni, gate_r=gate_r, **{k: t(v) for k, v in rnd.items()})
    lo, _ = oracle.nerf_loss(o, t(target), 1.0); lo.backward(); return sk.grad, float(lo)
ref, lo = orc(True)
print("loss", float(loss), lo)
d = (got - ref).abs(); m = float(ref.abs().max())
i = int(d.argmax()); idx = np.unravel_index(i, d.shape)
print("max err / max", float(d.max()) / m, "at", idx, "got", float(got[idx]), "ref", float(ref[idx]))
print("err by column (rows 0..2): rotation cols", float(d[:, :, :3, :3].max()) / m, "translation col", float(d[:, :, :3, 3].max()) / m)
print("frob rel", float((got - ref).norm() / ref.norm()))
per_joint = d.reshape(n, 24, 16).amax((0, 2)) / m
print("per joint max err:", np.round(per_joint.numpy(), 5))
gc, gf, _, _, _ = ops.backward(state, g, pk(Pc, 1)[0], pk(Pf, 1)[0], ap.perm_tables(cfg, torch.device("cuda")), shapes, shapes, pk(Pc, 2)[0], pk(Pf, 2)[0], want_skts=True) if False else (None, None, None, None, None)
out2, state2 = ops.train_forward(cfg, pk(Pc, 0), pk(Pf, 0), pipeline.make_ray_batch(dev(ro), dev(rd)), dev(skts), dev(cyls), S, Ni, **{k: dev(v) for k, v in rnd.items()})
gc, gf, _, _, _ = ops.backward(state2, g, pk(Pc, 1)[0], pk(Pf, 1)[0], ap.perm_tables(cfg, torch.device("cuda")), shapes, shapes, pk(Pc, 2)[0], pk(Pf, 2)[0], want_skts=True)
oc, of = oracle.params_from_numpy(Pc_np, True), oracle.params_from_numpy(Pf_np, True)
o = oracle.render_rays(ocfg, oc, of, oracle.make_ray_batch(t(ro), t(rd)), t(skts), t(cyls), S, Ni, gate_r=True, **{k: t(v) for k, v in rnd.items()})
lo, _ = oracle.nerf_loss(o, t(target), 1.0); lo.backward()
for tag, got_l, P in (("c", gc, oc), ("f", gf, of)):
    for i, nm in enumerate(ops.PARAM_ORDER):
        for j2, sfx in enumerate((".weight", ".bias")):
            a, w = got_l[2 * i + j2].cpu(), P[nm + sfx].grad
            e = float((a - w).abs().max() / w.abs().max())
            if e > 2e-4:
                dd = (a - w).abs()
                cols = dd.amax(0) / float(w.abs().max()) if dd.dim() == 2 else None
                print(tag, nm + sfx, "err", e, "frob", float((a - w).norm() / w.norm()), "" if cols is None else ("worst cols " + str(torch.topk(cols, 5).indices.tolist()) + " n cols > 1e-4: " + str(int((cols > 1e-4).sum()))))
