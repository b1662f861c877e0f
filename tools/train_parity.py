"""Convergence parity of the two training precisions: fit a student network to a teacher's renders for a few hundred
steps with the exact-fp32 kernels and with the split-bf16 ones (same seeds, same batches, same perturbation noise) and
compare the loss / PSNR trajectories.  Writes a small text table (run on the GPU box)."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("a-nerf_amd.synth"); networks = importlib.import_module("a-nerf_amd.networks")
raycaster = importlib.import_module("a-nerf_amd.raycaster"); render_mod = importlib.import_module("a-nerf_amd.render")
optim = importlib.import_module("a-nerf_amd.optim")
dev = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32, device="cuda")
STEPS, N_RAND, S, NI = int(sys.argv[1]) if len(sys.argv) > 1 else 300, 1024, 64, 16
pk = {"density_scale": 1.0, "density_fn": torch.nn.functional.relu}


def make_caster(seed_c, seed_f):
    kw = dict(D=8, W=256, input_ch=360, input_ch_bones=72, input_ch_views=648, use_viewdirs=True)
    nc, nf = networks.NeRF(**kw), networks.NeRF(**kw)
    nc.load_state_dict({k: torch.tensor(v) for k, v in synth.make_net_params(seed_c).items()})
    nf.load_state_dict({k: torch.tensor(v) for k, v in synth.make_net_params(seed_f).items()})
    ck = {"cutoff": True, "cutoff_dist": 0.5, "cutoff_inputs": True, "cutoff_dim": 24}
    e_v, _ = networks.get_embedder(7, input_dims=24, cutoff_kwargs=dict(ck, dist_inputs=False))
    e_b, _ = networks.get_embedder(0, input_dims=72, cutoff_kwargs={"cutoff": False})
    e_d, _ = networks.get_embedder(4, input_dims=72, cutoff_kwargs=dict(ck, dist_inputs=True))
    return raycaster.RayCaster(nc, e_v, e_b, e_d, network_fine=nf).cuda()


ro, rd, kp, skts, bones, cyls, _ = synth.scene_batch(N_RAND * 8, list(range(8)), H=512, W=512, focal=600.0, ray_seed=3, per_ray_pose=True)
teacher = make_caster(101, 102).eval()
with torch.no_grad():
    tgt = render_mod.render(512, 512, 600.0, chunk=4096, rays=(dev(ro), dev(rd)), use_viewdirs=True, ray_caster=teacher, cams=None,
                            subject_idxs=None, N_samples=S, N_importance=NI, perturb=0.0, raw_noise_std=0.0, preproc_kwargs=pk,
                            kp_batch=dev(kp), skts=dev(skts), cyls=dev(cyls), bones=dev(bones))
    target_all = (tgt["rgb_map"] + (1 - tgt["acc_map"])[:, None]).clone()
curves = {}
for prec in ["fp32", "bf16x3"]:
    torch.manual_seed(0)
    caster = make_caster(11, 12)
    caster.train()
    caster.train_precision = prec
    opt = optim.FusedAdam([p for p in caster.parameters() if p.requires_grad], lr=5e-4)
    rng = np.random.default_rng(7)
    hist = []
    for it in range(STEPS):
        sel = torch.tensor(rng.choice(N_RAND * 8, N_RAND, replace=False), device="cuda")
        out = render_mod.render(512, 512, 600.0, chunk=4096, rays=(dev(ro)[sel], dev(rd)[sel]), use_viewdirs=True, ray_caster=caster,
                                cams=None, subject_idxs=None, N_samples=S, N_importance=NI, perturb=1.0, raw_noise_std=1.0,
                                preproc_kwargs=pk, kp_batch=dev(kp)[sel], skts=dev(skts)[sel], cyls=dev(cyls)[sel], bones=dev(bones)[sel])
        loss, stats = optim.fused_nerf_loss(out, target_all[sel], bgs=1.0)
        loss.backward()
        opt.step(zero_grad=True)
        if it % 25 == 0 or it == STEPS - 1:
            hist.append((it, float(loss.detach()), float(render_mod.mse2psnr(stats[3]))))
    curves[prec] = hist
print(f"# student fit to a teacher's renders, {N_RAND} rays/step, {S}+{NI} samples, Adam 5e-4, identical seeds/batches/noise")
print(f"{'step':>5} {'loss fp32':>12} {'loss bf16x3':>12} {'psnr fp32':>10} {'psnr bf16x3':>12}")
for (i, lf, pf), (_, lb, pb) in zip(curves["fp32"], curves["bf16x3"]):
    print(f"{i:5d} {lf:12.6f} {lb:12.6f} {pf:10.3f} {pb:12.3f}")
