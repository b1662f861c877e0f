#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE itself (/root/reference, build container only).

Inputs come from a-nerf_amd/synth.py (numpy seeds), so both boxes regenerate them; only the
reference's OUTPUTS are stored (tests/golden/*.npz).  The reference source never travels.

Import recipe = SURVEY.md Appendix B: stub cv2/pytorch3d/configargparse, exec run_nerf.config_parser
from its AST, create_raycaster() on configs/surreal/surreal.txt (or mixamo.txt), load numpy-seeded
weights, call core.trainer.render().

Run:  python tests/golden/gen_golden.py        (rewrites tests/golden/*.npz)
"""
import sys, os, types, ast, argparse, tempfile, importlib
from unittest import mock
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, ROOT)
synth = importlib.import_module("a-nerf_amd.synth")

OUT = os.path.join(ROOT, "tests", "golden")


def import_reference():
    for m in ["cv2", "pytorch3d", "pytorch3d.transforms", "pytorch3d.transforms.rotation_conversions"]:
        sys.modules[m] = mock.MagicMock(name=m)
    cap = types.ModuleType("configargparse")

    class AP(argparse.ArgumentParser):
        def add_argument(self, *a, **k):
            k.pop("is_config_file", None)
            return super().add_argument(*a, **k)
    cap.ArgumentParser = AP
    sys.modules["configargparse"] = cap
    sys.path.insert(0, REF)
    tree = ast.parse(open(os.path.join(REF, "run_nerf.py")).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "config_parser"][0]
    ns = {}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "cfg", "exec"), ns)
    return ns["config_parser"]


def make_args(config_parser, cfg_file, extra=()):
    argv = []
    for line in open(os.path.join(REF, cfg_file)):
        line = line.strip()
        if not line or line.startswith("#") or "=" not in line:
            continue
        k, v = [s.strip() for s in line.split("=", 1)]
        if v == "True":
            argv.append("--" + k)
        elif v == "False":
            continue
        else:
            argv += ["--" + k, v]
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "g"), exist_ok=True)
    argv += ["--no_reload", "--basedir", tmp, "--expname", "g"] + list(extra)
    return config_parser().parse_args(argv)


def build_caster(config_parser, cfg_file, seed_c, seed_f, n_views=8, extra=()):
    from core.raycasters import create_raycaster
    from core.utils.skeleton_utils import SMPLSkeleton, get_per_joint_coords, smpl_rest_pose
    args = make_args(config_parser, cfg_file, extra)
    data_attrs = {"skel_type": SMPLSkeleton, "near": 0.0, "far": 1.0, "n_views": n_views,
                  "joint_coords": get_per_joint_coords(smpl_rest_pose * synth.SURREAL_SCALE)}
    rk_train, rk_test, _, _, _, _ = create_raycaster(args, data_attrs)
    caster = rk_test["ray_caster"]
    fc = 16 if args.opt_framecode else 0
    for net, seed in [(caster.network, seed_c), (caster.network_fine, seed_f)]:
        if net is None:
            continue
        P = synth.make_net_params(seed, args.multires, args.multires_views, fc, n_views)
        sd = {k: torch.tensor(v) for k, v in P.items()}
        net.load_state_dict(sd, strict=True)
    rk_train["ray_caster"] = caster          # bare module instead of DataParallel (CPU)
    return args, caster, rk_train, rk_test


scene_batch = synth.scene_batch


def t(x):
    return torch.tensor(np.asarray(x), dtype=torch.float32)


def run_render(rk, ro, rd, kp, skts, bones, cyls, cams=None, chunk=4096, **over):
    from core.trainer import render
    kw = dict(rk)
    kw.update(over)
    return render(64, 64, 75.0, chunk=chunk, rays=(t(ro), t(rd)), kp_batch=t(kp), skts=skts if torch.is_tensor(skts) else t(skts),
                  bones=t(bones), cyls=t(cyls), cams=cams, subject_idxs=None, **kw)


def np_dict(d):
    return {k: v.detach().cpu().numpy() for k, v in d.items() if torch.is_tensor(v)}


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    cp = import_reference()
    from core.utils.skeleton_utils import get_smpl_l2ws, get_kp_bounding_cylinder, smpl_rest_pose
    from core.utils.ray_utils import get_rays, kp_to_valid_rays, get_near_far_in_cylinder, \
        sample_from_lineseg, isample_from_lineseg
    from core.trainer import img2mse, img2l1

    # ---------------- G0: pin the synthetic-scene generator against the reference's helpers
    pose = synth.make_pose(3)
    l2w_ref = get_smpl_l2ws(pose["bones"], smpl_rest_pose * np.float32(synth.SURREAL_SCALE))
    cyl_ref = get_kp_bounding_cylinder(pose["kp"], ext_scale=0.001, extend_mm=250, top_expand_ratio=1.6,
                                       bot_expand_ratio=1.1, head="-y")
    c2w = synth.default_c2w()
    ro_ref, rd_ref = get_rays(64, 64, 75.0, torch.tensor(c2w))
    rays_v, vidx, _, bb = kp_to_valid_rays(torch.tensor(c2w)[None], 64, 64, 75.0, kps=torch.tensor(pose["kp"])[None],
                                           ext_scale=0.001)
    rays512, vidx512, _, _ = kp_to_valid_rays(torch.tensor(c2w)[None], 512, 512, 600.0, kps=torch.tensor(pose["kp"])[None],
                                              ext_scale=0.001)
    np.savez_compressed(os.path.join(OUT, "synth_pins.npz"), l2ws=l2w_ref.astype(np.float32), cyl=cyl_ref.astype(np.float32),
                        rays_d_64=rd_ref.numpy(), valid_idx_64=vidx[0].numpy(), n_valid_512=np.array(len(vidx512[0])),
                        valid_idx_512_head=vidx512[0][:8].numpy(), valid_idx_512_tail=vidx512[0][-8:].numpy())
    print("synth pins done; 512 valid rays:", len(vidx512[0]))

    # ---------------- surreal config caster (two nets, 7/4 freqs)
    args, caster, rk_train, rk_test = build_caster(cp, "configs/surreal/surreal.txt", 11, 12)
    caster.eval()

    # G1: eval, S=32, Ni=0, shared pose (BASELINE config 1 slice) + per-stage intermediates
    ro, rd, kp, skts, bones, cyls, _ = scene_batch(96, [0], ray_seed=1)
    out = run_render(rk_test, ro, rd, kp, skts, bones, cyls, N_samples=32, N_importance=0)
    near, far = get_near_far_in_cylinder(t(ro), t(rd), t(cyls), near=torch.zeros(96, 1), far=torch.ones(96, 1))
    z = sample_from_lineseg(near, far, 96, 32, perturb=0.)
    pts = t(ro)[:, None] + t(rd)[:, None] * z[..., None]
    enc = caster.encode_inputs(pts, [t(ro)[:, None], t(rd)[:, None]], t(kp), t(skts), t(bones),
                               joint_coords=caster.get_subject_joint_coords(None, "cpu"),
                               network=caster.network, **rk_test["preproc_kwargs"])
    X = torch.cat([enc["v"], enc["r"], enc["d"]], -1)
    raw = caster.run_network(enc, caster.network)
    g1 = np_dict(out)
    g1.update(near=near.numpy(), far=far.numpy(), z_vals=z.numpy(), X_head=X[:4].detach().numpy(),
              raw=raw.detach().numpy())
    np.savez_compressed(os.path.join(OUT, "eval_s32.npz"), **g1)
    print("G1", {k: v.shape for k, v in g1.items()})

    # G2: eval, hierarchical S=64 + Ni=16 (surreal.txt), shared pose
    ro, rd, kp, skts, bones, cyls, _ = scene_batch(64, [1], ray_seed=2)
    out = run_render(rk_test, ro, rd, kp, skts, bones, cyls)
    np.savez_compressed(os.path.join(OUT, "eval_hier.npz"), **np_dict(out))
    # importance-sampling stage vectors from the coarse weights of this case
    near, far = get_near_far_in_cylinder(t(ro), t(rd), t(cyls), near=torch.zeros(64, 1), far=torch.ones(64, 1))
    z = sample_from_lineseg(near, far, 64, 64, perturb=0.)
    w_c = None
    with torch.no_grad():
        pts = t(ro)[:, None] + t(rd)[:, None] * z[..., None]
        enc = caster.encode_inputs(pts, [t(ro)[:, None], t(rd)[:, None]], t(kp), t(skts), t(bones),
                                   joint_coords=caster.get_subject_joint_coords(None, "cpu"),
                                   network=caster.network, **rk_test["preproc_kwargs"])
        raw = caster.run_network(enc, caster.network)
        r0 = caster.network.raw2outputs(raw, z, t(rd), 0., B=1.0, act_fn=torch.nn.functional.relu)
        zm, zs, sidx = isample_from_lineseg(z, r0["weights"], 16, det=True)
        zm128, zs128, _ = isample_from_lineseg(z, r0["weights"], 128, det=True)
    np.savez_compressed(os.path.join(OUT, "importance.npz"), z=z.numpy(), weights=r0["weights"].numpy(),
                        z_samples=zs.numpy(), z_merged=zm.numpy(), sorted_idx=sidx.numpy(),
                        z_samples128=zs128.numpy(), z_merged128=zm128.numpy())

    # G3: NaN fallback: shrink the cylinder radius so part of the rays miss it
    ro, rd, kp, skts, bones, cyls, _ = scene_batch(64, [2], ray_seed=3)
    cyl_small = cyls.copy()
    cyl_small[:, 2] *= 0.45
    nn_, ff_ = get_near_far_in_cylinder(t(ro), t(rd), t(cyl_small), near=torch.zeros(64, 1), far=torch.ones(64, 1))
    out = run_render(rk_test, ro, rd, kp, skts, bones, cyl_small, N_samples=16, N_importance=0)
    g3 = np_dict(out)
    g3.update(near=nn_.numpy(), far=ff_.numpy())
    assert np.isnan((cyl_small[:, 2:3] ** 2)).sum() == 0
    np.savez_compressed(os.path.join(OUT, "nan_fallback.npz"), **g3)
    print("G3 bounds changed rows:", int((np.abs(nn_.numpy() - nn_.numpy().mean()) < 1e-7).sum()))

    # G4: train mode, pytest=True (numpy-seeded jitter/noise), per-ray poses, loss + grads incl. skts
    caster.train()
    ro, rd, kp, skts, bones, cyls, which = scene_batch(48, [4, 5, 6], ray_seed=4, per_ray_pose=True)
    skts_t = t(skts).requires_grad_(True)
    out = run_render(rk_train, ro, rd, kp, skts_t, bones, cyls, pytest=True)
    target = t(np.random.default_rng(1).random((48, 3)))
    bgs = torch.ones(48, 3)
    loss = img2mse(out["rgb_map"] + (1 - out["acc_map"])[..., None] * bgs, target) + \
        img2mse(out["rgb0"] + (1 - out["acc0"])[..., None] * bgs, target)
    caster.zero_grad()
    loss.backward()
    g4 = np_dict(out)
    g4["loss"] = np.array(loss.item())
    g4["dskts"] = skts_t.grad.numpy()
    for tag, net in [("c", caster.network), ("f", caster.network_fine)]:
        for n, p in net.named_parameters():
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            g4[f"gnorm_{tag}.{n}"] = np.array(g.norm().item())
            g4[f"gslice_{tag}.{n}"] = g.reshape(-1)[:64].numpy().copy()
        g4[f"gfull_{tag}.pts_linears.5.bias"] = net.pts_linears[5].bias.grad.numpy().copy()
        g4[f"gfull_{tag}.rgb_linear.weight"] = net.rgb_linear.weight.grad.numpy().copy()
        g4[f"gfull_{tag}.alpha_linear.weight"] = net.alpha_linear.weight.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "train_pytest.npz"), **g4)
    print("G4 loss", loss.item(), "dskts norm", float(skts_t.grad.norm()))

    # G5: mixamo config: frame codes (920-wide view layer), L1 loss, cams
    args_m, caster_m, rk_train_m, rk_test_m = build_caster(cp, "configs/mixamo/mixamo.txt", 21, 22, n_views=8)
    caster_m.train()
    ro, rd, kp, skts, bones, cyls, which = scene_batch(40, [7, 8], ray_seed=5, per_ray_pose=True)
    cams = t(np.arange(40) % 8)
    skts_t = t(skts).requires_grad_(True)
    out = run_render(rk_train_m, ro, rd, kp, skts_t, bones, cyls, cams=cams, pytest=True)
    target = t(np.random.default_rng(2).random((40, 3)))
    bgs = torch.ones(40, 3)
    loss = img2l1(out["rgb_map"] + (1 - out["acc_map"])[..., None] * bgs, target) + \
        img2l1(out["rgb0"] + (1 - out["acc0"])[..., None] * bgs, target)
    caster_m.zero_grad()
    loss.backward()
    g5 = np_dict(out)
    g5["loss"] = np.array(loss.item())
    g5["dskts"] = skts_t.grad.numpy()
    for tag, net in [("c", caster_m.network), ("f", caster_m.network_fine)]:
        for n, p in net.named_parameters():
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            g5[f"gnorm_{tag}.{n}"] = np.array(g.norm().item())
            g5[f"gslice_{tag}.{n}"] = g.reshape(-1)[:64].numpy().copy()
        g5[f"gfull_{tag}.framecodes.codes.weight"] = net.framecodes.codes.weight.grad.numpy().copy()
    # eval with cams=-1 -> mean code path
    caster_m.eval()
    with torch.no_grad():
        out_e = run_render(rk_test_m, ro, rd, kp, t(skts), bones, cyls, cams=-torch.ones(40))
    g5.update({"eval_" + k: v for k, v in np_dict(out_e).items()})
    np.savez_compressed(os.path.join(OUT, "mixamo_train.npz"), **g5)
    print("G5 loss", loss.item())

    # G6: 64x64 frame, S=32, Ni=0 (BASELINE config 1) -> image-level rgb for the PSNR check
    caster.eval()
    sc = synth.make_scene(0, 64, 64, 75.0)
    n = len(sc["rays_o"])
    rep = lambda a: np.broadcast_to(a[None], (n,) + a.shape).copy()
    with torch.no_grad():
        out = run_render(rk_test, sc["rays_o"], sc["rays_d"], rep(sc["pose"]["kp"]), rep(sc["pose"]["skts"]),
                         rep(sc["pose"]["bones"]), rep(sc["cyl"]), N_samples=32, N_importance=0, chunk=4096)
    np.savez_compressed(os.path.join(OUT, "frame64.npz"), rgb_map=out["rgb_map"].numpy(), acc_map=out["acc_map"].numpy(),
                        disp_map=out["disp_map"].numpy(), n_rays=np.array(n))
    print("G6 rays", n)

    # G8: density / mesh query path (fwd_type='density' | 'mesh'), fine network, single pose
    caster.eval()
    pose = synth.make_pose(10)
    kps1, skts1, bones1 = t(pose["kp"])[None], t(pose["skts"])[None], t(pose["bones"])[None]
    qpts = t(np.random.default_rng(3).uniform(-0.8, 0.8, (333, 1, 3)).astype(np.float32)) + kps1[0, 0]
    with torch.no_grad():
        dens = caster(qpts, kps1, skts1, bones1, render_kwargs=rk_test["preproc_kwargs"], fwd_type="density")
        mesh = caster(kps1, skts1, bones1, radius=0.9, res=6, render_kwargs=rk_test["preproc_kwargs"], fwd_type="mesh")
    np.savez_compressed(os.path.join(OUT, "density.npz"), density=dens.numpy(), mesh=mesh.numpy())
    print("G8", dens.shape, mesh.shape, float(dens.abs().mean()))

    # G7: surreal_single (single_net, multires_views=0)
    args_s, caster_s, rk_train_s, rk_test_s = build_caster(cp, "configs/surreal/surreal_single.txt", 31, 31)
    caster_s.eval()
    ro, rd, kp, skts, bones, cyls, _ = scene_batch(32, [9], ray_seed=6)
    with torch.no_grad():
        out = run_render(rk_test_s, ro, rd, kp, skts, bones, cyls)
    g7 = np_dict(out)
    g7["cfg"] = np.array([args_s.multires, args_s.multires_views, int(args_s.single_net), args_s.N_samples, args_s.N_importance])
    np.savez_compressed(os.path.join(OUT, "single_net.npz"), **g7)
    print("G7", {k: v.shape for k, v in g7.items()})
    print("sizes:", {f: os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT) if f.endswith(".npz")})


if __name__ == "__main__":
    main()
