#!/usr/bin/env python3
"""End to end through the drop-in surface, the way run_nerf.py drives the reference (run_nerf.py:504-640): a dataset in the
reference's on-disk layout -> the dataset reader -> create_raycaster -> Trainer.train_batch in a loop (optionally replayed
from a captured hipGraph) -> checkpoint -> reload -> render_path of a held-out camera.

There is no dataset in the image, so the dataset is SYNTHETIC: a teacher caster (numpy-seeded weights) renders N_kps poses from
N_cams cameras with render_path; images, masks and poses are written by dataset.write_npz_twin in the layout of
core/process_spin.py:234-297 (SURREAL arrangement: images (N_cams, N_kps), poses shared by the cameras).  The student starts
from other seeds and is trained on pixels sampled by H5PoseData (the reference's sampling, collate and index arithmetic).

  python tools/train_synthetic.py [--iters 300] [--hw 96] [--graph on|off] [--out DIR]
  python tools/train_synthetic.py --subject spheres --pose-noise 0.05 [--pretrain 1500] --iters 800 --pose-step 1

The second form is A-NeRF's own use, pose refinement (mixamo.txt's arrangement: rot6d pose layer from create_popt, its Adam folded
into the one-bucket optimiser by FusedAdam.from_torch, per-camera frame codes, pose regulariser): the images show the TRUE poses
of an analytic ball-and-stick body (`render_spheres`: shape and colours follow the pose, unlike the NeRF teacher's smooth volume),
the dataset's kp3d / bones / skts are perturbed estimates; the summary reports the mean per-joint error before / after.

Prints one line per 50 iterations and a final JSON summary (PSNR against the teacher's pixels at start / end, held-out view
PSNR, it/s, host time per iteration).  Used by tests/test_end_to_end.py with a short run.
"""
import argparse
import importlib
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
synth = importlib.import_module("a-nerf_amd.synth")
dataset = importlib.import_module("a-nerf_amd.dataset")
raycaster = importlib.import_module("a-nerf_amd.raycaster")
render_mod = importlib.import_module("a-nerf_amd.render")
optim = importlib.import_module("a-nerf_amd.optim")
trainer_mod = importlib.import_module("a-nerf_amd.trainer")
pose_opt = importlib.import_module("a-nerf_amd.pose_opt")
checkpoint = importlib.import_module("a-nerf_amd.checkpoint")


class Skel:
    joint_names = ["j%d" % i for i in range(24)]
    joint_trees = np.asarray(synth.SMPL_PARENTS)


def ref_args(config="surreal", **over):
    """the reference's own parsed arguments of configs/<config>/<config>.txt (tests/golden/args_<config>.json)"""
    d = json.load(open(os.path.join(ROOT, "tests", "golden", f"args_{config}.json")))
    d.pop("_config_file")
    d.update(basedir="/nonexistent", **over)
    return argparse.Namespace(**d)


def cameras(n_cams, tz=3.0):
    """n_cams cameras on a circle around the subject, looking at it (c2w, NeRF convention of synth.default_c2w)"""
    out = []
    base = synth.default_c2w(tz)
    for k in range(n_cams):
        a = 2 * np.pi * k / n_cams * 0.35            # a 126-degree arc: the views overlap
        R = np.array([[np.cos(a), 0, np.sin(a), 0], [0, 1, 0, 0], [-np.sin(a), 0, np.cos(a), 0], [0, 0, 0, 1]], dtype=np.float32)
        out.append((R @ base).astype(np.float32))
    return np.stack(out)


PALETTE = np.array([[0.85, 0.2, 0.2], [0.2, 0.7, 0.25], [0.2, 0.3, 0.85], [0.9, 0.75, 0.15], [0.7, 0.25, 0.75], [0.15, 0.75, 0.75]], np.float32)


def sphere_subject(kp, radius=0.1):
    """a body whose SHAPE follows the pose: one ball per joint and two along every bone (70 balls), coloured by body part"""
    parents = np.asarray(synth.SMPL_PARENTS)
    c, col = [kp], [PALETTE[np.arange(24) % len(PALETTE)]]
    for f in (1 / 3, 2 / 3):
        c.append(kp[1:] * f + kp[parents[1:]] * (1 - f))
        col.append(PALETTE[np.arange(1, 24) % len(PALETTE)])
    return np.concatenate(c).astype(np.float32), np.full(70, radius, np.float32), np.concatenate(col)


def render_spheres(kp, c2w, hw, focal):
    """(rgb [hw,hw,3] in [0,1] over white, mask [hw,hw,1]): nearest ray-ball hit per pixel, Lambert-shaded by a head light"""
    centres, rad, col = sphere_subject(kp)
    o, d = synth.camera_rays(hw, hw, focal, c2w)
    d = d / np.linalg.norm(d, axis=-1, keepdims=True)
    oc = o[..., None, :] - centres                                   # [H,W,S,3]
    b = (oc * d[..., None, :]).sum(-1)
    disc = b * b - ((oc * oc).sum(-1) - rad * rad)
    t = np.where(disc > 0, -b - np.sqrt(np.maximum(disc, 0)), np.inf)
    t = np.where(t > 0, t, np.inf)
    k = t.argmin(-1)
    tk = np.take_along_axis(t, k[..., None], -1)[..., 0]
    hit = np.isfinite(tk)
    pt = o + d * np.where(hit, tk, 0)[..., None]
    nrm = (pt - centres[k]) / rad[k][..., None]
    shade = 0.35 + 0.65 * np.clip(-(nrm * d).sum(-1), 0, 1)
    rgb = np.where(hit[..., None], col[k] * shade[..., None], 1.0).astype(np.float32)
    return rgb, hit[..., None].astype(np.uint8)


def dilate(m, n):
    for _ in range(n):
        g = m.copy()
        g[:, 1:] |= m[:, :-1]; g[:, :-1] |= m[:, 1:]; g[:, :, 1:] |= m[:, :, :-1]; g[:, :, :-1] |= m[:, :, 1:]
        m = g
    return m


def perturbed(poses, sigma, dev, seed=5):
    """the poses a pose ESTIMATOR would hand over: every joint rotation off by N(0, sigma) rad per axis, the pelvis by sigma/10
    (scene units); keypoints / skts re-derived from them by the library's forward kinematics"""
    r = np.random.RandomState(seed)
    bones = np.stack([q["bones"] for q in poses]).astype(np.float32)
    bones = bones + (r.randn(*bones.shape) * sigma).astype(np.float32)
    pelvis = np.stack([q["kp"][0] for q in poses]).astype(np.float32) + (r.randn(len(poses), 3) * sigma * 0.1).astype(np.float32)
    rest = torch.tensor((synth.SMPL_REST_POSE * synth.SURREAL_SCALE).astype(np.float32), device=dev)[None]
    with torch.no_grad():
        kp, skts, _, _ = pose_opt.calculate_kinematic(torch.tensor(bones, device=dev), torch.tensor(pelvis, device=dev), rest)
    return [{"kp": k, "bones": b, "skts": s} for k, b, s in zip(kp.cpu().numpy(), bones, skts.cpu().numpy())]


def build_dataset(path, hw, focal, n_kps, n_cams, dev, chunk=4096, pose_noise=0.0, subject="nerf"):
    """teacher renders -> the reference's dataset layout on disk; returns the held-out view's (c2w, pose, teacher image).
    subject "nerf": a teacher caster's renders (a smooth coloured volume filling the bounding cylinder -- its appearance hardly
    depends on the pose); "spheres": an analytic ball-and-stick body (render_spheres) whose silhouette and colours follow the pose.
    pose_noise > 0: the images show the TRUE poses, the file's kp3d / bones / skts are perturbed estimates (gt_kp3d stays true);
    a twin file `*_truepose.npz` holds the same images with the true poses."""
    args = ref_args()
    poses = [synth.make_pose(300 + k) for k in range(n_kps)]
    kp = np.stack([q["kp"] for q in poses])
    attrs = {"skel_type": Skel, "near": 0.0, "far": 1.0, "n_views": n_cams, "hwf": (hw, hw, focal),
             "joint_coords": dataset.per_joint_coords((synth.SMPL_REST_POSE * synth.SURREAL_SCALE).astype(np.float32), Skel.joint_trees)}
    _, rk_test, *_ = raycaster.create_raycaster(args, attrs, device=dev)
    teacher = rk_test["ray_caster"]
    for net, seed in ((teacher.network, 101), (teacher.network_fine, 102)):
        P = synth.make_net_params(seed, alpha_bias=6.0)                  # an opaque, strongly coloured subject: the student (alpha
        P["rgb_linear.bias"] = np.array([2.5, -2.5, 0.5], np.float32)   # bias 1, grey) starts far from it
        net.load_state_dict({k: torch.tensor(v) for k, v in P.items()})
    teacher.eval()
    c2ws = cameras(n_cams + 1)                       # the last camera is held out
    t = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32, device=dev)
    imgs, masks, c2w_all = [], [], []
    cyl = np.stack([synth.bounding_cylinder(k) for k in kp])
    for c in range(n_cams + 1):
        if subject == "spheres":
            out = [render_spheres(k, c2ws[c], hw, focal) for k in kp]
            rgbs, accs = np.stack([x[0] for x in out]), np.stack([x[1] for x in out]).astype(np.float32)
        else:
            rgbs, _, accs, _, _ = render_mod.render_path([c2ws[c]] * n_kps, (hw, hw, focal), chunk, rk_test, kp=t(kp), skts=t(np.stack([q["skts"] for q in poses])),
                                                         cyls=t(cyl), bones=t(np.stack([q["bones"] for q in poses])), white_bkgd=True, ret_acc=True,
                                                         ext_scale=args.ext_scale)
        if c == n_cams:
            held = (c2ws[c], rgbs)
            break
        imgs.append((np.clip(rgbs, 0, 1) * 255).round().astype(np.uint8))
        masks.append((accs > 0.05).astype(np.uint8))
        c2w_all.append(np.repeat(c2ws[c][None], n_kps, 0))
    imgs, masks = np.concatenate(imgs), np.concatenate(masks)            # (N_cams, N_kps) arrangement, flattened
    # sample where the body is, plus a one-pixel rim (the reference dilates its masks the same way for sampling)
    rim = dilate(masks[..., 0].astype(bool), 1 if subject == "nerf" else max(2, hw // 16))     # spheres: background around the body too
    est = perturbed(poses, pose_noise, dev) if pose_noise > 0 else poses
    data = {"imgs": imgs, "masks": masks, "sampling_masks": rim[..., None].astype(np.uint8),
            "bkgds": np.full((1, hw, hw, 3), 255, np.uint8), "bkgd_idxs": np.zeros(len(imgs), np.int64),
            "kp3d": np.stack([q["kp"] for q in est]), "gt_kp3d": kp, "bones": np.stack([q["bones"] for q in est]),
            "skts": np.stack([q["skts"] for q in est]), "cyls": np.stack([synth.bounding_cylinder(q["kp"]) for q in est]), "rest_pose": synth.SMPL_REST_POSE * synth.SURREAL_SCALE, "betas": np.zeros((1, 10)),
            "c2ws": np.concatenate(c2w_all), "focals": np.full(len(imgs), focal), "ext_scale": args.ext_scale}
    dataset.write_npz_twin(path, data)
    if pose_noise > 0:
        data.update(kp3d=kp, bones=np.stack([q["bones"] for q in poses]), skts=np.stack([q["skts"] for q in poses]), cyls=cyl)
        dataset.write_npz_twin(path.replace(".npz", "_truepose.npz"), data)
    return held, poses, teacher


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--hw", type=int, default=96)
    ap.add_argument("--n-kps", type=int, default=4)
    ap.add_argument("--n-cams", type=int, default=4)
    ap.add_argument("--n-rand", type=int, default=1024)
    ap.add_argument("--n-sample-images", type=int, default=8)
    ap.add_argument("--graph", default="on", choices=["on", "off"])
    ap.add_argument("--out", default=None)
    ap.add_argument("--staged-uploads", default="on", choices=["on", "off"], help="the reader's one-arena asynchronous upload (off: pageable copies)")
    ap.add_argument("--pose-noise", type=float, default=0.0, help="rad: the dataset's poses are the true ones perturbed by this much; "
                    "> 0 trains mixamo.txt's way (pose layer from create_popt, rot6d, frame codes, L1, pose regulariser)")
    ap.add_argument("--pose-step", type=int, default=4, help="opt_pose_step of the pose-refinement run (mixamo.txt: 20, over 500k iterations)")
    ap.add_argument("--pose-lrate", type=float, default=None, help="opt_pose_lrate (default: the config's)")
    ap.add_argument("--from-teacher", action="store_true", help="the student starts from the teacher's weights: isolates pose refinement")
    ap.add_argument("--subject", default="nerf", choices=["nerf", "spheres"], help="what the images show (see build_dataset)")
    ap.add_argument("--pretrain", type=int, default=0, help="pose refinement only: this many iterations on the TRUE poses first (no pose "
                    "layer), then the perturbed poses take over -- the finetune arrangement of the reference's *_finetune configs")
    ap.add_argument("--net-lrate", type=float, default=None, help="lrate of the networks (default: the config's; 0 freezes them)")
    ap.add_argument("--pose-coef", type=float, default=None, help="opt_pose_coef, the weight of the pose regulariser (default: the config's)")
    a = ap.parse_args(argv)
    dev = torch.device("cuda")
    out_dir = a.out or tempfile.mkdtemp(prefix="anerf_synth_")
    os.makedirs(out_dir, exist_ok=True)
    focal = 600.0 * a.hw / 512.0
    path = os.path.join(out_dir, "synthetic_train_h5py.npz")
    held, poses, teacher = build_dataset(path, a.hw, focal, a.n_kps, a.n_cams, dev, pose_noise=a.pose_noise, subject=a.subject)

    # ---- the student, as run_nerf.py builds it: dataset -> data_attrs -> create_raycaster -> optimiser -> Trainer
    ds = dataset.H5PoseData(path, device=dev, kind="surreal")
    ds.staged_uploads = a.staged_uploads == "on"
    attrs = ds.data_attrs(skel_type=Skel)
    refine = a.pose_noise > 0
    over = dict(N_rand=a.n_rand, N_sample_images=a.n_sample_images)
    if refine:              # mixamo.txt, whose 500k-iteration schedule is compressed to this run's length; its per-image frame codes stay
        # on for the analytic subject (off against the NeRF teacher, which has none and shares the student's architecture)
        over.update(opt_pose_step=a.pose_step, opt_framecode=a.subject == "spheres", loss_fn="MSE", lrate_decay=500, decay_unit=1000)
        if a.pose_lrate is not None:
            over["opt_pose_lrate"] = a.pose_lrate
        if a.pose_coef is not None:
            over["opt_pose_coef"] = a.pose_coef
    if a.net_lrate is not None:
        over["lrate"] = a.net_lrate
    args = ref_args("mixamo" if refine else "surreal", **over)
    torch.manual_seed(0)
    rk_train, rk_test, start, grad_vars, torch_opt, _ = raycaster.create_raycaster(args, attrs, device=dev)
    caster = rk_test["ray_caster"]
    fc = args.framecode_size if args.opt_framecode else 0
    for net, src, seed in ((caster.network, teacher.network, 11), (caster.network_fine, teacher.network_fine, 12)):
        net.load_state_dict(src.state_dict() if a.from_teacher else {k: torch.tensor(v) for k, v in synth.make_net_params(
            seed, args.multires, args.multires_views, fc, attrs["n_views"] if fc else 0).items()})
    attrs_t = dict(attrs, hwf=(a.hw, a.hw, focal))
    true_kp = torch.tensor(np.stack([q["kp"] for q in poses]), dtype=torch.float32, device=dev)
    n_per = a.n_rand // a.n_sample_images
    np.random.seed(0)
    popt_kwargs, layer = None, None

    def pose_error_mm():
        """mean per-joint position error of the layer's poses against the true ones, in mm (the reference's MPJPE / ext_scale)"""
        with torch.no_grad():
            kp = layer(np.arange(a.n_kps))[0]
        return float((kp - true_kp).norm(dim=-1).mean() / args.ext_scale)

    def run(tr, data, iters, tag=""):
        """the reference's loop (run_nerf.py:560-640): its sampling (numpy's global generator) and collate, inline -- the GPU step
        bounds this loop (prefetch > 0 would assemble batches ahead on a background thread: worth it only when sampling is the
        bottleneck)"""
        if a.graph == "on":
            tr.enable_graph(eager_steps=2)
        caster.train()
        hist, host = [], []
        t0 = time.perf_counter()
        for i, batch in enumerate(data.batches(dataset.image_batches(len(data), a.n_sample_images, iters), n_per, prefetch=0), 1):
            h0 = time.perf_counter()
            loss_dict, stats = tr.train_batch(batch, i=i, global_step=i)
            host.append(time.perf_counter() - h0)
            if i % 50 == 0 or i in (1, iters):
                hist.append((i, float(loss_dict["total_loss"].detach()), float(stats["psnr"])))
                print(f"{tag}iter {i:5d}  loss {hist[-1][1]:.5f}  psnr {hist[-1][2]:.2f} dB  lr {float(stats['lrate']):.2e}  tau {float(stats['cutoff']):.1f}"
                      + (f"  pose error {pose_error_mm():.1f} mm" if tr.popt_kwargs is not None else ""), flush=True)
        torch.cuda.synchronize()
        return hist, host, time.perf_counter() - t0

    pre = None
    if refine and a.pretrain > 0:      # the subject is learnt on the true poses first; the perturbed estimates take over afterwards
        ds0 = dataset.H5PoseData(path.replace(".npz", "_truepose.npz"), device=dev, kind="surreal")
        ds0.staged_uploads = ds.staged_uploads
        args0 = argparse.Namespace(**dict(vars(args), opt_pose=False, opt_pose_coef=0.0, lrate=ref_args("mixamo").lrate))     # same caster, no pose layer
        fused0 = optim.FusedAdam.from_torch(torch_opt).attach(caster)
        tr0 = trainer_mod.Trainer(args0, attrs_t, fused0.group_optimizer(0), None, rk_train, rk_test, popt_kwargs=None, device=dev)
        h0, _, dt0 = run(tr0, ds0, a.pretrain, tag="pretrain ")
        pre = {"iters": a.pretrain, "first": h0[0], "last": h0[-1], "it_per_s": a.pretrain / dt0}
        fused0.detach()
        if a.net_lrate is not None:
            for g in torch_opt.param_groups:
                g["lr"] = a.net_lrate
    if refine:              # run_nerf.py:523: the pose layer, its Adam and the regulariser's anchors from the dataset's attributes
        pose_optimizer, popt_kwargs = pose_opt.create_popt(args, attrs, ckpt=None, device=dev)
        layer = popt_kwargs["popt_layer"]
        fused = optim.FusedAdam.from_torch(torch_opt, pose_optimizer, pose_step_every=args.opt_pose_step).attach(caster, pose_layer=layer)
    else:
        fused = optim.FusedAdam.from_torch(torch_opt).attach(caster)
    tr = trainer_mod.Trainer(args, attrs_t, fused.group_optimizer(0), fused.group_optimizer(1) if refine else None, rk_train, rk_test,
                             popt_kwargs=popt_kwargs, device=dev)
    mpjpe0 = pose_error_mm() if refine else None
    hist, host, dt = run(tr, ds, a.iters)

    # ---- checkpoint in the reference's format, reload into a fresh caster, render the held-out camera
    ck = os.path.join(out_dir, f"{a.iters:06d}.tar")
    tr.save_nerf(ck, a.iters)
    _, rk2, *_ = raycaster.create_raycaster(args, attrs, device=dev)
    checkpoint.load_nerf(ck, rk2["ray_caster"])
    t = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32, device=dev)
    kw = dict(kp=t(np.stack([q["kp"] for q in poses])), skts=t(np.stack([q["skts"] for q in poses])),
              cyls=t(np.stack([synth.bounding_cylinder(q["kp"]) for q in poses])), bones=t(np.stack([q["bones"] for q in poses])),
              white_bkgd=True, ext_scale=args.ext_scale)
    pose_res = None
    if refine:              # the checkpoint restores the pose layer, its Adam state and the anchors through create_popt (pose_opt.py:51-60)
        ckd = torch.load(ck, map_location="cpu", weights_only=False)
        popt2, kw2 = pose_opt.create_popt(args, attrs, ckpt=ckd, device=dev)
        same = all(torch.equal(x, y) for x, y in zip(kw2["popt_layer"].state_dict().values(), layer.state_dict().values()))
        pose_res = {"mpjpe_mm_start": mpjpe0, "mpjpe_mm_end": pose_error_mm(), "pose_steps": int(fused._steps[1]),
                    "reloaded_layer_identical": bool(same), "reloaded_pose_adam_steps": int(next(iter(popt2.state_dict()["state"].values()))["step"])}
        with torch.no_grad():       # the held-out view is rendered with the REFINED poses
            rkp, rbones, rskts, _, _ = layer(np.arange(a.n_kps))
        kw.update(kp=rkp, skts=rskts, bones=layer.to_bones3d(rbones), cyls=t(np.stack([synth.bounding_cylinder(k) for k in rkp.cpu().numpy()])))
    if args.opt_framecode:      # a camera no frame code was learnt for: index -1 = the mean code (embedding.py:21-22)
        kw["cams"] = torch.full((a.n_kps,), -1, dtype=torch.int64, device=dev)
    img_a, *_ = render_mod.render_path([held[0]] * a.n_kps, (a.hw, a.hw, focal), args.chunk, rk_test, **kw)
    img_b, *_ = render_mod.render_path([held[0]] * a.n_kps, (a.hw, a.hw, focal), args.chunk, rk2, **kw)
    psnr_held = float(-10 * np.log10(np.mean((np.clip(img_a, 0, 1) - np.clip(held[1], 0, 1)) ** 2)))
    # ---- run_nerf.py's periodic test render (run_nerf.py:553-590, render_testset :148-180): the dataset's render subset --
    # cameras, poses, ground-truth images, backgrounds by index -- through render_path, PSNR against the dataset's own images
    rd = ds.render_data()
    tt = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32, device=dev)
    cyl_rd = torch.tensor(np.stack([synth.bounding_cylinder(k, ext_scale=args.ext_scale) for k in rd["kp3d"]]), dtype=torch.float32, device=dev)
    kp_rd, skts_rd, bones_rd = tt(rd["kp3d"]), tt(rd["skts"]), tt(rd["bones"])
    if refine and rd.get("kp_idxs") is not None:          # run_nerf.py:555-559: a refining run renders its test set with the layer's poses
        with torch.no_grad():
            kp_rd, b, skts_rd, _, _ = layer(np.asarray(rd["kp_idxs"]))
        bones_rd = layer.to_bones3d(b)
        cyl_rd = torch.tensor(np.stack([synth.bounding_cylinder(k, ext_scale=args.ext_scale) for k in kp_rd.cpu().numpy()]), dtype=torch.float32, device=dev)
    rgbs_rd, *_ = render_mod.render_path(rd["c2ws"], rd["hwf"], args.chunk // 8, rk_test, bg_imgs=rd["bgs"], bg_indices=rd["bg_idxs"],
                                          centers=rd["center"], kp=kp_rd, skts=skts_rd, cyls=cyl_rd, bones=bones_rd,
                                          # run_nerf.py:571; the SURREAL arrangement reports camera-DATA indices (one per image), its
                                          # per-view code is the camera's number (load_surreal.py:330: image index // N_kps)
                                          cams=torch.tensor(np.asarray(rd["cam_idxs"]) // a.n_kps, device=dev) if args.opt_framecode else None,
                                          ext_scale=args.ext_scale, white_bkgd=args.white_bkgd)
    psnr_testset = float(-10 * np.log10(np.mean((np.clip(rgbs_rd, 0, 1) - rd["imgs"]) ** 2)))
    res = {"iters": a.iters, "graph": a.graph == "on", "first": hist[0], "last": hist[-1], "psnr_gain_db": hist[-1][2] - hist[0][2],
           "held_out_psnr_db": psnr_held, "testset_psnr_db": psnr_testset, "testset_frames": int(len(rd["imgs"])), "reload_max_abs_diff": float(np.abs(img_a - img_b).max()), "it_per_s": a.iters / dt,
           "host_ms_per_train_batch_median": float(np.median(host) * 1e3), "dataset": path, "checkpoint": ck,
           "max_memory_allocated_mb": torch.cuda.max_memory_allocated() / 2 ** 20, "reserved_mb": torch.cuda.memory_reserved() / 2 ** 20,
           "param_checksum": float(fused.flat.double().sum()), "pose_refinement": pose_res, "pretrain": pre, "subject": a.subject, "staged_uploads": ds.staged_uploads,
           "graphs": None if tr._gs is None else {"captures": tr._gs.captures, "replays": tr._gs.replays, "eager": tr._gs.eager_calls}}
    print(json.dumps(res))
    return res


if __name__ == "__main__":
    main()
