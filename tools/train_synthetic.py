#!/usr/bin/env python3
"""End to end through the drop-in surface, the way run_nerf.py drives the reference (run_nerf.py:504-640): a dataset in the
reference's on-disk layout -> the dataset reader -> create_raycaster -> Trainer.train_batch in a loop (optionally replayed
from a captured hipGraph) -> checkpoint -> reload -> render_path of a held-out camera.

There is no dataset in the image, so the dataset is SYNTHETIC: a teacher caster (numpy-seeded weights) renders N_kps poses from
N_cams cameras with render_path; images, masks and poses are written by dataset.write_npz_twin in the layout of
core/process_spin.py:234-297 (SURREAL arrangement: images (N_cams, N_kps), poses shared by the cameras).  The student starts
from other seeds and is trained on pixels sampled by H5PoseData (the reference's sampling, collate and index arithmetic).

  python tools/train_synthetic.py [--iters 300] [--hw 96] [--graph on|off] [--out DIR]

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
checkpoint = importlib.import_module("a-nerf_amd.checkpoint")


class Skel:
    joint_names = ["j%d" % i for i in range(24)]
    joint_trees = np.asarray(synth.SMPL_PARENTS)


def ref_args(**over):
    """the reference's own parsed arguments of configs/surreal/surreal.txt (tests/golden/args_surreal.json)"""
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "args_surreal.json")))
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


def build_dataset(path, hw, focal, n_kps, n_cams, dev, chunk=4096):
    """teacher renders -> the reference's dataset layout on disk; returns the held-out view's (c2w, pose, teacher image)"""
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
    m = masks[..., 0].astype(bool)
    rim = m.copy()
    rim[:, 1:] |= m[:, :-1]; rim[:, :-1] |= m[:, 1:]; rim[:, :, 1:] |= m[:, :, :-1]; rim[:, :, :-1] |= m[:, :, 1:]
    data = {"imgs": imgs, "masks": masks, "sampling_masks": rim[..., None].astype(np.uint8),
            "bkgds": np.full((1, hw, hw, 3), 255, np.uint8), "bkgd_idxs": np.zeros(len(imgs), np.int64),
            "kp3d": kp, "gt_kp3d": kp, "bones": np.stack([q["bones"] for q in poses]), "skts": np.stack([q["skts"] for q in poses]),
            "cyls": cyl, "rest_pose": synth.SMPL_REST_POSE * synth.SURREAL_SCALE, "betas": np.zeros((1, 10)),
            "c2ws": np.concatenate(c2w_all), "focals": np.full(len(imgs), focal), "ext_scale": args.ext_scale}
    dataset.write_npz_twin(path, data)
    return held, poses


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
    a = ap.parse_args(argv)
    dev = torch.device("cuda")
    out_dir = a.out or tempfile.mkdtemp(prefix="anerf_synth_")
    os.makedirs(out_dir, exist_ok=True)
    focal = 600.0 * a.hw / 512.0
    path = os.path.join(out_dir, "synthetic_train_h5py.npz")
    held, poses = build_dataset(path, a.hw, focal, a.n_kps, a.n_cams, dev)

    # ---- the student, as run_nerf.py builds it: dataset -> data_attrs -> create_raycaster -> optimiser -> Trainer
    ds = dataset.H5PoseData(path, device=dev, kind="surreal")
    attrs = ds.data_attrs(skel_type=Skel)
    args = ref_args(N_rand=a.n_rand, N_sample_images=a.n_sample_images)
    torch.manual_seed(0)
    rk_train, rk_test, start, grad_vars, _, _ = raycaster.create_raycaster(args, attrs, device=dev)
    caster = rk_test["ray_caster"]
    for net, seed in ((caster.network, 11), (caster.network_fine, 12)):
        net.load_state_dict({k: torch.tensor(v) for k, v in synth.make_net_params(seed).items()})
    fused = optim.FusedAdam([{"params": grad_vars, "lr": args.lrate}], betas=(0.9, 0.999)).attach(caster)
    attrs_t = dict(attrs, hwf=(a.hw, a.hw, focal))
    tr = trainer_mod.Trainer(args, attrs_t, fused.group_optimizer(0), None, rk_train, rk_test, popt_kwargs=None, device=dev)
    if a.graph == "on":
        tr.enable_graph(eager_steps=2)
    caster.train()
    n_per = a.n_rand // a.n_sample_images
    np.random.seed(0)
    hist, host = [], []
    t0 = time.perf_counter()
    # the reference's sampling (numpy's global generator) and collate; inline -- the GPU step bounds this loop (prefetch > 0 would
    # assemble batches ahead on a background thread: worth it only when sampling is the bottleneck)
    batches = ds.batches(dataset.image_batches(len(ds), a.n_sample_images, a.iters), n_per, prefetch=0)
    for i, batch in enumerate(batches, 1):
        h0 = time.perf_counter()
        loss_dict, stats = tr.train_batch(batch, i=i, global_step=i)
        host.append(time.perf_counter() - h0)
        if i % 50 == 0 or i in (1, a.iters):
            hist.append((i, float(loss_dict["total_loss"].detach()), float(stats["psnr"])))
            print(f"iter {i:5d}  loss {hist[-1][1]:.5f}  psnr {hist[-1][2]:.2f} dB  lr {float(stats['lrate']):.2e}  tau {float(stats['cutoff']):.1f}", flush=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    # ---- checkpoint in the reference's format, reload into a fresh caster, render the held-out camera
    ck = os.path.join(out_dir, f"{a.iters:06d}.tar")
    tr.save_nerf(ck, a.iters)
    _, rk2, *_ = raycaster.create_raycaster(args, attrs, device=dev)
    checkpoint.load_nerf(ck, rk2["ray_caster"])
    t = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32, device=dev)
    kw = dict(kp=t(np.stack([q["kp"] for q in poses])), skts=t(np.stack([q["skts"] for q in poses])),
              cyls=t(np.stack([synth.bounding_cylinder(q["kp"]) for q in poses])), bones=t(np.stack([q["bones"] for q in poses])),
              white_bkgd=True, ext_scale=args.ext_scale)
    img_a, *_ = render_mod.render_path([held[0]] * a.n_kps, (a.hw, a.hw, focal), args.chunk, rk_test, **kw)
    img_b, *_ = render_mod.render_path([held[0]] * a.n_kps, (a.hw, a.hw, focal), args.chunk, rk2, **kw)
    psnr_held = float(-10 * np.log10(np.mean((np.clip(img_a, 0, 1) - np.clip(held[1], 0, 1)) ** 2)))
    # ---- run_nerf.py's periodic test render (run_nerf.py:553-590, render_testset :148-180): the dataset's render subset --
    # cameras, poses, ground-truth images, backgrounds by index -- through render_path, PSNR against the dataset's own images
    rd = ds.render_data()
    tt = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32, device=dev)
    cyl_rd = torch.tensor(np.stack([synth.bounding_cylinder(k, ext_scale=args.ext_scale) for k in rd["kp3d"]]), dtype=torch.float32, device=dev)
    rgbs_rd, *_ = render_mod.render_path(rd["c2ws"], rd["hwf"], args.chunk // 8, rk_test, bg_imgs=rd["bgs"], bg_indices=rd["bg_idxs"],
                                          centers=rd["center"], kp=tt(rd["kp3d"]), skts=tt(rd["skts"]), cyls=cyl_rd, bones=tt(rd["bones"]),
                                          ext_scale=args.ext_scale, white_bkgd=args.white_bkgd)
    psnr_testset = float(-10 * np.log10(np.mean((np.clip(rgbs_rd, 0, 1) - rd["imgs"]) ** 2)))
    res = {"iters": a.iters, "graph": a.graph == "on", "first": hist[0], "last": hist[-1], "psnr_gain_db": hist[-1][2] - hist[0][2],
           "held_out_psnr_db": psnr_held, "testset_psnr_db": psnr_testset, "testset_frames": int(len(rd["imgs"])), "reload_max_abs_diff": float(np.abs(img_a - img_b).max()), "it_per_s": a.iters / dt,
           "host_ms_per_train_batch_median": float(np.median(host) * 1e3), "dataset": path, "checkpoint": ck,
           "max_memory_allocated_mb": torch.cuda.max_memory_allocated() / 2 ** 20, "reserved_mb": torch.cuda.memory_reserved() / 2 ** 20,
           "param_checksum": float(fused.flat.double().sum()),
           "graphs": None if tr._gs is None else {"captures": tr._gs.captures, "replays": tr._gs.replays, "eager": tr._gs.eager_calls}}
    print(json.dumps(res))
    return res


if __name__ == "__main__":
    main()
