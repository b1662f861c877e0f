#!/usr/bin/env python3
"""Golden vectors for the embedder variants OUTSIDE the shipped configs, from the REFERENCE itself (build container only):

  variants_freq_schedule.npz   --freq_schedule (BARF-style band schedule, core/cutoff_embedder.py:185-197) at global_step 2750
                               (alpha = 3.3 of 6, tau = 20.51): eval outputs, and a pytest=True training step with loss,
                               parameter gradients and d loss / d skts
  variants_no_cutoff.npz       use_cutoff off (the argparse DEFAULT of run_nerf.py: plain Embedder for distances and views)
  variants_no_view_cutoff.npz  use_cutoff on, cutoff_viewdir off (plain Embedder for the view directions only)
  variants_cutoff_bones.npz    --cutoff_bones (multires_bones = 0): the bone directions gated by the distance gate, at global_step
                               60 000 (tau = 34.8: a gate sharp enough to matter)
  variants_noop_flags.npz      --opt_cutoff --normalize_cutoff: the reference's outputs are BIT-IDENTICAL to the run without them
                               (asserted here; the flags are stored / mis-keyed and never read)

Same import recipe and synthetic inputs as gen_golden.py.  Run:  python tests/golden/gen_golden_variants.py
"""
import os, sys, tempfile
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G          # noqa: E402

synth, t, OUT = G.synth, G.t, G.OUT


def make_args(config_parser, cfg_file, drop=(), extra=()):
    argv = []
    for line in open(os.path.join(G.REF, cfg_file)):
        line = line.strip()
        if not line or line.startswith("#") or "=" not in line:
            continue
        k, v = [s.strip() for s in line.split("=", 1)]
        if k in drop or v == "False":
            continue
        argv += ["--" + k] if v == "True" else ["--" + k, v]
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "g"), exist_ok=True)
    return config_parser().parse_args(argv + ["--no_reload", "--basedir", tmp, "--expname", "g"] + list(extra))


def build(cp, drop=(), extra=(), seeds=(11, 12)):
    from core.raycasters import create_raycaster
    from core.utils.skeleton_utils import SMPLSkeleton, get_per_joint_coords, smpl_rest_pose
    args = make_args(cp, "configs/surreal/surreal.txt", drop, extra)
    data_attrs = {"skel_type": SMPLSkeleton, "near": 0.0, "far": 1.0, "n_views": 8,
                  "joint_coords": get_per_joint_coords(smpl_rest_pose * synth.SURREAL_SCALE)}
    rk_train, rk_test, _, _, _, _ = create_raycaster(args, data_attrs)
    caster = rk_test["ray_caster"]
    for net, seed in [(caster.network, seeds[0]), (caster.network_fine, seeds[1])]:
        P = synth.make_net_params(seed, args.multires, args.multires_views, 0, 8)
        net.load_state_dict({k: torch.tensor(v) for k, v in P.items()}, strict=True)
    rk_train["ray_caster"] = caster
    return args, caster, rk_train, rk_test


def eval_and_train(caster, rk_train, rk_test, pose_eval, pose_train, ray_seed):
    """eval outputs (shared pose) + one pytest=True training step (per-ray poses): loss, gradients, dskts"""
    from core.trainer import img2mse
    g = {}
    caster.eval()
    ro, rd, kp, skts, bones, cyls, _ = synth.scene_batch(48, pose_eval, ray_seed=ray_seed)
    with torch.no_grad():
        out = G.run_render(rk_test, ro, rd, kp, skts, bones, cyls)
    g.update({"eval_" + k: v for k, v in G.np_dict(out).items()})
    caster.train()
    ro, rd, kp, skts, bones, cyls, _ = synth.scene_batch(40, pose_train, ray_seed=ray_seed + 1, per_ray_pose=True)
    skts_t = t(skts).requires_grad_(True)
    out = G.run_render(rk_train, ro, rd, kp, skts_t, bones, cyls, pytest=True)
    target = t(np.random.default_rng(ray_seed).random((40, 3)))
    bgs = torch.ones(40, 3)
    loss = img2mse(out["rgb_map"] + (1 - out["acc_map"])[..., None] * bgs, target) + \
        img2mse(out["rgb0"] + (1 - out["acc0"])[..., None] * bgs, target)
    caster.zero_grad()
    loss.backward()
    g.update({"train_" + k: v for k, v in G.np_dict(out).items()})
    g["loss"] = np.array(loss.item())
    g["dskts"] = skts_t.grad.numpy()
    for tag, net in [("c", caster.network), ("f", caster.network_fine)]:
        for n, p in net.named_parameters():
            gr = p.grad if p.grad is not None else torch.zeros_like(p)
            g[f"gnorm_{tag}.{n}"] = np.array(gr.norm().item())
            g[f"gslice_{tag}.{n}"] = gr.reshape(-1)[:64].numpy().copy()
        # the tensors whose COLUMNS carry the schedule / the gates: 24 full rows each (every input column)
        g[f"grows_{tag}.pts_linears.0.weight"] = net.pts_linears[0].weight.grad[:24].numpy().copy()
        g[f"grows_{tag}.pts_linears.5.weight"] = net.pts_linears[5].weight.grad[:24].numpy().copy()
        g[f"grows_{tag}.views_linears.0.weight"] = net.views_linears[0].weight.grad[:24].numpy().copy()
    return g


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    cp = G.import_reference()

    # ---- frequency schedule
    args, caster, rk_train, rk_test = build(cp, extra=["--freq_schedule"])
    step = 2750
    caster.update_embed_fns(step, args)
    g = eval_and_train(caster, rk_train, rk_test, [3], [4, 5], ray_seed=21)
    g["global_step"] = np.array(step)
    g["tau"] = np.array(caster.embed_fn.get_tau())
    g["alpha_v"] = np.array(caster.embed_fn.sched_alpha.item())
    g["alpha_d"] = np.array(caster.embeddirs_fn.sched_alpha.item())
    g["sched_w_v"] = caster.embed_fn.get_schedule_w().reshape(-1).numpy()
    g["sched_w_d"] = caster.embeddirs_fn.get_schedule_w().reshape(-1).numpy()
    np.savez_compressed(os.path.join(OUT, "variants_freq_schedule.npz"), **g)
    print("freq_schedule: alpha", g["alpha_v"], g["alpha_d"], "tau", g["tau"], "loss", g["loss"], "w_v", g["sched_w_v"][::2])

    # ---- no cutoff at all (argparse defaults of run_nerf.py:412-425)
    args, caster, rk_train, rk_test = build(cp, drop=("use_cutoff", "cutoff_viewdir", "cutoff_inputs"))
    assert type(caster.embed_fn).__name__ == "Embedder" and type(caster.embeddirs_fn).__name__ == "Embedder"
    g = eval_and_train(caster, rk_train, rk_test, [6], [7, 8], ray_seed=23)
    np.savez_compressed(os.path.join(OUT, "variants_no_cutoff.npz"), **g)
    print("no_cutoff loss", g["loss"])

    # ---- distance cutoff, plain view embedder
    args, caster, rk_train, rk_test = build(cp, drop=("cutoff_viewdir",))
    assert type(caster.embed_fn).__name__ == "CutoffEmbedder" and type(caster.embeddirs_fn).__name__ == "Embedder"
    g = eval_and_train(caster, rk_train, rk_test, [9], [10, 11], ray_seed=25)
    np.savez_compressed(os.path.join(OUT, "variants_no_view_cutoff.npz"), **g)
    print("no_view_cutoff loss", g["loss"])

    # ---- gated bone directions
    args, caster, rk_train, rk_test = build(cp, extra=["--cutoff_bones"])
    assert type(caster.embedbones_fn).__name__ == "CutoffEmbedder" and caster.embedbones_fn.out_dim == 72
    caster.update_embed_fns(60000, args)
    g = eval_and_train(caster, rk_train, rk_test, [15], [16, 17], ray_seed=29)
    g["tau"] = np.array(caster.embedbones_fn.get_tau())
    assert caster.embed_fn.get_tau() == caster.embedbones_fn.get_tau()
    np.savez_compressed(os.path.join(OUT, "variants_cutoff_bones.npz"), **g)
    print("cutoff_bones tau", g["tau"], "loss", g["loss"])

    # ---- flags the reference stores and never reads
    _, caster0, rk_train0, rk_test0 = build(cp)
    g0 = eval_and_train(caster0, rk_train0, rk_test0, [12], [13, 14], ray_seed=27)
    _, caster1, rk_train1, rk_test1 = build(cp, extra=["--opt_cutoff", "--normalize_cutoff"])
    g1 = eval_and_train(caster1, rk_train1, rk_test1, [12], [13, 14], ray_seed=27)
    assert set(g0) == set(g1)
    for k in g0:
        assert np.array_equal(g0[k], g1[k]), k
    assert not any(p.requires_grad for p in caster1.embed_fn.parameters())
    keep = {k: v for k, v in g1.items() if k.startswith(("eval_rgb_map", "eval_acc_map", "train_rgb_map", "loss", "gnorm_"))}
    np.savez_compressed(os.path.join(OUT, "variants_noop_flags.npz"), **keep)
    print("noop flags: bit-identical to the plain run; loss", g1["loss"])
    print({f: os.path.getsize(os.path.join(OUT, f)) for f in sorted(os.listdir(OUT)) if f.startswith("variants_")})


if __name__ == "__main__":
    main()
