#!/usr/bin/env python3
"""Golden TRAJECTORIES: the reference's own training iteration, run by its own `Trainer.train_batch` (core/trainer.py:228-277)
for five iterations on CPU -- render() -> compute_loss (_compute_nerf_loss [+ _compute_kp_loss]) -> optimize (backward, Adam,
zero_grad, pose optimiser every `opt_pose_step` iterations) -> decay_optimizer_lrate -> RayCaster.update_embed_fns -- with
`pytest=True` (the reference's numpy-seeded jitter / noise overrides), SURREAL (surreal.txt) and Mixamo (mixamo.txt: frame
codes, L1, rot6d pose refinement through the reference's PoseOptLayer, the pose regulariser, the pose cadence).

Per iteration the file stores: total loss and its parts, PSNR of both heads, the learning rate and tau AFTER the iteration's
schedule updates, the gradient norms get_gradnorm reports, and a few parameter tensors after the step (plus frame codes and
the pose parameters for Mixamo).  tests/test_trajectory.py drives the HIP path (RayCaster + fused loss + FusedAdam + PoseOptLayer
mirrors) through the same five iterations.

Build container only (imports /root/reference; the reference never travels).  Inputs are regenerated from numpy seeds on both
sides (a-nerf_amd/synth.py); only the reference's outputs are stored.

Two accommodations, both outside the arithmetic under test:
  * core.pose_opt imports smplx / process_spin at module top (absent here): stubbed, as in gen_golden_fk.py; pytorch3d's
    axis_angle_to_matrix (absent) is the oracle's restatement, asserted against scipy there.
  * the reference was written for torch 1.x, where `optimizer.zero_grad()` ZEROES the gradients; torch >= 2.0 sets them to
    None, and the pose branch of `Trainer.optimize` then divides by zero in get_gradnorm (trainer.py:192-203 counts tensors
    with a gradient).  `_optim_step` is therefore run with `zero_grad(set_to_none=False)` -- the semantics the code was
    written against.

Schedules are made visible inside five iterations by flags, not by code changes: `--decay_unit 1 --lrate_decay 5` (the
learning rate falls by 10^(-1/5) per iteration), global_step = 50 000 * i (tau = 20 * 10^(global_step / 250 000)),
`--opt_pose_step 2` (pose parameters step at i = 2, 4 and accumulate in between).

Run:  python tests/golden/gen_golden_trajectory.py [case ...]      (writes tests/golden/trajectory_{surreal,mixamo,surreal_freq}.npz)
"""
import importlib
import os
import sys
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import gen_golden
from gen_golden import t

synth = importlib.import_module("a-nerf_amd.synth")
oracle = importlib.import_module("oracle.anerf_oracle")

N_ITERS = 5
GLOBAL_STEP_PER_ITER = 50000
N_POSES = 8
# parameter tensors stored after every iteration (full when small, leading slice otherwise)
WATCH = ["pts_linears.0.weight", "pts_linears.5.bias", "alpha_linear.weight", "views_linears.0.bias", "rgb_linear.weight"]
EXTRA = ["--decay_unit", "1", "--lrate_decay", "5"]


def batch_for(case):
    """the collated training batch (dataset.py:813-820) of the case, regenerated identically by the test"""
    ro, rd, kp, skts, bones, cyls, which = synth.scene_batch(case["n"], case["poses"], ray_seed=case["ray_seed"], per_ray_pose=True)
    n = case["n"]
    return dict(rays=t(np.stack([ro, rd])), target_s=t(np.random.default_rng(case["target_seed"]).random((n, 3))),
                kp_idx=torch.tensor(np.asarray(which), dtype=torch.int64), kp3d=t(kp), bones=t(bones), skts=t(skts), cyls=t(cyls),
                cam_idxs=t(np.asarray(which, dtype=np.float32)), fgs=torch.ones(n, 1), bgs=torch.ones(n, 3))


def pose_perturbation(shape):
    return (np.random.RandomState(77).randn(*shape) * 0.12).astype(np.float32)


CASES = {
    "surreal": dict(cfg="configs/surreal/surreal.txt", seeds=(11, 12), n=64, poses=[0, 1, 2, 3], ray_seed=21, target_seed=5, mixamo=False),
    "mixamo": dict(cfg="configs/mixamo/mixamo.txt", seeds=(21, 22), n=64, poses=[0, 1, 2, 3, 4, 5, 6, 7], ray_seed=22, target_seed=6,
                   mixamo=True),
    # --freq_schedule (outside the shipped configs): alpha = 6 * global_step / 500 000 = 0.6 i, so the bands open while the
    # trajectory runs (band 0 partly at i = 1, bands 0..2 at i = 5) and the closed ones must not move at all
    "surreal_freq": dict(cfg="configs/surreal/surreal.txt", seeds=(11, 12), n=64, poses=[0, 1, 2, 3], ray_seed=23, target_seed=7,
                         mixamo=False, extra=["--freq_schedule", "--freq_schedule_step", "500"]),
}


def run_case(cp, name, case):
    from core.raycasters import create_raycaster
    from core.trainer import Trainer
    from core.utils.skeleton_utils import SMPLSkeleton, get_per_joint_coords, smpl_rest_pose
    extra = list(EXTRA) + (["--opt_pose_step", "2"] if case["mixamo"] else []) + list(case.get("extra", []))
    args = gen_golden.make_args(cp, case["cfg"], extra)
    rest = (smpl_rest_pose * synth.SURREAL_SCALE).astype(np.float32)
    data_attrs = {"skel_type": SMPLSkeleton, "near": 0.0, "far": 1.0, "n_views": N_POSES, "hwf": (512, 512, 600.0),
                  "joint_coords": get_per_joint_coords(rest)}
    rk_train, rk_test, _, grad_vars, optimizer, _ = create_raycaster(args, data_attrs)
    caster = rk_test["ray_caster"]
    fc = 16 if args.opt_framecode else 0
    for net, seed in ((caster.network, case["seeds"][0]), (caster.network_fine, case["seeds"][1])):
        P = synth.make_net_params(seed, args.multires, args.multires_views, fc, N_POSES)
        net.load_state_dict({k: torch.tensor(v) for k, v in P.items()}, strict=True)
    assert rk_train["ray_caster"].module is caster            # nn.DataParallel without devices: a pass-through on the CPU
    rk_train["pytest"] = True                                 # reaches render_rays through render()'s **kwargs
    pose_optimizer = popt_kwargs = None
    if case["mixamo"]:
        import core.pose_opt as po
        poses = [synth.make_pose(k) for k in range(N_POSES)]
        data_attrs.update(rest_pose=rest[None], betas=np.zeros((1, 10), np.float32), kp3d=np.stack([q["kp"] for q in poses]),
                          bones=np.stack([q["bones"] for q in poses]))
        pose_optimizer, popt_kwargs = po.create_popt(args, data_attrs, ckpt=None, device="cpu")
        # move the pose parameters off their anchors by a seeded perturbation, as some thousand iterations of refinement would
        # have: the regulariser (_compute_kp_loss: zero below opt_pose_tol) is then active from the first iteration
        with torch.no_grad():
            layer = popt_kwargs["popt_layer"]
            layer.bones.add_(torch.tensor(pose_perturbation(tuple(layer.bones.shape))))
    trainer = Trainer(args, data_attrs, optimizer, pose_optimizer, rk_train, rk_test, popt_kwargs=popt_kwargs, device="cpu")

    def optim_step_torch1():
        optimizer.step()
        optimizer.zero_grad(set_to_none=False)
    trainer._optim_step = optim_step_torch1
    caster.train()
    batch = batch_for(case)
    out = {"n_iters": np.array(N_ITERS), "global_step_per_iter": np.array(GLOBAL_STEP_PER_ITER), "lrate0": np.array(args.lrate),
           "opt_pose_step": np.array(args.opt_pose_step), "N_samples": np.array(args.N_samples), "N_importance": np.array(args.N_importance)}
    for i in range(1, N_ITERS + 1):
        loss_dict, stats = trainer.train_batch(batch, i=i, global_step=GLOBAL_STEP_PER_ITER * i)
        rec = {"loss": loss_dict["total_loss"].item(), "rgb_loss": loss_dict["rgb_loss"].item(), "rgb_loss0": loss_dict["rgb_loss0"].item(),
               "psnr": stats["psnr"], "psnr0": stats["psnr0"], "lrate": stats["lrate"], "tau": float(stats["cutoff"]),
               "tau_d": float(caster.embeddirs_fn.get_tau()), "total_norm": stats["total_norm"], "avg_norm": stats["avg_norm"],
               "alpha_mean": stats["alpha"]}
        if case["mixamo"]:
            rec.update(kp_loss=loss_dict["kp_loss"].item(), mpjpc=stats["MPJPC"])
        if i == 1:          # what train_batch hands back, by name (trainer.py:262-277)
            out["keys_loss_dict"] = np.array(sorted(loss_dict))
            out["keys_stats"] = np.array(sorted(stats))
        if args.freq_schedule:
            rec.update(sched_alpha=caster.embed_fn.sched_alpha.item(), sched_alpha_d=caster.embeddirs_fn.sched_alpha.item())
        for k, v in rec.items():
            out[f"it{i}.{k}"] = np.array(v, dtype=np.float64)
        for tag, net in (("c", caster.network), ("f", caster.network_fine)):
            sd = dict(net.named_parameters())
            for w in WATCH:
                p = sd[w].detach().numpy()
                out[f"it{i}.{tag}.{w}"] = (p if p.size <= 4096 else p.reshape(-1)[:4096]).copy()
            if case["mixamo"]:
                out[f"it{i}.{tag}.framecodes.codes.weight"] = sd["framecodes.codes.weight"].detach().numpy().copy()
        if case["mixamo"]:
            layer = popt_kwargs["popt_layer"]
            out[f"it{i}.popt.bones"] = layer.bones.detach().numpy().copy()
            out[f"it{i}.popt.pelvis"] = layer.pelvis.detach().numpy().copy()
        print(name, i, {k: (round(v, 6) if isinstance(v, float) else v) for k, v in rec.items()})
    np.savez_compressed(os.path.join(gen_golden.OUT, f"trajectory_{name}.npz"), **out)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    cp = gen_golden.import_reference()
    for m in ["smplx", "h5py", "imageio", "core.process_spin", "core.load_data", "tensorboard"]:
        sys.modules.setdefault(m, mock.MagicMock(name=m))
    import core.utils.skeleton_utils as su
    su.p3dr.axis_angle_to_matrix = oracle.axis_angle_to_matrix
    for name, case in CASES.items():
        if len(sys.argv) < 2 or name in sys.argv[1:]:
            run_case(cp, name, case)


if __name__ == "__main__":
    main()
