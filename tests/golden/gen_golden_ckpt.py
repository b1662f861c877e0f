#!/usr/bin/env python3
"""Checkpoint `.tar` written by the REFERENCE's own `Trainer.save_nerf` (core/trainer.py:485-506), build container only.

`reference_checkpoint(path, config)` builds the reference's caster (create_raycaster on the parsed config), its
PoseOptLayer (rot6d, when the config has opt_pose) and the two torch Adam optimizers, takes one optimizer step with
synthetic gradients so that the optimizer states exist, and calls the reference's unmodified `save_nerf` on a Trainer
shell (its __init__ wants a dataset; save_nerf reads five attributes).  Used two ways:
  * here: tests/golden/ckpt_manifest_<config>.json = key -> shape/dtype of that file (DATA; the .tar is not committed);
  * tests/test_checkpoint.py, when /root/reference is present: the file itself is written at test time and loaded by
    a-nerf_amd/checkpoint.load_nerf, and our save_nerf output is loaded back by the reference's modules.

Run:  python tests/golden/gen_golden_ckpt.py
"""
import importlib
import json
import os
import sys
import tempfile
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import gen_golden   # noqa: E402

synth = importlib.import_module("a-nerf_amd.synth")
N_POSES = 5


def pose_inputs():
    poses = [synth.make_pose(30 + k) for k in range(N_POSES)]
    return (np.stack([q["kp"] for q in poses]).astype(np.float32), np.stack([q["bones"] for q in poses]).astype(np.float32),
            (synth.SMPL_REST_POSE * synth.SURREAL_SCALE).astype(np.float32)[None])


def reference_modules(config="mixamo"):
    """(args, ref caster, ref render_kwargs_train, optimizer, popt layer | None, pose optimizer | None, anchors | None)"""
    cp = gen_golden.import_reference()
    for m in ["smplx", "h5py", "imageio", "core.process_spin", "core.load_data", "tensorboard", "torch.utils.tensorboard"]:
        sys.modules.setdefault(m, mock.MagicMock(name=m))
    oracle = importlib.import_module("oracle.anerf_oracle")
    import core.utils.skeleton_utils as su
    su.p3dr.axis_angle_to_matrix = oracle.axis_angle_to_matrix      # pytorch3d is absent (see gen_golden_fk.py)
    from core.raycasters import create_raycaster
    from core.utils.skeleton_utils import SMPLSkeleton, get_per_joint_coords, smpl_rest_pose
    cfg_file = {"mixamo": "configs/mixamo/mixamo.txt", "surreal": "configs/surreal/surreal.txt"}[config]
    args = gen_golden.make_args(cp, cfg_file)
    data_attrs = {"skel_type": SMPLSkeleton, "near": 0.0, "far": 1.0, "n_views": N_POSES,
                  "joint_coords": get_per_joint_coords(smpl_rest_pose * np.float32(synth.SURREAL_SCALE))}
    rk_train, rk_test, start, grad_vars, optimizer, _ = create_raycaster(args, data_attrs)
    caster = rk_test["ray_caster"]
    g = torch.Generator().manual_seed(0)
    for p in grad_vars:                                   # one step so that exp_avg / exp_avg_sq / step exist
        p.grad = torch.randn(p.shape, generator=g) * 1e-3
    optimizer.step()
    optimizer.zero_grad()
    popt = popt_optim = anchors = None
    if args.opt_pose:
        import core.pose_opt as po
        kps, bones, rest = pose_inputs()
        popt = po.PoseOptLayer(torch.tensor(kps), torch.tensor(bones), torch.tensor(rest), use_rot6d=args.opt_rot6d)
        popt_optim = torch.optim.Adam(params=list(popt.parameters()), lr=args.opt_pose_lrate, betas=(0.9, 0.999))
        for p in popt.parameters():
            p.grad = torch.randn(p.shape, generator=g) * 1e-3
        popt_optim.step()
        popt_optim.zero_grad()
        rots = su.axisang_to_rot(torch.tensor(bones).view(-1, 3)).view(N_POSES, 24, 3, 3)
        anchors = {"kps": torch.tensor(kps), "bones": torch.tensor(bones), "rots": rots, "beta": None}   # pose_opt.py:77-79
    return args, caster, rk_train, optimizer, popt, popt_optim, anchors


def reference_checkpoint(path, config="mixamo", global_step=4321):
    args, caster, rk_train, optimizer, popt, popt_optim, anchors = reference_modules(config)
    from core.trainer import Trainer
    tr = Trainer.__new__(Trainer)
    tr.args = args
    wrapper = rk_train["ray_caster"]
    if not hasattr(wrapper, "module"):                   # CPU: create_raycaster still wraps in DataParallel; be safe
        wrapper = mock.MagicMock(module=caster)
    tr.render_kwargs_train = {"ray_caster": wrapper}
    tr.optimizer, tr.pose_optimizer = optimizer, popt_optim
    tr.popt_kwargs = None if popt is None else {"popt_layer": popt, "popt_anchors": anchors}
    tr.save_nerf(path, global_step)                      # the reference's writer, unmodified
    return args, caster, optimizer, popt, popt_optim, anchors


def main():
    checkpoint = importlib.import_module("a-nerf_amd.checkpoint")
    for config in ("mixamo", "surreal"):
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "ref.tar")
            reference_checkpoint(path, config)
            ck = torch.load(path, map_location="cpu", weights_only=False)
        m = checkpoint.manifest(ck)
        with open(os.path.join(HERE, f"ckpt_manifest_{config}.json"), "w") as f:
            json.dump(m, f, indent=1, sort_keys=True)
        print(config, sorted(ck), "| optimizer params:", len(ck["optimizer_state_dict"]["state"]))


if __name__ == "__main__":
    main()
