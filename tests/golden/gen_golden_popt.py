#!/usr/bin/env python3
"""`create_popt` fixtures from the REFERENCE's own function (core/pose_opt.py:14-83), build container only.

The data attributes are what a dataset's get_meta() hands over (dataset.py:433-484) for N_POSES synthetic poses; the parsed
configs are the reference's own (mixamo.txt: rot6d pose layer; perfcap.txt with --opt_pose_cache for the cached layer).  Cases:

  fresh          no checkpoint: layer parameters = the dataset's poses, anchors = the dataset's poses
  reload         ckpt = a layer / optimiser pair one Adam step further on, with anchors of its own: all three are restored
  no_reload      the same ckpt under --no_poseopt_reload: ignored
  multiview      kp_map / kp_uidxs (h36m's shared-body layout): root rotations per view, body rotations per distinct pose

Per case: the layer's state_dict, the optimiser's state (step / exp_avg / exp_avg_sq per parameter, lr, betas), the four
anchors, the cached FK outputs when the layer caches.  `--use_ckpt_anchor` is not a case: the reference's branch unpacks four of
forward's five return values (pose_opt.py:65 against :311-316) and raises before producing anything.

Run:  python tests/golden/gen_golden_popt.py      (writes tests/golden/popt_cases.npz)
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
import gen_golden           # noqa: E402
import gen_golden_ckpt      # noqa: E402

N_POSES = gen_golden_ckpt.N_POSES
KP_MAP = np.array([0, 1, 2, 0, 1])            # 5 images of 3 distinct poses (two cameras see poses 0 and 1)
KP_UIDXS = np.array([0, 1, 2])


def attrs(skel, multiview=False):
    kps, bones, rest = gen_golden_ckpt.pose_inputs()
    a = {"skel_type": skel, "rest_pose": rest[0], "betas": np.linspace(-1, 1, 10, dtype=np.float32)[None], "kp3d": kps, "bones": bones}
    if multiview:
        a.update(kp_map=KP_MAP, kp_uidxs=KP_UIDXS)
    return a


def record(out, tag, optim, kw):
    layer, anchors = kw["popt_layer"], kw["popt_anchors"]
    for k, v in layer.state_dict().items():
        out[f"{tag}.layer.{k}"] = v.detach().cpu().numpy().copy()          # (a view of the parameter otherwise)
    sd = optim.state_dict()
    out[f"{tag}.optim.lr_betas"] = np.array([sd["param_groups"][0]["lr"], *sd["param_groups"][0]["betas"]], dtype=np.float64)
    out[f"{tag}.optim.n_params"] = np.array(len(sd["param_groups"][0]["params"]))
    for pi, st in sd["state"].items():
        for k, v in st.items():
            out[f"{tag}.optim.{pi}.{k}"] = np.asarray(v.detach().cpu().numpy().copy() if torch.is_tensor(v) else v)
    for k, v in anchors.items():
        out[f"{tag}.anchor.{k}"] = v.detach().cpu().numpy().copy()
    out[f"{tag}.grads_none_or_zero"] = np.array(all(p.grad is None or not p.grad.any() for p in layer.parameters()))
    if layer.use_cache:
        for k in ("cache_kps", "cache_bones", "cache_skts", "cache_l2ws", "cache_rots"):
            out[f"{tag}.{k}"] = getattr(layer, k).detach().cpu().numpy().copy()


def main():
    cp = gen_golden.import_reference()
    for m in ["smplx", "h5py", "imageio", "core.process_spin", "core.load_data", "tensorboard", "torch.utils.tensorboard"]:
        sys.modules.setdefault(m, mock.MagicMock(name=m))
    oracle = importlib.import_module("oracle.anerf_oracle")
    import core.utils.skeleton_utils as su
    su.p3dr.axis_angle_to_matrix = oracle.axis_angle_to_matrix      # pytorch3d is absent (see gen_golden_fk.py)
    import core.pose_opt as po
    from core.utils.skeleton_utils import SMPLSkeleton

    out = {}
    args = gen_golden.make_args(cp, "configs/mixamo/mixamo.txt")
    assert args.opt_rot6d and not args.opt_pose_cache
    optim, kw = po.create_popt(args, attrs(SMPLSkeleton))
    record(out, "fresh", optim, kw)

    # a checkpoint one Adam step further on, with anchors of its own (what Trainer.save_nerf stores, trainer.py:497-503)
    g = torch.Generator().manual_seed(7)
    for p in kw["popt_layer"].parameters():
        p.grad = torch.randn(p.shape, generator=g) * 1e-2
    optim.step()
    shifted = {k: (v + 0.01 * (i + 1) if k != "rots" else v) for i, (k, v) in enumerate(kw["popt_anchors"].items())}
    ckpt = {"poseopt_layer_state_dict": kw["popt_layer"].state_dict(), "pose_optimizer_state_dict": optim.state_dict(),
            "poseopt_anchors": shifted}
    for k, v in ckpt["poseopt_layer_state_dict"].items():
        out[f"ckpt.layer.{k}"] = v.detach().cpu().numpy().copy()
    for pi, st in ckpt["pose_optimizer_state_dict"]["state"].items():
        for k, v in st.items():
            out[f"ckpt.optim.{pi}.{k}"] = np.asarray(v.detach().cpu().numpy().copy() if torch.is_tensor(v) else v)
    for k, v in shifted.items():
        out[f"ckpt.anchor.{k}"] = v.detach().cpu().numpy().copy()

    optim2, kw2 = po.create_popt(args, attrs(SMPLSkeleton), ckpt=ckpt)
    record(out, "reload", optim2, kw2)
    args.no_poseopt_reload = True
    optim3, kw3 = po.create_popt(args, attrs(SMPLSkeleton), ckpt=ckpt)
    record(out, "no_reload", optim3, kw3)

    # cached layer, axis-angle parameters (perfcap.txt's pose refinement + --opt_pose_cache), and the multi-view layout
    args = gen_golden.make_args(cp, "configs/perfcap/perfcap.txt")
    args.opt_pose_cache = True
    out["cached.opt_rot6d"] = np.array(bool(args.opt_rot6d))
    optim4, kw4 = po.create_popt(args, attrs(SMPLSkeleton))
    record(out, "cached", optim4, kw4)
    args.opt_pose_cache = False
    optim5, kw5 = po.create_popt(args, attrs(SMPLSkeleton, multiview=True))
    record(out, "multiview", optim5, kw5)
    out["multiview.kp_map"], out["multiview.kp_uidxs"] = KP_MAP, KP_UIDXS

    np.savez_compressed(os.path.join(HERE, "popt_cases.npz"), **out)
    print(len(out), "arrays;", sorted(k for k in out if k.startswith("reload.optim") or k.startswith("multiview.layer")))


if __name__ == "__main__":
    main()
