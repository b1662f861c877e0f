#!/usr/bin/env python3
"""Golden pins for the caller-side frame helpers with the camera variants the h36m / perfcap loaders produce: per-frame
image sizes, (fx, fy) focal pairs and explicit principal points -- from the REFERENCE's kp_to_valid_rays / get_rays /
cylinder_to_box_2d (/root/reference/core/utils/ray_utils.py:6-28,83-136, skeleton_utils.py:607-690; build container only).

Run:  python tests/golden/gen_golden_frame.py      (writes tests/golden/frame_pins.npz)
"""
import importlib, os, sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import gen_golden

synth = importlib.import_module("a-nerf_amd.synth")


def frame_inputs():
    """regenerated identically by the tests"""
    pose = synth.make_pose(5)
    c2w = synth.default_c2w()
    H, W = np.array([48, 80]), np.array([72, 56])
    focal = np.array([[70.0, 74.0], [90.0, 86.0]], dtype=np.float32)
    centers = np.array([[33.0, 25.0], [30.0, 41.0]], dtype=np.float32)
    return pose, c2w, H, W, focal, centers


def main():
    gen_golden.import_reference()
    from core.utils.ray_utils import kp_to_valid_rays
    pose, c2w, H, W, focal, centers = frame_inputs()
    poses = torch.tensor(np.stack([c2w, c2w]))
    rays, vidx, cyl, bb = kp_to_valid_rays(poses, H, W, focal, kps=torch.tensor(pose["kp"])[None], ext_scale=0.001, centers=centers)
    out = {"cyl": cyl.numpy()}
    for i in range(2):
        out[f"valid_idx_{i}"] = vidx[i].numpy()
        out[f"rays_o_{i}"] = rays[i][0].numpy()
        out[f"rays_d_{i}"] = rays[i][1].numpy()
        out[f"tl_{i}"], out[f"br_{i}"] = np.asarray(bb[i][0]), np.asarray(bb[i][1])
    np.savez_compressed(os.path.join(HERE, "frame_pins.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
