#!/usr/bin/env python3
"""Golden vectors for ray_noise_std > 0 (`pts + randn_like(pts) * ray_noise_std`, core/raycasters.py:660,674), from the
REFERENCE itself (build container only).  No shipped config sets it, so this is the only pin of that branch.

The random part is made reproducible by replacing `torch.randn_like` for the duration of the call with numpy-seeded
values (seed 1000 + number of samples of the point set): tests regenerate the same arrays (`cases.ray_noise_arrays`).
ray_noise.npz: surreal-config caster, train mode, pytest=True, 24 + 8 samples, ray_noise_std = 0.15 on 40 rays of three
poses: output dict, MSE loss on both heads, per-tensor gradient norms + slices, dskts.

Run:  python tests/golden/gen_golden_raynoise.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import import_reference, build_caster, scene_batch, run_render, np_dict, t, OUT   # noqa: E402

STD = 0.15


def seeded_randn_like(x):
    return torch.tensor(np.random.RandomState(1000 + x.shape[1]).randn(*x.shape), dtype=x.dtype)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    cp = import_reference()
    from core.trainer import img2mse
    args, caster, rk_train, rk_test = build_caster(cp, "configs/surreal/surreal.txt", 11, 12)
    caster.train()
    n, S, Ni = 40, 24, 8
    ro, rd, kp, skts, bones, cyls, which = scene_batch(n, [13, 14, 15], ray_seed=10, per_ray_pose=True)
    skts_t = t(skts).requires_grad_(True)
    real = torch.randn_like
    torch.randn_like = seeded_randn_like
    try:
        out = run_render(rk_train, ro, rd, kp, skts_t, bones, cyls, pytest=True, N_samples=S, N_importance=Ni, ray_noise_std=STD)
    finally:
        torch.randn_like = real
    target = t(np.random.default_rng(4).random((n, 3)))
    bgs = torch.ones(n, 3)
    loss = img2mse(out["rgb_map"] + (1 - out["acc_map"])[..., None] * bgs, target) + \
        img2mse(out["rgb0"] + (1 - out["acc0"])[..., None] * bgs, target)
    caster.zero_grad()
    loss.backward()
    g = np_dict(out)
    g["loss"] = np.array(loss.item())
    g["dskts"] = skts_t.grad.numpy()
    for tag, net in [("c", caster.network), ("f", caster.network_fine)]:
        for name, p in net.named_parameters():
            gr = p.grad if p.grad is not None else torch.zeros_like(p)
            g[f"gnorm_{tag}.{name}"] = np.array(gr.norm().item())
            g[f"gslice_{tag}.{name}"] = gr.reshape(-1)[:64].numpy().copy()
    # the same call without the offsets, to show the vectors do pin the branch
    out0 = run_render(rk_train, ro, rd, kp, t(skts), bones, cyls, pytest=True, N_samples=S, N_importance=Ni, ray_noise_std=0.0)
    g["rgb_map_no_noise"] = out0["rgb_map"].detach().numpy()
    np.savez_compressed(os.path.join(OUT, "ray_noise.npz"), **g)
    print("loss", loss.item(), "max |rgb - rgb_no_noise|", float((out["rgb_map"] - out0["rgb_map"]).abs().max()))


if __name__ == "__main__":
    main()
