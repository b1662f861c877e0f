#!/usr/bin/env python3
"""Golden vectors for BASELINE config 5's shape (64 coarse + 128 importance samples), from the REFERENCE itself
(build container only; same import recipe and caster as gen_golden.py).

eval_hier128.npz: render() of the surreal-config caster (two networks) on 96 rays of pose 12 with N_samples=64,
N_importance=128 -- the full output dict (rgb_map, disp_map, acc_map, alpha [96,192], rgb0, disp0, acc0, alpha0).

Run:  python tests/golden/gen_golden_hier128.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import import_reference, build_caster, scene_batch, run_render, np_dict, OUT   # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    cp = import_reference()
    args, caster, rk_train, rk_test = build_caster(cp, "configs/surreal/surreal.txt", 11, 12)
    caster.eval()
    ro, rd, kp, skts, bones, cyls, _ = scene_batch(96, [12], ray_seed=9)
    with torch.no_grad():
        out = run_render(rk_test, ro, rd, kp, skts, bones, cyls, N_samples=64, N_importance=128)
    g = np_dict(out)
    np.savez_compressed(os.path.join(OUT, "eval_hier128.npz"), **g)
    print({k: v.shape for k, v in g.items()}, "acc mean", float(g["acc_map"].mean()))


if __name__ == "__main__":
    main()
