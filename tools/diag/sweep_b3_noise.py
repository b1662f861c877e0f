"""Diagnostic for tests/test_hip_sweep.py's bf16x3 draws on very few samples (seeds 16 / 17: 31 rays x 8 and 15 rays x 22):
is the distance to the oracle arithmetic noise of single ReLU decisions, or structure?  Runs the same draw through the fp32
kernels and the bf16x3 kernels and prints, per precision, the worst parameter-gradient tensor, its five largest element errors
(a flipped unit shows as isolated bias entries / weight rows, structure as a spread) and the share of the squared error they hold.
Run on the GPU box:  python tools/diag/sweep_b3_noise.py [seed ...]"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
T = importlib.import_module("test_hip_sweep")
oracle = importlib.import_module("anerf_oracle")
ops = T.ops

for seed in [int(a) for a in sys.argv[1:]] or [16, 17]:
    for prec in ("fp32", "bf16x3"):
        d = T.draw(seed)
        if d["mv"] == 0:
            d["code"] = 0
        R = T.run_case(oracle, seed, prec, d)
        rows = []
        for got, P, which in ((R["gc"], R["oc"], "coarse"),) + (((R["gf"], R["of"], "fine"),) if d["Ni"] else ()):
            for i, nm in enumerate(ops.PARAM_ORDER):
                for j, sfx in enumerate((".weight", ".bias")):
                    a, w = got[2 * i + j].cpu().double(), P[nm + sfx].grad.double()
                    rows.append((float((a - w).abs().max() / (w.abs().max() + 1e-30)), float((a - w).norm() / (w.norm() + 1e-30)),
                                 f"{which} {nm}{sfx}", a, w))
        rows.sort(key=lambda x: -x[0])
        e, fro, name, a, w = rows[0]
        err = (a - w).abs().reshape(-1)
        top = torch.topk(err, min(5, err.numel()))
        share = float((top.values ** 2).sum() / (err ** 2).sum())
        tag = {k: v for k, v in d.items() if k not in ("rng", "cut_v", "cut_d")}
        print(f"seed {seed} [{prec}] {tag}")
        print(f"   worst tensor {name} {tuple(a.shape)}: element / max {e:.2e}, Frobenius {fro:.2e}; top-5 |err| / max "
              f"{[f'{float(v / w.abs().max()):.1e}' for v in top.values]} at {top.indices.tolist()} hold {100 * share:.0f} % of the squared error")
        print(f"   all tensors: median element / max {np.median([r[0] for r in rows]):.2e}, median Frobenius {np.median([r[1] for r in rows]):.2e}; "
              f"dskts {T.rel_max(R['g_skts'], R['sk'].grad):.2e}; outputs max |d rgb| {float((R['out']['rgb_map'].cpu() - R['o']['rgb_map']).abs().max()):.1e}")
