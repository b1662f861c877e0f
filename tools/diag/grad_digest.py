"""SHA-256 of every gradient of one seeded training step (one-call entry points), for A/B runs of two builds of the library
(ANERF_LIB=... python tools/diag/grad_digest.py): equal digests = the same bits.  Configurations: config-3 shape at 384 and 3072 rays,
Mixamo shape (frame codes, per-ray poses, dskts) at 384 rays, fp32 and bf16x3."""
import hashlib
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ops = importlib.import_module("a-nerf_amd.ops")
pipeline = importlib.import_module("a-nerf_amd.pipeline")
synth = importlib.import_module("a-nerf_amd.synth")
ap = importlib.import_module("a-nerf_amd.autograd_path")
dev = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32, device="cuda")

for n, code, precision in ((384, 0, "fp32"), (3072, 0, "fp32"), (384, 16, "fp32"), (384, 16, "bf16x3")):
    b3 = precision == "bf16x3"
    cfg = ops.PathConfig(framecode_ch=code)
    mk = dict(framecode_ch=code, n_codes=8) if code else {}
    Pc = {k: dev(v) for k, v in synth.make_net_params(11, **mk).items()}
    Pf = {k: dev(v) for k, v in synth.make_net_params(12, **mk).items()}
    ro, rd, kp, skts, bones, cyls, pidx = synth.scene_batch(n, list(range(8)), H=512, W=512, focal=600.0, ray_seed=3, per_ray_pose=True)
    r = np.random.RandomState(5)
    S, Ni = 64, 16
    rnd = {"t_rand": r.rand(n, S), "u_imp": r.rand(n, Ni), "noise": r.randn(n, S), "noise_fine": r.randn(n, S + Ni)}
    pk = lambda P, w: ops.pack_params(cfg, P, w)
    shapes = [tuple(Pc[nm + sfx].shape) for nm in ops.PARAM_ORDER for sfx in (".weight", ".bias")]
    out, state = ops.train_forward(cfg, pk(Pc, 3 if b3 else 0), pk(Pf, 3 if b3 else 0), pipeline.make_ray_batch(dev(ro), dev(rd)), dev(skts), dev(cyls),
                                   S, Ni, cam_idx=dev(np.asarray(pidx, np.float32)) if code else None, codes_c=Pc.get("framecodes.codes.weight"),
                                   codes_f=Pf.get("framecodes.codes.weight"), precision=precision, **{k: dev(v) for k, v in rnd.items()})
    tgt = dev(r.rand(n, 3))
    g = {"rgb_map": 2.0 * (out["rgb_map"] - tgt) / n, "rgb0": 2.0 * (out["rgb0"] - tgt) / n}
    gc, gf, g_skts, gcc, gcf = ops.backward(state, g, pk(Pc, 4 if b3 else 1)[0], pk(Pf, 4 if b3 else 1)[0], ap.perm_tables(cfg, torch.device("cuda"), b3=b3),
                                            shapes, shapes, pk(Pc, 5 if b3 else 2)[0], pk(Pf, 5 if b3 else 2)[0], want_skts=True, want_codes_c=code > 0,
                                            want_codes_f=code > 0)
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for tns in list(gc) + list(gf) + [g_skts] + ([gcc, gcf] if code else []):
        h.update(tns.detach().cpu().numpy().tobytes())
    print(f"{n} rays, framecode_ch {code}, {precision}: {h.hexdigest()[:32]}  (|dW| of coarse pts_linears.0.weight {float(gc[0].norm()):.6e})")
