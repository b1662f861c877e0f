#!/usr/bin/env python3
"""Time the fused MLP kernel alone: fused-encode path vs pre-encoded path (ANERF_LIB selects an ablation build)."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("a-nerf_amd.synth"); ops = importlib.import_module("a-nerf_amd.ops")
pipeline = importlib.import_module("a-nerf_amd.pipeline")
dev = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32, device="cuda")
cfg = ops.PathConfig()
net = ops.pack_params(cfg, {k: dev(v) for k, v in synth.make_net_params(11).items()})
sc = synth.make_scene(0, 512, 512, 600.0)
n = 65536
rb = pipeline.make_ray_batch(dev(sc["rays_o"][:n]), dev(sc["rays_d"][:n]))
cyl = dev(sc["cyl"])[None].expand(n, -1).contiguous(); skt = dev(sc["pose"]["skts"])[None]
cut = torch.full((24,), 0.5, device="cuda")
nf, st = ops.ray_bounds(rb, cyl); z, _ = ops.coarse_z(nf, st, rb, 64)
def timeit(f, reps=3):
    f(); torch.cuda.synchronize(); t = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); t.append(a.elapsed_time(b))
    return min(t)
P = n * 64
ms = timeit(lambda: ops.mlp_raw(cfg, net[0], net[1], rb, z, skt, 20.0, 20.0, cut, cut))
out = {"lib": os.environ.get("ANERF_LIB", "default"), "fused_ms": ms, "fused_TF": P * 1.723648e6 / ms / 1e9}
if "--b3" in sys.argv:
    net3 = ops.pack_params(cfg, {k: dev(v) for k, v in synth.make_net_params(11).items()}, 3)
    ms3 = timeit(lambda: ops.mlp_raw(cfg, net3[0], net3[1], rb, z, skt, 20.0, 20.0, cut, cut, precision="bf16x3"))
    out.update(b3_ms=ms3, b3_algorithmic_TF=P * 1.723648e6 / ms3 / 1e9)
if "--pre" in sys.argv:
    X = torch.rand(P // 4, 1080, device="cuda") - 0.5
    ms2 = timeit(lambda: ops.mlp_forward(cfg, net[0], net[1], X))
    out.update(pre_ms=ms2, pre_TF=(P // 4) * 1.723648e6 / ms2 / 1e9)
print(out)
