#!/usr/bin/env python3
"""Time the training forward kernel (k_mlp_fwd<TRAIN>, anerf_mlp_raw_train) alone on BASELINE config 3's fine pass
(3072 rays x 80 samples = 245 760 samples, per-ray poses); ANERF_LIB selects an ablation build (tools/ablate.sh).
Use it for same-box A/B only: launched alone between host synchronisations the chip runs these 3 ms kernels at ~2.15 GHz
(clock ramp); inside a training step the same launch runs at 2.36 GHz and ~8 % faster."""
import ctypes as C, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_lib = importlib.import_module("a-nerf_amd._lib"); ops = importlib.import_module("a-nerf_amd.ops")
ap = importlib.import_module("a-nerf_amd.autograd_path"); synth = importlib.import_module("a-nerf_amd.synth")
pipeline = importlib.import_module("a-nerf_amd.pipeline")
dev = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32, device="cuda")
n, S = int(os.environ.get("N_RAYS", 3072)), 80
cfg = ops.PathConfig(); cc = cfg.c(); lib = _lib.load()
ro, rd, kp, skts, bones, cyls, _ = synth.scene_batch(n, list(range(8)), H=512, W=512, focal=600.0, ray_seed=3, per_ray_pose=True)
rb = pipeline.make_ray_batch(dev(ro), dev(rd)); skts, cyls = dev(skts), dev(cyls)
nf, st = ops.ray_bounds(rb, cyls); z, _ = ops.coarse_z(nf, st, rb, S)
packed, aux = ops.pack_params(cfg, {k: dev(v) for k, v in synth.make_net_params(11).items()})
P = n * S
T = ap.train_layout(cfg, P); pp = T.p_pad
sv = {k: torch.zeros(sh, device="cuda") for k, sh in [("h", (8, pp, 256)), ("f", (pp, 256)), ("g", (pp, 128)), ("x", (pp, T.x_width)), ("u", (pp, T.u_width))]}
p = lambda t: C.c_void_p(t.data_ptr())
stt = _lib.AnerfSaved(p(sv["h"]), p(sv["f"]), p(sv["g"]), p(sv["x"]), p(sv["u"]), pp)
raw = torch.empty(n, S, 4, device="cuda"); cut = torch.full((24,), 0.5, device="cuda")
def go():
    _lib.check(lib.anerf_mlp_raw_train(C.byref(cc), p(packed), p(aux), p(rb), 11, p(z), p(skts), 384, None, None, 0, 20.0, 20.0, p(cut), p(cut),
                                       n, S, p(raw), C.byref(stt), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "train fwd")
if os.environ.get("SHARED") == "1":      # one pose for every ray (stride 0) instead of per-ray replicated poses
    skts = skts[:1].contiguous()
STRIDE = 0 if skts.shape[0] == 1 else 384
def go():
    _lib.check(lib.anerf_mlp_raw_train(C.byref(cc), p(packed), p(aux), p(rb), 11, p(z), p(skts), STRIDE, None, None, 0, 20.0, 20.0, p(cut), p(cut),
                                       n, S, p(raw), C.byref(stt), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "train fwd")
def render():
    _lib.check(lib.anerf_mlp_raw(C.byref(cc), p(packed), p(aux), p(rb), 11, p(z), p(skts), STRIDE, None, None, 0, 20.0, 20.0, p(cut), p(cut),
                                 n, S, p(raw), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "render fwd")
if os.environ.get("RENDER") == "1":
    go = render
go(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); go(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ms = min(ts)
print(f"{os.path.basename(os.environ.get('ANERF_LIB', 'default')):28s} shared={os.environ.get('SHARED','0')} render={os.environ.get('RENDER','0')} P={P} fwd {ms:.3f} ms  {P * 1.723648e6 / ms / 1e9:.1f} TFLOP/s  ({100 * P * 1.723648e6 / ms / 1e9 / 157.3:.1f} % of fp32 MFMA peak)")
