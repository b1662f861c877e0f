#!/usr/bin/env python3
"""Debug build only (-DANERF_EXP_STAGE_TIMING): per-wave clocks around every stage barrier of the render kernel.
   build: tools/ablate.sh stime "-DANERF_EXP_STAGE_TIMING"; run with ANERF_LIB=tools/exp/libanerf_stime.so python tools/stage_timing.py"""
import ctypes, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("a-nerf_amd.synth"); ops = importlib.import_module("a-nerf_amd.ops")
pipeline = importlib.import_module("a-nerf_amd.pipeline"); _lib = importlib.import_module("a-nerf_amd._lib")
dev = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32, device="cuda")
import ctypes as C
ap = importlib.import_module("a-nerf_amd.autograd_path")
cfg = ops.PathConfig(); cc = cfg.c(); lib = _lib.load()
TRAIN = os.environ.get("TRAIN") == "1"
if TRAIN:
    n, S = 3072, 80
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
    def f():
        _lib.check(lib.anerf_mlp_raw_train(C.byref(cc), p(packed), p(aux), p(rb), 11, p(z), p(skts), 384, None, None, 0, 20.0, 20.0, p(cut), p(cut),
                                           n, S, p(raw), C.byref(stt), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "train fwd")
    ntile = P // 128
else:
    net = ops.pack_params(cfg, {k: dev(v) for k, v in synth.make_net_params(11).items()})
    sc = synth.make_scene(0, 512, 512, 600.0)
    n = int(os.environ.get("N_RAYS", 65536))
    rb = pipeline.make_ray_batch(dev(sc["rays_o"][:n]), dev(sc["rays_d"][:n]))
    cyl = dev(sc["cyl"])[None].expand(n, -1).contiguous(); skt = dev(sc["pose"]["skts"])[None]
    cut = torch.full((24,), 0.5, device="cuda")
    nf, st = ops.ray_bounds(rb, cyl); z, _ = ops.coarse_z(nf, st, rb, 64)
    ntile = n * 64 // 128
    if os.environ.get("B3") == "1":
        net3 = ops.pack_params(cfg, {k: dev(v) for k, v in synth.make_net_params(11).items()}, 3)
        f = lambda: ops.mlp_raw(cfg, net3[0], net3[1], rb, z, skt, 20.0, 20.0, cut, cut, precision="bf16x3")
    else:
        f = lambda: ops.mlp_raw(cfg, net[0], net[1], rb, z, skt, 20.0, 20.0, cut, cut)
nrec = ntile // 997 + 1
assert nrec <= 64
buf = torch.zeros(64 * 4 * 128 * 3 + 6 * ntile, dtype=torch.int64, device="cuda")
lib.anerf_debug_set_timing_buf.argtypes = [ctypes.c_void_p]
f(); f(); torch.cuda.synchronize()
lib.anerf_debug_set_timing_buf(ctypes.c_void_p(buf.data_ptr()))
f(); torch.cuda.synchronize()
lib.anerf_debug_set_timing_buf(ctypes.c_void_p(0))
full = buf.cpu().numpy()
tiles = full[64 * 4 * 128 * 3:].reshape(ntile, 6)
t = full[:nrec * 4 * 128 * 3].reshape(nrec, 4, 128, 3)
ns = int((t[0, 0, :, 0] != 0).sum())
t = t[:, :, :ns, :].astype(np.float64)
arrive, landed, leave = t[..., 0], t[..., 1], t[..., 2]
print("tiles", nrec, "stages", ns)
stage_len = np.diff(leave, axis=2)            # leave-to-leave per wave
print("stage length (clocks) median %.0f mean %.0f" % (np.median(stage_len), stage_len.mean()))
park = leave - arrive
print("park per wave-stage: mean %.0f median %.0f  (vmcnt part mean %.0f, barrier part mean %.0f)" %
      (park.mean(), np.median(park), (landed - arrive).mean(), (leave - landed).mean()))
last = landed.max(axis=1, keepdims=True)
print("barrier latency after the last arrival: mean %.0f median %.0f" % ((leave - last).mean(), np.median(leave - last)))
skew = landed.max(axis=1) - landed.min(axis=1)
print("arrival spread (max-min over the 4 waves): mean %.0f median %.0f p90 %.0f" % (skew.mean(), np.median(skew), np.percentile(skew, 90)))
who = landed.argmax(axis=1)
print("last-arriving wave histogram", np.bincount(who.ravel(), minlength=4) / who.size)
first = landed.argmin(axis=1)
print("first-arriving wave histogram", np.bincount(first.ravel(), minlength=4) / first.size)
# per-stage profile of mean spread (which stages are bad?)
ms = skew.mean(axis=0)
print("mean spread by stage:", " ".join("%d" % x for x in ms))
mp = park.mean(axis=(0, 1))
print("mean park by stage:", " ".join("%d" % x for x in mp))
# compute-phase length per wave (leave[s-1] -> arrive[s]) relative to the mean of the 4 waves: persistent or random?
work = arrive[:, :, 1:] - leave[:, :, :-1]
dev_w = work - work.mean(axis=1, keepdims=True)
print("per-wave work deviation: std %.0f ; mean by wave %s" % (dev_w.std(), np.round(dev_w.mean(axis=(0, 2)), 0)))
c = np.corrcoef(dev_w[:, :, :-1].ravel(), dev_w[:, :, 1:].ravel())[0, 1]
print("lag-1 autocorrelation of a wave's deviation: %.2f" % c)
print("tile total clocks median %.0f" % np.median(leave[:, :, -1] - arrive[:, :, 0]))
sl = stage_len.mean(axis=(0, 1))
print("mean stage length by stage (stage s = leave[s] - leave[s-1]):", " ".join("%d" % x for x in sl))
print("first stage arrive - tile start unknown; compute phase of stage 0 not shown")

# ---- whole tiles: duration, and the gap between consecutive tiles on the same CU (HW_ID: cu 8..11, sh 12, se 13..15; XCC_ID)
dur = (tiles[:, 1] - tiles[:, 0]).astype(np.float64)
rdur = (tiles[:, 3] - tiles[:, 2]).astype(np.float64) * 10.0     # ns (100 MHz)
rec_tiles = np.arange(nrec) * 997
pro = arrive[:, :, 0].min(axis=1) - tiles[rec_tiles, 0]
epi = tiles[rec_tiles, 1] - leave[:, :, -1].max(axis=1)
print("prologue + stage 0 (tile start -> first stage barrier): median %.0f ; epilogue (last barrier -> end stamp): median %.0f" % (np.median(pro), np.median(epi)))
print("tiles %d: duration clocks median %.0f mean %.0f ; realtime median %.0f ns -> %.3f GHz" % (ntile, np.median(dur), dur.mean(), np.median(rdur), np.median(dur) / np.median(rdur)))
hw = tiles[:, 4]; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; xcc = tiles[:, 5] & 15
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print("distinct CUs", len(np.unique(key)))
gaps = []; per_cu = []
for k in np.unique(key):
    idx = np.where(key == k)[0]
    o = idx[np.argsort(tiles[idx, 2])]
    per_cu.append(len(o))
    g = (tiles[o[1:], 2] - tiles[o[:-1], 3]) * 10.0
    gaps.extend(g.tolist())
gaps = np.array(gaps)
print("tiles per CU: min %d max %d" % (min(per_cu), max(per_cu)))
print("gap between tiles on one CU: median %.0f ns mean %.0f ns p90 %.0f ns" % (np.median(gaps), gaps.mean(), np.percentile(gaps, 90)))
span = (tiles[:, 3].max() - tiles[:, 2].min()) * 10.0
print("kernel span %.1f us ; sum of tile realtime / CUs %.1f us" % (span / 1e3, rdur.sum() / len(np.unique(key)) / 1e3))
