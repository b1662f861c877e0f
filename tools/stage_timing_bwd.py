#!/usr/bin/env python3
"""Debug build only (-DANERF_EXP_STAGE_TIMING): stage / tile clocks of k_mlp_bwd during a 3072-ray training step (the
records of the last launch = the coarse pass, 1536 tiles, survive).  build: tools/ablate.sh stime "-DANERF_EXP_STAGE_TIMING"; run with ANERF_LIB=tools/exp/libanerf_stime.so"""
import ctypes, importlib, os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
_lib = importlib.import_module("a-nerf_amd._lib")
lib = _lib.load()
ntile = 3072 * 80 // 128
buf = torch.zeros(64 * 4 * 128 * 3 + 6 * ntile, dtype=torch.int64, device="cuda")
lib.anerf_debug_set_bwd_timing_buf.argtypes = [ctypes.c_void_p]
lib.anerf_debug_set_bwd_timing_buf(ctypes.c_void_p(buf.data_ptr()))
import bench
sys.argv = ["bench.py", "--workload", "train", "--steps", "2", "--warmup", "1", "--cpu-rays", "0", "--extra", "off"]
bench.main()
torch.cuda.synchronize()
lib.anerf_debug_set_bwd_timing_buf(ctypes.c_void_p(0))
full = buf.cpu().numpy()
tiles = full[64 * 4 * 128 * 3:].reshape(ntile, 6)
live = tiles[:, 0] != 0
# fine-pass records beyond the coarse pass's 1536 tiles are stale (fine launch ran first): keep the coarse tiles
nt = 3072 * 64 // 128
tiles = tiles[:nt]
nrec = nt // 97 + 1
t = full[:64 * 4 * 128 * 3].reshape(64, 4, 128, 3)[:nrec]
ns = int((t[0, 0, :, 0] != 0).sum())
t = t[:, :, :ns, :].astype(np.float64)
arrive, landed, leave = t[..., 0], t[..., 1], t[..., 2]
print("recorded tiles", nrec, "stages", ns)
park = leave - arrive
print("park per wave-stage: mean %.0f (vmcnt part %.0f, barrier part %.0f)" % (park.mean(), (landed - arrive).mean(), (leave - landed).mean()))
sl = np.diff(leave, axis=2).mean(axis=(0, 1))
print("mean stage length by stage:", " ".join("%d" % x for x in sl))
print("stage-loop clocks median %.0f" % np.median(leave[:, :, -1] - arrive[:, :, 0]))
dur = (tiles[:, 1] - tiles[:, 0]).astype(np.float64)
rdur = (tiles[:, 3] - tiles[:, 2]).astype(np.float64) * 10.0
print("tiles %d: duration clocks median %.0f ; realtime median %.0f ns -> %.3f GHz" % (nt, np.median(dur), np.median(rdur), np.median(dur) / np.median(rdur)))
hw = tiles[:, 4]; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; xcc = tiles[:, 5] & 15
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
gaps = []
for k in np.unique(key):
    idx = np.where(key == k)[0]
    o = idx[np.argsort(tiles[idx, 2])]
    gaps.extend(((tiles[o[1:], 2] - tiles[o[:-1], 3]) * 10.0).tolist())
gaps = np.array(gaps)
print("distinct CUs %d ; gap between tiles on one CU: median %.0f ns mean %.0f ns" % (len(np.unique(key)), np.median(gaps), gaps.mean()))
print("kernel span %.1f us" % ((tiles[:, 3].max() - tiles[:, 2].min()) * 10.0 / 1e3))
