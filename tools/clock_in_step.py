#!/usr/bin/env python3
"""Debug build only (tools/ablate.sh stime "-DANERF_EXP_STAGE_TIMING", ANERF_LIB=tools/exp/libanerf_stime.so): clock and span of
the training forward's fine pass INSIDE a training step (NR=<N_rand>), from the per-tile s_memtime / s_memrealtime stamps --
the stand-alone microbenchmarks see a lower clock."""
import ctypes, importlib, os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
_lib = importlib.import_module("a-nerf_amd._lib")
lib = _lib.load()
NR = int(os.environ.get("NR", 3072)); ntile = NR * 80 // 128
buf = torch.zeros(64 * 4 * 128 * 3 + 6 * ntile, dtype=torch.int64, device="cuda")
lib.anerf_debug_set_timing_buf.argtypes = [ctypes.c_void_p]
lib.anerf_debug_set_timing_buf(ctypes.c_void_p(buf.data_ptr()))
import bench
sys.argv = ["bench.py", "--workload", "train", "--n-rand", str(NR), "--steps", "12", "--warmup", "3", "--cpu-rays", "0", "--extra", "off"]
bench.main()
torch.cuda.synchronize()
lib.anerf_debug_set_timing_buf(ctypes.c_void_p(0))
tiles = buf.cpu().numpy()[64 * 4 * 128 * 3:].reshape(ntile, 6)
dur = (tiles[:, 1] - tiles[:, 0]).astype(np.float64); rdur = (tiles[:, 3] - tiles[:, 2]).astype(np.float64) * 10.0
print("TRAIN fwd fine pass inside a training step: tile clocks median %.0f, realtime %.0f ns -> %.3f GHz ; span %.1f us" %
      (np.median(dur), np.median(rdur), np.median(dur) / np.median(rdur), (tiles[:, 3].max() - tiles[:, 2].min()) * 10.0 / 1e3))
