#!/usr/bin/env python3
"""Debug build only (-DANERF_EXP_STAGE_TIMING): clocks around every stage wait / barrier of k_gemm_tn's heavy blocks during a
3072-ray training step (last launch = coarse pass).  build: tools/ablate.sh stime "-DANERF_EXP_STAGE_TIMING"; run with ANERF_LIB=tools/exp/libanerf_stime.so"""
import ctypes, importlib, os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
_lib = importlib.import_module("a-nerf_amd._lib")
lib = _lib.load()
buf = torch.zeros(16 * 4 * 128 * 3, dtype=torch.int64, device="cuda")
lib.anerf_debug_set_gemm_timing_buf.argtypes = [ctypes.c_void_p]
lib.anerf_debug_set_gemm_timing_buf(ctypes.c_void_p(buf.data_ptr()))
import bench
sys.argv = ["bench.py", "--workload", "train", "--steps", "2", "--warmup", "1", "--cpu-rays", "0", "--extra", "off"]
bench.main()
torch.cuda.synchronize()
lib.anerf_debug_set_gemm_timing_buf(ctypes.c_void_p(0))
t = buf.cpu().numpy().reshape(16, 4, 128, 3).astype(np.float64)
ok = t[:, 0, 5, 0] != 0
t = t[ok]
print("recorded blocks", t.shape[0])
arrive, landed, leave = t[..., 0], t[..., 1], t[..., 2]
print("per wave-stage: vmcnt wait mean %.0f median %.0f ; barrier wait mean %.0f median %.0f" %
      ((landed - arrive).mean(), np.median(landed - arrive), (leave - landed).mean(), np.median(leave - landed)))
sl = np.diff(leave, axis=2)
print("stage length: mean %.0f median %.0f p10 %.0f p90 %.0f   (8192 MFMA clocks)" % (sl.mean(), np.median(sl), np.percentile(sl, 10), np.percentile(sl, 90)))
work = arrive[:, :, 1:] - leave[:, :, :-1]
print("compute phase (leave -> next arrive): mean %.0f median %.0f" % (work.mean(), np.median(work)))
print("by wave: compute", np.round(work.mean(axis=(0, 2))), "vmcnt", np.round((landed - arrive).mean(axis=(0, 2))), "barrier", np.round((leave - landed).mean(axis=(0, 2))))
