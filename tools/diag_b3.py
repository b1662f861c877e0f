"""Debug aid: where do bf16x3 and fp32 logits differ on a small golden case? (run on the GPU box)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_hip_parity as T
import cases
c = cases.build(sys.argv[1] if len(sys.argv) > 1 else "eval_s32")
ref = T.run_hip(c, precision="fp32")["_extras"]["raw"].cpu().numpy()
for it in range(6):
    out = T.run_hip(c, precision="bf16x3")["_extras"]["raw"].cpu().numpy()
    d = np.abs(out - ref).max(-1).reshape(-1)
    bad = np.nonzero(d > 1e-4)[0]
    print(it, "max", d.max(), "nbad", bad.size, "P", d.size)
    if bad.size:
        tiles = bad // 128
        print("   tiles", np.unique(tiles)[:20], "waves", np.unique((bad % 128) // 32), "first", bad[:16], "errs", d[bad[:8]])
