"""Time anerf_weight_grads alone (k_gemm_tn + k_reduce_dw) over a range of sample counts."""
import ctypes as C, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_lib = importlib.import_module("a-nerf_amd._lib"); ops = importlib.import_module("a-nerf_amd.ops")
ap = importlib.import_module("a-nerf_amd.autograd_path")
cfg = ops.PathConfig(); cc = cfg.c(); lib = _lib.load(); dev = torch.device("cuda")
p = lambda t: C.c_void_p(t.data_ptr())
B3 = "--b3" in sys.argv
fn = lib.anerf_weight_grads_b3 if B3 else lib.anerf_weight_grads
for P in [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [24576, 49152, 98304, 196608, 245760]:
    T = _lib.AnerfTrainLayout(); lib.anerf_train_layout(C.byref(cc), P, C.byref(T)); pp = T.p_pad
    g = torch.Generator(device="cuda").manual_seed(0)
    r = lambda *s: torch.randn(*s, device=dev, generator=g) * 0.1
    sv = {"h": r(8, pp, 256), "f": r(pp, 256), "g": r(pp, 128), "x": r(pp, T.x_width), "u": r(pp, T.u_width)}
    st = _lib.AnerfSaved(p(sv["h"]), p(sv["f"]), p(sv["g"]), p(sv["x"]), p(sv["u"]), pp)
    dz, df, dzv, draw = r(8, pp, 256), r(pp, 256), r(pp, 128), r(pp, 4)
    shapes = [(256, 432), (256,)] + [(256, 256), (256,)] * 4 + [(256, 688), (256,)] + [(256, 256), (256,)] * 2 + \
             [(1, 256), (1,), (256, 256), (256,), (128, 256 + T.u_width), (128,), (3, 128), (3,)]
    grads = [torch.empty(s, device=dev) for s in shapes]
    gs = _lib.AnerfNetGrads()
    for i in range(12):
        gs.w[i] = grads[2 * i].data_ptr(); gs.b[i] = grads[2 * i + 1].data_ptr()
    ws = torch.empty(T.gemm_ws_floats, device=dev)
    px, pu = ap.perm_tables(cfg, dev)
    def go():
        _lib.check(fn(C.byref(cc), C.byref(st), p(dz), p(df), p(dzv), p(draw), P, p(px), p(pu), C.byref(gs),
                                          p(ws), T.gemm_ws_floats, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "wg")
    go(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        go()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    # reference check of one layer
    ref = dz[1][:P].T @ sv["h"][0][:P]
    err = float((grads[2] - ref).abs().max() / ref.abs().max())
    ref0 = torch.zeros(256, T.x_width, device=dev); ref0[:, px.long()] = dz[0][:P].T @ sv["x"][:P]
    err = max(err, float((grads[0] - ref0).abs().max() / ref0.abs().max()), float((grads[3] - dz[1][:P].sum(0)).abs().max() / dz[1][:P].sum(0).abs().max()))
    print(f"P={P:7d} chunks={T.gemm_chunks:3d} {ms:7.3f} ms  {1.7236e6 * P / ms / 1e9:7.1f} TFLOP/s  rel.err {err:.1e}")
