#!/usr/bin/env python3
"""what is GEMM-chain || tile-chain concurrency worth at the 8-GPU shard size (384 rays)?

The backward of a 384-ray step is a chain of ONE-round tile kernels (k_mlp_bwd: 240 fine / 192 coarse tiles on 256 CUs, k_mlp_bwd_in
the same) and the weight-gradient GEMM of each pass (252 blocks).  Tile blocks and GEMM blocks both take a whole CU (512 registers per
wave), so concurrency can only put GEMM blocks on the CUs a tile kernel leaves idle (16 / 64 of 256).  Independent pairs:
  k_gemm_tn(fine)   ||  k_mlp_bwd_in(fine) .. k_composite_bwd(coarse) -> k_mlp_bwd(coarse)
  k_gemm_tn(coarse) ||  k_mlp_bwd_in(coarse)
This prices the pairs with the REAL kernels through the staged entry points on two HIP streams (tile chain on a high-priority stream,
enqueued FIRST so that its workgroups are dispatched first), against the same kernels back to back on one stream; the GEMM with its
own chunking (18 row chunks) and re-chunked to ~36 / ~72 (library built with -DANERF_EXP_GEMM_ROWS, ANERF_GEMM_ROWS=<rows>).
Usage: ANERF_LIB=tools/exp/libanerf_gemmrows.so python tools/diag/concurrency_probe.py"""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
_lib = importlib.import_module("a-nerf_amd._lib")
ops = importlib.import_module("a-nerf_amd.ops")
ap = importlib.import_module("a-nerf_amd.autograd_path")
synth = importlib.import_module("a-nerf_amd.synth")
cfg = ops.PathConfig(framecode_ch=16)
cc, lib, dev = cfg.c(), _lib.load(), torch.device("cuda")
p = lambda t: C.c_void_p(t.data_ptr())
P_nets = {k: torch.tensor(v, device=dev) for k, v in synth.make_net_params(11, framecode_ch=16, n_codes=8).items()}
packed_t, aux_t = ops.pack_params(cfg, P_nets, which=1)
packed_i, _ = ops.pack_params(cfg, P_nets, which=2)
_, aux = ops.pack_params(cfg, P_nets, which=0)
px, pu = ap.perm_tables(cfg, dev)


class Pass:
    """buffers of one network pass over P points"""

    def __init__(self, P, seed):
        self.P = P
        T = _lib.AnerfTrainLayout()
        lib.anerf_train_layout(C.byref(cc), P, C.byref(T))
        self.T, pp = T, T.p_pad
        g = torch.Generator(device="cuda").manual_seed(seed)
        r = lambda *s: torch.randn(*s, device=dev, generator=g) * 0.1
        self.sv = {"h": r(8, pp, 256).abs(), "f": r(pp, 256), "g": r(pp, 128).abs(), "x": r(pp, T.x_width), "u": r(pp, T.u_width)}
        self.st = _lib.AnerfSaved(p(self.sv["h"]), p(self.sv["f"]), p(self.sv["g"]), p(self.sv["x"]), p(self.sv["u"]), pp)
        self.dz, self.df, self.dzv, self.draw = r(8, pp, 256), r(pp, 256), r(pp, 128), r(pp, 4)
        self.dx, self.du = torch.empty(pp, T.x_width, device=dev), torch.empty(pp, T.u_width, device=dev)
        shapes = [(256, 432), (256,)] + [(256, 256), (256,)] * 4 + [(256, 688), (256,)] + [(256, 256), (256,)] * 2 + \
                 [(1, 256), (1,), (256, 256), (256,), (128, 256 + T.u_width), (128,), (3, 128), (3,)]
        self.grads = [torch.empty(s, device=dev) for s in shapes]
        self.gs = _lib.AnerfNetGrads()
        for i in range(12):
            self.gs.w[i], self.gs.b[i] = self.grads[2 * i].data_ptr(), self.grads[2 * i + 1].data_ptr()
        self.ws = torch.empty(8 * T.gemm_ws_floats, device=dev)          # room for re-chunked partials

    def bwd(self, s):
        _lib.check(lib.anerf_mlp_backward(C.byref(cc), p(packed_t), p(aux), p(self.draw), C.byref(self.st), p(self.dz), p(self.df), p(self.dzv),
                                          self.P, C.c_void_p(s.cuda_stream)), "bwd")

    def bwd_in(self, s):
        _lib.check(lib.anerf_input_grads(C.byref(cc), p(packed_i), p(self.dz), p(self.dzv), self.T.p_pad, self.P, p(self.dx), p(self.du),
                                         C.c_void_p(s.cuda_stream)), "bwd_in")

    def gemm(self, s):
        _lib.check(lib.anerf_weight_grads(C.byref(cc), C.byref(self.st), p(self.dz), p(self.df), p(self.dzv), p(self.draw), self.P, p(px), p(pu),
                                          C.byref(self.gs), p(self.ws), 8 * self.T.gemm_ws_floats, C.c_void_p(s.cuda_stream)), "gemm")


def timed(fn, reps=30):
    ts = []
    for _ in range(reps + 3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(e0, e1)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts[3:]))


n = int(os.environ.get("PROBE_RAYS", 384))
fine, coarse = Pass(n * 80, 1), Pass(n * 64, 2)
hi = torch.cuda.Stream(priority=-1)
lo = torch.cuda.Stream(priority=0)


def serial(*calls):
    def f(e0, e1):
        with torch.cuda.stream(hi):
            e0.record(hi)
            for c in calls:
                c(hi)
            e1.record(hi)
    return f


def forked(tile_chain, gemm_chain, gemm_first=False):
    def f(e0, e1):
        e0.record(hi)
        lo.wait_event(e0)
        if gemm_first:
            for c in gemm_chain:
                c(lo)
        for c in tile_chain:
            c(hi)
        if not gemm_first:
            for c in gemm_chain:
                c(lo)
        hi.wait_stream(lo)
        e1.record(hi)
    return f


print(f"# {n} rays: fine pass {fine.P} points ({fine.P // 128} tiles), coarse pass {coarse.P} points ({coarse.P // 128} tiles); medians of 30, us")
rows_default = None
for rows in (None, 864, 432):           # rows per GEMM block: library's own plan, ~36 chunks, ~72 chunks of the fine pass
    if rows is None:
        os.environ.pop("ANERF_GEMM_ROWS", None)
    else:
        os.environ["ANERF_GEMM_ROWS"] = str(rows)
    tag = "library plan" if rows is None else f"ANERF_GEMM_ROWS={rows}"
    t = {"bwd_f": timed(serial(fine.bwd)), "bwd_c": timed(serial(coarse.bwd)), "in_f": timed(serial(fine.bwd_in)), "in_c": timed(serial(coarse.bwd_in)),
         "gemm_f": timed(serial(fine.gemm)), "gemm_c": timed(serial(coarse.gemm))}
    print(f"[{tag}] alone: " + ", ".join(f"{k} {v:.1f}" for k, v in t.items()))
    # pair 1: GEMM(fine) beside bwd_in(fine) -> bwd(coarse)
    s1 = timed(serial(fine.gemm, fine.bwd_in, coarse.bwd))
    f1 = timed(forked([fine.bwd_in, coarse.bwd], [fine.gemm]))
    f1g = timed(forked([fine.bwd_in, coarse.bwd], [fine.gemm], gemm_first=True))
    # pair 2: GEMM(coarse) beside bwd_in(coarse)
    s2 = timed(serial(coarse.gemm, coarse.bwd_in))
    f2 = timed(forked([coarse.bwd_in], [coarse.gemm]))
    # the whole backward's MFMA kernels: serial order of the library vs both GEMMs on the side stream
    s3 = timed(serial(fine.bwd, fine.gemm, fine.bwd_in, coarse.bwd, coarse.gemm, coarse.bwd_in))

    def whole(e0, e1):
        e0.record(hi)
        fine.bwd(hi)
        ev = torch.cuda.Event()
        ev.record(hi)
        lo.wait_event(ev)
        fine.bwd_in(hi)
        coarse.bwd(hi)
        fine.gemm(lo)
        ev2 = torch.cuda.Event()
        ev2.record(hi)
        lo.wait_event(ev2)
        coarse.bwd_in(hi)
        coarse.gemm(lo)
        hi.wait_stream(lo)
        e1.record(hi)
    f3 = timed(whole)
    print(f"[{tag}] gemm(f) || bwd_in(f)->bwd(c): serial {s1:.1f}, forked (tile chain enqueued first) {f1:.1f}  gain {s1 - f1:+.1f}; GEMM enqueued first {f1g:.1f}  gain {s1 - f1g:+.1f}")
    print(f"[{tag}] gemm(c) || bwd_in(c):          serial {s2:.1f}, forked {f2:.1f}  gain {s2 - f2:+.1f}")
    print(f"[{tag}] all six MFMA kernels of the backward: library order on one stream {s3:.1f}, GEMMs on the side stream {f3:.1f}  gain {s3 - f3:+.1f}")
