"""Training path: the same kernel pipeline as pipeline.py, bridged into torch.autograd so that the reference's
trainer (`loss.backward()` -> Adam, core/trainer.py:451-483) drives it unchanged.

Two-network configurations run as ONE autograd node, one C call each way:
  _RenderRaysFn parameters (+ skts, frame codes) -> the output dict      fwd: anerf_train_forward   bwd: anerf_backward
single_net (one shared network evaluated twice + a gather-merge of the raw outputs) composes the staged nodes, which wrap
HIP launches only (no torch math inside):
  _MlpRawFn     parameters -> raw [N,S,4]   fwd: k_mlp_fwd<TRAIN> (saves activations)
                                            bwd: k_mlp_bwd + grouped k_gemm_tn/k_reduce_dw (weight gradients)
  _CompositeFn  raw -> rgb/disp/acc/alpha/weights      fwd: k_composite   bwd: k_composite_bwd
Both routes enqueue the same kernels in the same order and give bit-identical results (tests/test_hip_backward.py).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, ops
from .ops import _p, _stream, PARAM_ORDER

_perm_cache = {}


def train_layout(cfg, n_points):
    T = _lib.AnerfTrainLayout()
    cc = cfg.c()
    _lib.check(_lib.load().anerf_train_layout(C.byref(cc), n_points, C.byref(T)), "anerf_train_layout")
    return T


def perm_tables(cfg, device, b3=False):
    """saved-plane column -> torch input column, for the fp32 training forward or (b3) the split-bf16 one"""
    k = (cfg.key(), str(device), bool(b3))
    if k not in _perm_cache:
        T = train_layout(cfg, 128)
        px, pu = np.empty(T.x_width, np.int32), np.empty(T.u_width, np.int32)
        cc = cfg.c()
        fn = _lib.load().anerf_build_perm_tables_b3 if b3 else _lib.load().anerf_build_perm_tables
        _lib.check(fn(C.byref(cc), px.ctypes.data_as(C.c_void_p), pu.ctypes.data_as(C.c_void_p)), "anerf_build_perm_tables")
        _perm_cache[k] = (torch.from_numpy(px).to(device), torch.from_numpy(pu).to(device))
    return _perm_cache[k]


def _planes(rows_pad, rows, width, n_planes, device):
    """[n_planes][rows_pad][width] buffer whose pad rows are zero (the GEMM reads them)."""
    buf = torch.empty(n_planes, rows_pad, width, dtype=torch.float32, device=device)
    if rows_pad > rows:
        buf[:, rows:].zero_()
    return buf


class _MlpRawFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, meta, skts_in, codes_in, *params):
        cfg, dev = meta["cfg"], meta["rays"].device
        z = meta["z"]
        n, s = z.shape
        P = n * s
        T = train_layout(cfg, P)
        pp = T.p_pad
        sv = {"h": _planes(pp, P, 256, 8, dev), "f": _planes(pp, P, 256, 1, dev), "g": _planes(pp, P, 128, 1, dev),
              "x": _planes(pp, P, T.x_width, 1, dev), "u": _planes(pp, P, T.u_width, 1, dev)}
        st = _lib.AnerfSaved(_p(sv["h"]), _p(sv["f"]), _p(sv["g"]), _p(sv["x"]), _p(sv["u"]), pp)
        raw = torch.empty(n, s, 4, dtype=torch.float32, device=dev)
        skts = meta["skts"]
        stride = 0 if skts.shape[0] == 1 else 16 * cfg.n_joints
        codes = meta.get("codes")
        cc = cfg.c()
        b3 = meta.get("precision", "fp32") == "bf16x3"
        packed, aux = meta["packed_b3"] if b3 else meta["packed"]
        fwd = _lib.load().anerf_mlp_raw_train_b3 if b3 else _lib.load().anerf_mlp_raw_train
        _lib.check(fwd(
            C.byref(cc), _p(packed), _p(aux), _p(meta["rays"]), meta["rays"].shape[1], _p(z), _p(skts), stride,
            _p(meta.get("cam")), _p(codes), 0 if codes is None else codes.shape[0], float(meta["tau_v"]),
            float(meta["tau_d"]), _p(meta["cut_v"]), _p(meta["cut_d"]), n, s, _p(raw), C.byref(st), _stream()),
            "anerf_mlp_raw_train")
        ctx.meta, ctx.sv, ctx.T, ctx.P = meta, sv, T, P
        ctx.shapes = [tuple(p.shape) for p in params]
        return raw

    @staticmethod
    def backward(ctx, g_raw):
        meta, sv, T, P = ctx.meta, ctx.sv, ctx.T, ctx.P
        cfg, dev, pp = meta["cfg"], g_raw.device, T.p_pad
        draw = torch.zeros(pp, 4, dtype=torch.float32, device=dev)
        draw[:P] = g_raw.reshape(P, 4)
        dz = _planes(pp, P, 256, 8, dev)
        df = _planes(pp, P, 256, 1, dev)
        dzv = _planes(pp, P, 128, 1, dev)
        st = _lib.AnerfSaved(_p(sv["h"]), _p(sv["f"]), _p(sv["g"]), _p(sv["x"]), _p(sv["u"]), pp)
        cc = cfg.c()
        b3 = meta.get("precision", "fp32") == "bf16x3"
        packed_t, _ = meta["packed_t"]
        _, aux = meta["packed_b3"] if b3 else meta["packed"]
        lib = _lib.load()
        _lib.check((lib.anerf_mlp_backward_b3 if b3 else lib.anerf_mlp_backward)(C.byref(cc), _p(packed_t), _p(aux), _p(draw), C.byref(st), _p(dz), _p(df), _p(dzv),
                                          P, _stream()), "anerf_mlp_backward")
        grads = [torch.empty(sh, dtype=torch.float32, device=dev) for sh in ctx.shapes]
        gs = _lib.AnerfNetGrads()
        for i in range(12):
            gs.w[i] = grads[2 * i].data_ptr()
            gs.b[i] = grads[2 * i + 1].data_ptr()
        if meta.get("sched") is not None:
            meta["sched"].fill(gs)
        ws = torch.empty(T.gemm_ws_floats, dtype=torch.float32, device=dev)
        px, pu = perm_tables(cfg, dev, b3=b3)
        _lib.check((lib.anerf_weight_grads_b3 if b3 else lib.anerf_weight_grads)(C.byref(cc), C.byref(st), _p(dz), _p(df), _p(dzv), _p(draw), P, _p(px), _p(pu),
                                          C.byref(gs), _p(ws), T.gemm_ws_floats, _stream()), "anerf_weight_grads")
        g_skts = g_codes = None
        need_skts, need_codes = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        if need_skts or need_codes:
            packed_i, _ = meta["packed_i"]()
            dx = torch.empty(pp, T.x_width, dtype=torch.float32, device=dev)
            du = torch.empty(pp, T.u_width, dtype=torch.float32, device=dev)
            _lib.check((lib.anerf_input_grads_b3 if b3 else lib.anerf_input_grads)(
                C.byref(cc), _p(packed_i), _p(dz), _p(dzv), pp, P, _p(dx), _p(du), _stream()), "anerf_input_grads")
            z, rays, skts = meta["z"], meta["rays"], meta["skts"]
            n, s = z.shape
            if need_skts:
                if skts.shape[0] != n:
                    raise NotImplementedError("skts.requires_grad needs per-ray skts [N,24,4,4]")
                g_skts = torch.empty_like(skts)          # every element is written (row 3 as zeros) by k_pose_reduce
                dyw = torch.empty(P, 72, dtype=torch.float32, device=dev)
                dqw = torch.empty(P, 72, dtype=torch.float32, device=dev)
                _lib.check(lib.anerf_encode_backward(
                    C.byref(cc), _p(dx), _p(du), _p(rays), rays.shape[1], _p(z), _p(skts), 16 * cfg.n_joints,
                    float(meta["tau_v"]), float(meta["tau_d"]), _p(meta["cut_v"]), _p(meta["cut_d"]), n, s, _p(dyw), _p(dqw),
                    _p(g_skts), _stream()), "anerf_encode_backward")
            if need_codes:
                codes = meta["codes"]
                g_codes = torch.zeros_like(codes)
                rowsum = torch.empty(n, 16, dtype=torch.float32, device=dev)
                _lib.check(lib.anerf_code_grads(C.byref(cc), _p(du), _p(meta["cam"]), n, s, _p(g_codes), codes.shape[0],
                                                _p(rowsum), _stream()), "anerf_code_grads")
        ctx.sv = None
        return (None, g_skts, g_codes, *grads)


class _CompositeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, meta, raw):
        out = ops.composite(meta["cfg"], raw, meta["z"], meta["rays"], meta.get("noise"))
        ctx.meta = meta
        ctx.save_for_backward(raw)
        ctx.set_materialize_grads(False)
        return out["rgb_map"], out["disp_map"], out["acc_map"], out["alpha"], out["weights"]

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, g_alpha, g_w):
        meta = ctx.meta
        (raw,) = ctx.saved_tensors
        z, rays = meta["z"], meta["rays"]
        n, s = z.shape
        if g_rgb is None:
            g_rgb = torch.zeros(n, 3, dtype=torch.float32, device=raw.device)
        c = lambda t: None if t is None else t.contiguous()
        draw = torch.empty(n, s, 4, dtype=torch.float32, device=raw.device)
        cc = meta["cfg"].c()
        _lib.check(_lib.load().anerf_composite_backward(
            C.byref(cc), _p(raw), _p(z), _p(rays), rays.shape[1], _p(meta.get("noise")), n, s, _p(c(g_rgb)), _p(c(g_acc)),
            _p(c(g_disp)), _p(c(g_alpha)), _p(c(g_w)), _p(draw), _stream()), "anerf_composite_backward")
        return None, draw


_OUT_KEYS = ("rgb_map", "disp_map", "acc_map", "alpha", "rgb0", "disp0", "acc0", "alpha0")


class _RenderRaysFn(torch.autograd.Function):
    """RayCaster.render_rays (raycasters.py:361-474) as one autograd node over anerf_train_forward / anerf_backward."""

    @staticmethod
    def forward(ctx, meta, skts, codes_c, codes_f, *params):
        kw = meta["kw"]
        out, state = ops.train_forward(
            kw["cfg"], meta["net_c"], meta["net_f"], kw["ray_batch"], skts, kw["cyls"], kw["n_samples"], kw["n_importance"],
            kw["tau_v"], kw["tau_d"], kw["cut_v"], kw["cut_d"], meta["cam"], codes_c, codes_f, kw["t_rand"], kw["u_imp"],
            kw["noise"], kw["noise_fine"], kw["lindisp"], meta["precision"], kw.get("pts_noise"), kw.get("pts_noise_is"))
        ctx.state, ctx.meta = state, meta
        ctx.keys = _OUT_KEYS if kw["n_importance"] > 0 else _OUT_KEYS[:4]
        ctx.shapes = [tuple(p.shape) for p in params]
        ctx.set_materialize_grads(False)
        return tuple(out[k] for k in ctx.keys)

    @staticmethod
    def backward(ctx, *gs):
        meta, state = ctx.meta, ctx.state
        if state is None:
            raise RuntimeError("render_rays: backward called twice (the saved activations are released after the first)")
        b3 = meta["precision"] == "bf16x3"
        hier = len(ctx.keys) == 8
        want_skts, want_cc, want_cf = ctx.needs_input_grad[1], ctx.needs_input_grad[2], ctx.needs_input_grad[3] and hier
        if want_skts and state["io"].skt_ray_stride == 0:
            raise NotImplementedError("skts.requires_grad needs per-ray skts [N,24,4,4]")
        want_in = want_skts or want_cc or want_cf
        pi = meta["packed_i"]() if want_in else (None, None)
        # In-place gradient accumulation is an explicit opt-in: `FusedAdam.attach(caster)` registers the optimiser as the
        # caster's gradient sink.  Only then -- and only while every parameter's .grad really is a view into that optimiser's
        # flat bucket -- does the reduction kernel add into the bucket directly (what AccumulateGrad would otherwise do with
        # 48 extra launches per step) and this node reports no parameter gradients to autograd.  Everything else (torch
        # optimisers, an un-attached FusedAdam, torch.autograd.grad, hooks, DDP-style reducers) gets fresh gradient tensors
        # through autograd as usual.
        dev = state["ws"].device
        sink = meta.get("grad_sink")
        into = [p.grad for p in meta["params"]]
        direct = sink is not None and sink.owns_grads(meta["params"], dev)
        # frame codes ride along: when the tables handed to the kernels ARE the embedding weights (training mode) and their .grad
        # lives in the same bucket, the code gradients are added in place as well (no zero fill, no AccumulateGrad add)
        codes_into = None
        if direct and (want_cc or want_cf):
            cp = meta.get("code_params", (None, None))
            ok = all(p is None or (p.grad is not None and sink.owns_grads([p], dev)) for p in cp) and \
                (not want_cc or cp[0] is not None) and (not want_cf or cp[1] is not None)
            if ok:
                codes_into = (cp[0].grad if cp[0] is not None else None, cp[1].grad if cp[1] is not None else None)
        # data-parallel overlap (opt-in on the attached optimiser): the fine network's path gradients are complete when the
        # first half of the backward is enqueued; their all-reduce starts there and runs under the coarse pass
        hook = hook_c = hook_w = None
        if direct and getattr(sink, "overlap", False):
            sink.check_one_backward()        # raises on a second backward of the same step (nothing is enqueued yet)
        if direct and hier and getattr(sink, "overlap", False):
            # each network's bucket range = its 24 path tensors + its frame-code table right behind them (NeRF registers
            # `framecodes` last); a table whose gradient is not added in place stays out (it reaches the bucket later, through autograd)
            cp = meta.get("code_params", (None, None))
            fine_params = meta["params"][24:] + ([cp[1]] if (codes_into is not None and want_cf) else [])
            coarse_params = meta["params"][:24] + ([cp[0]] if (codes_into is not None and want_cc) else [])
            hook = lambda: sink.begin_async_all_reduce(fine_params)
            # ... and the coarse network's, once ITS parameter gradients are enqueued: under the pose-gradient tail of the coarse
            # pass and the pose layer's backward when there is one (Mixamo-type configurations), else at the end of the pass
            hook_c = lambda: sink.begin_async_all_reduce(coarse_params)
            # round 6: with input gradients in the step (pose refinement / frame codes) the coarse network's WEIGHTS -- 99.9 % of its
            # bytes -- are final behind the GEMM, a whole input-gradient kernel (180 us at the 384-ray shard) before the frame codes:
            # their all-reduce starts there (AnerfBackwardIO.passes = 16), and hook_c is left with the frame-code table alone
            if want_in:
                hook_w = lambda: sink.begin_async_all_reduce(meta["params"][:24])
                hook_c = (lambda: sink.begin_async_all_reduce([cp[0]])) if (codes_into is not None and want_cc) else None
        grads_c, grads_f, g_skts, g_cc, g_cf = ops.backward(
            state, dict(zip(ctx.keys, gs)), meta["packed_t_c"], meta["packed_t_f"], perm_tables(meta["kw"]["cfg"], dev, b3=b3),
            ctx.shapes[:24], ctx.shapes[24:], pi[0], pi[1], want_skts, want_cc, want_cf,
            accumulate_into=(into[:24], into[24:]) if direct else None, after_fine=hook, codes_into=codes_into,
            sched=meta.get("sched"), after_coarse_params=hook_c, after_coarse_weights=hook_w)
        ctx.state = None
        if direct:
            if codes_into is not None:
                g_cc = g_cf = None
            return (None, g_skts, g_cc, g_cf) + (None,) * len(ctx.shapes)
        if not hier:
            grads_f = [None] * (len(ctx.shapes) - 24)
        return (None, g_skts, g_cc, g_cf, *grads_c, *grads_f)


def _render_rays_one_node(caster, kw, prec):
    net_c, net_f = caster.network, caster.network_fine
    hier = kw["n_importance"] > 0
    b3 = prec == "bf16x3"
    cam = kw["cam_idx"].contiguous() if net_c.use_framecode else None
    codes_c = _codes_with_grad(net_c)
    codes_f = _codes_with_grad(net_f) if hier else None
    nets = [net_c] + ([net_f] if hier else [])
    from .networks import prepack
    want_in = kw["skts"].requires_grad or any(n.use_framecode and n.framecodes.codes.weight.requires_grad for n in nets)
    prepack([(n, w) for n in nets for w in ((3, 4) if b3 else (0, 1)) + (((5,) if b3 else (2,)) if want_in else ())])
    meta = dict(kw=kw, precision=prec, cam=cam,
                net_c=net_c.packed(3 if b3 else 0), net_f=net_f.packed(3 if b3 else 0) if hier else None,
                packed_t_c=net_c.packed(4 if b3 else 1)[0], packed_t_f=net_f.packed(4 if b3 else 1)[0] if hier else None,
                packed_i=lambda: tuple(n.packed(5 if b3 else 2)[0] for n in nets) + ((None,) if not hier else ()))
    params = [p for n in nets for p in _net_params(n)]
    meta["params"] = params
    meta["sched"] = net_c.input_schedule()     # what the images above were packed with (set by RayCaster.render_rays)
    # the embedding weights behind codes_c / codes_f when the kernels index them directly (training mode), else None
    code_w = lambda n: n.framecodes.codes.weight if (n is not None and n.use_framecode and n.training) else None
    meta["code_params"] = (code_w(net_c), code_w(net_f) if hier else None)
    sink = getattr(caster, "_anerf_grad_sink", None)
    meta["grad_sink"] = sink() if sink is not None else None        # weakref to an attached FusedAdam, or nothing
    out = _RenderRaysFn.apply(meta, kw["skts"].contiguous(), codes_c, codes_f, *params)
    return dict(zip(_OUT_KEYS, out))


def _codes_with_grad(net):
    """The frame-code table the kernels index, as an autograd input.  Training: the embedding weight.  Eval-mode rendering with
    gradients enabled: RayCaster.render_rays has pointed negative camera indices at row n_codes (Optcodes.table_for; the
    reference's "mean code" rule, embedding.py:21-22), so the table gets that row here too -- built with torch ops, the gradient
    of the mean code flows back into every code through the cat / mean."""
    if not net.use_framecode:
        return None
    w = net.framecodes.codes.weight
    return w if net.training else torch.cat([w, w.mean(0, keepdim=True)], 0)


def _net_params(net):
    P = net.named_path_params()
    out = []
    for n in PARAM_ORDER:
        out += [P[n + ".weight"], P[n + ".bias"]]
    return out


def render_rays_train(caster, kw):
    """Differentiable RayCaster.render_rays (raycasters.py:361-474); kw as assembled by RayCaster.render_rays."""
    cfg, rays, skts, cyls = kw["cfg"], kw["ray_batch"], kw["skts"], kw["cyls"]
    S, Ni = kw["n_samples"], kw["n_importance"]
    net_c, net_f = caster.network, caster.network_fine
    prec = getattr(caster, "train_precision", "fp32")
    if prec not in ("fp32", "bf16x3"):
        raise ValueError(f"train_precision must be 'fp32' or 'bf16x3', got {prec!r}")
    if not kw["single_net"] and getattr(caster, "train_route", "one_call") == "one_call":
        return _render_rays_one_node(caster, kw, prec)
    if kw.get("pts_noise") is not None:
        raise NotImplementedError("ray_noise_std > 0 trains through the one-call route (two-network configurations)")
    skts_c = skts.contiguous()
    with torch.no_grad():
        nf_raw, stats = ops.ray_bounds(rays, cyls)
        z, _ = ops.coarse_z(nf_raw, stats, rays, S, kw["t_rand"], kw["lindisp"])

    def mlp(net, zz):
        codes = _codes_with_grad(net)
        cam = kw["cam_idx"].contiguous() if net.use_framecode else None
        meta = dict(cfg=cfg, rays=rays, z=zz, skts=skts_c.detach(), tau_v=kw["tau_v"], tau_d=kw["tau_d"],
                    cut_v=kw["cut_v"], cut_d=kw["cut_d"], cam=cam, codes=None if codes is None else codes.detach(),
                    packed=net.packed(0) if prec == "fp32" else None, packed_t=net.packed(4 if prec == "bf16x3" else 1), packed_i=lambda: net.packed(5 if prec == "bf16x3" else 2),
                    precision=prec, packed_b3=net.packed(3) if prec == "bf16x3" else None, sched=net.input_schedule())
        return _MlpRawFn.apply(meta, skts_c, codes, *_net_params(net))

    def comp(raw, zz, noise):
        return _CompositeFn.apply(dict(cfg=cfg, rays=rays, z=zz, noise=noise), raw)

    raw = mlp(net_c, z)
    rgb, disp, acc, alpha, w = comp(raw, z, kw["noise"])
    ret = {"rgb_map": rgb, "disp_map": disp, "acc_map": acc, "alpha": alpha}
    if Ni > 0:
        with torch.no_grad():
            zs, zm, idx = ops.importance(z, w.detach(), Ni, kw["u_imp"], kw["single_net"], want_idx=True)
        if kw["single_net"]:
            raw_is = mlp(net_f, zs)
            raw_f = torch.gather(torch.cat([raw, raw_is], 1), 1, idx[..., None].expand(-1, -1, 4)).contiguous()
        else:
            raw_f = mlp(net_f, zm)
        rgb_f, disp_f, acc_f, alpha_f, _ = comp(raw_f, zm, kw["noise_fine"])
        ret = {"rgb_map": rgb_f, "disp_map": disp_f, "acc_map": acc_f, "alpha": alpha_f,
               "rgb0": rgb, "disp0": disp, "acc0": acc, "alpha0": alpha}
    return ret
