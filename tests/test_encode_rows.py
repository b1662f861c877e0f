"""Rows A4-A6 alone (world->bone transform, RelDist / VecNorm features, cutoff positional encodings): the network INPUT the
fused kernels produce, compared directly with the reference's encoded rows (tests/golden/encode_rows.npz, written by
tests/golden/gen_golden_encode.py from core/raycasters.py:476-577, core/encoders.py:8-193, core/cutoff_embedder.py:111-174).

The training forward saves exactly that tensor for the backward (`AnerfSaved.x` [P,432] and `.u` [P,648] in stream column
order); the perm tables map a saved column to the reference's column.  Gate: 2e-6 absolute per element for the fp32 kernel
(the reference's own fp32 rows sit within 3e-7 of a float64 evaluation of the same encoders; the kernel's double-angle sin /
cos recurrences are specified to 1.3e-6), observed maxima printed."""
import ctypes as C
import importlib

import numpy as np
import pytest
import torch

from cases import build

ops = importlib.import_module("a-nerf_amd.ops")
_lib = importlib.import_module("a-nerf_amd._lib")
ap = importlib.import_module("a-nerf_amd.autograd_path")

GATE_FP32 = 2e-6
GATE_B3 = 2e-6          # the split-bf16 kernel uses the same fp32 encode arithmetic: same rows


def test_oracle_encode_equals_the_reference_rows(oracle, golden):
    """CPU: the oracle's encode() against the reference's rows (pins the oracle's A4-A6 directly)."""
    g = golden("encode_rows")
    t = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32)
    cut = torch.full((24,), 0.5)
    c = build("eval_s32")
    rays = g["eval_rays"]
    z = t(g["eval_z"])[rays]
    ro, rd = t(c["rays_o"])[rays], t(c["rays_d"])[rays]
    pts = ro[:, None] + rd[:, None] * z[..., None]
    sk = t(c["skts"])
    X = oracle.encode(oracle.OracleConfig(), pts, rd / rd.norm(dim=-1, keepdim=True), sk[rays] if sk.shape[0] > 1 else sk, 20.0, 20.0,
                      cut, cut)
    np.testing.assert_allclose(X.numpy(), g["eval_X"], atol=1e-6)
    c = build("train_pytest")
    rays = g["train_rays"]
    ro, rd, sk = t(c["rays_o"])[rays], t(c["rays_d"])[rays], t(c["skts"])[rays]
    for tag in ("coarse", "fine"):
        z = t(g[f"train_z_{tag}"])[rays]
        pts = ro[:, None] + rd[:, None] * z[..., None]
        X = oracle.encode(oracle.OracleConfig(), pts, rd / rd.norm(dim=-1, keepdim=True), sk, 20.0, 20.0, cut, cut)
        np.testing.assert_allclose(X.numpy(), g[f"train_X_{tag}"], atol=1e-6, err_msg=tag)


def _saved_rows(cfg, P, rays_o, rays_d, skts, z, b3):
    """run the training forward kernel on (rays, z) and return its saved input planes un-permuted to the reference's column
    order: X [N,S,1080] = cat(v-PE 360, r 72, d-PE 648)"""
    dev = torch.device("cuda")
    d = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32, device=dev)
    rb = ops.make_ray_batch(d(rays_o), d(rays_d))
    zz = d(z)
    n, S = zz.shape
    T = ap.train_layout(cfg, n * S)
    pp = T.p_pad
    sv = {"h": torch.zeros(8, pp, 256, device=dev), "f": torch.zeros(pp, 256, device=dev), "g": torch.zeros(pp, 128, device=dev),
          "x": torch.full((pp, T.x_width), float("nan"), device=dev), "u": torch.full((pp, T.u_width), float("nan"), device=dev)}
    st = _lib.AnerfSaved(sv["h"].data_ptr(), sv["f"].data_ptr(), sv["g"].data_ptr(), sv["x"].data_ptr(), sv["u"].data_ptr(), pp)
    packed, aux = ops.pack_params(cfg, {k: d(v) for k, v in P.items()}, 3 if b3 else 0)
    raw = torch.empty(n, S, 4, device=dev)
    sk = d(skts)
    cut = torch.full((24,), 0.5, device=dev)
    cc = cfg.c()
    fn = _lib.load().anerf_mlp_raw_train_b3 if b3 else _lib.load().anerf_mlp_raw_train
    _lib.check(fn(C.byref(cc), packed.data_ptr(), aux.data_ptr(), rb.data_ptr(), rb.shape[1], zz.data_ptr(), sk.data_ptr(),
                  0 if sk.shape[0] == 1 else 384, None, None, 0, 20.0, 20.0, cut.data_ptr(), cut.data_ptr(), n, S, raw.data_ptr(),
                  C.byref(st), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "anerf_mlp_raw_train")
    torch.cuda.synchronize()
    px, pu = ap.perm_tables(cfg, dev, b3=b3)
    X = torch.empty(n * S, 1080, device=dev)
    X[:, px.long()] = sv["x"][:n * S]                  # saved column j holds the reference's column perm[j]
    X[:, 432 + pu.long()] = sv["u"][:n * S]
    assert sorted(px.tolist()) == list(range(432)) and sorted(pu.tolist()) == list(range(648))
    return X.view(n, S, 1080).cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("b3", [False, True], ids=["fp32", "bf16x3"])
def test_kernel_encoded_rows_equal_the_reference_rows(golden, b3):
    g = golden("encode_rows")
    cfg = ops.PathConfig()
    gate = GATE_B3 if b3 else GATE_FP32
    worst = {}
    c = build("eval_s32")
    X = _saved_rows(cfg, c["Pc"], c["rays_o"], c["rays_d"], c["skts"], g["eval_z"], b3)[g["eval_rays"]]
    worst["eval S=32, shared pose"] = np.abs(X - g["eval_X"])
    c = build("train_pytest")
    for tag, P in (("coarse", c["Pc"]), ("fine", c["Pf"])):
        X = _saved_rows(cfg, P, c["rays_o"], c["rays_d"], c["skts"], g[f"train_z_{tag}"], b3)[g["train_rays"]]
        worst[f"train 64+16 {tag}, per-ray poses"] = np.abs(X - g[f"train_X_{tag}"])
    for k, e in worst.items():
        parts = {"v-PE": e[..., :360].max(), "r": e[..., 360:432].max(), "d-PE": e[..., 432:].max()}
        print(f"encode rows [{'bf16x3' if b3 else 'fp32'}] {k}: max |kernel - reference| " +
              ", ".join(f"{n} {v:.2e}" for n, v in parts.items()))
    for k, e in worst.items():
        assert e.max() <= gate, (k, float(e.max()))
