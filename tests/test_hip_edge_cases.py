"""GPU: edge cases and configuration coverage of the HIP path vs the CPU oracle."""
import importlib

import numpy as np
import pytest
import torch

from cases import build

pytestmark = pytest.mark.gpu

ops = importlib.import_module("a-nerf_amd.ops")
pipeline = importlib.import_module("a-nerf_amd.pipeline")
synth_mod = importlib.import_module("a-nerf_amd.synth")


def dev(x):
    return torch.tensor(np.asarray(x), dtype=torch.float32, device="cuda")


def t(x):
    return torch.tensor(np.asarray(x), dtype=torch.float32)


def cuda_params(P):
    return {k: dev(v) for k, v in P.items()}


def both(oracle, c, n, S, Ni, cfg_kw=None, lindisp=False, seed_noise=None):
    """Run n rays of case c through HIP and oracle with the given sampling config."""
    cfg_kw = cfg_kw or {}
    cfg = ops.PathConfig(**cfg_kw)
    ocfg = oracle.OracleConfig(**cfg_kw)
    Pc, Pf = c["Pc"], c["Pf"]
    net_c, net_f = ops.pack_params(cfg, cuda_params(Pc)), ops.pack_params(cfg, cuda_params(Pf))
    ro, rd, skts, cyls = c["rays_o"][:n], c["rays_d"][:n], c["skts"][:n], c["cyls"][:n]
    rb = pipeline.make_ray_batch(dev(ro), dev(rd))
    kw, okw = {}, {}
    if seed_noise is not None:
        g = np.random.default_rng(seed_noise)
        tr, u = g.random((n, S)).astype(np.float32), g.random((n, max(Ni, 1))).astype(np.float32)
        nz, nzf = g.standard_normal((n, S)).astype(np.float32), g.standard_normal((n, S + Ni)).astype(np.float32)
        kw = dict(t_rand=dev(tr), u_imp=dev(u) if Ni else None, noise=dev(nz), noise_fine=dev(nzf) if Ni else None)
        okw = dict(t_rand=t(tr), u_imp=t(u) if Ni else None, noise=t(nz), noise_fine=t(nzf) if Ni else None)
    out = pipeline.render_rays_forward(cfg, net_c, net_f, rb, dev(skts), dev(cyls), S, Ni, lindisp=lindisp, extras=True, **kw)
    with torch.no_grad():
        ref = oracle.render_rays(ocfg, oracle.params_from_numpy(Pc), oracle.params_from_numpy(Pf),
                                 oracle.make_ray_batch(t(ro), t(rd)), t(skts), t(cyls), S, Ni, lindisp=lindisp,
                                 return_extras=True, **okw)
    return out, ref


def assert_out(out, ref, keys, atol=1e-4):
    for k in keys:
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k].numpy(), atol=atol, rtol=0, err_msg=k)


def test_empty_batch():
    cfg = ops.PathConfig()
    net = ops.pack_params(cfg, cuda_params(synth_mod.make_net_params(11)))
    rb = torch.zeros(0, 11, device="cuda")
    out = pipeline.render_rays_forward(cfg, net, net, rb, torch.zeros(0, 24, 4, 4, device="cuda"), torch.zeros(0, 5, device="cuda"), 16, 8)
    assert out["rgb_map"].shape == (0, 3) and out["alpha"].shape == (0, 24) and out["alpha0"].shape == (0, 16)


@pytest.mark.parametrize("n,S,Ni", [(1, 8, 0), (3, 8, 8), (5, 17, 9), (7, 33, 0), (2, 64, 128), (1, 256, 256)])
def test_ragged_and_extreme_sizes(oracle, n, S, Ni):
    """single ray, minimum / odd / maximum sample counts, per-ray poses straddling tiles, config-5 shape 64+128."""
    c = build("train_pytest")       # per-ray poses
    out, ref = both(oracle, c, n, S, Ni)
    keys = ["rgb_map", "acc_map", "alpha"] + (["rgb0", "alpha0"] if Ni else [])
    assert_out(out, ref, keys)
    assert out["alpha"].shape == (n, S + Ni)


def test_softplus_density_scale_lindisp_and_noise(oracle):
    c = build("eval_hier")
    out, ref = both(oracle, c, 24, 32, 16, cfg_kw=dict(density_scale=0.5, softplus_shift=1.0), lindisp=True, seed_noise=11)
    assert_out(out, ref, ["rgb_map", "acc_map", "alpha", "rgb0", "alpha0"])
    np.testing.assert_allclose(out["disp_map"].cpu().numpy(), ref["disp_map"].numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(out["_extras"]["z_vals"].cpu().numpy(), ref["_extras"]["z_vals"].numpy(), rtol=3e-6)


def test_nerf_forward_seam_with_framecodes(oracle):
    """NeRF.forward(x) (nerf.py:133-148) through the mirror module, frame-code column included."""
    networks = importlib.import_module("a-nerf_amd.networks")
    c = build("mixamo_train")
    net = networks.NeRF(D=8, W=256, input_ch=360, input_ch_bones=72, input_ch_views=648, use_viewdirs=True,
                        use_framecode=True, framecode_ch=16, n_framecodes=8)
    net.load_state_dict({k: t(v) for k, v in c["Pc"].items()})
    net = net.cuda().train()
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(5, 41, 1080, generator=g) - 0.5)
    idx = torch.randint(0, 8, (5, 41, 1), generator=g).float()
    X = torch.cat([x, idx], -1)
    with torch.no_grad():
        got = net(X.cuda())
        assert got.shape == (5, 41, 4)
        ref = oracle.mlp(oracle.OracleConfig(framecode_ch=16), oracle.params_from_numpy(c["Pc"]), X)
        np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), atol=3e-5)
        assert torch.equal(net.forward_batchify(X.cuda().reshape(-1, 1081), chunk=64), got.reshape(-1, 4))
        # eval with idx < 0 -> mean code
        net.eval()
        Xe = torch.cat([x, -torch.ones_like(idx)], -1)
        ref_e = oracle.mlp(oracle.OracleConfig(framecode_ch=16), oracle.params_from_numpy(c["Pc"]), Xe, eval_mean_code=True)
        np.testing.assert_allclose(net(Xe.cuda()).cpu().numpy(), ref_e.numpy(), atol=3e-5)


def test_raw2outputs_seam(oracle):
    """NeRF.raw2outputs (nerf.py:150-205) through the mirror, incl. the pytest noise override."""
    networks = importlib.import_module("a-nerf_amd.networks")
    net = networks.NeRF(D=8, W=256, input_ch=360, input_ch_bones=72, input_ch_views=648, use_viewdirs=True).cuda()
    g = torch.Generator().manual_seed(2)
    raw, rd = torch.randn(9, 40, 4, generator=g), torch.randn(9, 3, generator=g)
    z = torch.sort(torch.rand(9, 40, generator=g) * 2 + 1, -1)[0]
    out = net.raw2outputs(raw.cuda(), z.cuda(), rd.cuda(), raw_noise_std=1.0, pytest=True, B=1.0)
    np.random.seed(0)
    noise = t(np.random.rand(9, 40))
    ref = oracle.composite(oracle.OracleConfig(), raw, z, rd, noise)
    for k in ["rgb_map", "disp_map", "acc_map", "weights", "alpha"]:
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k].numpy(), atol=2e-5, rtol=1e-4, err_msg=k)


def test_chunking_invariance_and_determinism(synth):
    """batchify_rays chunking must not change a ray's result (tile alignment differs per chunk), and two identical
    training steps must give bitwise-identical gradients (fixed-order chunk reduction)."""
    render_mod = importlib.import_module("a-nerf_amd.render")
    tb = importlib.import_module("test_hip_backward")
    c = build("train_pytest")
    caster = tb.make_caster(c)
    caster.eval()
    kw = dict(rays=(dev(c["rays_o"]), dev(c["rays_d"])), use_viewdirs=True, ray_caster=caster, kp_batch=dev(c["kp"]),
              skts=dev(c["skts"]), cyls=dev(c["cyls"]), bones=dev(c["bones"]), cams=None, subject_idxs=None, N_samples=24,
              N_importance=8, preproc_kwargs={"density_scale": 1.0, "density_fn": torch.nn.functional.relu})
    a = render_mod.render(64, 64, 75.0, chunk=4096, **kw)
    b = render_mod.render(64, 64, 75.0, chunk=7, **kw)
    for k in ["rgb_map", "acc_map", "alpha", "rgb0"]:
        assert torch.equal(a[k], b[k]), k
    caster.train()
    grads = []
    for _ in range(2):
        caster.zero_grad()
        out = render_mod.render(64, 64, 75.0, chunk=4096, perturb=1.0, raw_noise_std=1.0, pytest=True, **kw)
        render_mod.nerf_loss(out, torch.full((c["n"], 3), 0.3, device="cuda"))[0].backward()
        grads.append([p.grad.clone() for p in caster.parameters() if p.grad is not None])
    assert len(grads[0]) == 48
    for g0, g1 in zip(*grads):
        assert torch.equal(g0, g1)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_cutoff_gate_at_the_end_of_the_tau_schedule(oracle, precision):
    """tau reaches 2000 at the end of training (cutoff_embedder.py:192-197): tau * (dist - cutoff) is then in the thousands,
    exp overflows for every joint beyond its cutoff and the gate is 0 or 1 almost everywhere.  The kernels evaluate the gate as
    rcp(1 + exp(min(a, 80))) with a Newton step: no inf * 0, finite outputs, same maps as the oracle's 1 - sigmoid."""
    c = build("eval_hier")
    cfg, ocfg = ops.PathConfig(), oracle.OracleConfig()
    Pc, Pf = c["Pc"], c["Pf"]
    which = 3 if precision == "bf16x3" else 0
    net_c, net_f = ops.pack_params(cfg, cuda_params(Pc), which), ops.pack_params(cfg, cuda_params(Pf), which)
    n, S, Ni = 24, 32, 16
    ro, rd, skts, cyls = c["rays_o"][:n], c["rays_d"][:n], c["skts"][:n], c["cyls"][:n]
    out = pipeline.render_rays_forward(cfg, net_c, net_f, pipeline.make_ray_batch(dev(ro), dev(rd)), dev(skts), dev(cyls), S, Ni,
                                       tau_v=2000.0, tau_d=2000.0, precision=precision)
    with torch.no_grad():
        ref = oracle.render_rays(ocfg, oracle.params_from_numpy(Pc), oracle.params_from_numpy(Pf),
                                 oracle.make_ray_batch(t(ro), t(rd)), t(skts), t(cyls), S, Ni, tau_v=2000.0, tau_d=2000.0)
    for k in ("rgb_map", "acc_map", "alpha", "rgb0", "alpha0"):
        assert torch.isfinite(out[k]).all(), k
    assert_out(out, ref, ["rgb_map", "acc_map", "rgb0"], atol=2e-4)


@pytest.mark.parametrize("lindisp", [False, True])
@pytest.mark.parametrize("miss", ["some", "all", "none"])
def test_one_launch_bounds_and_depths_equal_the_staged_pair(oracle, miss, lindisp):
    """Round 6: inside the one-call entry points A2 + A3 are ONE launch (k_bounds_z: each sample's thread computes its ray's bounds;
    a block that holds a ray missing the cylinder recomputes the call-wide NaN-mean statistics itself, in k_ray_bounds' exact form).
    Against the staged k_ray_bounds -> k_coarse_z route: every output BIT-equal -- rays that miss (the NaN-mean fallback,
    ray_utils.py:327-342), a call in which ALL rays miss (no valid row: the placeholder bounds 0 / 1 stay), stratified jitter,
    lindisp, per-ray and shared cylinders, a block that straddles rays (S = 24 does not divide 256)."""
    c = build("nan_fallback")
    cfg = ops.PathConfig()
    net = ops.pack_params(cfg, cuda_params(c["Pc"]))
    n, S = c["n"], 24
    rb = pipeline.make_ray_batch(dev(c["rays_o"]), dev(c["rays_d"]))
    cyl = dev(c["cyls"]).clone()
    if miss == "all":
        cyl[:, 0] += 100.0                    # the cylinder is nowhere near any ray
    elif miss == "none":
        cyl[:, 2] *= 10.0
    skt = dev(c["skts"])
    tr = torch.rand(n, S, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    nf_raw, _ = ops.ray_bounds(rb, cyl)
    n_nan = int(torch.isnan(nf_raw[:, 0]).sum())
    assert {"some": 0 < n_nan < n, "all": n_nan == n, "none": n_nan == 0}[miss], n_nan
    for t_rand in (None, tr):
        staged = pipeline.render_rays_forward(cfg, net, None, rb, skt, cyl, S, 0, t_rand=t_rand, lindisp=lindisp, extras=True)
        fused = ops.forward(cfg, net, None, rb, skt, cyl, S, 0, t_rand=t_rand, lindisp=lindisp)
        shared = ops.forward(cfg, net, None, rb, skt, cyl[:1], S, 0, t_rand=t_rand, lindisp=lindisp)
        bits = lambda x: x.contiguous().view(torch.int32)        # (lindisp with the placeholder near = 0 divides by zero, in the
        for k in fused:                                          # reference too: NaN depths must be the SAME NaNs)
            assert torch.equal(bits(fused[k]), bits(staged[k])) and torch.equal(bits(shared[k]), bits(staged[k])), (k, miss, t_rand is not None)
        if not lindisp:
            assert torch.isfinite(fused["rgb_map"]).all()


def test_small_batch_loss_is_one_launch_with_the_same_bits():
    """anerf_loss for n <= 1024 rays runs k_loss_one (one block walking k_loss's blocks in turn, k_loss_final's sums behind them):
    the same four outputs and gradient maps, bit for bit, as the two-launch route takes for the same rays -- checked by evaluating
    each size both ways: directly (one launch) and as the leading rows of a padded 1025+-ray call whose extra rows contribute
    exact zeros (target == prediction, background 0) to every sum."""
    g = torch.Generator(device="cuda").manual_seed(11)
    for n in (1, 255, 256, 257, 384, 1024):
        for kind in (0, 1, 2):
            rgb, rgb0, tgt = (torch.rand(n, 3, device="cuda", generator=g) for _ in range(3))
            acc, acc0 = torch.rand(n, device="cuda", generator=g), torch.rand(n, device="cuda", generator=g)
            bg = torch.rand(n, 3, device="cuda", generator=g)
            out, gr = ops.loss(rgb, acc, tgt, rgb0, acc0, bg, kind, 0.7)
            # reference arithmetic of the same sums in float64, then the scale 1 / (3 n): agreement to fp32 rounding
            d = (rgb + (1 - acc)[:, None] * bg - tgt).double()
            d0 = (rgb0 + (1 - acc0)[:, None] * bg - tgt).double()
            term = (lambda x: x * x) if kind == 0 else (lambda x: x.abs()) if kind == 1 else \
                (lambda x: torch.where(x.abs() >= 0.1, x.abs() - 0.05, 0.5 * x * x / 0.1))
            want = torch.stack([term(d).mean() + 0.7 * term(d0).mean(), term(d).mean(), term(d0).mean(), (d * d).mean()])
            assert torch.allclose(out.double(), want, rtol=2e-6, atol=1e-9), (n, kind, out, want)
            assert gr["rgb"].shape == (n, 3) and torch.isfinite(gr["flat"]).all()
    # the same rays through both routes: n = 1024 (one launch) vs the first 1024 rows of a 1280-ray call (two launches) whose
    # other rows add exact zeros; 1 / (3 n) differs between the calls, so compare sums: out * 3n
    n, m = 1024, 1280
    rgb, rgb0, tgt = (torch.rand(m, 3, device="cuda", generator=g) for _ in range(3))
    acc, acc0 = torch.rand(m, device="cuda", generator=g), torch.rand(m, device="cuda", generator=g)
    rgb[n:], rgb0[n:] = tgt[n:], tgt[n:]
    bg = torch.zeros(3, device="cuda")
    one, _ = ops.loss(rgb[:n].contiguous(), acc[:n].contiguous(), tgt[:n].contiguous(), rgb0[:n].contiguous(), acc0[:n].contiguous(), bg, 0, 1.0)
    two, _ = ops.loss(rgb, acc, tgt, rgb0, acc0, bg, 0, 1.0)
    assert torch.allclose(one[1:] * (3 * n), two[1:] * (3 * m), rtol=3e-7, atol=0)
