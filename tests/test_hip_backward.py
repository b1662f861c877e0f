"""GPU: training path (forward with saved activations + HIP backward) through the reference-shaped API
(create-free: RayCaster/NeRF mirrors + render()), vs the reference's golden gradients and the oracle's autograd."""
import importlib

import numpy as np
import pytest
import torch

from cases import build

pytestmark = pytest.mark.gpu

ops = importlib.import_module("a-nerf_amd.ops")
networks = importlib.import_module("a-nerf_amd.networks")
raycaster = importlib.import_module("a-nerf_amd.raycaster")
render_mod = importlib.import_module("a-nerf_amd.render")


def dev(x):
    return torch.tensor(np.asarray(x), dtype=torch.float32, device="cuda")


def t(x):
    return torch.tensor(np.asarray(x), dtype=torch.float32)


def make_caster(c):
    mv = c["cfg"].get("multires_views", 4)
    fc = c["cfg"].get("framecode_ch", 0)
    kw = dict(D=8, W=256, input_ch=360, input_ch_bones=72, input_ch_views=72 * (1 + 2 * mv), use_viewdirs=True,
              use_framecode=fc > 0, framecode_ch=16, n_framecodes=c.get("n_codes", 0))
    net_c, net_f = networks.NeRF(**kw), networks.NeRF(**kw)
    net_c.load_state_dict({k: t(v) for k, v in c["Pc"].items()})
    net_f.load_state_dict({k: t(v) for k, v in c["Pf"].items()})
    ck = {"cutoff": True, "cutoff_dist": 0.5, "cutoff_inputs": True, "cutoff_dim": 24}
    e_v, _ = networks.get_embedder(7, input_dims=24, cutoff_kwargs=dict(ck, dist_inputs=False))
    e_b, _ = networks.get_embedder(0, input_dims=72, cutoff_kwargs={"cutoff": False})
    e_d, _ = networks.get_embedder(mv, input_dims=72, cutoff_kwargs=dict(ck, dist_inputs=True))
    return raycaster.RayCaster(net_c, e_v, e_b, e_d, network_fine=net_f).cuda()


GRAD_BAR = 5e-4        # fp32 kernels: max |got - ref| over a tensor, relative to the tensor's largest element (observed ~1e-4)
NORM_BAR = 5e-4        # fp32 kernels: relative difference of a gradient tensor's norm


def grad_err(got, ref):
    """max element error of a gradient tensor relative to the tensor's largest reference element"""
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    ref = ref.detach().cpu().numpy() if torch.is_tensor(ref) else np.asarray(ref)
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))


def test_train_step_gradients_vs_golden_and_oracle(oracle, golden):
    g = golden("train_pytest")
    c = build("train_pytest")
    caster = make_caster(c)
    caster.train()
    n = c["n"]
    out = render_mod.render(64, 64, 75.0, chunk=4096, rays=(dev(c["rays_o"]), dev(c["rays_d"])), use_viewdirs=True,
                            ray_caster=caster, kp_batch=dev(c["kp"]), skts=dev(c["skts"]), cyls=dev(c["cyls"]),
                            bones=dev(c["bones"]), cams=None, subject_idxs=None, N_samples=64, N_importance=16,
                            perturb=1.0, raw_noise_std=1.0, pytest=True,
                            preproc_kwargs={"density_scale": 1.0, "density_fn": torch.nn.functional.relu})
    for k in ["rgb_map", "acc_map", "alpha", "rgb0", "alpha0"]:
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), g[k], atol=1e-4, err_msg=k)
    target = dev(np.random.default_rng(1).random((n, 3)))
    loss, _ = render_mod.nerf_loss(out, target, bgs=torch.ones(n, 3, device="cuda"))
    assert abs(float(loss.detach()) - float(g["loss"])) < 2e-6
    loss.backward()
    # reference golden: per-tensor norms, leading slices, a few full tensors
    seen = {"norm": 0.0, "slice": 0.0, "full": 0.0, "oracle": 0.0}
    for tag, net in [("c", caster.network), ("f", caster.network_fine)]:
        for name, p in net.named_parameters():
            ref_n = float(g[f"gnorm_{tag}.{name}"])
            got_n = float(p.grad.norm())
            seen["norm"] = max(seen["norm"], abs(got_n - ref_n) / (ref_n + 1e-30))
            assert abs(got_n - ref_n) <= NORM_BAR * ref_n + 1e-9, (tag, name, got_n, ref_n)
            # leading 64 elements, against the tensor's RMS element (the slice may miss the tensor's large entries)
            rms = ref_n / max(np.sqrt(p.numel()), 1.0)
            e = float(np.abs(p.grad.reshape(-1)[:64].cpu().numpy() - g[f"gslice_{tag}.{name}"]).max() / (rms + 1e-30))
            seen["slice"] = max(seen["slice"], e)
            assert e <= 2e-3, (tag, name, e)
        for name in ["pts_linears.5.bias", "rgb_linear.weight", "alpha_linear.weight"]:
            e = grad_err(dict(net.named_parameters())[name].grad, g[f"gfull_{tag}.{name}"])
            seen["full"] = max(seen["full"], e)
            assert e <= GRAD_BAR, (tag, name, e)
    # oracle autograd: every element of all 48 tensors
    ocfg = oracle.OracleConfig()
    Pc, Pf = oracle.params_from_numpy(c["Pc"], True), oracle.params_from_numpy(c["Pf"], True)
    o = oracle.render_rays(ocfg, Pc, Pf, oracle.make_ray_batch(t(c["rays_o"]), t(c["rays_d"])), t(c["skts"]), t(c["cyls"]),
                           64, 16, t_rand=t(c["t_rand"]), u_imp=t(c["u_imp"]), noise=t(c["noise"]), noise_fine=t(c["noise_fine"]))
    lo, _ = oracle.nerf_loss(o, target.cpu(), torch.ones(n, 3))
    lo.backward()
    for P, net in [(Pc, caster.network), (Pf, caster.network_fine)]:
        for name, p in net.named_parameters():
            e = grad_err(p.grad, P[name].grad)
            seen["oracle"] = max(seen["oracle"], e)
            assert e <= GRAD_BAR, (name, e)
    print("fp32 training-step gradients, observed maxima: " + ", ".join(f"{k} {v:.2e}" for k, v in seen.items()) +
          f"  (bars: norm {NORM_BAR:g}, element / tensor max {GRAD_BAR:g})")


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_single_net_training_gradients_vs_oracle(oracle, precision):
    """surreal_single.txt shape of the path (single_net: the coarse network also evaluates the importance samples, merged by
    sorted index; multires_views = 0): training-step outputs and all parameter gradients against the pinned oracle's autograd."""
    c = build("single_net")
    mv = c["cfg"]["multires_views"]
    kw = dict(D=8, W=256, input_ch=360, input_ch_bones=72, input_ch_views=72 * (1 + 2 * mv), use_viewdirs=True)
    net = networks.NeRF(**kw)
    net.load_state_dict({k: t(v) for k, v in c["Pc"].items()})
    ck = {"cutoff": True, "cutoff_dist": 0.5, "cutoff_inputs": True, "cutoff_dim": 24}
    e_v, _ = networks.get_embedder(7, input_dims=24, cutoff_kwargs=dict(ck, dist_inputs=False))
    e_b, _ = networks.get_embedder(0, input_dims=72, cutoff_kwargs={"cutoff": False})
    e_d, _ = networks.get_embedder(mv, input_dims=72, cutoff_kwargs=dict(ck, dist_inputs=True))
    caster = raycaster.RayCaster(net, e_v, e_b, e_d, network_fine=net, single_net=True).cuda()
    caster.train()
    caster.train_precision = precision
    n, S, Ni = c["n"], c["S"], c["Ni"]
    rng = np.random.RandomState(4)
    rnd = {"t_rand": rng.rand(n, S).astype(np.float32), "u_imp": rng.rand(n, Ni).astype(np.float32),
           "noise": rng.randn(n, S).astype(np.float32), "noise_fine": rng.randn(n, S + Ni).astype(np.float32)}
    rb = importlib.import_module("a-nerf_amd.pipeline").make_ray_batch(dev(c["rays_o"]), dev(c["rays_d"]))
    skts_d = dev(c["skts"]).requires_grad_(True)      # pose gradients too: the 72-wide view input of k_mlp_bwd_in (one column group)
    kwargs = dict(cfg=ops.PathConfig(**c["cfg"]), ray_batch=rb, skts=skts_d, cyls=dev(c["cyls"]), n_samples=S, n_importance=Ni,
                  tau_v=20.0, tau_d=20.0, cut_v=torch.full((24,), 0.5, device="cuda"), cut_d=torch.full((24,), 0.5, device="cuda"),
                  cam_idx=None, t_rand=dev(rnd["t_rand"]), u_imp=dev(rnd["u_imp"]), noise=dev(rnd["noise"]),
                  noise_fine=dev(rnd["noise_fine"]), lindisp=False, single_net=True)
    out = importlib.import_module("a-nerf_amd.autograd_path").render_rays_train(caster, kwargs)
    target = dev(np.random.default_rng(5).random((n, 3)))
    loss, _ = render_mod.nerf_loss(out, target, bgs=1.0)
    loss.backward()
    ocfg = oracle.OracleConfig(**c["cfg"])
    P = oracle.params_from_numpy(c["Pc"], True)
    sk = t(c["skts"]).requires_grad_(True)
    o = oracle.render_rays(ocfg, P, P, oracle.make_ray_batch(t(c["rays_o"]), t(c["rays_d"])), sk, t(c["cyls"]), S, Ni,
                           t_rand=t(rnd["t_rand"]), u_imp=t(rnd["u_imp"]), noise=t(rnd["noise"]), noise_fine=t(rnd["noise_fine"]),
                           single_net=True)
    lo, _ = oracle.nerf_loss(o, target.cpu(), 1.0)
    lo.backward()
    for k in ["rgb_map", "acc_map", "alpha", "rgb0", "alpha0"]:
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), o[k].detach().numpy(), atol=1e-4, err_msg=k)
    assert abs(float(loss.detach()) - float(lo.detach())) < 5e-6
    # fp32 kernels: summation-order noise only.  bf16x3: every product carries ~2^-17 relative error (two bf16 = 16 mantissa
    # bits per operand), which the heavy cancellation inside a weight gradient (sum over thousands of samples) amplifies to
    # ~1e-3 of the tensor's largest element -- the bar the reference golden-vector tests use (2e-3) still holds.
    bar, nbar = (GRAD_BAR, NORM_BAR) if precision == "fp32" else (6e-3, 2e-3)
    worst = wn = 0.0
    for name, p in net.named_parameters():
        ref = P[name].grad.numpy()
        e = grad_err(p.grad, ref)
        en = abs(float(p.grad.norm()) - float(np.linalg.norm(ref))) / (float(np.linalg.norm(ref)) + 1e-30)
        worst, wn = max(worst, e), max(wn, en)
        assert e <= bar, (name, e)
        assert en <= nbar, (name, en)
    e_sk = grad_err(skts_d.grad, sk.grad)
    print(f"single_net [{precision}] gradients, observed maxima: element / tensor max {worst:.2e}, norm {wn:.2e}, dskts {e_sk:.2e}")
    assert e_sk <= (1e-3 if precision == "fp32" else 6e-3), e_sk


@pytest.mark.parametrize("mv,code,gate_bones", [(0, 0, False), (4, 0, False), (4, 16, False), (4, 0, True)])
def test_fused_input_gradient_kernel_variants_vs_oracle(oracle, mv, code, gate_bones):
    """k_mlp_bwd_in_enc<LD, CODE> (the encoding's backward as the input-gradient kernel's epilogue, round 6) in all three
    instantiations + the cutoff_bones gate, through the one-call entry points with two networks: dskts, frame-code and parameter
    gradients against the oracle's autograd on 96 rays x (24 + 8) samples with per-ray poses (rays straddle tiles and waves)."""
    fk = {"multires_views": mv, "framecode_ch": code}
    cfg = ops.PathConfig(cutoff_bones=gate_bones, **fk)
    ocfg = oracle.OracleConfig(**fk)
    synth = importlib.import_module("a-nerf_amd.synth")
    pipeline = importlib.import_module("a-nerf_amd.pipeline")
    ap = importlib.import_module("a-nerf_amd.autograd_path")
    n, S, Ni, n_codes = 96, 24, 8, 5
    mk = dict(multires_views=mv, **(dict(framecode_ch=code, n_codes=n_codes) if code else {}))
    Pc_np, Pf_np = synth.make_net_params(51, **mk), synth.make_net_params(52, **mk)
    ro, rd, kp, skts, bones, cyls, pidx = synth.scene_batch(n, [3, 4, 5, 6], ray_seed=21, per_ray_pose=True)
    rng = np.random.RandomState(8)
    rnd = {"t_rand": rng.rand(n, S).astype(np.float32), "u_imp": rng.rand(n, Ni).astype(np.float32),
           "noise": rng.randn(n, S).astype(np.float32), "noise_fine": rng.randn(n, S + Ni).astype(np.float32)}
    cam = (np.asarray(pidx) % n_codes).astype(np.float32)
    target = np.random.default_rng(9).random((n, 3)).astype(np.float32)
    Pc, Pf = {k: dev(v) for k, v in Pc_np.items()}, {k: dev(v) for k, v in Pf_np.items()}
    pk = lambda P, w: ops.pack_params(cfg, P, w)
    shapes = [tuple(Pc[nm + sfx].shape) for nm in ops.PARAM_ORDER for sfx in (".weight", ".bias")]
    out, state = ops.train_forward(cfg, pk(Pc, 0), pk(Pf, 0), pipeline.make_ray_batch(dev(ro), dev(rd)), dev(skts), dev(cyls), S, Ni,
                                   cam_idx=dev(cam) if code else None, codes_c=Pc.get("framecodes.codes.weight"),
                                   codes_f=Pf.get("framecodes.codes.weight"), **{k: dev(v) for k, v in rnd.items()})
    leaf = {k: out[k].detach().clone().requires_grad_(True) for k in ("rgb_map", "acc_map", "rgb0", "acc0")}
    loss, _ = render_mod.nerf_loss(leaf, dev(target), bgs=1.0, loss_fn="MSE")
    g = dict(zip(leaf, torch.autograd.grad(loss, list(leaf.values()))))
    gc, gf, g_skts, gcc, gcf = ops.backward(state, g, pk(Pc, 1)[0], pk(Pf, 1)[0], ap.perm_tables(cfg, torch.device("cuda")), shapes, shapes,
                                            pk(Pc, 2)[0], pk(Pf, 2)[0], want_skts=True, want_codes_c=code > 0, want_codes_f=code > 0)
    oc, of = oracle.params_from_numpy(Pc_np, True), oracle.params_from_numpy(Pf_np, True)
    sk = t(skts).requires_grad_(True)
    o = oracle.render_rays(ocfg, oc, of, oracle.make_ray_batch(t(ro), t(rd)), sk, t(cyls), S, Ni, cam_idx=t(cam) if code else None,
                           gate_r=gate_bones, **{k: t(v) for k, v in rnd.items()})
    lo, _ = oracle.nerf_loss(o, t(target), 1.0)
    lo.backward()
    assert abs(float(loss.detach()) - float(lo.detach())) < 5e-6
    e_sk = grad_err(g_skts, sk.grad)
    assert e_sk <= 1e-3 and float(g_skts[:, :, 3].abs().max()) == 0.0, e_sk
    # parameter gradients (not this kernel's output; checked so that the step as a whole is the oracle's): Frobenius-relative, and a
    # loose element bar -- on 96 rays a single view-layer ReLU flipping between the two float32 evaluations moves a bias entry by
    # 2.6e-3 of the tensor's largest (seen with cutoff_bones: fine views_linears.0.bias, Frobenius 6.8e-4, identical with the
    # unfused kernels; tools/diag/gate_bones_dskts.py)
    worst = wfro = 0.0
    for got, P in ((gc, oc), (gf, of)):
        for i, nm in enumerate(ops.PARAM_ORDER):
            for j2, sfx in enumerate((".weight", ".bias")):
                a, w = got[2 * i + j2].cpu(), P[nm + sfx].grad
                worst = max(worst, grad_err(a, w))
                wfro = max(wfro, float((a - w).norm() / (w.norm() + 1e-30)))
    assert worst <= 5e-3 and wfro <= 1e-3, (worst, wfro)
    e_code = 0.0
    if code:
        e_code = max(grad_err(gcc, oc["framecodes.codes.weight"].grad), grad_err(gcf, of["framecodes.codes.weight"].grad))
        assert e_code <= GRAD_BAR, e_code
    print(f"k_mlp_bwd_in_enc<{mv},{code}> cutoff_bones={gate_bones}: dskts {e_sk:.2e} (bar 1e-3), parameters {worst:.2e} / Frobenius {wfro:.2e}, frame codes {e_code:.2e} (bar {GRAD_BAR:g})")


@pytest.mark.parametrize("name", ["train_pytest", "mixamo_train"])
def test_pose_and_framecode_gradients(oracle, golden, name):
    """d(loss)/d(skts) (pose optimisation) and frame-code gradients vs the reference golden."""
    g = golden(name)
    c = build(name)
    caster = make_caster(c)
    caster.train()
    n = c["n"]
    skts = dev(c["skts"]).requires_grad_(True)
    cams = None if "cams" not in c else dev(c["cams"])
    out = render_mod.render(64, 64, 75.0, chunk=4096, rays=(dev(c["rays_o"]), dev(c["rays_d"])), use_viewdirs=True,
                            ray_caster=caster, kp_batch=dev(c["kp"]), skts=skts, cyls=dev(c["cyls"]),
                            bones=dev(c["bones"]), cams=cams, subject_idxs=None, N_samples=64, N_importance=16,
                            perturb=1.0, raw_noise_std=1.0, pytest=True,
                            preproc_kwargs={"density_scale": 1.0, "density_fn": torch.nn.functional.relu})
    seed = 1 if name == "train_pytest" else 2
    target = dev(np.random.default_rng(seed).random((n, 3)))
    loss, _ = render_mod.nerf_loss(out, target, bgs=torch.ones(n, 3, device="cuda"), loss_fn=c.get("loss", "MSE"))
    assert abs(float(loss.detach()) - float(g["loss"])) < 2e-6
    loss.backward()
    ref = g["dskts"]
    got = skts.grad.cpu().numpy()
    assert np.abs(got[:, :, 3, :]).max() == 0.0
    e_sk = grad_err(got, ref)
    assert e_sk <= 1e-3, e_sk
    assert abs(np.linalg.norm(got) - np.linalg.norm(ref)) < NORM_BAR * np.linalg.norm(ref)
    wn = wc = 0.0
    for tag, net in [("c", caster.network), ("f", caster.network_fine)]:
        for pname, p in net.named_parameters():
            ref_n = float(g[f"gnorm_{tag}.{pname}"])
            wn = max(wn, abs(float(p.grad.norm()) - ref_n) / (ref_n + 1e-30))
            assert abs(float(p.grad.norm()) - ref_n) <= NORM_BAR * ref_n + 1e-9, (tag, pname)
        if name == "mixamo_train":
            e = grad_err(net.framecodes.codes.weight.grad, g[f"gfull_{tag}.framecodes.codes.weight"])
            wc = max(wc, e)
            assert e <= GRAD_BAR, (tag, e)
    print(f"{name}: observed maxima dskts {e_sk:.2e} (bar 1e-3), parameter-gradient norms {wn:.2e} (bar {NORM_BAR:g}), "
          f"frame codes {wc:.2e} (bar {GRAD_BAR:g})")


def test_eval_mode_with_gradients_uses_the_mean_code(golden):
    """render_rays in eval mode with autograd on and cams = -1 (the reference's "mean code" rule,
    core/networks/embedding.py:21-22): the differentiable route must index the table that carries the mean code as its extra row
    -- outputs equal the reference's eval golden, and the gradient reaches every code equally through the mean."""
    g = golden("mixamo_train")
    c = build("mixamo_train")
    caster = make_caster(c)
    caster.eval()
    n = c["n"]
    out = render_mod.render(64, 64, 75.0, chunk=4096, rays=(dev(c["rays_o"]), dev(c["rays_d"])), use_viewdirs=True,
                            ray_caster=caster.render_rays, kp_batch=dev(c["kp"]), skts=dev(c["skts"]), cyls=dev(c["cyls"]),
                            bones=dev(c["bones"]), cams=-torch.ones(n, device="cuda"), subject_idxs=None, N_samples=64, N_importance=16,
                            perturb=0.0, raw_noise_std=0.0,
                            preproc_kwargs={"density_scale": 1.0, "density_fn": torch.nn.functional.relu})
    assert out["rgb_map"].requires_grad
    for k in ["rgb_map", "acc_map", "rgb0"]:
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), g["eval_" + k], atol=1e-4, err_msg=k)
    (out["rgb_map"].sum() + out["rgb0"].sum()).backward()
    for net in (caster.network, caster.network_fine):
        gc = net.framecodes.codes.weight.grad
        assert gc is not None and float(gc.abs().max()) > 0
        assert torch.allclose(gc, gc[:1].expand_as(gc), rtol=0, atol=0)       # d/d(code_i) of the mean code: the same row for all i


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("name,n,S,Ni", [("train_pytest", None, 64, 16), ("mixamo_train", None, 64, 16),
                                          ("mixamo_train", 37, 24, 8), ("train_pytest", 41, 40, 0),
                                          ("mixamo_train", 1, 8, 1), ("train_pytest", 3, 300, 212)])
def test_one_call_training_step_equals_the_staged_nodes(name, n, S, Ni, precision):
    """anerf_train_forward / anerf_backward (one autograd node, one C call each way) vs the staged per-kernel autograd nodes:
    same kernels in the same order, so outputs and all parameter gradients are bit-identical (dskts: see below); ragged sizes exercise
    the zeroed pad rows of the saved planes (P not a multiple of 128) and N_importance = 0 the single-pass form."""
    c = build(name)
    n = c["n"] if n is None else n
    res = {}
    for route in ("one_call", "staged", "one_call_flat", "one_call_managed"):
        caster = make_caster(c)
        caster.train()
        caster.train_precision, caster.train_route = precision, route.replace("_flat", "").replace("_managed", "")
        if route in ("one_call_flat", "one_call_managed"):
            # FusedAdam-managed parameters.  "_flat": the optimiser is ATTACHED to the caster, so the backward adds the
            # gradients into its flat bucket in place; "_managed": not attached, gradients arrive through autograd
            # (AccumulateGrad adds them into the same views) -- the explicit opt-in of ADVICE r01.
            opt = importlib.import_module("a-nerf_amd.optim").FusedAdam([p for p in caster.parameters() if p.requires_grad])
            opt.materialize()
            assert float(opt.flat_grad.abs().max()) == 0.0
            if route == "one_call_flat":
                opt.attach(caster)
                assert caster._anerf_grad_sink() is opt
        skts = dev(c["skts"][:n]).requires_grad_(True)
        cams = None if "cams" not in c else dev(c["cams"][:n])
        out = render_mod.render(64, 64, 75.0, chunk=4096, rays=(dev(c["rays_o"][:n]), dev(c["rays_d"][:n])), use_viewdirs=True,
                                ray_caster=caster, kp_batch=dev(c["kp"][:n]), skts=skts, cyls=dev(c["cyls"][:n]),
                                bones=dev(c["bones"][:n]), cams=cams, subject_idxs=None, N_samples=S, N_importance=Ni,
                                perturb=1.0, raw_noise_std=1.0, pytest=True,
                                preproc_kwargs={"density_scale": 1.0, "density_fn": torch.nn.functional.relu})
        target = dev(np.random.default_rng(3).random((n, 3)))
        loss, _ = render_mod.nerf_loss(out, target, bgs=torch.ones(n, 3, device="cuda"), loss_fn=c.get("loss", "MSE"))
        (loss + 0.1 * out["disp_map"].mean() + 0.05 * out["alpha"].mean()).backward()     # every output gets a gradient
        nets = [("c", caster.network)] + ([("f", caster.network_fine)] if Ni > 0 else [])
        res[route] = ({k: v.detach().clone() for k, v in out.items()}, skts.grad.clone(),
                      {f"{tag}.{k}": p.grad.clone() for tag, net in nets for k, p in net.named_parameters()})
    o2, s2, g2 = res["staged"]
    for route in ("one_call", "one_call_flat", "one_call_managed"):
        o1, s1, g1 = res[route]
        assert set(o1) == set(o2) and set(g1) == set(g2)
        for k in o1:
            assert torch.equal(o1[k], o2[k]), (route, k)
        # dskts: the fp32 one-call backward applies the encoding's backward inside k_mlp_bwd_in (k_mlp_bwd_in_enc, round 6: band 0 / 4
        # anchors + double-angle steps, sums per column group); the staged nodes and the split-bf16 path keep k_encode_bwd (one
        # sincos per band, one pass over the row).  The same derivative in another order of operations: equal to rounding, not bits.
        if precision == "fp32":
            scale = float(s2.abs().max())
            assert float((s1 - s2).abs().max()) <= 2e-5 * scale and scale > 0, (route, float((s1 - s2).abs().max()), scale)
            assert torch.equal(s1, res["one_call"][1])            # ... and bit-identical among the one-call routes
        else:
            assert torch.equal(s1, s2) and float(s1.abs().max()) > 0
        for k in g1:          # incl. the frame-code tables: k_code_reduce is a fixed-order reduction (no atomics)
            assert torch.equal(g1[k], g2[k]), (route, k)
        assert all(float(v.abs().max()) > 0 for k, v in g1.items())


def test_autograd_grad_on_fused_adam_parameters_and_detach():
    """torch.autograd.grad(loss, params) on FusedAdam-managed parameters returns the real gradients (and leaves p.grad
    alone) unless the optimiser was explicitly attached; attach() -> in-place accumulation; detach() / dropping the
    optimiser restores the autograd route; a replaced p.grad falls back to autograd as well."""
    optim = importlib.import_module("a-nerf_amd.optim")
    c = build("train_pytest")
    n = 16

    def run(caster):
        out = render_mod.render(64, 64, 75.0, chunk=4096, rays=(dev(c["rays_o"][:n]), dev(c["rays_d"][:n])), use_viewdirs=True,
                                ray_caster=caster, kp_batch=dev(c["kp"][:n]), skts=dev(c["skts"][:n]), cyls=dev(c["cyls"][:n]),
                                bones=dev(c["bones"][:n]), cams=None, subject_idxs=None, N_samples=24, N_importance=8,
                                perturb=0.0, raw_noise_std=0.0,
                                preproc_kwargs={"density_scale": 1.0, "density_fn": torch.nn.functional.relu})
        return (out["rgb_map"] ** 2).sum() + (out["rgb0"] ** 2).sum()

    ref_caster = make_caster(c)
    ref_caster.train()
    ref_caster.train_route = "staged"
    params_ref = [p for p in ref_caster.parameters() if p.requires_grad]
    want = torch.autograd.grad(run(ref_caster), params_ref)
    caster = make_caster(c)
    caster.train()
    params = [p for p in caster.parameters() if p.requires_grad]
    opt = optim.FusedAdam(params)
    opt.materialize()
    got = torch.autograd.grad(run(caster), params)                       # managed, not attached
    assert all(g is not None and torch.equal(g, w) for g, w in zip(got, want))
    assert float(opt.flat_grad.abs().max()) == 0.0                        # p.grad untouched by autograd.grad
    opt.attach(caster)
    run(caster).backward()                                                # attached: lands in the bucket
    assert all(torch.equal(p.grad, w) for p, w in zip(params, want))
    assert opt.owns_grads(params, params[0].device)
    opt.zero_grad()
    params[3].grad = torch.zeros_like(params[3])                          # someone replaced a gradient tensor
    assert not opt.owns_grads(params, params[0].device)
    run(caster).backward()                                                # -> autograd route, values still right
    assert all(torch.equal(p.grad, w) for p, w in zip(params, want))
    opt.detach()
    assert caster._anerf_grad_sink is None
    opt.attach(caster)
    del opt
    import gc
    gc.collect()
    assert caster._anerf_grad_sink is None or caster._anerf_grad_sink() is None      # dropped optimiser: no dangling sink
    got = torch.autograd.grad(run(caster), params)
    assert all(torch.equal(g, w) for g, w in zip(got, want))


@pytest.mark.parametrize("name", ["train_pytest", "mixamo_train"])
def test_bf16x3_training_forward_keeps_the_gradient_bar(golden, name):
    """train_precision = 'bf16x3': split-bf16 forward (fp32 activations saved in its own column order) + the fp32
    backward: outputs, loss, weight / frame-code / pose gradients against the same reference golden vectors and bars."""
    g = golden(name)
    c = build(name)
    caster = make_caster(c)
    caster.train()
    caster.train_precision = "bf16x3"
    n = c["n"]
    skts = dev(c["skts"]).requires_grad_(True)
    cams = None if "cams" not in c else dev(c["cams"])
    out = render_mod.render(64, 64, 75.0, chunk=4096, rays=(dev(c["rays_o"]), dev(c["rays_d"])), use_viewdirs=True,
                            ray_caster=caster, kp_batch=dev(c["kp"]), skts=skts, cyls=dev(c["cyls"]),
                            bones=dev(c["bones"]), cams=cams, subject_idxs=None, N_samples=64, N_importance=16,
                            perturb=1.0, raw_noise_std=1.0, pytest=True,
                            preproc_kwargs={"density_scale": 1.0, "density_fn": torch.nn.functional.relu})
    for k in ["rgb_map", "acc_map", "alpha", "rgb0", "alpha0"]:
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), g[k], atol=1e-4, err_msg=k)
    target = dev(np.random.default_rng(1 if name == "train_pytest" else 2).random((n, 3)))
    loss, _ = render_mod.nerf_loss(out, target, bgs=torch.ones(n, 3, device="cuda"), loss_fn=c.get("loss", "MSE"))
    assert abs(float(loss.detach()) - float(g["loss"])) < 5e-6
    loss.backward()
    ref = g["dskts"]
    np.testing.assert_allclose(skts.grad.cpu().numpy(), ref, rtol=5e-3, atol=2e-3 * np.abs(ref).max(), err_msg="dskts")
    for tag, net in [("c", caster.network), ("f", caster.network_fine)]:
        for pname, p in net.named_parameters():
            ref_n = float(g[f"gnorm_{tag}.{pname}"])
            assert abs(float(p.grad.norm()) - ref_n) <= 2e-3 * ref_n + 1e-9, (tag, pname)
            key = f"gfull_{tag}.{pname}"
            if key in g:
                np.testing.assert_allclose(p.grad.cpu().numpy(), g[key], rtol=5e-3, atol=2e-3 * np.abs(g[key]).max(), err_msg=key)


def test_bf16x3_training_kernels_are_bitwise_repeatable_and_save_clean_planes():
    """Guards the two hazards found while writing the split-bf16 training forward (store data overwritten by a trailing
    ds_read; stores inside a stage): repeated launches must agree bit for bit -- outputs AND saved planes -- and the saved
    X' / U' must equal the fp32 forward's (same values, different column order) after undoing the permutations."""
    import ctypes as C
    _lib = importlib.import_module("a-nerf_amd._lib")
    ap = importlib.import_module("a-nerf_amd.autograd_path")
    pipeline = importlib.import_module("a-nerf_amd.pipeline")
    c = build("mixamo_train")
    cfg = ops.PathConfig(**c["cfg"])
    P_ = {k: dev(v) for k, v in c["Pc"].items()}
    codes = P_["framecodes.codes.weight"]
    cams = dev(c["cams"])
    rb = pipeline.make_ray_batch(dev(c["rays_o"]), dev(c["rays_d"]))
    skts, cyl = dev(c["skts"]), dev(c["cyls"])
    nf, st = ops.ray_bounds(rb, cyl)
    z, _ = ops.coarse_z(nf, st, rb, 64, dev(c["t_rand"]))
    cut = torch.full((24,), 0.5, device="cuda")
    n, s = z.shape
    T = ap.train_layout(cfg, n * s)
    pp, lib, cc = T.p_pad, _lib.load(), cfg.c()
    p = lambda t: C.c_void_p(t.data_ptr())
    stream = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(which, fn):
        packed, aux = ops.pack_params(cfg, P_, which)
        sv = {k: torch.zeros(sh, device="cuda") for k, sh in [("h", (8, pp, 256)), ("f", (pp, 256)), ("g", (pp, 128)),
                                                               ("x", (pp, T.x_width)), ("u", (pp, T.u_width))]}
        stt = _lib.AnerfSaved(p(sv["h"]), p(sv["f"]), p(sv["g"]), p(sv["x"]), p(sv["u"]), pp)
        raw = torch.empty(n, s, 4, device="cuda")
        _lib.check(fn(C.byref(cc), p(packed), p(aux), p(rb), 11, p(z), p(skts), 384, p(cams), p(codes), codes.shape[0], 20.0, 20.0,
                      p(cut), p(cut), n, s, p(raw), C.byref(stt), stream()), "train forward")
        return raw, sv

    raw0, sv0 = run(3, lib.anerf_mlp_raw_train_b3)
    for _ in range(4):
        raw1, sv1 = run(3, lib.anerf_mlp_raw_train_b3)
        assert torch.equal(raw0, raw1)
        for k in sv0:
            assert torch.equal(sv0[k], sv1[k]), k
    rawf, svf = run(0, lib.anerf_mlp_raw_train)
    assert float((raw0 - rawf).abs().max()) < 2e-5
    for nm, b3 in [("x", True), ("u", True)]:
        pa = ap.perm_tables(cfg, torch.device("cuda"), b3=True)[0 if nm == "x" else 1].long()
        pb = ap.perm_tables(cfg, torch.device("cuda"))[0 if nm == "x" else 1].long()
        a, b = torch.zeros_like(sv0[nm]), torch.zeros_like(svf[nm])
        a[:, pa] = sv0[nm]
        b[:, pb] = svf[nm]
        assert float((a - b).abs().max()) < 2e-6, nm          # same inputs to the net: no fragment bits in the planes
    for l in range(8):
        assert float((sv0["h"][l] - svf["h"][l]).abs().max()) < 2e-5, l


def test_backward_kernels_are_bitwise_repeatable_and_precisions_agree():
    """k_mlp_bwd / k_mlp_bwd_b3, k_mlp_bwd_in / _b3 and both weight-gradient GEMMs on random (fixed-seed) activations:
    every kernel must reproduce its own output bit for bit (no store / register-reuse races), and the split-bf16 twin
    must agree with the fp32 kernel to ~1e-5 relative."""
    import ctypes as C
    _lib = importlib.import_module("a-nerf_amd._lib")
    ap = importlib.import_module("a-nerf_amd.autograd_path")
    cfg = ops.PathConfig(framecode_ch=16)
    lib, cc = _lib.load(), cfg.c()
    g = torch.Generator(device="cuda").manual_seed(3)
    r = lambda *sh: torch.randn(*sh, device="cuda", generator=g)
    P = 5000                                   # ragged: 39 full tiles + 8 samples
    T = ap.train_layout(cfg, P)
    pp = T.p_pad
    shapes = synth_shapes = importlib.import_module("a-nerf_amd.synth").net_shapes(7, 4, 16)
    params = {}
    for name, (o, k) in shapes.items():
        params[name + ".weight"] = r(o, k) * (1.0 / k ** 0.5)
        params[name + ".bias"] = r(o) * 0.1
    params["framecodes.codes.weight"] = r(4, 16)
    p = lambda t: C.c_void_p(t.data_ptr())
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    zpad = lambda t: torch.cat([t, torch.zeros((pp - P,) + t.shape[1:], device="cuda")], 0)
    sv = {"h": torch.stack([zpad(torch.relu(r(P, 256))) for _ in range(8)]), "f": zpad(r(P, 256)), "g": zpad(torch.relu(r(P, 128))),
          "x": zpad(r(P, T.x_width)), "u": zpad(r(P, T.u_width))}
    stt = _lib.AnerfSaved(p(sv["h"]), p(sv["f"]), p(sv["g"]), p(sv["x"]), p(sv["u"]), pp)
    draw = zpad(r(P, 4))
    outs = {}
    for tag, which_t, which_i, f_bwd, f_in, f_wg in [("fp32", 1, 2, lib.anerf_mlp_backward, lib.anerf_input_grads, lib.anerf_weight_grads),
                                                     ("b3", 4, 5, lib.anerf_mlp_backward_b3, lib.anerf_input_grads_b3, lib.anerf_weight_grads_b3)]:
        pt, _ = ops.pack_params(cfg, params, which_t)
        pi, _ = ops.pack_params(cfg, params, which_i)
        _, aux = ops.pack_params(cfg, params, 0)
        runs = []
        for _ in range(3):
            dz, df, dzv = torch.zeros(8, pp, 256, device="cuda"), torch.zeros(pp, 256, device="cuda"), torch.zeros(pp, 128, device="cuda")
            _lib.check(f_bwd(C.byref(cc), p(pt), p(aux), p(draw), C.byref(stt), p(dz), p(df), p(dzv), P, st()), "bwd")
            dx, du = torch.zeros(pp, T.x_width, device="cuda"), torch.zeros(pp, T.u_width, device="cuda")
            _lib.check(f_in(C.byref(cc), p(pi), p(dz), p(dzv), pp, P, p(dx), p(du), st()), "in")
            grads = [torch.zeros(sh, device="cuda") for n_ in ops.PARAM_ORDER for sh in (params[n_ + ".weight"].shape, params[n_ + ".bias"].shape)]
            gs = _lib.AnerfNetGrads()
            for i in range(12):
                gs.w[i], gs.b[i] = grads[2 * i].data_ptr(), grads[2 * i + 1].data_ptr()
            ws = torch.empty(T.gemm_ws_floats, device="cuda")
            px, pu = ap.perm_tables(cfg, torch.device("cuda"))
            _lib.check(f_wg(C.byref(cc), C.byref(stt), p(dz), p(df), p(dzv), p(draw), P, p(px), p(pu), C.byref(gs), p(ws),
                            T.gemm_ws_floats, st()), "wg")
            runs.append([dz, df, dzv, dx, du] + grads)
        for a, b in zip(runs[0], runs[1]):
            assert torch.equal(a, b)
        for a, b in zip(runs[0], runs[2]):
            assert torch.equal(a, b)
        outs[tag] = runs[0]
        # the two head problems ride as fp32 FMAs inside two of the GEMM's blocks (GemmHead, anerf_gemm.hip): rows of `draw`
        # against h7 / g, summed in sample order per wave, then over the waves' row slots and the row chunks -- against float64
        dz, df, dzv = runs[0][0], runs[0][1], runs[0][2]
        grads = runs[0][5:]
        d64 = draw[:P].double()
        ix = {n_: 2 * i for i, n_ in enumerate(ops.PARAM_ORDER)}
        checks = [("alpha_linear", d64[:, 3:4].T @ sv["h"][7][:P].double(), d64[:, 3:4].sum(0)),
                  ("rgb_linear", d64[:, :3].T @ sv["g"][:P].double(), d64[:, :3].sum(0)),
                  ("pts_linears.1", dz[1][:P].double().T @ sv["h"][0][:P].double(), dz[1][:P].double().sum(0)),
                  ("feature_linear", df[:P].double().T @ sv["h"][7][:P].double(), df[:P].double().sum(0)),
                  ("pts_linears.7", dz[7][:P].double().T @ sv["h"][6][:P].double(), dz[7][:P].double().sum(0))]
        for name, w64, b64 in checks:
            gw, gb = grads[ix[name]].double(), grads[ix[name] + 1].double()
            assert gw.shape == w64.shape and gb.shape == b64.shape, name
            tol = 2e-6 if tag == "fp32" else 3e-5
            assert float((gw - w64).abs().max()) <= tol * float(w64.abs().max()), (tag, name, float((gw - w64).abs().max()))
            assert float((gb - b64).abs().max()) <= tol * float(b64.abs().max()) + 1e-9, (tag, name)
    for k, (a, b) in enumerate(zip(outs["fp32"], outs["b3"])):
        scale = float(a.abs().max())
        assert float((a - b).abs().max()) <= 3e-5 * scale + 1e-12, (k, float((a - b).abs().max()), scale)


def test_composite_backward_vs_autograd(oracle):
    """k_composite_bwd alone against torch autograd of the oracle's composite (softplus density too)."""
    autograd_path = importlib.import_module("a-nerf_amd.autograd_path")
    gen = torch.Generator().manual_seed(3)
    n, s = 37, 80
    raw = torch.randn(n, s, 4, generator=gen)
    z = torch.sort(torch.rand(n, s, generator=gen) * 3 + 1, -1)[0]
    rd = torch.randn(n, 3, generator=gen)
    rays = torch.cat([torch.zeros(n, 3), rd], -1)
    noise = torch.randn(n, s, generator=gen) * 0.3
    g_rgb, g_acc, g_disp = torch.randn(n, 3, generator=gen), torch.randn(n, generator=gen), torch.randn(n, generator=gen) * 0.1
    g_alpha = torch.randn(n, s, generator=gen) * 0.1
    for shift in [None, 1.0]:
        ocfg = oracle.OracleConfig(softplus_shift=shift, density_scale=0.7)
        rr = raw.clone().requires_grad_(True)
        o = oracle.composite(ocfg, rr, z, rd, noise)
        (o["rgb_map"] * g_rgb).sum().add((o["acc_map"] * g_acc).sum()).add((o["disp_map"] * g_disp).sum()) \
            .add((o["alpha"] * g_alpha).sum()).backward()
        cfg = ops.PathConfig(density_scale=0.7, softplus_shift=shift)
        rc = raw.cuda().requires_grad_(True)
        outs = autograd_path._CompositeFn.apply(dict(cfg=cfg, rays=rays.cuda(), z=z.cuda(), noise=noise.cuda()), rc)
        (outs[0] * g_rgb.cuda()).sum().add((outs[2] * g_acc.cuda()).sum()).add((outs[1] * g_disp.cuda()).sum()) \
            .add((outs[3] * g_alpha.cuda()).sum()).backward()
        ref = rr.grad.numpy()
        np.testing.assert_allclose(rc.grad.cpu().numpy(), ref, rtol=2e-3, atol=2e-5 * np.abs(ref).max())


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_ray_noise_std_training_step_vs_reference_golden(golden, monkeypatch, precision):
    """ray_noise_std > 0 (`pts + randn_like(pts) * ray_noise_std`, raycasters.py:660,674) through RayCaster.render_rays: outputs,
    loss, all gradient norms and dskts against vectors from the reference itself (its torch.randn_like replaced by the same
    numpy-seeded arrays, tests/golden/gen_golden_raynoise.py); eval with the offsets through the no-grad one-call route too."""
    from cases import ray_noise_arrays, RAY_NOISE_STD
    g = golden("ray_noise")
    c = build("ray_noise")
    caster = make_caster(c)
    caster.train()
    caster.train_precision = precision
    n, S, Ni = c["n"], c["S"], c["Ni"]
    real_randn = torch.randn

    def seeded_randn(*shape, **kw):          # what RayCaster.render_rays draws for the point offsets: unit normals [n, S|Ni, 3]
        if len(shape) == 3 and shape[2] == 3:
            return torch.tensor(np.random.RandomState(1000 + shape[1]).randn(*shape), dtype=torch.float32, device=kw.get("device"))
        return real_randn(*shape, **kw)
    monkeypatch.setattr(torch, "randn", seeded_randn)
    skts = dev(c["skts"]).requires_grad_(True)
    out = render_mod.render(64, 64, 75.0, chunk=4096, rays=(dev(c["rays_o"]), dev(c["rays_d"])), use_viewdirs=True,
                            ray_caster=caster, kp_batch=dev(c["kp"]), skts=skts, cyls=dev(c["cyls"]), bones=dev(c["bones"]),
                            cams=None, subject_idxs=None, N_samples=S, N_importance=Ni, perturb=1.0, raw_noise_std=1.0,
                            ray_noise_std=RAY_NOISE_STD, pytest=True,
                            preproc_kwargs={"density_scale": 1.0, "density_fn": torch.nn.functional.relu})
    for k in ["rgb_map", "acc_map", "alpha", "rgb0", "alpha0"]:
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), g[k], atol=1e-4, err_msg=k)
    assert np.abs(out["rgb_map"].detach().cpu().numpy() - g["rgb_map_no_noise"]).max() > 1e-3      # the offsets were applied
    target = dev(np.random.default_rng(4).random((n, 3)))
    loss, _ = render_mod.nerf_loss(out, target, bgs=torch.ones(n, 3, device="cuda"))
    assert abs(float(loss.detach()) - float(g["loss"])) < 5e-6
    loss.backward()
    ref = g["dskts"]
    np.testing.assert_allclose(skts.grad.cpu().numpy(), ref, rtol=5e-3, atol=2e-3 * np.abs(ref).max(), err_msg="dskts")
    for tag, net in [("c", caster.network), ("f", caster.network_fine)]:
        for name, p in net.named_parameters():
            ref_n = float(g[f"gnorm_{tag}.{name}"])
            assert abs(float(p.grad.norm()) - ref_n) <= 2e-3 * ref_n + 1e-9, (tag, name)
            if precision == "fp32":      # element level; the split-bf16 kernels carry ~1e-3 of a tensor's largest element
                np.testing.assert_allclose(p.grad.reshape(-1)[:64].cpu().numpy(), g[f"gslice_{tag}.{name}"], rtol=5e-3,
                                           atol=2e-3 * ref_n / max(np.sqrt(p.numel()), 1.0) + 1e-9, err_msg=f"{tag}.{name}")
    # no-grad route (anerf_forward) with the same offsets == the training forward's outputs
    a, b = ray_noise_arrays(n, S, Ni)
    kw = dict(cfg=ops.PathConfig(), ray_batch=importlib.import_module("a-nerf_amd.pipeline").make_ray_batch(dev(c["rays_o"]), dev(c["rays_d"])),
              skts=dev(c["skts"]), cyls=dev(c["cyls"]), n_samples=S, n_importance=Ni, t_rand=dev(c["t_rand"]), u_imp=dev(c["u_imp"]),
              noise=dev(c["noise"]), noise_fine=dev(c["noise_fine"]), pts_noise=dev(a), pts_noise_is=dev(b))
    which = 3 if precision == "bf16x3" else 0
    ev = importlib.import_module("a-nerf_amd.pipeline").render_rays_forward(
        net_c=caster.network.packed(which), net_f=caster.network_fine.packed(which), precision=precision, **kw)
    for k in ["rgb_map", "alpha", "rgb0"]:
        np.testing.assert_allclose(ev[k].cpu().numpy(), out[k].detach().cpu().numpy(), atol=2e-6, err_msg=k)
    with pytest.raises(NotImplementedError):       # staged route (single_net / extras): loud, not silently without offsets
        importlib.import_module("a-nerf_amd.pipeline").render_rays_forward(net_c=caster.network.packed(which),
                                                                          net_f=caster.network_fine.packed(which), extras=True,
                                                                          precision=precision, **kw)


@pytest.mark.parametrize("n,S,Ni", [(1, 8, 8), (5, 17, 9), (3, 33, 8), (130, 8, 16), (2, 64, 128)])
def test_ragged_training_sizes_vs_oracle_autograd(oracle, synth, n, S, Ni):
    """Edge sizes on the TRAINING path (the forward-only edge cases are in test_hip_edge_cases.py): one ray, ray counts and sample
    counts that leave tiles ragged (n * S and n * (S + Ni) not multiples of the 128-sample tile, S + Ni odd), more importance than
    coarse samples -- outputs and every element of all 48 parameter gradients + dskts against the pinned oracle's autograd,
    with jitter and density noise as explicit inputs (anerf_train_forward / anerf_backward through the one-node route)."""
    ap = importlib.import_module("a-nerf_amd.autograd_path")
    ro, rd, kp, skts, bones, cyls, _ = synth.scene_batch(n, [3, 4], ray_seed=40 + n, per_ray_pose=True)
    Pc, Pf = synth.make_net_params(11), synth.make_net_params(12)
    g = torch.Generator().manual_seed(n * 1000 + S)
    rnd = dict(t_rand=torch.rand(n, S, generator=g), u_imp=torch.rand(n, Ni, generator=g), noise=torch.randn(n, S, generator=g),
               noise_fine=torch.randn(n, S + Ni, generator=g))
    cfg = ops.PathConfig()
    pipeline = importlib.import_module("a-nerf_amd.pipeline")
    rb = pipeline.make_ray_batch(dev(ro), dev(rd))
    net_c = ops.pack_params(cfg, {k: dev(v) for k, v in Pc.items()})
    net_f = ops.pack_params(cfg, {k: dev(v) for k, v in Pf.items()})
    out, state = ops.train_forward(cfg, net_c, net_f, rb, dev(skts), dev(cyls), S, Ni, **{k: v.cuda() for k, v in rnd.items()})
    gmaps = {"rgb_map": 2.0 * out["rgb_map"], "rgb0": 2.0 * out["rgb0"], "acc_map": torch.full_like(out["acc_map"], 0.3)}
    shapes = [tuple(np.asarray(Pc[nm + sfx]).shape) for nm in ops.PARAM_ORDER for sfx in (".weight", ".bias")]
    pk = lambda P, w: ops.pack_params(cfg, {k: dev(v) for k, v in P.items()}, which=w)[0]
    gc, gf, g_skts, _, _ = ops.backward(state, gmaps, pk(Pc, 1), pk(Pf, 1), ap.perm_tables(cfg, rb.device), shapes, shapes,
                                        pk(Pc, 2), pk(Pf, 2), want_skts=True)
    torch.cuda.synchronize()
    oc, of = oracle.params_from_numpy(Pc, True), oracle.params_from_numpy(Pf, True)
    sk = t(skts).requires_grad_(True)
    o = oracle.render_rays(oracle.OracleConfig(), oc, of, oracle.make_ray_batch(t(ro), t(rd)), sk, t(cyls), S, Ni, **rnd)
    for k in ("rgb_map", "acc_map", "rgb0"):
        np.testing.assert_allclose(out[k].cpu().numpy(), o[k].detach().numpy(), atol=1e-4, err_msg=k)
    ((o["rgb_map"] ** 2).sum() + (o["rgb0"] ** 2).sum() + 0.3 * o["acc_map"].sum()).backward()
    worst = 0.0
    for got, P in ((gc, oc), (gf, of)):
        for i, nm in enumerate(ops.PARAM_ORDER):
            for j, sfx in enumerate((".weight", ".bias")):
                ref = P[nm + sfx].grad
                e = grad_err(got[2 * i + j], ref)
                # a one-element tensor (alpha_linear.bias) summed over a handful of samples is a cancelling sum: the relative bar
                # gets an absolute floor of a few fp32 ulps of the terms
                floor = 3e-6 / (float(ref.abs().max()) + 1e-30)
                worst = max(worst, e)
                assert e <= GRAD_BAR + floor, (nm + sfx, e, float(ref.abs().max()))
    e_sk = grad_err(g_skts, sk.grad)
    assert e_sk <= 1e-3, e_sk
    print(f"n={n} S={S} Ni={Ni}: worst parameter-gradient error {worst:.2e}, dskts {e_sk:.2e}")
