"""GPU: fused loss / Adam / grad-norm kernels against torch's own implementations of the same
formulae (core/trainer.py:353-380 _compute_nerf_loss, :173-203 Adam + get_gradnorm)."""
import importlib

import numpy as np
import pytest
import torch

from cases import build
from test_hip_backward import dev, make_caster

pytestmark = pytest.mark.gpu

ops = importlib.import_module("a-nerf_amd.ops")
optim = importlib.import_module("a-nerf_amd.optim")
render_mod = importlib.import_module("a-nerf_amd.render")


def _preds(n, seed, coarse=True):
    g = torch.Generator(device="cuda").manual_seed(seed)
    p = {"rgb_map": torch.rand(n, 3, device="cuda", generator=g), "acc_map": torch.rand(n, device="cuda", generator=g)}
    if coarse:
        p["rgb0"] = torch.rand(n, 3, device="cuda", generator=g)
        p["acc0"] = torch.rand(n, device="cuda", generator=g)
    for v in p.values():
        v.requires_grad_(True)
    return p, torch.rand(n, 3, device="cuda", generator=g)


@pytest.mark.parametrize("loss_fn", ["MSE", "L1", "Huber"])
@pytest.mark.parametrize("bg", ["scalar", "per_ray", "none"])
@pytest.mark.parametrize("coarse", [True, False])
def test_fused_loss_value_and_gradients(loss_fn, bg, coarse):
    n = 3071          # ragged: not a multiple of the block size
    p, target = _preds(n, 5, coarse)
    bgs = {"scalar": 1.0, "per_ray": torch.rand(n, 3, device="cuda"), "none": 1.0}[bg]
    use_bg = bg != "none"
    ref, _ = render_mod.nerf_loss(p, target, bgs=bgs, loss_fn=loss_fn, coarse_weight=0.7, use_background=use_bg)
    gref = torch.autograd.grad(ref * 3.0, list(p.values()), allow_unused=True)
    out, stats = optim.fused_nerf_loss(p, target, bgs=bgs, loss_fn=loss_fn, coarse_weight=0.7, use_background=use_bg)
    got = torch.autograd.grad(out * 3.0, list(p.values()), allow_unused=True)
    assert abs(float(out.detach()) - float(ref.detach())) < 2e-7 * max(1.0, abs(float(ref.detach())))
    for k, a, b in zip(p.keys(), got, gref):
        if b is None:
            assert a is None or float(a.abs().max()) == 0.0, k
        else:
            # atol: d = pred - target carries ~6e-8 of rounding (fma vs mul+add); Huber's quadratic zone divides it by beta
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-6, atol=1e-9, err_msg=k)
    pred = p["rgb_map"] + (1 - p["acc_map"])[:, None] * bgs if use_bg else p["rgb_map"]
    mse = float(((pred - target) ** 2).mean().detach())
    assert abs(float(stats[3]) - mse) < 2e-7
    assert abs(float(render_mod.mse2psnr(stats[3])) - float(render_mod.mse2psnr(torch.tensor(mse)))) < 1e-3   # dB bar


def test_fused_loss_is_deterministic_and_handles_empty():
    p, target = _preds(50000, 9)
    a = optim.fused_nerf_loss(p, target)[1].clone()
    b = optim.fused_nerf_loss(p, target)[1].clone()
    assert torch.equal(a, b)
    e = {"rgb_map": torch.zeros(0, 3, device="cuda"), "acc_map": torch.zeros(0, device="cuda")}
    out, _ = ops.loss(e["rgb_map"], e["acc_map"], torch.zeros(0, 3, device="cuda"))
    assert float(out.abs().sum()) == 0.0
    with pytest.raises(ValueError):
        ops.loss(p["rgb_map"], p["acc_map"], target[:10])


def test_fused_adam_matches_torch_adam_and_gradnorm():
    torch.manual_seed(0)
    shapes = [(256, 432), (256,), (1, 256), (1,), (3, 128), (3,), (7, 5)]          # total not a multiple of 4
    ref = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    o_ref = torch.optim.Adam(ref, lr=5e-4, betas=(0.9, 0.999))
    o_my = optim.FusedAdam(mine, lr=5e-4, betas=(0.9, 0.999))
    for it in range(6):
        grads = [torch.randn(s, device="cuda") * (0.1 + it) for s in shapes]
        for p, q, g in zip(ref, mine, grads):
            p.grad = g.clone()
            if q.grad is None:
                q.grad = g.clone()
            else:
                q.grad.copy_(g)
        if it == 3:                       # decay_optimizer_lrate-style LR change through param_groups
            for o in (o_ref, o_my):
                o.param_groups[0]["lr"] = 2e-4
        v0 = mine[0]._version
        norms = o_my.step(zero_grad=True, want_norms=True)
        o_ref.step()
        assert mine[0]._version > v0                                    # weight-image caches see the update
        total = sum(float(g.norm(2)) ** 2 for g in grads)
        np.testing.assert_allclose(norms.cpu().numpy(), [total ** 0.5, (total / len(shapes)) ** 0.5], rtol=2e-6)
        for p, q in zip(ref, mine):
            np.testing.assert_allclose(q.detach().cpu().numpy(), p.detach().cpu().numpy(), rtol=2e-6, atol=2e-8)
            assert float(q.grad.abs().max()) == 0.0                     # zero_grad fused into the step
    assert o_my.state[mine[0]]["step"] == 6
    # torch-format state round trip: continue in a fresh optimiser, still identical to torch
    sd = o_my.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and sd["param_groups"][0]["lr"] == 2e-4
    again = [torch.nn.Parameter(q.detach().clone()) for q in mine]
    o2 = optim.FusedAdam(again, lr=1.0)
    o2.load_state_dict(sd)
    g = [torch.randn(s, device="cuda") for s in shapes]
    for p, q, gg in zip(ref, again, g):
        p.grad = gg.clone()
        q.grad = gg.clone()
    o2.step()
    o_ref.step()
    for p, q in zip(ref, again):
        np.testing.assert_allclose(q.detach().cpu().numpy(), p.detach().cpu().numpy(), rtol=2e-6, atol=2e-8)
    # torch.optim.Adam accepts our state dict too (reference checkpoints round-trip both ways)
    o3 = torch.optim.Adam([torch.nn.Parameter(q.detach().clone()) for q in mine], lr=1.0)
    o3.load_state_dict(sd)
    assert o3.param_groups[0]["lr"] == 2e-4


def test_training_steps_fused_tail_equals_torch_tail():
    """Three optimisation steps of the full path: (render -> fused loss -> backward -> FusedAdam) vs
    (render -> torch loss -> backward -> torch.optim.Adam) from identical weights; pytest-mode randomness."""
    c = build("train_pytest")
    n = c["n"]
    target = dev(np.random.default_rng(1).random((n, 3)))
    bgs = torch.ones(n, 3, device="cuda")
    kw = dict(chunk=4096, rays=(dev(c["rays_o"]), dev(c["rays_d"])), use_viewdirs=True, kp_batch=dev(c["kp"]),
              skts=dev(c["skts"]), cyls=dev(c["cyls"]), bones=dev(c["bones"]), cams=None, subject_idxs=None, N_samples=64,
              N_importance=16, perturb=1.0, raw_noise_std=1.0, pytest=True,
              preproc_kwargs={"density_scale": 1.0, "density_fn": torch.nn.functional.relu})
    casters, losses = [], []
    for fused in (False, True):
        caster = make_caster(c)
        caster.train()
        params = [p for p in caster.parameters() if p.requires_grad]
        opt = optim.FusedAdam(params, lr=5e-4) if fused else torch.optim.Adam(params, lr=5e-4)
        ls = []
        for _ in range(3):
            out = render_mod.render(64, 64, 75.0, ray_caster=caster, **kw)
            if fused:
                loss, _ = optim.fused_nerf_loss(out, target, bgs=bgs)
                loss.backward()
                opt.step(zero_grad=True)
            else:
                loss, _ = render_mod.nerf_loss(out, target, bgs=bgs)
                loss.backward()
                opt.step()
                opt.zero_grad()
            ls.append(float(loss.detach()))
        casters.append(caster)
        losses.append(ls)
    np.testing.assert_allclose(losses[1], losses[0], rtol=2e-5)
    assert losses[0][2] < losses[0][0]                                  # it actually trains
    for (k, a), (_, b) in zip(casters[0].state_dict()["network_fn_state_dict"].items(),
                              casters[1].state_dict()["network_fn_state_dict"].items()):
        d = np.abs(b.cpu().numpy() - a.cpu().numpy())
        # Adam divides by sqrt(v): elements whose gradient is rounding noise may move differently; all others agree
        # Adam divides by sqrt(v): the few elements whose gradient is rounding noise may move differently (bounded by the
        # 3 x lr = 1.5e-3 a weight can move at all); everything else agrees to ~1e-5
        assert np.mean(d > 3e-5) < 1e-2 and d.max() < 1.6e-3, (k, float(np.mean(d > 3e-5)), float(d.max()))


def test_fused_adam_groups_cadence_and_checkpoint_round_trip(tmp_path):
    """Two parameter groups in ONE flat bucket (networks + a `step_every` group, the pose-optimiser cadence of
    trainer.py:476-478) against two torch.optim.Adam instances driven the reference's way; then the reference's checkpoint
    layout: checkpoint.save_nerf splits the state into optimizer_state_dict / pose_optimizer_state_dict (torch format),
    torch optimisers load them, and load_nerf restores a fresh FusedAdam to continue bit for bit."""
    checkpoint = importlib.import_module("a-nerf_amd.checkpoint")
    c = build("train_pytest")
    g = torch.Generator(device="cuda").manual_seed(0)

    def fresh():
        caster = make_caster(c)
        extra = [torch.nn.Parameter(torch.linspace(-1, 1, 37 * 3, device="cuda").reshape(37, 3).clone()),
                 torch.nn.Parameter(torch.linspace(0, 2, 5, device="cuda").clone())]
        return caster, [p for p in caster.parameters() if p.requires_grad], extra

    caster, net_p, pose_p = fresh()
    ref_c, ref_net, ref_pose = fresh()
    opt = optim.FusedAdam([{"params": net_p, "lr": 5e-4}, {"params": pose_p, "lr": 2e-3, "step_every": 3}])
    t_net, t_pose = torch.optim.Adam(ref_net, lr=5e-4), torch.optim.Adam(ref_pose, lr=2e-3)
    opt.materialize()
    grads = [[torch.randn(p.shape, device="cuda", generator=g) * 1e-2 for p in net_p + pose_p] for _ in range(7)]

    def drive(opt_f, params_f, i):
        for p, gr in zip(params_f, grads[i - 1]):
            p.grad.add_(gr)
        opt_f.step(zero_grad=True, i=i)

    def drive_ref(i):
        for p, gr in zip(ref_net + ref_pose, grads[i - 1]):
            p.grad = gr.clone() if p.grad is None else p.grad + gr
        t_net.step()
        t_net.zero_grad()
        if i % 3 == 0:                      # trainer.py:476-478
            t_pose.step()
            t_pose.zero_grad()

    for i in range(1, 5):
        drive(opt, net_p + pose_p, i)
        drive_ref(i)
    assert opt._steps == [4, 1]
    for a, b in zip(net_p + pose_p, ref_net + ref_pose):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)
    assert float(opt.flat_grad[opt._segments()[1][0]:].abs().max()) > 0          # pose bucket is accumulating (i = 4)
    # ---- checkpoint in the reference's layout
    path = str(tmp_path / "ck.tar")
    checkpoint.save_nerf(path, 4, caster, opt)
    ck = torch.load(path, map_location="cuda", weights_only=False)
    assert len(ck["optimizer_state_dict"]["state"]) == len(net_p) and ck["optimizer_state_dict"]["param_groups"][0]["params"] == list(range(len(net_p)))
    t2 = torch.optim.Adam(ref_net, lr=1.0)
    t2.load_state_dict(ck["optimizer_state_dict"])                                # torch accepts it and agrees with its own state
    for p_ref, i in zip(ref_net[:3], range(3)):
        np.testing.assert_allclose(t2.state[p_ref]["exp_avg"].cpu().numpy(), t_net.state[p_ref]["exp_avg"].cpu().numpy(), rtol=2e-6, atol=1e-9)
    # the pose group has no checkpoint slot without a pose layer (trainer.py:491-496); its state round-trips through group views
    pose_sd = opt.group_optimizer(1).state_dict()
    assert len(pose_sd["state"]) == 2 and pose_sd["param_groups"][0]["step_every"] == 3 and pose_sd["param_groups"][0]["lr"] == 2e-3
    caster2, net2, pose2 = fresh()
    with torch.no_grad():
        for a, b in zip(pose2, pose_p):
            a.copy_(b)
    opt2 = optim.FusedAdam([{"params": net2, "lr": 5e-4}, {"params": pose2, "lr": 2e-3, "step_every": 3}])
    r = checkpoint.load_nerf(path, caster2, opt2)
    opt2.group_optimizer(1).load_state_dict(pose_sd)
    assert r["global_step"] == 4 and opt2._steps == [4, 1]
    o1, n1 = opt._segments()[1]
    opt2.flat_grad[o1:o1 + n1].copy_(opt.flat_grad[o1:o1 + n1])                   # the half-accumulated pose gradients are trainer state
    for i in range(5, 8):
        drive(opt, net_p + pose_p, i)
        drive(opt2, net2 + pose2, i)
    assert torch.equal(opt.flat, opt2.flat) and opt._steps == opt2._steps == [7, 2]


@pytest.mark.gpu
def test_from_torch_takes_over_hyper_parameters_and_restored_state():
    """run_nerf.py hands the trainer two torch Adams (create_raycaster's, create_popt's), possibly restored from a checkpoint:
    FusedAdam.from_torch builds the one-bucket optimiser from them -- groups, learning rates, cadence, moments, step counts --
    and continues exactly where they would (torch's own steps as the reference)."""
    c = build("train_pytest")
    g = torch.Generator(device="cuda").manual_seed(3)

    def fresh():
        caster = make_caster(c)
        extra = [torch.nn.Parameter(torch.linspace(-1, 1, 24 * 6, device="cuda").reshape(24, 6).clone()),
                 torch.nn.Parameter(torch.linspace(0, 2, 9, device="cuda").clone())]
        return caster, [p for p in caster.parameters() if p.requires_grad], extra

    (_, net_a, pose_a), (_, net_b, pose_b) = fresh(), fresh()
    grads = [[torch.randn(p.shape, device="cuda", generator=g) * 1e-2 for p in net_a + pose_a] for _ in range(6)]

    def torch_pair(net, pose):
        return torch.optim.Adam(net, lr=5e-4, betas=(0.9, 0.999)), torch.optim.Adam(pose, lr=2e-3, betas=(0.9, 0.999))

    def torch_iteration(t_net, t_pose, params, i):
        for p, gr in zip(params, grads[i - 1]):
            p.grad = gr.clone() if p.grad is None else p.grad + gr
        t_net.step()
        t_net.zero_grad()
        if i % 2 == 0:
            t_pose.step()
            t_pose.zero_grad()

    ta_net, ta_pose = torch_pair(net_a, pose_a)
    tb_net, tb_pose = torch_pair(net_b, pose_b)
    for i in (1, 2, 3):                                   # three iterations on torch's optimisers: net at step 3, pose at step 1,
        torch_iteration(ta_net, ta_pose, net_a + pose_a, i)          # one pose gradient accumulated and pending
        torch_iteration(tb_net, tb_pose, net_b + pose_b, i)
    pending = [p.grad.clone() for p in pose_b]
    fused = optim.FusedAdam.from_torch(tb_net, tb_pose, pose_step_every=2)
    assert [gr["lr"] for gr in fused.param_groups] == [5e-4, 2e-3] and [gr["step_every"] for gr in fused.param_groups] == [1, 2]
    assert fused._steps == [3, 1] and all(p.data_ptr() >= fused.flat.data_ptr() for p in net_b + pose_b)
    sd = fused.group_optimizer(1).state_dict()
    for j, p in enumerate(pose_b):
        assert torch.equal(sd["state"][j]["exp_avg"], tb_pose.state[p]["exp_avg"]) and float(sd["state"][j]["step"]) == 1.0
    for p, gr in zip(pose_b, pending):                    # materialize() keeps gradients that were pending
        assert torch.equal(p.grad, gr)
    for i in (4, 5, 6):
        torch_iteration(ta_net, ta_pose, net_a + pose_a, i)
        for p, gr in zip(net_b + pose_b, grads[i - 1]):
            p.grad.add_(gr)
        fused.step(zero_grad=True, i=i)
    assert fused._steps == [6, 3]
    for a, b in zip(net_a + pose_a, net_b + pose_b):
        np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)
    with pytest.raises(TypeError):
        optim.FusedAdam.from_torch(torch.optim.SGD(net_a, lr=0.1))
    with pytest.raises(TypeError):
        optim.FusedAdam.from_torch(torch.optim.Adam(net_a, lr=0.1, amsgrad=True))


@pytest.mark.gpu
def test_backward_with_the_cached_unit_seed_equals_plain_backward():
    """optim.backward(loss) seeds the graph with a cached 1.0 and the fused losses skip their `gradient * 1` launches: the
    gradients w.r.t. the rendered maps (and through the pose regulariser) are bit-identical to loss.backward()'s, also when the
    loss is a sum of both and when it is scaled afterwards (then the seed is not what reaches the losses, and they scale)."""
    pose_opt = importlib.import_module("a-nerf_amd.pose_opt")
    g = torch.Generator(device="cuda").manual_seed(3)
    n = 257
    mk = lambda *sh: torch.rand(*sh, device="cuda", generator=g)
    target = mk(n, 3)
    vals0, anchors, w = torch.randn(5, 24, 3, 3, device="cuda", generator=g), torch.randn(5, 24, 6, device="cuda", generator=g), mk(5)
    base = {k: (mk(n, 3) if "rgb" in k else mk(n)) for k in ("rgb_map", "acc_map", "rgb0", "acc0")}
    res = {}
    for mode in ("plain", "seeded", "scaled_plain", "scaled_seeded"):
        preds = {k: v.clone().requires_grad_(True) for k, v in base.items()}
        vals = vals0.clone().requires_grad_(True)
        loss, _ = optim.fused_nerf_loss(preds, target, bgs=1.0, loss_fn="L1")
        loss = loss + pose_opt.kp_loss(vals, anchors, w / w.sum(), True, 0.01, 2.0)
        if mode.startswith("scaled"):
            loss = loss * 0.37
        (optim.backward if mode.endswith("seeded") else torch.autograd.backward)(loss)
        res[mode] = [preds[k].grad.clone() for k in sorted(preds)] + [vals.grad.clone()]
    for a, b in (("plain", "seeded"), ("scaled_plain", "scaled_seeded")):
        assert all(torch.equal(x, y) for x, y in zip(res[a], res[b])), (a, b)
    assert not torch.equal(res["plain"][0], res["scaled_plain"][0])
