"""CPU: host-side mirror of the reference interface (no HIP compute): constructor parity, checkpoint key layout,
tau schedule, configuration guards."""
import argparse
import importlib
import os

import numpy as np
import pytest
import torch

networks = importlib.import_module("a-nerf_amd.networks")
raycaster = importlib.import_module("a-nerf_amd.raycaster")
synth = importlib.import_module("a-nerf_amd.synth")


def surreal_args(**over):
    """The attributes create_raycaster reads, with run_nerf.config_parser defaults + configs/surreal/surreal.txt."""
    a = dict(n_framecodes=None, use_cutoff=True, normalize_cutoff=False, cutoff_mm=500.0, ext_scale=0.001, cutoff_inputs=True,
             opt_cutoff=False, freq_schedule=False, init_freq=0.0, cut_to_dist=False, cutoff_shift=False, multires=7, i_embed=0,
             cutoff_bones=False, multires_bones=0, use_viewdirs=True, cutoff_viewdir=True, multires_views=4, N_importance=16,
             netdepth=8, netwidth=256, opt_framecode=False, framecode_size=16, density_scale=1.0, single_net=False, lrate=5e-4,
             basedir="/nonexistent", expname="x", ft_path=None, no_reload=True, finetune=False, fix_layer=0, debug=False,
             perturb=1.0, N_samples=64, raw_noise_std=1.0, ray_noise_std=0.0, lindisp=False, nerf_type="nerf",
             pts_tr_type="local", kp_dist_type="reldist", bone_type="reldir", view_type="relray", density_type="relu",
             softplus_shift=0.0, weight_decay=None, cutoff_step=250, cutoff_rate=10.0, freq_schedule_step=50)
    a.update(over)
    return argparse.Namespace(**a)


class Skel:
    joint_names = ["j%d" % i for i in range(24)]
    joint_trees = np.asarray(synth.SMPL_PARENTS)


def data_attrs():
    return {"skel_type": Skel, "near": 0.0, "far": 1.0, "n_views": 8, "joint_coords": np.tile(np.eye(3, dtype=np.float32), (24, 1, 1))}


def test_create_raycaster_tuple_and_checkpoint_layout():
    rk_train, rk_test, start, grad_vars, optimizer, ckpt = raycaster.create_raycaster(surreal_args(), data_attrs(), device="cpu")
    assert start == 0 and ckpt is None and len(grad_vars) == 48
    assert sum(p.numel() for p in grad_vars) == 2 * 864260
    assert set(rk_train) == {"ray_caster", "perturb", "N_importance", "N_samples", "use_viewdirs", "raw_noise_std",
                             "ray_noise_std", "ext_scale", "preproc_kwargs", "lindisp", "nerf_type"}
    assert rk_test["perturb"] is False and rk_test["raw_noise_std"] == 0.0
    caster = rk_test["ray_caster"]
    assert rk_train["ray_caster"].module is caster           # trainer.py:265,270,504 reach through .module
    assert rk_train["ray_caster"].sync_gradients() is None    # single process: nothing to reduce
    sd = caster.state_dict()
    assert set(sd) == {"network_fn_state_dict", "network_fine_state_dict", "embed_state_dict", "embedbones_state_dict",
                       "embeddirs_state_dict"}
    ref_names = ["pts_linears.%d.%s" % (i, s) for i in range(8) for s in ("weight", "bias")] + \
        ["alpha_linear.weight", "alpha_linear.bias", "views_linears.0.weight", "views_linears.0.bias",
         "feature_linear.weight", "feature_linear.bias", "rgb_linear.weight", "rgb_linear.bias"]
    assert sorted(sd["network_fn_state_dict"]) == sorted(ref_names)
    assert sd["network_fn_state_dict"]["pts_linears.5.weight"].shape == (256, 688)
    assert sd["network_fn_state_dict"]["views_linears.0.weight"].shape == (128, 904)
    assert sorted(sd["embed_state_dict"]) == ["cutoff_dist", "tau"] and sd["embed_state_dict"]["cutoff_dist"].shape == (24,)
    assert float(sd["embed_state_dict"]["cutoff_dist"][0]) == pytest.approx(0.5)
    assert len(sd["embedbones_state_dict"]) == 0
    # round trip through the reference's checkpoint dict layout (trainer.py:498-505)
    ck = {"global_step": 7, **{k: {n: v.clone() + 1 for n, v in d.items()} for k, d in sd.items()}}
    caster.load_state_dict(ck)
    assert torch.equal(caster.network.pts_linears[0].bias, ck["network_fn_state_dict"]["pts_linears.0.bias"])
    # shape-mismatched tensors are skipped like run_nerf_helpers.filter_state_dict
    ck["network_fine_state_dict"]["rgb_linear.weight"] = torch.zeros(3, 64)
    caster.load_state_dict(ck)


def test_mixamo_and_single_net_variants():
    _, rk, _, gv, _, _ = raycaster.create_raycaster(surreal_args(opt_framecode=True), data_attrs(), device="cpu")
    net = rk["ray_caster"].network
    assert net.views_linears[0].weight.shape == (128, 920) and net.framecodes.codes.weight.shape == (8, 16)
    assert len(gv) == 50
    _, rk, _, gv, _, _ = raycaster.create_raycaster(surreal_args(single_net=True, multires_views=0, N_samples=96, N_importance=48),
                                                    data_attrs(), device="cpu")
    c = rk["ray_caster"]
    assert c.network_fine is c.network and c.single_net and len(gv) == 24
    assert c.network.views_linears[0].weight.shape == (128, 328)


def test_tau_schedule_matches_reference_formula():
    _, rk, *_ = raycaster.create_raycaster(surreal_args(), data_attrs(), device="cpu")
    c = rk["ray_caster"]
    args = surreal_args()
    for step in [0, 1000, 150000, 10 ** 7]:
        c.update_embed_fns(step, args)
        expect = min(20.0 * 10.0 ** (step / 250000.0), 2000.0)       # cutoff_embedder.py:181-183
        assert c.embed_fn.get_tau() == pytest.approx(expect, rel=1e-5)
        assert c.embeddirs_fn.get_tau() == pytest.approx(expect, rel=1e-5)
    assert c.embedbones_fn.get_tau() == 0.0


def test_unsupported_configurations_fail_loudly():
    for bad in [dict(kp_dist_type="relpos"), dict(view_type="rayangle"), dict(bone_type="axisang"),
                dict(multires_bones=2), dict(cutoff_inputs=False), dict(cut_to_dist=True), dict(cutoff_shift=True)]:
        with pytest.raises(NotImplementedError):
            raycaster.create_raycaster(surreal_args(**bad), data_attrs(), device="cpu")
    with pytest.raises(NotImplementedError):
        networks.NeRF(input_ch=360, input_ch_bones=72, input_ch_views=648, use_viewdirs=False)
    with pytest.raises(NotImplementedError):
        networks.get_embedder(7, input_dims=24, cutoff_kwargs={"cutoff": True, "cutoff_inputs": True, "cutoff_dim": 24,
                                                               "dist_inputs": False, "normalize": True})[0]


def test_fused_adam_host_contract():
    """FusedAdam mirrors torch.optim.Adam's surface; on CPU it refuses to run (no fallback for the product path)."""
    optim = importlib.import_module("a-nerf_amd.optim")
    ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(2), requires_grad=False)]
    opt = optim.FusedAdam(ps, lr=5e-4, betas=(0.9, 0.999))
    assert len(opt.param_groups[0]["params"]) == 2 and opt.param_groups[0]["lr"] == 5e-4 and opt.state == {}
    sd = opt.state_dict()
    assert sd["state"] == {} and sd["param_groups"][0]["params"] == [0, 1] and sd["param_groups"][0]["betas"] == (0.9, 0.999)
    torch.optim.Adam([p for p in ps if p.requires_grad], lr=1.0).load_state_dict(sd)      # torch accepts the format
    for p in ps[:2]:
        p.grad = torch.ones_like(p)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        opt.step()
    with pytest.raises(ValueError):
        optim.FusedAdam([ps[2]])
    with pytest.raises(NotImplementedError):
        optim.fused_nerf_loss({"rgb_map": torch.zeros(1, 3), "acc_map": torch.zeros(1)}, torch.zeros(1, 3), loss_fn="BCE")


def test_fused_adam_overlap_refuses_a_second_backward_and_pending_group_state():
    """(a) with overlap on, an early all-reduce that all_reduce_grads() has not consumed means a second backward of
    the same step -- refused before anything is enqueued; step() / zero_grad() join and forget a stale handle.  (b) a state
    loaded while the model is still on the host is returned per group by state_dict(group=g) (save_nerf splits it into the
    reference's optimizer_state_dict / pose_optimizer_state_dict)."""
    optim = importlib.import_module("a-nerf_amd.optim")

    class Work:
        waited = 0

        def wait(self):
            Work.waited += 1
    ps = [torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(6))]
    pose = [torch.nn.Parameter(torch.randn(2, 24, 6))]
    opt = optim.FusedAdam([{"params": ps}, {"params": pose, "step_every": 20}], lr=5e-4)
    opt.enable_overlap()
    opt.check_one_backward()                           # nothing pending: fine
    opt._async = [(Work(), 0, 12), (Work(), 12, 18)]   # the fine network's early all-reduce, then the coarse network's
    with pytest.raises(RuntimeError, match="second backward"):
        opt.check_one_backward()
    opt.zero_grad()                                    # a skipped all_reduce_grads(): the stale handles are joined and dropped
    assert opt._async == [] and Work.waited == 2
    opt.check_one_backward()
    opt.enable_overlap(False)
    opt._async = [(Work(), 0, 12)]
    opt.check_one_backward()                           # overlap off: accumulation over several backwards is allowed
    opt._async = []
    # (b) torch-format state for all three parameters, loaded before the buffers exist
    ref = torch.optim.Adam([{"params": ps}, {"params": pose}], lr=5e-4)
    for p in ps + pose:
        p.grad = torch.ones_like(p)
    ref.step()
    sd = ref.state_dict()
    opt.load_state_dict(sd)
    assert opt._pending is not None
    g0, g1, allg = opt.state_dict(group=0), opt.state_dict(group=1), opt.state_dict()
    assert sorted(g0["state"]) == [0, 1] and sorted(g1["state"]) == [0] and sorted(allg["state"]) == [0, 1, 2]
    assert torch.equal(g1["state"][0]["exp_avg"], sd["state"][2]["exp_avg"]) and g1["param_groups"][0]["params"] == [0]
    assert torch.equal(g0["state"][1]["exp_avg_sq"], sd["state"][1]["exp_avg_sq"])
    torch.optim.Adam(pose, lr=1.0).load_state_dict(g1)                     # what the reference's pose optimiser would load


def test_device_rng_follows_torch_seed_and_separates_streams():
    """ops.DeviceRng (host logic; the draws themselves are pinned against Philox known answers in test_step_glue.py): the key
    follows torch.manual_seed like the torch.rand path it replaces, ranks / instances / deep copies get their own streams, an
    explicit seed pins it, and the state round-trips."""
    import copy
    ops = importlib.import_module("a-nerf_amd.ops")
    torch.manual_seed(1234)
    a, b = ops.DeviceRng(), ops.DeviceRng()
    assert a.seed != b.seed and a.stream_id != b.stream_id            # two instances, same torch seed: different streams
    k0 = a.seed
    a.offset = 17
    torch.manual_seed(99)
    a.follow_torch_seed()
    assert a.seed != k0 and a.offset == 0                             # torch.manual_seed() re-derives the key and restarts the counter
    torch.manual_seed(1234)
    a.follow_torch_seed()
    assert a.seed == k0 and a.offset == 0                             # ... and the same seed reproduces the same stream
    r0, r1 = ops.DeviceRng(stream_id=(0 << 20) | 5), ops.DeviceRng(stream_id=(1 << 20) | 5)
    assert r0.seed != r1.seed                                         # same seed on two ranks: different shards draw different numbers
    c = copy.deepcopy(a)
    assert c.seed != a.seed and c.stream_id != a.stream_id            # a copied caster does not replay its original
    p = ops.DeviceRng().manual_seed(7)
    torch.manual_seed(5)
    kp = p.seed
    p.follow_torch_seed()
    assert p.seed == kp and p.pinned                                  # pinned: no longer follows torch
    p.offset = 3
    q = ops.DeviceRng()
    q.load_state_dict(p.state_dict())
    assert (q.seed, q.offset, q.stream_id, q.pinned) == (p.seed, 3, p.stream_id, True)
    # a checkpoint written by rank 0 and loaded on rank 1 keeps rank 1's own stream (different draws for its shard),
    # continues at the saved offset, and is pinned -- a different torch seed on resume does not restart it
    u = ops.DeviceRng()
    u.offset = 9
    sd = u.state_dict()
    os.environ["RANK"] = "1"
    try:
        v = ops.DeviceRng()
        v.load_state_dict(sd)
    finally:
        os.environ.pop("RANK")
    assert v.stream_id >> 20 == 1 and v.stream_id & 0xFFFFF == u.stream_id & 0xFFFFF and v.seed != u.seed and v.offset == 9
    w = ops.DeviceRng()
    w.load_state_dict(sd)                                             # the writing rank: exact continuation
    assert (w.seed, w.offset, w.stream_id) == (u.seed, 9, u.stream_id) and w.pinned
    torch.manual_seed(4321)
    w.follow_torch_seed()
    assert (w.seed, w.offset) == (u.seed, 9)
