"""The whole training iteration against the REFERENCE's own: tests/golden/trajectory_{surreal,mixamo}.npz hold five iterations
of `core.trainer.Trainer.train_batch` (render -> losses -> backward -> Adam [+ pose optimiser on its cadence] ->
decay_optimizer_lrate -> update_embed_fns; pytest=True randomness) run by tests/golden/gen_golden_trajectory.py in the build
container.  Here the mirror `anerf_amd.trainer.Trainer` drives the HIP path (RayCaster, fused loss, FusedAdam, PoseOptLayer,
pose regulariser) through the same five iterations from the same seeded inputs.

Bars: loss within 5e-6 and every watched parameter within 2 x lr of the reference at every iteration
(Adam moves a parameter by ~lr per step whatever the gradient's size, so a gradient that differs in the last bits around
zero may flip one update's sign: 2 x lr is one such flip; the observed fraction of elements beyond lr / 100 is printed).
CPU part: the schedule mirrors (learning-rate decay, tau) against the same file.
"""
import argparse
import importlib
import json
import os

import numpy as np
import pytest
import torch

raycaster = importlib.import_module("a-nerf_amd.raycaster")
trainer_mod = importlib.import_module("a-nerf_amd.trainer")
optim = importlib.import_module("a-nerf_amd.optim")
pose_opt = importlib.import_module("a-nerf_amd.pose_opt")
networks = importlib.import_module("a-nerf_amd.networks")
synth = importlib.import_module("a-nerf_amd.synth")

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
N_POSES = 8
WATCH = ["pts_linears.0.weight", "pts_linears.5.bias", "alpha_linear.weight", "views_linears.0.bias", "rgb_linear.weight"]
# the generator's cases (tests/golden/gen_golden_trajectory.py CASES / EXTRA), restated
CASES = {
    "surreal": dict(args="surreal", seeds=(11, 12), n=64, poses=[0, 1, 2, 3], ray_seed=21, target_seed=5, mixamo=False),
    "mixamo": dict(args="mixamo", seeds=(21, 22), n=64, poses=list(range(8)), ray_seed=22, target_seed=6, mixamo=True),
    # --freq_schedule: alpha = 0.6 i -- the weight images are re-folded every iteration, closed bands must not move
    "surreal_freq": dict(args="surreal", seeds=(11, 12), n=64, poses=[0, 1, 2, 3], ray_seed=23, target_seed=7, mixamo=False,
                         over=dict(freq_schedule=True, freq_schedule_step=500)),
}


def ref_args(name, **over):
    d = json.load(open(os.path.join(GOLDEN, f"args_{name}.json")))
    d.pop("_config_file")
    d.update(basedir="/nonexistent", decay_unit=1, lrate_decay=5, **over)
    return argparse.Namespace(**d)


class Skel:
    joint_names = ["j%d" % i for i in range(24)]
    joint_trees = np.asarray(synth.SMPL_PARENTS)


def test_schedule_mirrors_reproduce_the_reference_sequences():
    """decay_optimizer_lrate (trainer.py:173-183) on a torch Adam and CutoffEmbedder.update_tau (cutoff_embedder.py:181-183)
    against the lr / tau the reference's trainer reported after each of its five iterations"""
    g = dict(np.load(os.path.join(GOLDEN, "trajectory_surreal.npz")))
    p = torch.nn.Parameter(torch.zeros(3))
    opt = torch.optim.Adam([p], lr=float(g["lrate0"]))
    e_v, _ = networks.get_embedder(7, input_dims=24, cutoff_kwargs={"cutoff": True, "cutoff_dist": 0.5, "cutoff_inputs": True,
                                                                     "cutoff_dim": 24, "dist_inputs": False})
    args = ref_args("surreal")
    for i in range(1, int(g["n_iters"]) + 1):
        p.grad = torch.ones(3)
        opt.step()
        lr, _ = trainer_mod.decay_optimizer_lrate(args.lrate, args.lrate_decay, args.lrate_decay_rate, opt, decay_unit=args.decay_unit)
        assert lr == pytest.approx(float(g[f"it{i}.lrate"]), rel=1e-6)
        assert opt.param_groups[0]["lr"] == lr
        e_v.update_threshold(int(g["global_step_per_iter"]) * i, args.cutoff_step, args.cutoff_rate, args.freq_schedule_step, args.multires - 1)
        assert e_v.get_tau() == pytest.approx(float(g[f"it{i}.tau"]), rel=1e-6)


def _batch(case, dev):
    ro, rd, kp, skts, bones, cyls, which = synth.scene_batch(case["n"], case["poses"], ray_seed=case["ray_seed"], per_ray_pose=True)
    n = case["n"]
    t = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32, device=dev)
    return dict(rays=t(np.stack([ro, rd])), target_s=t(np.random.default_rng(case["target_seed"]).random((n, 3))),
                kp_idx=torch.tensor(np.asarray(which), dtype=torch.int64, device=dev), kp3d=t(kp), bones=t(bones), skts=t(skts),
                cyls=t(cyls), cam_idxs=t(np.asarray(which, dtype=np.float32)), fgs=torch.ones(n, 1, device=dev),
                bgs=torch.ones(n, 3, device=dev))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["surreal", "mixamo", "surreal_freq"])
@pytest.mark.parametrize("tail", ["fused", "torch"])
def test_five_training_iterations_follow_the_reference_trajectory(name, tail):
    case = CASES[name]
    g = dict(np.load(os.path.join(GOLDEN, f"trajectory_{name}.npz")))
    dev = torch.device("cuda")
    args = ref_args(case["args"], **({"opt_pose_step": 2} if case["mixamo"] else {}), **case.get("over", {}))
    assert int(g["opt_pose_step"]) == args.opt_pose_step and int(g["N_samples"]) == args.N_samples
    data_attrs = {"skel_type": Skel, "near": 0.0, "far": 1.0, "n_views": N_POSES, "hwf": (512, 512, 600.0),
                  "joint_coords": np.tile(np.eye(3, dtype=np.float32), (24, 1, 1))}
    rk_train, rk_test, _, grad_vars, torch_opt, _ = raycaster.create_raycaster(args, data_attrs, device=dev)
    caster = rk_test["ray_caster"]
    fc = 16 if args.opt_framecode else 0
    for net, seed in ((caster.network, case["seeds"][0]), (caster.network_fine, case["seeds"][1])):
        net.load_state_dict({k: torch.tensor(v) for k, v in synth.make_net_params(seed, args.multires, args.multires_views, fc, N_POSES).items()})
    rk_train["pytest"] = True
    popt_kwargs = pose_torch_opt = layer = None
    if case["mixamo"]:
        poses = [synth.make_pose(k) for k in range(N_POSES)]
        kps, bones = np.stack([q["kp"] for q in poses]), np.stack([q["bones"] for q in poses])
        layer = pose_opt.PoseOptLayer(kps, bones, (synth.SMPL_REST_POSE * synth.SURREAL_SCALE)[None], use_rot6d=args.opt_rot6d).to(dev)
        anchors = {"kps": torch.tensor(kps), "bones": torch.tensor(bones),
                   "rots": pose_opt.axisang_to_rot(torch.tensor(bones).reshape(-1, 3)).reshape(N_POSES, 24, 3, 3), "beta": None}
        with torch.no_grad():     # the generator's seeded perturbation of the pose parameters (regulariser active)
            layer.bones.add_(torch.tensor((np.random.RandomState(77).randn(*layer.bones.shape) * 0.12).astype(np.float32), device=dev))
        popt_kwargs = {"popt_layer": layer, "popt_anchors": anchors, "skel_type": Skel}
        pose_torch_opt = torch.optim.Adam(layer.parameters(), lr=args.opt_pose_lrate, betas=(0.9, 0.999))
    if tail == "fused":
        groups = [{"params": grad_vars, "lr": args.lrate}]
        if case["mixamo"]:
            groups.append({"params": list(layer.parameters()), "lr": args.opt_pose_lrate, "step_every": args.opt_pose_step})
        fused = optim.FusedAdam(groups, betas=(0.9, 0.999)).attach(caster, pose_layer=layer)     # networks, frame codes, pose: all in place
        opt, popt = fused.group_optimizer(0), (fused.group_optimizer(1) if case["mixamo"] else None)
    else:
        opt, popt = torch_opt, pose_torch_opt
    tr = trainer_mod.Trainer(args, data_attrs, opt, popt, rk_train, rk_test, popt_kwargs=popt_kwargs, device=dev)
    caster.train()
    batch = _batch(case, dev)
    worst = {"loss": 0.0, "param": 0.0, "beyond": 0.0}
    for i in range(1, int(g["n_iters"]) + 1):
        lr_used = float(g["lrate0"]) if i == 1 else float(g[f"it{i - 1}.lrate"])
        loss_dict, stats = tr.train_batch(batch, i=i, global_step=int(g["global_step_per_iter"]) * i)
        G = lambda k: float(g[f"it{i}.{k}"])
        if i == 1:      # train_batch hands back the reference's dictionaries, key for key (trainer.py:262-277)
            assert sorted(loss_dict) == [str(k) for k in g["keys_loss_dict"]] and sorted(stats) == [str(k) for k in g["keys_stats"]]
        d_loss = abs(float(loss_dict["total_loss"]) - G("loss"))
        worst["loss"] = max(worst["loss"], d_loss)
        assert d_loss <= 5e-6, (i, float(loss_dict["total_loss"]), G("loss"))
        assert abs(float(loss_dict["rgb_loss"]) - G("rgb_loss")) <= 5e-6 and abs(float(loss_dict["rgb_loss0"]) - G("rgb_loss0")) <= 5e-6
        assert abs(float(stats["psnr"]) - G("psnr")) <= 1e-3 and abs(float(stats["psnr0"]) - G("psnr0")) <= 1e-3      # dB (north_star)
        assert float(stats["lrate"]) == pytest.approx(G("lrate"), rel=1e-6)
        assert float(stats["cutoff"]) == pytest.approx(G("tau"), rel=1e-6)
        assert caster.embeddirs_fn.get_tau() == pytest.approx(G("tau_d"), rel=1e-6)
        assert float(stats["alpha"]) == pytest.approx(G("alpha_mean"), abs=1e-5)
        if args.freq_schedule:
            assert caster.embed_fn.get_alpha() == G("sched_alpha") and caster.embeddirs_fn.get_alpha() == G("sched_alpha_d")
            # bands >= 3 never opened in these five iterations (alpha <= 3.0 is reached AFTER the last step): their columns of
            # pts_linears.0 saw zero gradients only, so Adam left them exactly where they started
            w0 = synth.make_net_params(case["seeds"][0], args.multires, args.multires_views, fc, N_POSES)["pts_linears.0.weight"]
            assert np.array_equal(caster.network.pts_linears[0].weight.detach().cpu().numpy()[:, 24 + 48 * 3:24 + 48 * 7], w0[:, 24 + 48 * 3:24 + 48 * 7])
        if not case["mixamo"]:     # get_gradnorm before the step (the reference's pose branch reports it after zero_grad: zeros)
            assert float(stats["total_norm"]) == pytest.approx(G("total_norm"), rel=2e-3)
            assert float(stats["avg_norm"]) == pytest.approx(G("avg_norm"), rel=2e-3)
        else:
            assert abs(float(loss_dict["kp_loss"]) - G("kp_loss")) <= 5e-6
            assert float(stats["MPJPC"]) == pytest.approx(G("mpjpc"), rel=1e-4)
        watched = []
        for tag, net in (("c", caster.network), ("f", caster.network_fine)):
            sd = dict(net.named_parameters())
            watched += [(f"{tag}.{w}", sd[w]) for w in WATCH]
            if case["mixamo"]:
                watched.append((f"{tag}.framecodes.codes.weight", sd["framecodes.codes.weight"]))
        if case["mixamo"]:
            watched += [("popt.bones", layer.bones), ("popt.pelvis", layer.pelvis)]
        for key, p in watched:
            ref = g[f"it{i}.{key}"]
            got = p.detach().cpu().numpy()
            got = got if got.size <= 4096 or key.startswith("popt") or "framecodes" in key else got.reshape(-1)[:4096]
            d = np.abs(got.reshape(ref.shape) - ref)
            lr_p = args.opt_pose_lrate if key.startswith("popt") else lr_used
            worst["param"] = max(worst["param"], float(d.max() / lr_p))
            worst["beyond"] = max(worst["beyond"], float((d > 0.01 * lr_p).mean()))
            assert d.max() <= 2.0 * lr_p * 1.001, (i, key, float(d.max()), lr_p)
    print(f"{name}/{tail}: max |dloss| {worst['loss']:.2e}, max parameter distance {worst['param']:.3f} x lr, "
          f"largest fraction of a tensor's elements beyond lr/100: {worst['beyond']:.2e}")
    if case["mixamo"]:      # the pose cadence: stepped at i = 2 and 4 only -> Adam step count 2
        steps = (fused._steps[1] if tail == "fused" else int(pose_torch_opt.state[layer.bones]["step"]))
        assert steps == 2


def _mixamo_trainer(tail, dev, **over):
    """the mixamo case of the trajectory test as a reusable set-up: (trainer, caster, pose layer, fused-or-None, pose torch optimiser)"""
    case = CASES["mixamo"]
    args = ref_args("mixamo", **over)
    data_attrs = {"skel_type": Skel, "near": 0.0, "far": 1.0, "n_views": N_POSES, "hwf": (512, 512, 600.0),
                  "joint_coords": np.tile(np.eye(3, dtype=np.float32), (24, 1, 1))}
    rk_train, rk_test, _, grad_vars, torch_opt, _ = raycaster.create_raycaster(args, data_attrs, device=dev)
    caster = rk_test["ray_caster"]
    for net, seed in ((caster.network, case["seeds"][0]), (caster.network_fine, case["seeds"][1])):
        net.load_state_dict({k: torch.tensor(v) for k, v in synth.make_net_params(seed, args.multires, args.multires_views, 16, N_POSES).items()})
    rk_train["pytest"] = True
    poses = [synth.make_pose(k) for k in range(N_POSES)]
    kps, bones = np.stack([q["kp"] for q in poses]), np.stack([q["bones"] for q in poses])
    layer = pose_opt.PoseOptLayer(kps, bones, (synth.SMPL_REST_POSE * synth.SURREAL_SCALE)[None], use_rot6d=args.opt_rot6d).to(dev)
    anchors = {"kps": torch.tensor(kps), "bones": torch.tensor(bones),
               "rots": pose_opt.axisang_to_rot(torch.tensor(bones).reshape(-1, 3)).reshape(N_POSES, 24, 3, 3), "beta": None}
    with torch.no_grad():
        layer.bones.add_(torch.tensor((np.random.RandomState(77).randn(*layer.bones.shape) * 0.12).astype(np.float32), device=dev))
    popt_kwargs = {"popt_layer": layer, "popt_anchors": anchors, "skel_type": Skel}
    pose_torch_opt = torch.optim.Adam(layer.parameters(), lr=args.opt_pose_lrate, betas=(0.9, 0.999))
    fused = None
    if tail == "fused":
        fused = optim.FusedAdam([{"params": grad_vars, "lr": args.lrate},
                                 {"params": list(layer.parameters()), "lr": args.opt_pose_lrate, "step_every": args.opt_pose_step}],
                                betas=(0.9, 0.999)).attach(caster, pose_layer=layer)
        opt, popt = fused.group_optimizer(0), fused.group_optimizer(1)
    else:
        opt, popt = torch_opt, pose_torch_opt
    tr = trainer_mod.Trainer(args, data_attrs, opt, popt, rk_train, rk_test, popt_kwargs=popt_kwargs, device=dev)
    caster.train()
    return tr, caster, layer, fused, pose_torch_opt


@pytest.mark.gpu
def test_fresh_batches_reach_the_pose_layer_with_their_own_indices():
    """a loader yields NEW tensors every iteration, and the allocator hands the freed kp_idx block back at the same
    address -- the pose layer must still see each batch's own indices (the reference: `kp_idx.cpu().numpy()` per iteration,
    core/trainer.py:299), whether kp_idx arrives on the host (a DataLoader's batch) or already on the device."""
    dev = torch.device("cuda")
    tr, caster, layer, fused, _ = _mixamo_trainer("fused", dev, opt_pose_step=1)
    case = dict(CASES["mixamo"])
    base = {k: v.cpu() for k, v in _batch(case, dev).items()}
    rng = np.random.default_rng(3)
    for i in range(1, 7):
        which = np.sort(rng.choice(N_POSES, size=3 + i % 3, replace=False))
        idx = torch.tensor(which[rng.integers(0, len(which), size=case["n"])], dtype=torch.int64)
        idx[:len(which)] = torch.tensor(which)
        batch = dict(base, kp_idx=idx if i % 2 else idx.to(dev), cam_idxs=idx.to(torch.float32))
        tr.train_batch(batch, i=i, global_step=i)
        assert np.array_equal(layer.last_unique["idxs"], which), (i, layer.last_unique["idxs"], which)
        assert int(layer.last_unique["counts"].sum()) == case["n"]
        del batch, idx


@pytest.mark.gpu
@pytest.mark.parametrize("tail", ["fused", "torch"])
def test_pose_parameters_freeze_at_opt_pose_stop(tail):
    """from iteration opt_pose_stop on the reference never steps pose_optimizer again (trainer.py:441-483, the
    `not popt_detach` guard) -- bones / pelvis and the pose Adam step count stand still although Adam's first moments are not
    zero, in the fused tail as in the torch tail; the network group keeps training."""
    dev = torch.device("cuda")
    tr, caster, layer, fused, pose_torch_opt = _mixamo_trainer(tail, dev, opt_pose_step=1, opt_pose_stop=3)
    batch = _batch(CASES["mixamo"], dev)
    snaps, w = [], []
    for i in range(1, 6):
        tr.train_batch(batch, i=i, global_step=i)
        snaps.append((layer.bones.detach().clone(), layer.pelvis.detach().clone()))
        w.append(caster.network.pts_linears[3].weight.detach().clone())
    steps = fused._steps[1] if tail == "fused" else int(pose_torch_opt.state[layer.bones]["step"])
    assert steps == 2                                                       # iterations 1 and 2 only
    assert not torch.equal(snaps[0][0], snaps[1][0])                       # the pose did move while it was allowed to
    for s in snaps[2:]:
        assert torch.equal(s[0], snaps[1][0]) and torch.equal(s[1], snaps[1][1])
    assert not torch.equal(w[4], w[2])                                      # the networks go on
    if tail == "fused":
        assert fused._steps[0] == 5
        o, n = fused._segments()[1]
        assert float(fused.exp_avg[o:o + n].abs().max()) > 0                # moments are non-zero: a step would have moved the pose


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["surreal", "mixamo"])
def test_graphed_trainer_is_bit_identical_to_the_eager_trainer(name):
    """Trainer.enable_graph(): train_batch replays the device side of the iteration from a captured hipGraph.  Eight iterations on
    FRESH batches (new rays, new targets, a different set of poses with a different number of distinct poses) with the device
    generator's randomness, the reference's schedules moving tau and the learning rate every iteration, the pose cadence
    (opt_pose_step = 2) and the pose-optimisation stop (opt_pose_stop = 6): the same losses, statistics and parameters, bit for
    bit, as the eager trainer."""
    ops = importlib.import_module("a-nerf_amd.ops")
    dev = torch.device("cuda")
    case = CASES[name]
    n_iter, n = 8, 64
    runs = []
    for graph in (False, True):
        torch.manual_seed(3)
        if name == "mixamo":
            tr, caster, layer, fused, _ = _mixamo_trainer("fused", dev, opt_pose_step=2, opt_pose_stop=6)
        else:
            args = ref_args("surreal")
            data_attrs = {"skel_type": Skel, "near": 0.0, "far": 1.0, "n_views": N_POSES, "hwf": (512, 512, 600.0),
                          "joint_coords": np.tile(np.eye(3, dtype=np.float32), (24, 1, 1))}
            rk_train, rk_test, _, grad_vars, _, _ = raycaster.create_raycaster(args, data_attrs, device=dev)
            caster, layer = rk_test["ray_caster"], None
            for net, seed in ((caster.network, 11), (caster.network_fine, 12)):
                net.load_state_dict({k: torch.tensor(v) for k, v in synth.make_net_params(seed).items()})
            fused = optim.FusedAdam([{"params": grad_vars, "lr": args.lrate}], betas=(0.9, 0.999)).attach(caster)
            tr = trainer_mod.Trainer(args, data_attrs, fused.group_optimizer(0), None, rk_train, rk_test, popt_kwargs=None, device=dev)
            caster.train()
        tr.render_kwargs_train["pytest"] = False
        caster._rng = ops.DeviceRng(seed=99, stream_id=7)
        if graph:       # (the Mixamo case also takes the capture mode meant for runs with pin-memory threads beside the trainer)
            # warm_each_key=False: every new variant is captured at first sight -- this test is about many graphs in few iterations
            # (the default, one eager pass per new variant, is counted in tests/test_graph_step.py)
            tr.enable_graph(eager_steps=1, capture_error_mode="thread_local" if name == "mixamo" else "global", warm_each_key=False)
        trace = []
        rng = np.random.default_rng(5)
        for i in range(1, n_iter + 1):
            poses = sorted(rng.choice(N_POSES, size=2 + i % 3, replace=False).tolist())
            ro, rd, kp, skts, bones, cyls, which = synth.scene_batch(n, poses, ray_seed=100 + i, per_ray_pose=True)
            t = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32)
            which = np.asarray(poses)[np.asarray(which)]                  # scene_batch numbers the rays' poses 0..len(poses)-1
            batch = dict(rays=t(np.stack([ro, rd])), target_s=t(np.random.default_rng(200 + i).random((n, 3))),
                         kp_idx=torch.tensor(which, dtype=torch.int64), kp3d=t(kp), bones=t(bones), skts=t(skts), cyls=t(cyls),
                         cam_idxs=t(np.asarray(which, dtype=np.float32)), fgs=torch.ones(n, 1), bgs=torch.ones(n, 3))
            loss_dict, stats = tr.train_batch(batch, i=i, global_step=500 * i)
            trace.append({**{k: v.detach().clone() for k, v in loss_dict.items()},
                          **{k: (v.detach().clone() if torch.is_tensor(v) else torch.tensor(float(v))) for k, v in stats.items()}})
            if layer is not None:
                assert np.array_equal(layer.last_unique["idxs"], np.unique(which))
        torch.cuda.synchronize()
        runs.append(dict(trace=trace, flat=fused.flat.clone(), m=fused.exp_avg.clone(), steps=list(fused._steps), tr=tr))
    e, g = runs
    gs = g["tr"]._gs
    assert gs.eager_calls == 1 and gs.replays == n_iter - 1 and gs.captures >= (3 if name == "mixamo" else 1), (gs.eager_calls, gs.replays, gs.captures)
    assert e["steps"] == g["steps"] == ([n_iter, 2] if name == "mixamo" else [n_iter])       # pose steps at i = 2, 4; stopped from 6 on
    for i, (a, b) in enumerate(zip(e["trace"], g["trace"])):
        assert sorted(a) == sorted(b)
        for k in a:
            assert torch.equal(a[k].cpu(), b[k].cpu()), (i + 1, k, a[k], b[k])
    assert torch.equal(e["flat"], g["flat"]) and torch.equal(e["m"], g["m"])
    print(f"{name}: {gs.captures} graphs for keys {sorted(map(str, gs.graphs))}")


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [False, True])
def test_resumed_run_continues_bit_identically(graph, tmp_path):
    """Checkpoint / resume (SURVEY aux subsystems; trainer.py:485-517, raycasters.py:117-143, pose_opt.py:52-75): 4 iterations ->
    save_nerf -> fresh modules -> load_nerf -> 4 more iterations end on EXACTLY the parameters, Adam moments, pose parameters and
    random-stream position of 8 iterations straight -- networks, frame codes, embedder tau, both optimiser groups' step counts and
    moments, the regulariser anchors and the device generator's offset all travel in the reference's `.tar` layout.  Eager and
    captured-graph trainers."""
    ops = importlib.import_module("a-nerf_amd.ops")
    checkpoint = importlib.import_module("a-nerf_amd.checkpoint")
    dev = torch.device("cuda")
    n = 64

    def make():
        torch.manual_seed(3)
        tr, caster, layer, fused, _ = _mixamo_trainer("fused", dev, opt_pose_step=2)
        tr.render_kwargs_train["pytest"] = False
        caster._rng = ops.DeviceRng(seed=99, stream_id=7)
        if graph:
            tr.enable_graph(eager_steps=1, warm_each_key=False)
        return tr, caster, layer, fused

    def batch_of(i):
        rng = np.random.default_rng(1000 + i)
        poses = sorted(rng.choice(N_POSES, size=3, replace=False).tolist())
        ro, rd, kp, skts, bones, cyls, which = synth.scene_batch(n, poses, ray_seed=100 + i, per_ray_pose=True)
        which = np.asarray(poses)[np.asarray(which)]
        t = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32)
        return dict(rays=t(np.stack([ro, rd])), target_s=t(np.random.default_rng(200 + i).random((n, 3))), kp_idx=torch.tensor(which, dtype=torch.int64),
                    kp3d=t(kp), bones=t(bones), skts=t(skts), cyls=t(cyls), cam_idxs=t(np.asarray(which, dtype=np.float32)), fgs=torch.ones(n, 1),
                    bgs=torch.ones(n, 3))

    # (a) eight iterations straight
    tr, caster, layer, fused = make()
    for i in range(1, 9):
        tr.train_batch(batch_of(i), i=i, global_step=500 * i)
    torch.cuda.synchronize()
    want = dict(flat=fused.flat.clone(), m=fused.exp_avg.clone(), v=fused.exp_avg_sq.clone(), steps=list(fused._steps), offset=caster.rng().offset,
                tau=caster.embed_fn.get_tau())
    # (b) four, checkpoint, fresh modules, four more
    tr, caster, layer, fused = make()
    for i in range(1, 5):
        tr.train_batch(batch_of(i), i=i, global_step=500 * i)
    path = str(tmp_path / "002000.tar")
    tr.save_nerf(path, 2000)
    tr2, caster2, layer2, fused2 = make()
    caster2._rng = ops.DeviceRng(seed=5, stream_id=11)                      # whatever the fresh process had: the checkpoint's stream takes over
    r = checkpoint.load_nerf(path, tr2.render_kwargs_train["ray_caster"], fused2, layer2, None)
    assert r["global_step"] == 2000 and fused2._steps == [4, 2] and caster2.rng().offset == 4
    tr2.popt_kwargs["popt_anchors"] = r["poseopt_anchors"]
    for i in range(5, 9):
        tr2.train_batch(batch_of(i), i=i, global_step=500 * i)
    torch.cuda.synchronize()
    assert fused2._steps == want["steps"] and caster2.rng().offset == want["offset"] and caster2.embed_fn.get_tau() == want["tau"]
    assert torch.equal(fused2.flat, want["flat"]) and torch.equal(fused2.exp_avg, want["m"]) and torch.equal(fused2.exp_avg_sq, want["v"])


def _trainer_rccl_worker(port, q, log_path=None):
    """one rank, RCCL, collectives forced: the Trainer's data-parallel iteration (overlap on: three early collectives per step in
    this Mixamo arrangement, the pose group's own on its cadence, split Adam) eager and through Trainer.enable_graph()"""
    log = open(log_path, "w", buffering=1) if log_path else None

    def stage(msg):
        if log:
            log.write(msg + "\n")

    try:
        import faulthandler
        import os
        if log:      # a hang must say where: all threads' stacks into the log, then exit (the parent reports the log)
            faulthandler.dump_traceback_later(240, exit=True, file=log)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ANERF_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        stage("init_process_group ...")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        warm = torch.zeros(8, device=dev)
        dist.all_reduce(warm)
        torch.cuda.synchronize()
        stage("communicator up, first all-reduce done")
        ops = importlib.import_module("a-nerf_amd.ops")
        n_iter, n = 8, 64
        runs = []
        for graph in (False, True):
            torch.manual_seed(3)
            tr, caster, layer, fused, _ = _mixamo_trainer("fused", dev, opt_pose_step=2, opt_pose_stop=7)
            fused.enable_overlap()
            tr.render_kwargs_train["pytest"] = False
            caster._rng = ops.DeviceRng(seed=99, stream_id=7)
            if graph:
                tr.enable_graph(eager_steps=1, warm_each_key=False)
            trace = []
            rng = np.random.default_rng(5)
            for i in range(1, n_iter + 1):
                poses = sorted(rng.choice(N_POSES, size=3, replace=False).tolist())
                ro, rd, kp, skts, bones, cyls, which = synth.scene_batch(n, poses, ray_seed=100 + i, per_ray_pose=True)
                t = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32)
                which = np.asarray(poses)[np.asarray(which)]
                batch = dict(rays=t(np.stack([ro, rd])), target_s=t(np.random.default_rng(200 + i).random((n, 3))),
                             kp_idx=torch.tensor(which, dtype=torch.int64), kp3d=t(kp), bones=t(bones), skts=t(skts), cyls=t(cyls),
                             cam_idxs=t(np.asarray(which, dtype=np.float32)), fgs=torch.ones(n, 1), bgs=torch.ones(n, 3))
                loss_dict, stats = tr.train_batch(batch, i=i, global_step=500 * i)
                trace.append(loss_dict["total_loss"].detach().clone())
                stage(f"graph={graph} iteration {i} enqueued" + ("" if tr._gs is None else f" (captures {tr._gs.captures}, replays {tr._gs.replays})"))
            torch.cuda.synchronize()
            stage(f"graph={graph} synchronised")
            gs = tr._gs
            runs.append(dict(trace=trace, flat=fused.flat.clone(), m=fused.exp_avg.clone(), stats=dict(fused.overlap_stats),
                             replays=0 if gs is None else gs.replays, captures=0 if gs is None else gs.captures,
                             eager_only=[] if gs is None else list(gs.eager_only.values())))
        e, g = runs
        same = all(torch.equal(a, b) for a, b in zip(e["trace"], g["trace"])) and torch.equal(e["flat"], g["flat"]) and torch.equal(e["m"], g["m"])
        q.put({"same": bool(same), "eager_stats": e["stats"], "replays": g["replays"], "captures": g["captures"], "eager_only": g["eager_only"]})
        stage("result sent")
        faulthandler.cancel_dump_traceback_later()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put({"error": traceback.format_exc()})
        raise


@pytest.mark.gpu
def test_graphed_trainer_with_rccl_collectives_inside_the_capture(tmp_path):
    """Round 6: Trainer.enable_graph() no longer steps aside when a process group is up -- the data-parallel iteration is captured
    with its collectives.  As far as ONE GPU goes: a one-rank RCCL communicator with the collectives forced (sums over one rank are
    the identity; side stream, async work handles and RCCL kernels under capture are the real thing).  Eight train_batch calls on
    fresh batches crossing the pose cadence and opt_pose_stop: losses, parameters and Adam moments bit-identical to the eager
    data-parallel trainer; three early collectives per iteration while the pose layer is refined (fine network, coarse weights,
    coarse frame codes), two after it stopped."""
    import multiprocessing as mp
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    log_path = str(tmp_path / "rccl_worker.log")
    p = ctx.Process(target=_trainer_rccl_worker, args=(port, q, log_path))
    p.start()
    r = importlib.import_module("workers").await_worker(p, q, 300)
    if r is None:
        code = p.exitcode
        if p.is_alive():
            p.kill()
        raise AssertionError(f"the RCCL worker ended without an answer (exit code {code}); its stage log and stacks:\n" + open(log_path).read()[-6000:])
    p.join(timeout=120)
    assert "error" not in r, r.get("error")
    assert r["same"] and not r["eager_only"] and r["replays"] >= 5 and r["captures"] >= 2, r
    assert r["eager_stats"]["early_collectives"] >= 2 * 8, r
