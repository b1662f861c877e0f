"""The per-iteration host glue of ABI revision 3 (anerf_step.hip): one-launch weight-image gathers, the counter-based
random inputs of a caster call, render()'s ray batch, the projected cylinder boxes of render_path, the per-kernel profile
events, and bench.py launching its own ranks.

Checked against: the single-image entry points (pack), a numpy restatement of Philox4x32-10 (Salmon et al., SC'11; random123's
known-answer vectors pin the restatement itself), the reference's torch expression of the ray batch (core/trainer.py:116-135),
and synth.cylinder_bbox (pinned against the reference's cylinder_to_box_2d by tests/test_oracle_golden.py)."""
import importlib
import json
import os
import subprocess
import time
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ops = importlib.import_module("a-nerf_amd.ops")
synth = importlib.import_module("a-nerf_amd.synth")


# ---------------------------------------------------------------------------------------------------------------------
def philox4x32_10(ctr, key):
    """ctr [n,4] uint32, key (k0, k1) -> [n,4] uint32"""
    c = ctr.astype(np.uint64).copy()
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    M0, M1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[:, 0], M1 * c[:, 2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c = np.stack([hi1 ^ c[:, 1] ^ k0, lo1, hi0 ^ c[:, 3] ^ k1, lo0], 1)
        k0 = (k0 + np.uint64(0x9E3779B9)) & MASK
        k1 = (k1 + np.uint64(0xBB67AE85)) & MASK
    return c.astype(np.uint32)


def test_philox_restatement_known_answers():
    """Random123's kat_vectors for philox4x32-10"""
    z = philox4x32_10(np.zeros((1, 4), np.uint32), (0, 0))[0]
    assert [hex(int(v)) for v in z] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    f = philox4x32_10(np.full((1, 4), 0xFFFFFFFF, np.uint32), (0xFFFFFFFF, 0xFFFFFFFF))[0]
    assert [hex(int(v)) for v in f] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    p = philox4x32_10(np.array([[0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344]], np.uint32), (0xa4093822, 0x299f31d0))[0]
    assert [hex(int(v)) for v in p] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_bench_refuses_more_ranks_than_gpus_without_torchrun():
    """`python bench.py --gpus 2` on a box with fewer GPUs: a clear non-zero exit from the self-launcher, not a traceback from a
    half-initialised process group (CPU container: 0 GPUs)."""
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs present")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "ANERF_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 2 and "--gpus 2" in r.stderr and r.stdout.strip() == ""


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_rand_fill_is_philox_and_one_launch_per_call():
    rng = ops.DeviceRng(seed=0x1234567890ABCDEF)
    shapes = [((37, 64), "uniform", 1.0), None, ((37, 64), "normal", 0.25), ((5, 7, 3), "normal", 2.0)]
    for call in range(2):
        outs = rng.fill(shapes, torch.device("cuda"))
        assert outs[1] is None and rng.offset == call + 1
        j = 0
        for sp, t in zip(shapes, outs):
            if sp is None:
                continue
            n = t.numel()
            q = (n + 3) // 4
            ctr = np.zeros((q, 4), np.uint32)
            ctr[:, 0] = np.arange(q)
            ctr[:, 2] = j
            ctr[:, 3] = call
            r = philox4x32_10(ctr, (rng.seed & 0xFFFFFFFF, rng.seed >> 32))      # the key = the generator's derived 64-bit key
            if sp[1] == "uniform":
                want = ((r >> 8).astype(np.float32) * np.float32(2.0 ** -24)).reshape(-1)[:n]
                assert np.array_equal(t.cpu().numpy().reshape(-1), want)          # bit-exact
                assert want.min() >= 0.0 and want.max() < 1.0
            else:
                u1 = ((r[:, 0::2] >> 8).astype(np.float64) + 1.0) * 2.0 ** -24
                u2 = (r[:, 1::2] >> 8).astype(np.float64) * 2.0 ** -24
                rad = np.sqrt(-2.0 * np.log(u1)) * sp[2]
                want = np.stack([rad * np.cos(2 * np.pi * u2), rad * np.sin(2 * np.pi * u2)], -1).reshape(-1)[:n]
                np.testing.assert_allclose(t.cpu().numpy().reshape(-1), want, atol=2e-6 * sp[2] * 6, rtol=2e-6)
            j += 1
    # distribution sanity on a training-sized draw
    u, z = ops.DeviceRng(7).fill([((3072, 80), "uniform", 1.0), ((3072, 80), "normal", 1.0)], torch.device("cuda"))
    assert abs(float(u.mean()) - 0.5) < 2e-3 and abs(float(u.var()) - 1 / 12) < 1e-3
    assert abs(float(z.mean())) < 5e-3 and abs(float(z.var()) - 1.0) < 1e-2 and abs(float((z ** 4).mean()) - 3.0) < 0.1
    # two generators with the same seed AND stream id agree; consecutive calls differ; another stream id (another rank, another
    # caster instance: the default) differs
    a = ops.DeviceRng(5, stream_id=3).fill([((64, 64), "uniform", 1.0)], torch.device("cuda"))[0]
    g = ops.DeviceRng(5, stream_id=3)
    b = g.fill([((64, 64), "uniform", 1.0)], torch.device("cuda"))[0]
    c = g.fill([((64, 64), "uniform", 1.0)], torch.device("cuda"))[0]
    d = ops.DeviceRng(5).fill([((64, 64), "uniform", 1.0)], torch.device("cuda"))[0]
    e = ops.DeviceRng(5).fill([((64, 64), "uniform", 1.0)], torch.device("cuda"))[0]
    assert torch.equal(a, b) and not torch.equal(b, c) and not torch.equal(d, e) and not torch.equal(a, d)


@pytest.mark.gpu
def test_make_ray_batch_equals_the_reference_expression():
    g = torch.Generator(device="cuda").manual_seed(1)
    ro = torch.randn(1000, 3, device="cuda", generator=g)
    rd = torch.randn(1000, 3, device="cuda", generator=g) * 3.0
    for near, far, vd in ((0.0, 1.0, True), (0.5, 7.0, False)):
        cols = [ro, rd, near * torch.ones_like(rd[..., :1]), far * torch.ones_like(rd[..., :1])]     # core/trainer.py:116-135
        if vd:
            cols.append(rd / torch.norm(rd, dim=-1, keepdim=True))
        want = torch.cat(cols, -1)
        got = ops.make_ray_batch(ro, rd, near, far, vd)
        assert got.shape == want.shape
        assert torch.equal(got[:, :8], want[:, :8])
        if vd:
            np.testing.assert_allclose(got[:, 8:].cpu().numpy(), want[:, 8:].cpu().numpy(), rtol=0, atol=1.2e-7)
    assert ops.make_ray_batch(ro[:0], rd[:0]).shape == (0, 11)


@pytest.mark.gpu
def test_pack_params_multi_equals_the_single_image_gathers():
    dev = torch.device("cuda")
    for kw, nc in (({}, 0), ({"framecode_ch": 16}, 8)):
        cfg = ops.PathConfig(**kw)
        P1 = {k: torch.tensor(v, device=dev) for k, v in synth.make_net_params(3, 7, 4, cfg.framecode_ch, nc).items()}
        P2 = {k: torch.tensor(v, device=dev) for k, v in synth.make_net_params(4, 7, 4, cfg.framecode_ch, nc).items()}
        jobs, want = [], []
        for P in (P1, P2):
            for which in (0, 1, 2, 3, 4, 5):
                sf, af, _, _ = ops.layout(cfg, which)
                out = torch.full((sf + af,), float("nan"), device=dev)
                jobs.append((cfg, P, which, out))
                a, b = ops.pack_params(cfg, P, which)
                want.append(torch.cat([a, b]))
        ops.pack_params_multi(jobs)
        for (_, _, which, out), w in zip(jobs, want):
            assert torch.equal(out.view(torch.int32), w.view(torch.int32)), which
    # through the module mirror: one call refreshes every stale image, packed() afterwards is a cache hit with the same bytes
    networks = importlib.import_module("a-nerf_amd.networks")
    kwn = dict(D=8, W=256, input_ch=360, input_ch_bones=72, input_ch_views=648, use_viewdirs=True)
    n1, n2 = networks.NeRF(**kwn).to(dev), networks.NeRF(**kwn).to(dev)
    ref = [torch.cat(n.packed(w)).clone() for n in (n1, n2) for w in (0, 1)]
    with torch.no_grad():
        for n in (n1, n2):
            for p in n.parameters():
                p.add_(0.01)
    networks.prepack([(n, w) for n in (n1, n2) for w in (0, 1)])
    hits = [n._packed[w][1] for n in (n1, n2) for w in (0, 1)]
    ptrs = [n.packed(w)[0].data_ptr() for n in (n1, n2) for w in (0, 1)]
    got = [torch.cat(n.packed(w)) for n in (n1, n2) for w in (0, 1)]
    for h, q, g_, r in zip(hits, ptrs, got, ref):
        assert q == h.data_ptr() and not torch.equal(g_, r)          # cache hit on the buffer prepack() filled; new values
    for n in (n1, n2):
        n._packed.clear()
    again = [torch.cat(n.packed(w)) for n in (n1, n2) for w in (0, 1)]
    for g_, a in zip(got, again):
        assert torch.equal(g_, a)


@pytest.mark.gpu
def test_cyl_bbox_on_the_device_equals_the_host_restatement():
    dev = torch.device("cuda")
    rs = np.random.RandomState(0)
    cyls, c2ws, hwf, off, want = [], [], [], [], []
    for k in range(24):
        pose = synth.make_pose(k)
        cyl = synth.bounding_cylinder(pose["kp"])
        H, W = (512, 512) if k % 3 else (480, 640)
        focal = 600.0 if k % 2 else (550.0, 620.0)
        ang = rs.uniform(-0.6, 0.6)
        c2w = synth.default_c2w(3.0 + 0.5 * rs.rand())
        R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float32)
        c2w[:3, :3] = R
        c2w[:3, 3] = R @ c2w[:3, 3]
        center = None if k % 4 else (W * 0.5 + 3.7, H * 0.5 - 2.2)
        tl, br = synth.cylinder_bbox(cyl, H, W, focal, c2w, center=center)
        want.append([tl[0], tl[1], br[0], br[1]])
        fx, fy = (focal, focal) if np.ndim(focal) == 0 else focal
        cyls.append(cyl.astype(np.float64)); c2ws.append(c2w[:3, :4].astype(np.float64)); hwf.append([H, W, fx, fy])
        off.append([int(W * .5), int(H * .5)] if center is None else [int(center[0]), int(center[1])])
    got = ops.cyl_bbox(torch.tensor(np.stack(cyls), device=dev), torch.tensor(np.stack(c2ws), device=dev),
                       torch.tensor(np.array(hwf, np.float64), device=dev), torch.tensor(np.array(off, np.int32), device=dev))
    assert np.array_equal(got.cpu().numpy(), np.array(want, np.int32))


@pytest.mark.gpu
def test_profile_events_bracket_the_training_kernels():
    """AnerfProfile: the library records the caller's events around each MFMA kernel of the one-call step; the pieces are
    positive, ordered like the work they bracket, and sum to less than the step's own event time."""
    dev = torch.device("cuda")
    d = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32, device=dev)
    n, S, Ni = 512, 64, 16
    ro, rd, kp, skts, bones, cyls, _ = synth.scene_batch(n, [0, 1], ray_seed=2, per_ray_pose=True)
    cfg = ops.PathConfig()
    Pc = {k: d(v) for k, v in synth.make_net_params(11).items()}
    Pf = {k: d(v) for k, v in synth.make_net_params(12).items()}
    ap = importlib.import_module("a-nerf_amd.autograd_path")
    rb = ops.make_ray_batch(d(ro), d(rd))
    shapes = [tuple(Pc[nm + sfx].shape) for nm in ops.PARAM_ORDER for sfx in (".weight", ".bias")]
    prof = ops.Profile()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(2):
        with ops.profiling(prof):
            e0.record()
            out, state = ops.train_forward(cfg, ops.pack_params(cfg, Pc), ops.pack_params(cfg, Pf), rb, d(skts), d(cyls), S, Ni)
            g = {"rgb_map": 2.0 * out["rgb_map"], "rgb0": 2.0 * out["rgb0"]}
            ops.backward(state, g, ops.pack_params(cfg, Pc, which=1)[0], ops.pack_params(cfg, Pf, which=1)[0],
                         ap.perm_tables(cfg, dev), shapes, shapes)
            e1.record()
        torch.cuda.synchronize()
    total = e0.elapsed_time(e1)
    parts = {(k, p): prof.ms(k, p) for k in ("fwd", "bwd", "gemm") for p in (0, 1)}
    assert all(v is not None and v > 0.0 for v in parts.values()), parts
    assert sum(parts.values()) < total
    assert parts[("fwd", 1)] > 0.5 * parts[("fwd", 0)]           # 80 samples per ray against 64
    # without the context manager the structures carry no profile
    out2, state2 = ops.train_forward(cfg, ops.pack_params(cfg, Pc), ops.pack_params(cfg, Pf), rb, d(skts), d(cyls), S, Ni)
    assert not state2["io"].profile
    assert torch.equal(out2["rgb_map"], out["rgb_map"])


def test_bench_self_launch_watchdog_stops_hung_ranks():
    """Ranks that never reach the rendezvous (test hook) must not hold `python bench.py --gpus N` for ever: at the
    ANERF_BENCH_TIMEOUT deadline the launcher stops its children by PID, prints no record and exits 124."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(ANERF_BENCH_BACKEND="gloo", ANERF_BENCH_HANG_RANK="0,1", ANERF_BENCH_TIMEOUT="4")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--cpu-rays", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 124 and r.stdout.strip() == "" and "ANERF_BENCH_TIMEOUT" in r.stderr, (r.returncode, r.stderr[-500:])
    assert time.time() - t0 < 120


@pytest.mark.gpu
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without torchrun: two ranks (gloo, sharing the one GPU of the box), ONE JSON line on stdout
    with n_gpus = ranks = 2, both ranks' devices and wall times, and the collective's time; exit code 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["ANERF_BENCH_BACKEND"] = "gloo"
    for extra in (["--workload", "train", "--n-rand", "256"], ["--workload", "render64x64"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                            "--cpu-rays", "0"] + extra, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, r.stdout
        rec = json.loads(lines[0])
        assert rec["n_gpus"] == 2 and rec["ranks"] == 2 and rec["steps"] == 3 and rec["backend"].startswith("gloo")
        assert len(rec["devices"]) == 2 and len(rec["ms_per_step_per_rank"]) == 2 and len(rec["collective_ms_per_step"]) == 2
        assert rec["value"] > 0 and rec["roofline"]["frac"] > 0
        if extra[1] == "train":
            # --graph auto tries the captured step on every rank; gloo moves the bucket through the host (a stream synchronisation:
            # not capturable), so every rank's capture fails, is undone, and the ranks agree -- one tiny all-reduce outside any
            # capture -- to time the eager step; the record says which mode ran
            assert rec["config"]["graph"] is False and "error" in rec["graph"], (rec["config"], rec.get("graph"))
    # a rank that dies takes the launch down with a non-zero exit code and no record
    env["ANERF_BENCH_FAIL_RANK"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--workload", "render64x64",
                        "--cpu-rays", "0"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode != 0 and r.stdout.strip() == ""


@pytest.mark.gpu
def test_bench_under_torch_distributed_run():
    """The driver's N > 1 command line, verbatim: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr
    127.0.0.1 --master-port P bench.py --gpus 2 --steps K --warmup W` (gloo here: both ranks share the one GPU of the box).
    Exactly one JSON line on the launcher's stdout, from rank 0, with both ranks in it."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["ANERF_BENCH_BACKEND"] = "gloo"
    for extra, rays in ([], 261121), (["--workload", "train", "--n-rand", "256"], 256):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                            "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                            "--warmup", "1", "--cpu-rays", "0"] + extra, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        rec = json.loads(lines[0])
        assert rec["n_gpus"] == 2 and rec["ranks"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1
        assert rec["metric"] == "rays/sec" and rec["scaling"] == "strong" and rec["value"] > 0
        assert len(rec["ms_per_step_per_rank"]) == 2 and rec["config"]["rays_per_step"] == rays


@pytest.mark.gpu
def test_caster_draws_its_randomness_from_the_device_rng():
    """The non-pytest training call (perturb, raw_noise_std, ray_noise_std all on): every random input comes from ONE
    anerf_rand_fill launch of the caster's DeviceRng -- same seed and offset => bit-identical outputs and gradients, the next call
    differs, and the draws do what the reference's do (jittered depths differ from the deterministic ones)."""
    networks = importlib.import_module("a-nerf_amd.networks")
    raycaster = importlib.import_module("a-nerf_amd.raycaster")
    render_mod = importlib.import_module("a-nerf_amd.render")
    d = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32, device="cuda")
    kw = dict(D=8, W=256, input_ch=360, input_ch_bones=72, input_ch_views=648, use_viewdirs=True)
    net_c, net_f = networks.NeRF(**kw), networks.NeRF(**kw)
    net_c.load_state_dict({k: torch.tensor(v) for k, v in synth.make_net_params(11).items()})
    net_f.load_state_dict({k: torch.tensor(v) for k, v in synth.make_net_params(12).items()})
    ck = {"cutoff": True, "cutoff_dist": 0.5, "cutoff_inputs": True, "cutoff_dim": 24}
    e_v, _ = networks.get_embedder(7, input_dims=24, cutoff_kwargs=dict(ck, dist_inputs=False))
    e_b, _ = networks.get_embedder(0, input_dims=72, cutoff_kwargs={"cutoff": False})
    e_d, _ = networks.get_embedder(4, input_dims=72, cutoff_kwargs=dict(ck, dist_inputs=True))
    caster = raycaster.RayCaster(net_c, e_v, e_b, e_d, network_fine=net_f).cuda()
    caster.train()
    ro, rd, kp, skts, bones, cyls, _ = synth.scene_batch(96, [0, 1], ray_seed=7, per_ray_pose=True)

    def run(seed, ray_noise=0.05, perturb=1.0, raw_noise=1.0):
        if seed is not None:
            caster.manual_seed(seed)          # pins the caster's stream to `seed` (same stream id: re-seeding replays it)
        for p in caster.parameters():
            p.grad = None
        out = render_mod.render(64, 64, 75.0, chunk=4096, rays=(d(ro), d(rd)), use_viewdirs=True, ray_caster=caster, kp_batch=d(kp),
                                skts=d(skts), cyls=d(cyls), bones=d(bones), cams=None, subject_idxs=None, N_samples=24, N_importance=8,
                                perturb=perturb, raw_noise_std=raw_noise, ray_noise_std=ray_noise,
                                preproc_kwargs={"density_scale": 1.0, "density_fn": torch.nn.functional.relu})
        (out["rgb_map"].sum() + out["rgb0"].sum()).backward()
        g = torch.cat([p.grad.reshape(-1) for p in caster.parameters() if p.grad is not None])
        return {k: v.detach().clone() for k, v in out.items()}, g.clone()
    a, ga = run(123)
    off = caster._rng.offset
    assert off == 1                                   # one launch for the whole call
    b, gb = run(None)                                 # next call of the same generator: other draws
    c, gc = run(123)                                  # re-seeded: the first call again, bit for bit
    for k in a:
        assert torch.isfinite(a[k]).all(), k
        assert torch.equal(a[k], c[k]), k
    assert torch.equal(ga, gc)
    assert not torch.equal(a["rgb_map"], b["rgb_map"]) and not torch.equal(a["alpha"], b["alpha"])
    det, _ = run(123, ray_noise=0.0, perturb=0.0, raw_noise=0.0)
    assert caster._rng.offset == 0                    # nothing random asked for: no launch, no draw consumed
    assert float((det["rgb_map"] - a["rgb_map"]).abs().max()) > 1e-4


@pytest.mark.gpu
def test_driver_command_prints_one_bounded_strict_line():
    """The driver's N = 1 command, verbatim (`python3 bench.py --gpus 1 --steps 20 --warmup 5`): the LAST stdout line is the only
    one, <= 4096 bytes, strict JSON (no NaN / Infinity tokens), and carries the contract keys with `roofline`, `cpu_baseline` and
    the per-extra summary; the full record is in bench_detail.json.  (BENCH_r04 came back parsed = null on a 20 KB line.)"""
    def no_constants(tok):
        raise ValueError(f"non-strict JSON token {tok!r}")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    detail = os.path.join(ROOT, "bench_detail.json")
    if os.path.exists(detail):
        os.unlink(detail)
    r = subprocess.run(["python3", "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5"], env=env, capture_output=True, text=True,
                       timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.endswith("\n") and len(r.stdout.strip().splitlines()) == 1, r.stdout[-500:]
    line = r.stdout.rstrip("\n").splitlines()[-1]
    assert len(line.encode()) <= 4096, len(line.encode())
    j = json.loads(line, parse_constant=no_constants)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "parity", "ranks", "backend", "extras_summary"):
        assert k in j, k
    assert j["metric"] == "rays/sec" and j["n_gpus"] == 1 and j["steps"] == 20 and j["warmup"] == 5 and j["dtype"] == "f32"
    assert "BASELINE config 2" in j["config"]["workload"] and j["config"]["rays_per_step"] == 261121
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms", "flop_per_launch", "traffic"):
        assert k in j["roofline"], k
    assert j["roofline"]["bound"] == "mfma" and 0.5 < j["roofline"]["frac"] < 1.0 and j["roofline"]["peak"] == 157.3
    # traffic: measured in this very run (two rocprofv3 --pmc passes in child processes) or, if that failed, the committed profile's
    # value with the reason -- either way the record says which
    assert j["roofline"]["traffic"] > 2.8e8 and ("live" in j["roofline"]["traffic_source"] or "pmc_traffic.json" in j["roofline"]["traffic_source"])
    assert j["value"] == pytest.approx(261121 / (j["ms_per_step"] * 1e-3), rel=1e-4)
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in j["cpu_baseline"], k
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["value"] > 0
    ok = [e for e in j["extras_summary"] if "error" not in e]
    # (one failed extra would be visible in the record itself; the LINE is what this test is about)
    assert len(j["extras_summary"]) >= 7 and len(ok) >= len(j["extras_summary"]) - 1 and all(e["frac"] > 0 for e in ok), j["extras_summary"]
    full = json.load(open(detail))
    assert len(full["extra_workloads"]) == len(j["extras_summary"])
    assert any("kernels" in e.get("roofline", {}) for e in full["extra_workloads"])
