"""CPU: `python bench.py --gpus 8 --dry-run` -- the never-run N > 1 path exercised as far as a box without GPUs allows.

bench.py's own self-launcher starts 8 ranks (rendezvous on 127.0.0.1, gloo), each takes its shard by the workload's real
arithmetic, runs the step's collective with its real shape (frame all-gather / ONE all-reduce of the flat gradient bucket), the
barrier + max-over-ranks timing and the record assembly; the kernels are rank-tagged fills, so every rank checks the
collective's result exactly.  Reference: the single-process nn.DataParallel of core/raycasters.py:157 this replaces.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(*flags, env=None, timeout=300):
    e = dict(os.environ, **(env or {}))
    e.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "3", "--warmup", "1", *flags], env=e,
                       capture_output=True, text=True, timeout=timeout)
    return r


def record(r):
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout            # ONE JSON line on stdout, whatever gloo prints
    return json.loads(lines[0])


def test_world8_render_frame_is_sharded_gathered_and_recorded():
    j = record(run("--gpus", "8"))
    assert j["ranks"] == 8 and j["n_gpus"] == 8 and j["dry_run"] is True and j["value"] is None
    assert j["backend"].startswith("gloo")
    sh = j["shards"]
    assert sh[0] == [0, 32641] and sh[-1] == [228487, 261121]          # 261 121 rays: seven shards of 32 641, a ragged last one
    assert sum(b - a for a, b in sh) == 261121 == j["config"]["rays_per_step"]
    assert j["all_checks_ok"] and all(c["all_gather_frame_ok"] and c["gathered_rows"] == 8 * 32641 for c in j["checks_per_rank"])
    assert len(j["ms_per_step_per_rank"]) == 8 and j["ms_per_step"] >= max(j["ms_per_step_per_rank"]) * 0.999


@pytest.mark.parametrize("flags, shard0, floats, reduced", [
    (["--workload", "train"], [0, 384], 2 * 864260, 2 * 864260),                                             # config 3: 3072 -> 384 per rank
    (["--workload", "train", "--n-rand", "3070"], [0, 384], 2 * 864260, 2 * 864260),                         # ragged: last shard 382, weighted
    (["--workload", "train_mixamo", "--opt-pose-step", "3"], [0, 384], 2 * 866436 + 1176, 2 * 866436 + 1176),  # config 4: pose group due at the 3rd step
    (["--workload", "train_mixamo", "--opt-pose-step", "20"], [0, 384], 2 * 866436 + 1176, 2 * 866436),      # pose group accumulates: networks only
])
def test_world8_training_bucket_is_sharded_reduced_and_recorded(flags, shard0, floats, reduced):
    j = record(run("--gpus", "8", *flags))
    assert j["ranks"] == 8 and j["dry_run"] is True and j["all_checks_ok"]
    assert j["shards"][0] == shard0 and sum(b - a for a, b in j["shards"]) == j["config"]["rays_per_step"]
    n = j["config"]["rays_per_step"]
    # shard weights turn the plain 1/world average of per-rank mean gradients into the global-batch mean (parallel.shard_weight)
    assert sum(w * 1.0 for w in j["shard_weights"]) == pytest.approx(8.0, rel=1e-12)
    for (a, b), w in zip(j["shards"], j["shard_weights"]):
        assert w == pytest.approx((b - a) * 8 / n)
    for c in j["checks_per_rank"]:
        assert c["all_reduce_sum_ok"] and c["bucket_floats"] == floats and c["reduced_floats_last_step"] == reduced


def test_world2_and_a_dying_rank():
    j = record(run("--gpus", "2", "--workload", "render64x64"))
    assert j["ranks"] == 2 and j["all_checks_ok"]
    r = run("--gpus", "2", "--workload", "render64x64", env={"ANERF_BENCH_FAIL_RANK": "1", "ANERF_BENCH_PG_TIMEOUT": "20"}, timeout=120)
    assert r.returncode != 0 and "rank 1 exited" in r.stderr and not r.stdout.strip()


@pytest.mark.gpu
@pytest.mark.parametrize("workload, floats", [("train", 2 * 864260), ("train_mixamo", 2 * 866436 + 1176)])
def test_dry_run_bucket_is_the_real_flat_bucket(workload, floats):
    """the bucket the dry run reduces has the size of the flat gradient buffer the real bench all-reduces (`collective_bytes`
    of a one-rank RCCL run of the same workload)"""
    e = dict(os.environ, ANERF_BENCH_FORCE_DIST="1")
    e.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--n-rand", "64", "--steps", "2", "--warmup", "1",
                        "--cpu-rays", "0", "--extra", "off"], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["collective_bytes"] == 4 * floats and j["backend"].startswith("nccl")
