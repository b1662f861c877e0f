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
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(*flags, env=None, timeout=300):
    e = dict(os.environ, **(env or {}))
    e.pop("WORLD_SIZE", None)
    detail = tempfile.NamedTemporaryFile(suffix=".json", delete=False).name
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "3", "--warmup", "1", "--detail", detail,
                        *flags], env=e, capture_output=True, text=True, timeout=timeout)
    r.detail = detail
    return r


def _no_constants(tok):
    raise ValueError(f"non-strict JSON token {tok!r} on the bench line")


def strict_line(stdout):
    """what the driver does with the run: the LAST stdout line, bounded, strict JSON, contract keys present"""
    line = stdout.rstrip("\n").splitlines()[-1]
    assert len(line.encode()) <= 4096, len(line.encode())
    j = json.loads(line, parse_constant=_no_constants)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config"):
        assert k in j, k
    assert "workload" in j["config"]
    return j


def record(r):
    """the stdout line (checked as the driver reads it) merged over the full record of bench_detail.json"""
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout            # ONE JSON line on stdout, whatever gloo prints
    j = strict_line(r.stdout)
    full = json.load(open(r.detail))
    os.unlink(r.detail)
    assert j["detail"] == os.path.basename(r.detail)
    for k, v in j.items():                      # the line is a subset of the detail record (floats rounded to 6 digits)
        if k not in ("detail", "step_ms_median", "host_enqueue_ms_median") and not isinstance(v, (dict, list, float)):
            assert full[k] == v, k
    return {**full, **{k: v for k, v in j.items() if k not in full}}


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
    # round 6: the step's mode is agreed over all ranks (one 1-element all-reduce, MIN), and an even split issues the overlap path's
    # early collectives -- fine network, coarse network (config 4: its weights, then its frame-code table) -- before the rest
    assert j["config"]["graph"] is True and all(c["graph_mode_agreed"] and c["my_capture_succeeded"] for c in j["checks_per_rank"])
    per_step = 0 if n % 8 else (3 if "train_mixamo" in flags else 2)
    assert all(c["early_collectives"] == per_step * (j["steps"] + j["warmup"]) for c in j["checks_per_rank"])


def test_world8_one_failed_capture_sends_every_rank_to_the_eager_step():
    """bench_train's agreement (GraphedTrainStep.agree) at world 8: rank 5 reports a failed capture (test hook) -- EVERY rank then
    runs the eager step, the record says so, and the collectives still add up"""
    j = record(run("--gpus", "8", "--workload", "train_mixamo", "--opt-pose-step", "3", env={"ANERF_BENCH_FAIL_CAPTURE_RANK": "5"}))
    assert j["ranks"] == 8 and j["all_checks_ok"] and j["config"]["graph"] is False
    assert [c["my_capture_succeeded"] for c in j["checks_per_rank"]] == [r != 5 for r in range(8)]
    assert not any(c["graph_mode_agreed"] for c in j["checks_per_rank"])


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
    j = strict_line(r.stdout)
    assert j["collective_bytes"] == 4 * floats and j["backend"].startswith("nccl")
    # a process group is up: --graph auto still replays the step from the captured graph, collectives inside (round 6)
    assert j["config"]["graph"] is True, j["config"]
    if workload == "train":
        # ... unless a rank reports a failed capture: then every rank times the eager step, and the record says so
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--n-rand", "64", "--steps", "2", "--warmup", "1",
                            "--cpu-rays", "0", "--extra", "off"], env=dict(e, ANERF_BENCH_FAIL_CAPTURE_RANK="0"), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        j = strict_line(r.stdout)
        assert j["config"]["graph"] is False and "error" in j["graph"], (j["config"], j.get("graph"))


def test_line_budget_holds_for_a_full_default_record():
    """compact_line on a record shaped like the default run's (headline + 8 extras with per-kernel lists, NaN in a statistic):
    <= 4096 bytes, strict JSON, contract keys + roofline + cpu_baseline + extras_summary survive; detail keys do not."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    stats = {"n": 20, "mean": 19.09467144012451, "median": 19.0357084274292, "p95": 19.1477, "min": 19.01, "max": float("nan"), "over_1p5x_median": 0}
    roof = {"bound": "mfma", "kernel": "k_mlp_fwd<7,4,0,false,false,0>" + " x" * 40, "achieved": 142.81234567, "algorithmic": 142.8, "peak": 157.3,
            "unit": "TFLOP/s", "frac": 0.9079123456, "avg_launch_ms": 201.7, "launch_ms": stats, "flop_per_launch": 28805000000000,
            "traffic": 5.16e8, "traffic_note": "n" * 300, "kernels": [{"kernel": "k", "pass": "fine", "ms": 1.0, "flop": 1, "tflops": 1.0, "frac": float("inf")}] * 8}
    extra = {"workload": "w" * 200, "value": 1.0e6, "unit": "rays/s", "steps": 20, "ms_per_step": 2.93, "dtype": "f32", "roofline": roof,
             "step_ms": stats, "period_ms": stats, "host_enqueue_ms": stats, "slow_steps": [], "graph": True}
    res = {"metric": "rays/sec", "value": 1290000.123456789, "unit": "rays/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 202.4,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "SURREAL-shaped 512x512 frame, 64 samples/ray, forward render (BASELINE config 2)", "rays_per_step": 261121,
                      "samples_per_ray": 64, "n_importance": 0, "parallelism": "ray-sharded x1", "weights": "z" * 100},
           "roofline": roof, "step_ms": stats, "slow_steps": [], "alt_precision": {"x": 1},
           "cpu_baseline": {"value": 1430.0, "unit": "rays/s", "cores": 16, "kind": "port", "sample": "s" * 400},
           "parity": {"max_abs_rgb": 1e-6, "psnr_gpu_db": 10.0, "psnr_oracle_db": 10.0}, "ranks": 1, "backend": "none (single process)",
           "devices": ["cuda:0 AMD Instinct MI355X"], "ms_per_step_per_rank": [202.4], "collective_ms_per_step": [0.0],
           "scaling_model_8gpu": {"config3": {"predicted_speedup_8gpu": 6.7}, "config4": {"predicted_speedup_8gpu": 6.4}},
           "extra_workloads": [extra] * 8,
           "extras_summary": [{"workload": "train_mixamo_384_20", "value": 131000.123, "ms_per_step": 2.93, "frac": 0.632, "step_ms": 2.9,
                               "host_ms": 0.25, "graph": True}] * 8}
    line = b.compact_line(res)
    assert len(line.encode()) <= 4096
    j = json.loads(line, parse_constant=_no_constants)
    assert j["value"] == pytest.approx(1290000.123456789, rel=1e-5) and j["roofline"]["frac"] == pytest.approx(0.9079123, rel=1e-5)
    assert j["cpu_baseline"]["cores"] == 16 and len(j["cpu_baseline"]["sample"]) == 200 and len(j["extras_summary"]) == 8
    assert "extra_workloads" not in j and "kernels" not in j["roofline"] and "traffic_note" not in j["roofline"] and "launch_ms" not in j["roofline"]
    assert j["step_ms_median"] == pytest.approx(19.0357, rel=1e-5)
    # a record too fat even after the whitelist sheds the optional keys, never the contract ones
    res["extras_summary"] = [{"workload": "x" * 100, "value": 1.0}] * 60
    j = json.loads(b.compact_line(res), parse_constant=_no_constants)
    assert "extras_summary" not in j and j["roofline"]["frac"] > 0.9 and j["cpu_baseline"]["kind"] == "port"
