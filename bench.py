#!/usr/bin/env python3
"""bench.py -- rays/sec of the MI355X-native A-NeRF ray-march hot path (BASELINE.json metric).

Workload (BASELINE config 2): synthetic SURREAL-shaped 512x512 frame, 261 121 bbox rays, 64 samples/ray,
forward-only render (ray bounds -> z -> fused encode+MLP -> composite), fp32.  One "step" = one frame.
With --gpus N the frame's rays are split into N contiguous slices (strong scaling: total work fixed), one
process per GPU, and the per-ray outputs are all-gathered over RCCL at the end of every step.

  python bench.py [--gpus N --steps K --warmup W] [--workload render64|hier|hier128|render64x64|train|train_mixamo]

--gpus N > 1 from a plain shell starts N ranks of this script (one per GPU, RCCL, rendezvous on 127.0.0.1); under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it joins the ranks it is given.

Prints ONE strict-JSON line of <= 4 KB on rank 0 (contract keys only; the full record goes to bench_detail.json / --detail and
to stderr) with `roofline` for the dominant kernel
(k_mlp_fwd, MFMA-bound: algorithmic FLOPs / HIP-event time on the launch stream vs the 157.3 TFLOP/s fp32
matrix peak; training workloads: the FLOPs the kernels execute, with a per-kernel list), `cpu_baseline` (the torch-CPU
oracle = port of the reference path, timed on this host's cores over a bounded ray sample of the same frame), who ran
(`ranks`, `backend`, `devices`, per-rank and collective times) and, for the default invocation, `extra_workloads`
(BASELINE configs 3, 4 and 5 under the same clock; summarised per workload on the line) with the 8-GPU scaling model.
Training workloads replay the iteration from ONE captured hipGraph in a single process (--graph; a-nerf_amd/graph_step.py).
"""
import argparse
import importlib
import json
import os
import sys
import time

# dmabuf IPC: RCCL's intra-node transport needs it on this driver (already exported on the GPU boxes; set before the HIP runtime
# starts so that a rank launched by torch.distributed.run from a bare environment has it too)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F_MLP = 2 * 861824          # FLOP per network evaluation of one sample
PEAK_FP32_MFMA = 157.3e12   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
PEAK_BF16_MFMA = 2500e12    # dense bf16 MFMA peak (the bf16x3 path issues 3 bf16 MFMA FLOPs per algorithmic FLOP)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="render64", choices=["render64", "hier", "hier128", "render64x64", "api_render64", "train", "train_mixamo"])
    ap.add_argument("--cpu-rays", type=int, default=16384, help="ray sample for the CPU baseline (0 = skip); 16384 rays = ~10 s of host work")
    ap.add_argument("--n-rand", type=int, default=3072, help="train workload: global rays per step")
    ap.add_argument("--torch-tail", action="store_true",
                    help="train workload: torch loss + torch.optim.Adam + copy-bucket all-reduce instead of the fused kernels")
    ap.add_argument("--opt-pose-step", type=int, default=1,
                    help="train_mixamo: pose parameters are stepped (and their gradients all-reduced) every k-th iteration and "
                         "accumulate in between (trainer.py:476-478; mixamo.txt:48 uses 20).  Default 1 = every step, the more "
                         "expensive schedule")
    ap.add_argument("--extra", default="auto", choices=["auto", "on", "off"],
                    help="append `extra_workloads` (5 steps each of train N_rand=3072, train 384 rays, 64+128 bf16x3 render, each "
                         "with its own roofline) to the record; auto = only for the default invocation (render64, fp32, 1 GPU)")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="train workloads: replay the iteration as ONE captured hipGraph (a-nerf_amd/graph_step.py).  auto = on with the fused tail for "
                         "any number of ranks (the gradient collectives are captured with the step; all ranks fall back to the eager step "
                         "together if any rank's capture fails)")
    ap.add_argument("--dry-run", action="store_true",
                    help="multi-rank plumbing without a GPU: rendezvous (gloo), the real shard arithmetic of the workload, the "
                         "collectives with their real shapes (frame all-gather / gradient-bucket all-reduce), record assembly and "
                         "the watchdog; the kernels are replaced by rank-tagged fills.  The record carries dry_run = true and no value")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3"],
                    help="exact fp32 MFMA, or hi/lo-split bf16 MFMAs (3 per product, fp32 accumulate); train workloads: "
                         "bf16x3 applies to the forward kernel only, backward + weight-gradient GEMM stay fp32")
    ap.add_argument("--live-traffic", default="auto", choices=["auto", "on", "off"],
                    help="measure roofline.traffic in THIS run: two short rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE, WRITE_SIZE; "
                         "separate passes) of the same workload in child processes after the timed steps.  auto = for the default "
                         "invocation (render64, fp32, 1 GPU); otherwise the committed profiles/pmc_traffic.json entry is quoted")
    ap.add_argument("--detail", default=None, help="where the full record goes (default: bench_detail.json next to this script); "
                                                   "stdout carries only the <= 4 KB contract line")
    a = ap.parse_args()
    if a.detail:
        global DETAIL_PATH
        DETAIL_PATH = os.path.abspath(a.detail)
    return a


_JSON_FD = None


def step_stats(ms):
    """per-step HIP-event times of the timed steps: a mean alone cannot tell a hiccup from a regression"""
    a = np.asarray(ms, dtype=np.float64)
    if a.size == 0:
        return None
    med = float(np.median(a))
    return {"n": int(a.size), "mean": float(a.mean()), "median": med, "p95": float(np.percentile(a, 95)), "min": float(a.min()),
            "max": float(a.max()), "over_1p5x_median": int((a > 1.5 * med).sum())}


def outliers(gpu_ms, host_ms=None):
    """steps that took more than 1.5 x the median on the GPU timeline, with the host's enqueue time of the same step (a
    host-side stall -- allocator growth, a collector pass, a blocking copy -- shows as host_ms ~ gpu_ms; a slow kernel does not)"""
    a = np.asarray(gpu_ms, dtype=np.float64)
    if a.size == 0:
        return []
    med = float(np.median(a))
    return [{"step": int(i), "gpu_ms": float(a[i])} | ({"host_ms": float(host_ms[i])} if host_ms is not None else {})
            for i in np.nonzero(a > 1.5 * med)[0]]


class ClockSampler:
    """GPU shader clock (sclk, MHz) and socket power (W) sampled by a HOST thread through amdsmi while a workload runs
    (the bf16x3 kernel moves 9 % between leases under power management; a slow lease must be
    distinguishable from a regression).  Reads sysfs-backed counters only: nothing is enqueued on any stream.  Absent /
    failing amdsmi -> every field None."""

    def __init__(self, device_index=0, period_s=0.05):
        self.idx, self.period, self.samples, self._stop, self._th, self.err = device_index, period_s, [], None, None, None
        self.other = {}

    def _read(self, smi, h):
        clk = pw = None
        try:
            ci = smi.amdsmi_get_clock_info(h, smi.AmdSmiClkType.GFX)
            clk = float(ci.get("clk", ci.get("cur_clk")))
        except Exception as e:
            self.err = self.err or f"clock: {type(e).__name__}: {e}"
        # memory / data-fabric / SoC clocks: a lease that is slow with sclk unchanged shows in the tile kernels' L2 -> LDS weight
        # stream and in the store-heavy training forward first
        for name, typ in (("mclk", "MEM"), ("fclk", "DF"), ("socclk", "SOC")):
            try:
                mi = smi.amdsmi_get_clock_info(h, getattr(smi.AmdSmiClkType, typ))
                self.other.setdefault(name, []).append(float(mi.get("clk", mi.get("cur_clk"))))
            except Exception:
                pass
        try:
            pi = smi.amdsmi_get_power_info(h)
            for k in ("current_socket_power", "average_socket_power", "socket_power"):
                v = pi.get(k)
                if isinstance(v, (int, float)) and v > 0:
                    pw = float(v)
                    break
        except Exception as e:
            self.err = self.err or f"power: {type(e).__name__}: {e}"
        return clk, pw

    def __enter__(self):
        import threading
        try:
            import amdsmi as smi
            try:
                smi.amdsmi_init()
            except Exception:
                pass                                   # torch may have initialised it already
            h = smi.amdsmi_get_processor_handles()[self.idx]
        except Exception as e:
            self.err = f"amdsmi unavailable: {type(e).__name__}: {e}"
            return self
        self._stop = threading.Event()

        def run():
            while not self._stop.is_set():
                self.samples.append(self._read(smi, h))
                self._stop.wait(self.period)
        self._th = threading.Thread(target=run, daemon=True)
        self._th.start()
        return self

    def __exit__(self, *a):
        if self._th is not None:
            self._stop.set()
            self._th.join(timeout=2.0)

    def summary(self):
        clk = [c for c, _ in self.samples if c]
        pw = [p for _, p in self.samples if p]
        f = lambda xs, fn: float(fn(xs)) if xs else None
        return {"samples": len(self.samples), "sclk_mhz_min": f(clk, min), "sclk_mhz_max": f(clk, max), "sclk_mhz_mean": f(clk, np.mean),
                "mclk_mhz_min": f(self.other.get("mclk", []), min), "mclk_mhz_mean": f(self.other.get("mclk", []), np.mean),
                "fclk_mhz_mean": f(self.other.get("fclk", []), np.mean), "socclk_mhz_mean": f(self.other.get("socclk", []), np.mean),
                "power_w_mean": f(pw, np.mean), "power_w_max": f(pw, max), "error": self.err}


_ALLOC_KEYS = ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_sync_all_streams")


def alloc_counters(device):
    """the caching allocator's hipMalloc / hipFree / retry counters: a timed region that grows the pool pays milliseconds per call"""
    st = torch.cuda.memory_stats(device)
    return {k: int(st.get(k, 0)) for k in _ALLOC_KEYS}


def hbm_probe(device, mib=1024, reps=5):
    """device-to-device copy bandwidth (read + write bytes / HIP-event time): one number that tells a lease with slow HBM -- the
    store-heavy training forward drops first on such a box, with the shader clock unchanged -- from a kernel regression"""
    try:
        a = torch.empty(mib * 2 ** 20 // 4, dtype=torch.float32, device=device).normal_()
        b = torch.empty_like(a)
        b.copy_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        gbps = 2 * a.numel() * 4 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del a, b
        torch.cuda.empty_cache()
        return float(gbps)
    except Exception:
        return None


def _flush_c_stdio():
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


LINE_BUDGET = 4096           # bytes of the ONE stdout line (the driver reads a bounded tail of stdout: BENCH_r04 came back
                             # unparsed when the line had grown to 20 KB)
DETAIL_PATH = os.path.join(ROOT, "bench_detail.json")

# the contract keys of the task statement (+ who ran); everything else lives in bench_detail.json
_LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "dry_run", "config", "roofline", "cpu_baseline", "parity", "ranks", "backend", "devices",
              "ms_per_step_per_rank", "collective_ms_per_step", "collective_bytes", "collective_alone_ms", "shards", "shard_weights",
              "all_checks_ok", "graph", "alt_precision", "clocks", "extras_summary", "scaling_model_8gpu", "detail")
_ROOF_KEYS = ("bound", "kernel", "achieved", "algorithmic", "peak", "unit", "frac", "frac_at_observed_clock", "avg_launch_ms", "flop_per_launch",
              "traffic", "traffic_source")
_CONFIG_KEYS = ("workload", "rays_per_step", "samples_per_ray", "n_importance", "parallelism", "opt_pose_step", "chunk", "tail", "graph")
# dropped first (in this order) if a line would still exceed LINE_BUDGET
_OPTIONAL = ("shard_weights", "shards", "devices", "collective_ms_per_step", "ms_per_step_per_rank", "clocks", "alt_precision",
             "scaling_model_8gpu", "extras_summary", "parity")


def _strict(x, nd=6):
    """strict-JSON form of a record: no NaN / Infinity tokens (-> null), numpy scalars -> Python, floats to nd significant digits"""
    if isinstance(x, dict):
        return {str(k): _strict(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_strict(v, nd) for v in x]
    if isinstance(x, (bool, type(None), str)):
        return x
    if isinstance(x, (int, np.integer)):
        return int(x)
    if isinstance(x, (float, np.floating)):
        x = float(x)
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{nd}g}") if nd else x
    return str(x)


def compact_line(res):
    """The ONE stdout line: the contract keys only, <= LINE_BUDGET bytes of strict JSON.  Per-step statistics, per-kernel
    lists, traffic notes, the full extra-workload records and the scaling model's inputs go to bench_detail.json."""
    line = {k: res[k] for k in _LINE_KEYS if k in res}
    if "roofline" in line:
        line["roofline"] = {k: line["roofline"][k] for k in _ROOF_KEYS if k in line["roofline"]}
    if "config" in line:
        line["config"] = {k: line["config"][k] for k in _CONFIG_KEYS if k in line["config"]}
    if "step_ms" in res and res["step_ms"]:
        line["step_ms_median"] = res["step_ms"]["median"]
    if "host_enqueue_ms" in res and res["host_enqueue_ms"]:
        line["host_enqueue_ms_median"] = res["host_enqueue_ms"]["median"]
    if isinstance(line.get("alt_precision"), dict):              # the same frame through the split-bf16 kernels (never `value`)
        ap = line["alt_precision"]
        line["alt_precision"] = {"precision": "bf16x3", "value": ap.get("value"), "ms_per_step": ap.get("ms_per_step"),
                                 "max_abs_rgb_vs_f32": ap.get("max_abs_rgb_vs_f32")}
    if isinstance(line.get("clocks"), dict):
        line["clocks"] = {k: line["clocks"].get(k) for k in ("sclk_mhz_min", "sclk_mhz_max", "sclk_mhz_mean", "mclk_mhz_mean", "power_w_mean", "hbm_copy_GBps")}
    if isinstance(line.get("scaling_model_8gpu"), dict):         # the predictions only; terms and inputs are in the detail file
        sm = line["scaling_model_8gpu"]
        line["scaling_model_8gpu"] = {"note": "model, not a measurement: N = 1 timings + ring terms", "t_hop_us_assumed": sm.get("t_hop_us_assumed")} | {
            k: {"speedup_no_latency_no_skew": v["without_latency_and_skew"]["speedup_8gpu"], "speedup_with_latency": v["with_latency"]["speedup_8gpu"],
                "speedup_with_latency_and_skew": v["with_latency_and_skew"]["speedup_8gpu"], "exposed_collective_ms": v["exposed_collective_ms"],
                "skew_ms": v["skew_ms_p95_minus_median"]} |
               # the two forms the N > 1 command can run, each rank's step measured with the collectives live (one-rank RCCL):
               # [predicted speedup with latency + skew, shard step ms, host enqueue ms]
               {row: [(v.get(row) or {}).get("speedup_8gpu"), (v.get(row) or {}).get("shard_step_ms"), (v.get(row) or {}).get("host_enqueue_ms")]
                for row in ("graphed", "eager_overlap") if isinstance(v.get(row), dict) and "speedup_8gpu" in v[row]}
            for k, v in sm.items() if isinstance(v, dict) and "with_latency" in v}
    if "cpu_baseline" in line and isinstance(line["cpu_baseline"], dict):
        line["cpu_baseline"] = dict(line["cpu_baseline"], sample=str(line["cpu_baseline"].get("sample", ""))[:200])
    line = _strict(line)
    enc = lambda d: json.dumps(d, allow_nan=False, separators=(", ", ": "))
    for k in _OPTIONAL:
        if len(enc(line).encode()) <= LINE_BUDGET:
            break
        line.pop(k, None)
    out = enc(line)
    if len(out.encode()) > LINE_BUDGET:          # cannot happen with the keys above; never print an unparsable line
        line = {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                     "vs_baseline", "dtype", "data") if k in line}
        out = enc(line)
    return out


def emit(res):
    """The ONE JSON line of the bench contract (compact_line), written to the process's original stdout, nothing after it.
    The full record goes to bench_detail.json (--detail) and, as one line, to stderr."""
    _flush_c_stdio()
    full = _strict(res, nd=0)
    try:
        tmp = DETAIL_PATH + f".{os.getpid()}.tmp"
        with open(tmp, "w") as f:
            json.dump(full, f, allow_nan=False, indent=1)
        os.replace(tmp, DETAIL_PATH)
        res = dict(res, detail=os.path.basename(DETAIL_PATH))
    except OSError as e:                          # a read-only checkout: the line still goes out
        sys.stderr.write(f"bench.py: could not write {DETAIL_PATH}: {e}\n")
    sys.stderr.write("bench_detail: " + json.dumps(full, allow_nan=False) + "\n")
    sys.stderr.flush()
    line = (compact_line(res) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, line)


def self_launch(args):
    """`python bench.py --gpus N` from a plain shell (no torchrun; replaces the reference's single-command nn.DataParallel,
    core/raycasters.py:157): start N copies of this script, one rank per GPU, rendezvous on 127.0.0.1.  Rank 0 inherits this
    process's stdout (the ONE JSON line), the other ranks' stdout goes to stderr.  Exit code: 0 only if every rank exits 0;
    when a rank fails the others are stopped (by their exact PIDs) so that a dead peer cannot leave the rest hanging in a
    collective."""
    import socket
    import subprocess
    backend = "gloo" if args.dry_run else os.environ.get("ANERF_BENCH_BACKEND", "nccl")
    if backend == "nccl" and torch.cuda.device_count() < args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but this node has {torch.cuda.device_count()} GPU(s) "
                         f"(RCCL needs one device per rank; ANERF_BENCH_BACKEND=gloo shares one GPU for a control-flow test)\n")
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL's intra-node transport needs it on this driver
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // args.gpus)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    live = list(procs)
    # watchdog: a rank stuck in a rendezvous or a collective must not hold the caller for ever (ANERF_BENCH_TIMEOUT seconds)
    deadline = time.time() + float(os.environ.get("ANERF_BENCH_TIMEOUT", "1800"))
    while live:
        time.sleep(0.1)
        if time.time() > deadline:
            sys.stderr.write(f"bench.py: ranks {[procs.index(p) for p in live]} still running at the ANERF_BENCH_TIMEOUT deadline; stopping them\n")
            for q in live:
                q.terminate()
            t_end = time.time() + 10
            for q in live:
                try:
                    q.wait(max(0.1, t_end - time.time()))
                except subprocess.TimeoutExpired:
                    q.kill()
            return 124
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                sys.stderr.write(f"bench.py: rank {procs.index(p)} exited with code {code}; stopping the other ranks\n")
                for q in live:
                    q.terminate()
                t_end = time.time() + 10
                for q in live:
                    try:
                        q.wait(max(0.1, t_end - time.time()))
                    except subprocess.TimeoutExpired:
                        q.kill()
    return rc


def dist_record(res, dist, device, dt_local, steps, coll_ms, backend):
    """who ran: ranks / backend / devices as the process group saw them, every rank's own wall time, and the HIP-event time
    of the per-step collective.  Collective on every rank (all_gather_object); returns res on rank 0."""
    mine = {"rank": int(os.environ.get("RANK", 0)), "device": f"cuda:{device.index}", "name": torch.cuda.get_device_name(device),
            "ms_per_step": dt_local / max(steps, 1) * 1e3, "collective_ms_per_step": coll_ms}
    if dist is None:
        allr, ranks, bk = [mine], 1, "none (single process)"
    else:
        allr = [None] * dist.get_world_size()
        dist.all_gather_object(allr, mine)
        ranks, bk = dist.get_world_size(), dist.get_backend() + (" (RCCL over xGMI)" if backend == "nccl" else " (control-flow test: ranks share devices)")
    if res is not None:
        res["ranks"] = ranks
        res["backend"] = bk
        res["devices"] = [f"{r['device']} {r['name']}" for r in allr]
        res["ms_per_step_per_rank"] = [r["ms_per_step"] for r in allr]
        res["collective_ms_per_step"] = [r["collective_ms_per_step"] for r in allr]
    return res


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    if os.environ.get("RANK", "") in os.environ.get("ANERF_BENCH_HANG_RANK", "-").split(","):   # test hook: a rank that never
        time.sleep(3600)                                                                         # reaches the rendezvous
    # stdout carries exactly one line, the JSON record.  Libraries write there too -- RCCL prints a five-line version
    # banner through C stdio on communicator creation, gloo its connection notes -- and, being buffered, would land AFTER
    # the record at process exit.  So: keep the original stdout for the record only and point fd 1 at stderr for the run.
    global _JSON_FD
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)
    # Host hygiene for millisecond-scale steps: a full (generation-2) pass of Python's cycle collector over the ~10^6
    # objects that `import torch` leaves behind takes ~40 ms and was landing inside the timed window of the small-batch
    # training runs (2.8 -> 4.7 ms/step).  gc.freeze() after set-up moves those long-lived objects out of the
    # collector's reach; the per-step garbage is still collected.
    import gc
    gc.collect()
    gc.freeze()
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with `python bench.py --gpus {args.gpus}` (self-launching) "
                         f"or torch.distributed.run --nproc-per-node {args.gpus}")
    if args.dry_run:
        return dry_run(args, rank, world)
    # ANERF_BENCH_BACKEND=gloo: smoke-test the multi-rank control flow with all ranks on ONE GPU (RCCL refuses duplicate
    # devices); the driver's multi-GPU runs use the default, nccl (= RCCL), one rank per GPU
    backend = os.environ.get("ANERF_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    # ANERF_BENCH_FORCE_DIST=1: initialise the process group and run every collective even with ONE rank -- exercises the
    # RCCL code path (init with device_id, all_gather_into_tensor, all_reduce) on a single-GPU box
    if world > 1 or os.environ.get("ANERF_BENCH_FORCE_DIST") == "1":
        if world == 1:
            os.environ.setdefault("ANERF_FORCE_COLLECTIVES", "1")      # one rank: still run every collective of the DP path
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)
        import torch.distributed as dist
        import datetime
        pg_timeout = datetime.timedelta(seconds=float(os.environ.get("ANERF_BENCH_PG_TIMEOUT", "600")))   # a dead peer: error, not a hang
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device, timeout=pg_timeout)
        else:
            dist.init_process_group(backend, timeout=pg_timeout)
        # prime the communicator (RCCL builds its rings on the first collective) outside any timed region, whatever --warmup is
        prime = torch.zeros(4 * world, device=device)
        dist.all_reduce(prime)
        if backend == "nccl":
            dist.all_gather_into_tensor(prime, prime[:4].clone())
        torch.cuda.synchronize()
        _flush_c_stdio()          # the communicator banner leaves the C buffers now, not at exit behind the record
        if os.environ.get("ANERF_BENCH_FAIL_RANK") == str(rank):       # test hook: a rank dying mid-run (tests/test_step_glue.py)
            os._exit(3)

    synth = importlib.import_module("a-nerf_amd.synth")
    ops = importlib.import_module("a-nerf_amd.ops")
    pipeline = importlib.import_module("a-nerf_amd.pipeline")
    if args.workload in ("train", "train_mixamo"):
        res = bench_train(args, rank, world, device, dist, synth, mixamo=args.workload == "train_mixamo")
    elif args.workload == "api_render64":
        if world != 1:
            raise SystemExit("api_render64 is the single-process drop-in route (render() -> batchify_rays -> RayCaster); use render64 with --gpus N")
        res = bench_api_render(args, device, synth)
    else:
        res = bench_render(args, rank, world, device, dist, synth, ops, pipeline)
    if rank == 0:
        want_live = args.live_traffic == "on" or (args.live_traffic == "auto" and args.workload == "render64" and args.precision == "fp32"
                                                 and args.extra != "off")
        if want_live and world == 1 and dist is None and args.workload in ("render64", "hier", "hier128", "render64x64"):
            kern = "k_mlp_fwd_b3" if args.precision == "bf16x3" else "k_mlp_fwd"
            tr, why = live_traffic(["--workload", args.workload, "--precision", args.precision], kern)
            if tr is not None:
                quoted = res["roofline"].get("traffic")
                res["roofline"]["traffic"] = tr["hbm_bytes"]
                res["roofline"]["traffic_source"] = "live: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE passes of this run (FETCH x2)"
                res["roofline"]["traffic_live"] = dict(tr, committed_profile_value=quoted)
            else:
                res["roofline"]["traffic_source"] = f"profiles/pmc_traffic.json (live measurement failed: {why})"
        elif res["roofline"].get("traffic") is not None:
            res["roofline"]["traffic_source"] = "profiles/pmc_traffic.json (committed rocprofv3 --pmc passes)"
        want_extra = args.extra == "on" or (args.extra == "auto" and args.workload == "render64" and args.precision == "fp32")
        if want_extra and world == 1:
            ex = extra_workloads(args, device, synth, ops, pipeline)
            if dist is None:
                res["scaling_model_8gpu"] = scaling_model(ex, device, args, synth)
            summ = []
            for e in ex:
                key = e.pop("_key", None)
                if e.get("workload", "").startswith("reference-shaped") and "value" in e:
                    e["vs_headline_kernel_bench"] = e["value"] / res["value"]
                # <= ~150 bytes per extra on the stdout line; the full records are bench_detail.json's `extra_workloads`
                tag = "?" if key is None else "_".join(str(k) for k in key if k is not None)
                summ.append({"workload": tag, "error": e["error"][:80]} if "error" in e else
                            {"workload": tag, "value": e["value"], "ms_per_step": e["ms_per_step"], "frac": e["roofline"]["frac"]} |
                            ({"step_ms": e["step_ms"]["median"]} if e.get("step_ms") else {}) |
                            ({"host_ms": e["host_enqueue_ms"]["median"]} if e.get("host_enqueue_ms") else {}) |
                            ({"graph": bool(e["graph"]) and "error" not in e["graph"]} if "graph" in e else {}) |
                            ({"sclk_mhz": [e["clocks"]["sclk_mhz_min"], e["clocks"]["sclk_mhz_max"]]} if (e.get("clocks") or {}).get("sclk_mhz_min") else {}))
            res["extra_workloads"] = ex
            res["extras_summary"] = summ
        emit(res)
    if dist is not None:
        dist.destroy_process_group()


def dry_run(args, rank, world):
    """`--dry-run`: everything of a multi-rank run except the kernels, on the CPU over gloo -- what can be exercised of the N > 1
    path where no second GPU exists.  Same rendezvous (env RANK / WORLD_SIZE / MASTER_*), the workload's REAL shard arithmetic
    (render: the 261 121-ray frame in ceil-sized contiguous slices, ragged last shard; training: parallel.shard_rays /
    shard_weight of N_rand), the step's collective with its real shape and dtype (all-gather of the padded [per, 5] per-ray
    outputs / ONE all-reduce of the flat gradient bucket of both networks [+ frame codes + pose parameters], laid out by the
    same segment rule as FusedAdam), the barrier + max-over-ranks timing protocol and the record assembly.  Each rank's
    "kernel" is a fill with rank-tagged values, so the collectives' results are checked exactly on every rank."""
    import datetime
    import torch.distributed as dist
    for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1")):
        os.environ.setdefault(k, v)
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=float(os.environ.get("ANERF_BENCH_PG_TIMEOUT", "600"))))
    if os.environ.get("ANERF_BENCH_FAIL_RANK") == str(rank):
        os._exit(3)
    synth = importlib.import_module("a-nerf_amd.synth")
    parallel = importlib.import_module("a-nerf_amd.parallel")
    train = args.workload in ("train", "train_mixamo")
    checks = {}
    if train:
        N, mixamo = args.n_rand, args.workload == "train_mixamo"
        lo, hi = parallel.shard_rays(N, rank, world)
        w = parallel.shard_weight(N, rank, world)
        n_net = 864260 + (8 * 16 if mixamo else 0) + (128 * 16 if mixamo else 0)     # + frame codes [8,16] + 16 view-layer columns
        seg = [2 * n_net] + ([8 * 3 + 8 * 24 * 6] if mixamo else [])                 # group 1: pelvis [8,3] + rot6d bones [8,24,6]
        offs, o = [], 0
        for n in seg:                                                                # FusedAdam._segments: 16-byte padded groups
            offs.append((o, (n + 3) // 4 * 4))
            o += (n + 3) // 4 * 4
        bucket = torch.zeros(o)
        units, name = N, f"training step plumbing, N_rand={N}" + (" (config 4 bucket: + frame codes + pose group)" if mixamo else " (config 3 bucket)")
        # the step's MODE, agreed as bench_train agrees it (GraphedTrainStep.agree): every rank reports "my captures succeeded" through
        # ONE 1-element all-reduce (MIN) outside the step; if any rank failed, all ranks run the eager step.  (ANERF_BENCH_FAIL_CAPTURE_RANK:
        # the test hook bench_train has -- that rank reports a failed capture.)
        cap_ok = os.environ.get("ANERF_BENCH_FAIL_CAPTURE_RANK") != str(rank)
        flag = torch.tensor([1.0 if cap_ok else 0.0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        graph_mode = bool(float(flag.item()) >= 1.0)
        n_early = [0]
        # the collectives of ONE step in the order the overlap path issues them (autograd_path._RenderRaysFn.backward): the fine
        # network's range from behind the fine pass; the coarse network's -- in configurations with input gradients its 24 weight /
        # bias tensors from behind the GEMM, then its frame-code table from behind the input-gradient kernel (AnerfBackwardIO.passes =
        # 16 / 32) --; what is left of the due groups (the pose group on its iteration) in all_reduce_grads()
        codes = 8 * 16 if mixamo else 0
        early = [(n_net, 2 * n_net)] + ([(0, n_net - codes), (n_net - codes, n_net)] if mixamo else [(0, n_net)])

        def step(i):
            bucket.fill_(float(rank + 1) * (hi - lo) / max(N, 1))                    # this rank's mean-loss gradient x its ray share
            due = [0] + ([1] if mixamo and (i + 1) % args.opt_pose_step == 0 else [])
            lo_e, hi_e = offs[due[0]][0], offs[due[-1]][0] + offs[due[-1]][1]
            even = N % world == 0                                                    # (ragged shards: weighted, one collective, no overlap)
            done = []
            if even:
                for a, b in early:
                    dist.all_reduce(bucket[a:b])
                    done.append((a, b))
                n_early[0] += len(early)
            covered = max((b for _, b in done), default=lo_e) if done else lo_e
            if covered < hi_e:
                dist.all_reduce(bucket[covered:hi_e])                                # ONE collective over the rest of what is due
            return lo_e, hi_e
        expect = sum(float(r + 1) * (min(N, (r + 1) * ((N + world - 1) // world)) - min(N, r * ((N + world - 1) // world))) / max(N, 1)
                     for r in range(world))
    else:
        H, W, focal = (64, 64, 75.0) if args.workload == "render64x64" else (512, 512, 600.0)
        sc = synth.make_scene(0, H, W, focal)
        N = len(sc["rays_o"])
        per = (N + world - 1) // world
        lo, hi = parallel.shard_rays(N, rank, world)                                 # as bench_render
        w = 1.0
        gather_buf = torch.empty(world * per, 5)
        units, name = N, f"render plumbing, {H}x{W} frame, {N} bbox rays"

        def step(i):
            mine = torch.zeros(per, 5)
            mine[:hi - lo] = float(rank + 1)
            dist.all_gather(list(gather_buf.view(world, per, 5).unbind(0)), mine)
            return 0, world * per
    for _ in range(args.warmup):
        step(0)
    dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        span = step(i)
    dist.barrier()
    dt_local = time.perf_counter() - t0
    tt = torch.tensor([dt_local])
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    # exact checks of the last step's collective on EVERY rank
    if train:
        # (a ring all-reduce sums each chunk in its own rank order: elements may differ in the last bit for ragged shares)
        ok = float((bucket[span[0]:span[1]] - expect).abs().max()) < 1e-5 * max(1.0, expect)
        checks = {"all_reduce_sum_ok": ok, "expected": expect, "got": float(bucket[span[0]]), "bucket_floats": int(bucket.numel()),
                  "reduced_floats_last_step": int(span[1] - span[0]), "graph_mode_agreed": graph_mode, "my_capture_succeeded": cap_ok,
                  "early_collectives": int(n_early[0])}
    else:
        frame = gather_buf[:N]                       # the assembled frame: row r belongs to rank r // per
        owner = torch.arange(N) // per
        ok = bool(torch.all(frame[:, 0] == (owner + 1).float()).item())
        pad_ok = bool(torch.all(gather_buf.view(world, per, 5)[-1, (N - (world - 1) * per):] == 0).item()) if world * per > N else True
        checks = {"all_gather_frame_ok": ok and pad_ok, "gathered_rows": int(world * per), "frame_rows": int(N)}
    mine = {"rank": rank, "shard": [int(lo), int(hi)], "rays": int(hi - lo), "weight": float(w), "ms_per_step": dt_local / max(args.steps, 1) * 1e3,
            "checks": checks}
    allr = [None] * world
    dist.all_gather_object(allr, mine)
    if rank == 0:
        assert [r["rank"] for r in allr] == list(range(world))
        assert sum(r["rays"] for r in allr) == units and all(a["shard"][1] == b["shard"][0] for a, b in zip(allr, allr[1:]))
        res = {"metric": "rays/sec", "value": None, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": float(tt.item()) / max(args.steps, 1) * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic", "dry_run": True,
               "config": {"workload": "DRY RUN (no kernels, CPU, gloo): " + name, "rays_per_step": units,
                          "parallelism": f"ray-sharded x{world}" + (", 1 all-reduce/step" if train else ", 1 all-gather/step")} |
                         ({"graph": all(r["checks"]["graph_mode_agreed"] for r in allr)} if train else {}),
               "ranks": dist.get_world_size(), "backend": dist.get_backend() + " (dry run: rendezvous, sharding, collectives, record; no kernels)",
               "shards": [r["shard"] for r in allr], "shard_weights": [r["weight"] for r in allr],
               "ms_per_step_per_rank": [r["ms_per_step"] for r in allr], "checks_per_rank": [r["checks"] for r in allr],
               "all_checks_ok": all(all(v for k, v in r["checks"].items() if k.endswith("_ok")) for r in allr)}
        emit(res)
    dist.destroy_process_group()


def extra_workloads(args, device, synth, ops, pipeline):
    """BASELINE configs 3, 4 and 5 and the drop-in route inside the same driver-timed run: >= 20 timed steps each, same timing
    protocol, each with its own roofline (training: executed-FLOP, per kernel, priced at the MEDIAN step) and per-step
    statistics (median / p95 / max, slow steps listed).  Config 4 (Mixamo: frame codes + pose refinement) runs at the pose
    cadence of mixamo.txt:48 (opt_pose_step = 20) and at 1 (every step: the most expensive schedule), and at its 8-GPU shard
    size.  `api_render64` is config 2 through the reference-shaped API with the reference's own stride-0 expanded inputs.
    Compact records; the full ones come from `--workload ...`."""
    import copy
    out = []
    for over in (dict(workload="api_render64"),
                 dict(workload="train", n_rand=3072), dict(workload="train", n_rand=384),
                 dict(workload="train_mixamo", n_rand=3072, opt_pose_step=1),
                 dict(workload="train_mixamo", n_rand=3072, opt_pose_step=20),
                 dict(workload="train_mixamo", n_rand=384, opt_pose_step=20),
                 dict(workload="hier128", precision="bf16x3")):
        a = copy.copy(args)
        a.steps, a.warmup, a.cpu_rays, a.extra = max(20, args.steps), max(1, min(args.warmup, 3)), 0, "off"
        for k, v in over.items():
            setattr(a, k, v)
        try:
            if a.workload in ("train", "train_mixamo"):
                r = bench_train(a, 0, 1, device, None, synth, mixamo=a.workload == "train_mixamo")
            elif a.workload == "api_render64":
                r = bench_api_render(a, device, synth)
            else:
                r = bench_render(a, 0, 1, device, None, synth, ops, pipeline)
            rec = {"workload": r["config"]["workload"] + (f", opt_pose_step={a.opt_pose_step}" if a.workload == "train_mixamo" else ""),
                   "value": r["value"], "unit": r["unit"], "steps": a.steps,
                   "ms_per_step": r["ms_per_step"], "dtype": r["dtype"], "roofline": r["roofline"]}
            for k in ("step_ms", "period_ms", "host_enqueue_ms", "slow_steps", "allocator_in_timed_region",
                      "gc_collections_in_timed_region", "vs_headline_kernel_bench", "graph", "clocks"):
                if k in r:
                    rec[k] = r[k]
            train = a.workload in ("train", "train_mixamo")
            rec["_key"] = (a.workload + ("_bf16x3" if a.precision == "bf16x3" else ""), a.n_rand if train else None,
                           a.opt_pose_step if a.workload == "train_mixamo" else (1 if train else None))
            out.append(rec)
        except Exception as e:       # an extra must never take the headline record down with it
            out.append({"workload": str(over), "error": f"{type(e).__name__}: {e}"})
        torch.cuda.empty_cache()
    return out


def scaling_model(extras, device, args=None, synth=None):
    """What the N = 1 run can say about the 8-GPU strong-scaling target before a SCALE run exists: the 384-ray shard step
    measured here (= each rank's compute at N = 8), the gradient bucket's all-reduce on a ONE-rank RCCL communicator (the fixed
    launch + kernel cost of the collective; no xGMI traffic) and a ring model for the wire time (2 (G-1)/G x bytes over one
    xGMI link at ~153 GB/s -- RCCL's rings over the 7-link mesh can only be faster), a per-hop latency term, a rank-skew term and
    what this round's schedule hides of each network's collective.  A model on file, not a measurement."""
    by = {tuple(e["_key"]): e for e in extras if "_key" in e}
    out = {"note": "prediction from single-GPU measurements + a ring model; the measured curve is the driver's SCALE_rNN.json"}
    bucket_bytes = 2 * 864260 * 4
    coll1 = None
    try:
        import torch.distributed as dist
        created = False
        if not dist.is_initialized():
            import socket
            with socket.socket() as sck:
                sck.bind(("127.0.0.1", 0))
                port = sck.getsockname()[1]
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=device)
            created = True
        buf = torch.zeros(bucket_bytes // 4, device=device)
        for _ in range(3):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            dist.all_reduce(buf)
        e1.record()
        torch.cuda.synchronize()
        coll1 = e0.elapsed_time(e1) / 20
        if created and args is None:
            dist.destroy_process_group()
            created = False
        _flush_c_stdio()
    except Exception as e:
        out["collective_world1_error"] = f"{type(e).__name__}: {e}"
    # ---- the terms.  Per NETWORK (the bucket is reduced as two halves, see FusedAdam.enable_overlap):
    #   launch   the collective on a ONE-rank communicator, measured above (enqueue + kernel, no wire)
    #   latency  2 (G-1) ring steps x t_hop.  t_hop = 3 us ASSUMED: RCCL's LL / LL128 protocols move a small chunk per step with a
    #            flag-polling handshake; public rccl-tests all_reduce_perf figures for 8 x MI300X put the small-message (<= 64 KiB)
    #            all-reduce at 25-45 us = 14 steps x 1.8-3.2 us.  Nothing here can measure it (one GPU)
    #   wire     2 (G-1)/G x bytes over ONE xGMI link at 153 GB/s (a ring uses one link per direction; the 7-link mesh is idle otherwise)
    #   skew     p95 - median of the step period at the shard size, measured in this run: with 8 ranks in lockstep each step waits
    #            for the slowest one (the max of 8 draws sits near the p90 of one rank's distribution)
    # What is hidden: the FINE network's collective runs under the coarse backward (>= 1 ms: hidden whatever the terms).  The COARSE
    # network's starts when its parameter gradients are enqueued (AnerfBackwardIO.passes = 4): in config 4 the pose-gradient tail runs
    # on beside it and is not reduced on the 19 of 20 iterations that do not step the pose group.  Round 6: with the encoding's backward
    # inside k_mlp_bwd_in_enc that tail is k_pose_reduce + the pose layer's backward = 8.5 + 4.9 + 40.1 = 53 us at 384 rays
    # (profiles/r06_train_mixamo384_step_timeline_graph_c.txt; it was 115-127 us with k_encode_bwd in it: the step got 60 us shorter
    # and the window 62 us narrower) -- which is why the coarse WEIGHTS' collective now starts a kernel earlier, behind the GEMM (see the
    # loop below); in config 3 there is no tail, only the first network's share of k_adam (~5 us) runs under it.
    G, link, t_hop_ms = 8, 153e9, 3.0e-3
    net_bytes = bucket_bytes // 2
    wire_ms = 2 * (G - 1) / G * net_bytes / link * 1e3
    lat_ms = 2 * (G - 1) * t_hop_ms
    launch_ms = coll1 or 0.0
    coll_net_ms = launch_ms + lat_ms + wire_ms
    out.update({"collective_bytes": bucket_bytes, "collective_world1_ms": coll1, "xgmi_link_GBps_assumed": link / 1e9, "t_hop_us_assumed": t_hop_ms * 1e3,
                "per_network_collective_ms": {"launch": launch_ms, "latency_2(G-1)_hops": lat_ms, "ring_wire": wire_ms, "total": coll_net_ms}})
    # ---- each rank's step at N = 8, measured with the collectives LIVE on the one-rank communicator above, in BOTH forms the N > 1
    # command can take: the captured step (`--graph auto`'s default for any world since round 6) and the eager
    # overlap step every rank falls back to if a capture fails.  The shard runs of `extra_workloads` (no process group: no
    # collective kernels, no side stream) stay in the record as `shard_step_ms_no_collectives`.
    live = {}
    if args is not None and coll1 is not None:
        import copy
        import torch.distributed as dist
        prev = os.environ.get("ANERF_FORCE_COLLECTIVES")
        os.environ["ANERF_FORCE_COLLECTIVES"] = "1"
        try:
            for wl, every in (("train", 1), ("train_mixamo", 20)):
                for mode in ("on", "off"):
                    a = copy.copy(args)
                    a.workload, a.n_rand, a.opt_pose_step, a.graph = wl, 384, every, mode
                    # (60 steps: the skew term is p95 - median of the step period, and with 20 periods the p95 IS the maximum)
                    # 25 untimed steps (~60 ms): each capture is preceded by a 0.3 s pause that lets the process group's watchdog retire
                    # the eager collectives (graph_step.drain_process_group_watchdog) -- the device idles, its clocks drop, and the first
                    # steps after it belong to the ramp, not to the steady state a long run is in (seen as skew 103 us instead of 14-71)
                    a.steps, a.warmup, a.cpu_rays, a.extra, a.precision = max(60, args.steps), 25, 0, "off", "fp32"
                    try:
                        r = bench_train(a, 0, 1, device, dist, synth, mixamo=wl == "train_mixamo", per_kernel=False)
                        per = r.get("period_ms") or r["step_ms"]
                        live[(wl, mode)] = {"step_ms": r["step_ms"]["median"], "skew_ms": max(0.0, per["p95"] - per["median"]),
                                            "host_ms": r["host_enqueue_ms"]["median"], "graph": bool(r["config"].get("graph")),
                                            "overlap": r.get("overlap")}
                    except Exception as e:
                        live[(wl, mode)] = {"error": f"{type(e).__name__}: {e}"[:200]}
                    torch.cuda.empty_cache()
        except Exception as e:
            out["live_collective_runs_error"] = f"{type(e).__name__}: {e}"[:200]
        finally:
            if created:
                dist.destroy_process_group()
            _flush_c_stdio()
            if prev is None:
                os.environ.pop("ANERF_FORCE_COLLECTIVES", None)
            else:
                os.environ["ANERF_FORCE_COLLECTIVES"] = prev
    # Round 6, second step: in config 4 the coarse network's WEIGHT all-reduce (3.46 MB, the full `coll_net_ms`) starts behind the GEMM
    # (AnerfBackwardIO.passes = 16) and has the input-gradient kernel + the frame-code kernels + the pose tail over it: 183 + 14 + 53 =
    # 250 us (profiles/r06_train_mixamo384_step_timeline_graph_d.txt); what starts behind the input gradients (passes = 32) is the
    # frame-code table alone -- a latency-only collective (launch + 2 (G-1) hops) under the 53 us pose tail.
    for name, full, shard, n, window_ms, pose_every, codes_window_ms in (
            ("config3", ("train", 3072, 1), ("train", 384, 1), 3072, 0.005, 0, None),
            ("config4_opt_pose_step20", ("train_mixamo", 3072, 20), ("train_mixamo", 384, 20), 3072, 0.250, 20, 0.053)):
        if full in by and shard in by:
            t1, t8 = by[full]["step_ms"]["median"], by[shard]["step_ms"]["median"]
            per = by[shard].get("period_ms") or by[shard]["step_ms"]
            skew = max(0.0, per["p95"] - per["median"])
            small = (launch_ms + lat_ms) / pose_every if pose_every else 0.0      # the pose group's own collective, on the iterations it is due
            exp_old = launch_ms + wire_ms                                         # round-4 model: no latency, no skew, coarse half exposed
            exp_lat_seq = coll_net_ms                                             # + latency, the coarse collective behind the backward (round 4's schedule)
            exp_lat = max(0.0, coll_net_ms - window_ms) + small                   # + latency, this round's schedule
            if codes_window_ms is not None:
                exp_lat += max(0.0, launch_ms + lat_ms - codes_window_ms)         # the frame-code table's own small collective
            mk = lambda e, t=None: {"step_ms_8gpu": (t or t8) + e, "speedup_8gpu": t1 / ((t or t8) + e), "rays_per_s_8gpu": n / (((t or t8) + e) * 1e-3)}
            out[name] = {"step_ms_1gpu": t1, "shard_step_ms_no_collectives": t8, "speedup_before_allreduce": t1 / t8,
                         "hidden_window_ms": window_ms, "skew_ms_p95_minus_median": skew, "exposed_collective_ms": exp_lat,
                         "without_latency_and_skew": mk(exp_old), "with_latency_coarse_collective_after_backward": mk(exp_lat_seq),
                         "with_latency": mk(exp_lat), "with_latency_and_skew": mk(exp_lat + skew),
                         "predicted_speedup_8gpu": t1 / (t8 + exp_lat + skew)}
            # the two forms of the N = 8 step, collectives live: t8 = that run's median step (its launch cost is then inside t8: the
            # exposed term below keeps it, i.e. counts it twice -- the conservative side), skew = that run's own p95 - median
            wl = shard[0]
            for mode, row in (("on", "graphed"), ("off", "eager_overlap")):
                lv = live.get((wl, mode))
                if not lv or "error" in lv:
                    out[name][row] = lv or {"error": "not measured (no RCCL communicator in this process)"}
                    continue
                t8m = lv["step_ms"]
                # an eager rank cannot run faster than its host enqueues: the period is max(GPU step, host time)
                t8e = max(t8m, lv["host_ms"])
                out[name][row] = dict(mk(exp_lat + lv["skew_ms"], t8e), shard_step_ms=t8m, host_enqueue_ms=lv["host_ms"], skew_ms=lv["skew_ms"],
                                      ran_as_graph=lv["graph"], speedup_before_collective_terms=t1 / t8e)
            g_row, e_row = out[name].get("graphed") or {}, out[name].get("eager_overlap") or {}
            if "speedup_8gpu" in g_row:
                out[name]["predicted_speedup_8gpu"] = g_row["speedup_8gpu"]          # the default N > 1 form
            if "speedup_8gpu" in e_row:
                out[name]["predicted_speedup_8gpu_eager_fallback"] = e_row["speedup_8gpu"]
    return out


def bench_render(args, rank, world, device, dist, synth, ops, pipeline):
    backend = os.environ.get("ANERF_BENCH_BACKEND", "nccl")
    if args.workload == "render64x64":
        H = W = 64; focal = 75.0; S, Ni = 32, 0
        name = "SURREAL-shaped 64x64 frame, 32 samples/ray, forward render (BASELINE config 1)"
    elif args.workload == "hier":
        H = W = 512; focal = 600.0; S, Ni = 64, 16
        name = "SURREAL-shaped 512x512 frame, 64+16 samples/ray (surreal.txt), forward render"
    elif args.workload == "hier128":
        H = W = 512; focal = 600.0; S, Ni = 64, 128
        name = "SURREAL-shaped 512x512 frame, 64+128 samples/ray, forward render (BASELINE config 5)"
    else:
        H = W = 512; focal = 600.0; S, Ni = 64, 0
        name = "SURREAL-shaped 512x512 frame, 64 samples/ray, forward render (BASELINE config 2)"
    sc = synth.make_scene(0, H, W, focal)
    n_total = len(sc["rays_o"])
    dev = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32, device=device)
    cfg = ops.PathConfig()
    Pc, Pf = synth.make_net_params(11), synth.make_net_params(12)
    which = 3 if args.precision == "bf16x3" else 0
    net_c = ops.pack_params(cfg, {k: dev(v) for k, v in Pc.items()}, which)
    net_f = ops.pack_params(cfg, {k: dev(v) for k, v in Pf.items()}, which) if Ni else None

    # this rank's contiguous slice of the frame's rays (inputs resident in HBM before timing starts)
    per = (n_total + world - 1) // world
    lo, hi = importlib.import_module("a-nerf_amd.parallel").shard_rays(n_total, rank, world)     # contiguous ceil-sized slices
    rb = pipeline.make_ray_batch(dev(sc["rays_o"][lo:hi]), dev(sc["rays_d"][lo:hi]))
    cyl = dev(sc["cyl"])[None].expand(hi - lo, -1).contiguous()
    skt = dev(sc["pose"]["skts"])[None]          # one pose per frame: shared (stride-0) bone matrices
    cut = torch.full((24,), 0.5, device=device)
    gather_buf = torch.empty(world * per, 5, device=device) if dist is not None else None
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    ev_c = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    ev_s = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(i=None):
        # HIP events bracket the DOMINANT launch on its own stream: the only k_mlp_fwd launch (Ni = 0) or the fine pass
        # over the S+Ni merged samples (hierarchical workloads); ev_s brackets the whole step
        if i is not None:
            ev_s[i][0].record()
        nf_raw, stats = ops.ray_bounds(rb, cyl)
        z, _ = ops.coarse_z(nf_raw, stats, rb, S)
        if i is not None and not Ni:
            ev[i][0].record()
        raw = ops.mlp_raw(cfg, net_c[0], net_c[1], rb, z, skt, 20.0, 20.0, cut, cut, precision=args.precision)
        if i is not None and not Ni:
            ev[i][1].record()
        co = ops.composite(cfg, raw, z, rb)
        if Ni:
            zs, zm, _ = ops.importance(z, co["weights"], Ni, want_idx=False)
            if i is not None:
                ev[i][0].record()
            raw_f = ops.mlp_raw(cfg, net_f[0], net_f[1], rb, zm, skt, 20.0, 20.0, cut, cut, precision=args.precision)
            if i is not None:
                ev[i][1].record()
            co = ops.composite(cfg, raw_f, zm, rb)
        if dist is not None:
            mine = torch.zeros(per, 5, device=device)
            mine[:hi - lo, 0:3] = co["rgb_map"]; mine[:hi - lo, 3] = co["acc_map"]; mine[:hi - lo, 4] = co["disp_map"]
            if i is not None:
                ev_c[i][0].record()
            if backend == "nccl":
                dist.all_gather_into_tensor(gather_buf, mine)
            else:
                dist.all_gather(list(gather_buf.view(world, per, 5).unbind(0)), mine)
            if i is not None:
                ev_c[i][1].record()
        if i is not None:
            ev_s[i][1].record()
        return co

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    barrier()
    alloc0 = alloc_counters(device)
    clocks = ClockSampler(device.index or 0)
    with clocks:                                         # host thread, sysfs reads: nothing on the timed stream
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = step(i)
        barrier()
        dt = dt_local = time.perf_counter() - t0
    alloc1 = alloc_counters(device)
    tt = torch.tensor([dt], device=device)
    if dist is not None:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    mlp_all = [a.elapsed_time(b) for a, b in ev]
    step_all = [a.elapsed_time(b) for a, b in ev_s]
    # the roofline prices the MEDIAN launch (a mean over a handful of launches cannot tell a hiccup from a regression); the mean,
    # p95 and max are in `launch_ms`
    mlp_ms = float(np.median(mlp_all)) if args.steps else float("nan")
    coll_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_c])) if (args.steps and dist is not None) else 0.0

    # Side measurement (never the headline `value`): the same frame through the opt-in split-bf16 kernels, same timing
    # protocol, plus its agreement with the fp32 frame just rendered.
    alt = None
    if args.precision == "fp32" and args.steps > 0 and args.extra != "off":
        rgb_f32 = out["rgb_map"].clone()
        net_c3 = ops.pack_params(cfg, {k: dev(v) for k, v in Pc.items()}, 3)
        net_f3 = ops.pack_params(cfg, {k: dev(v) for k, v in Pf.items()}, 3) if Ni else None
        saved = (net_c, net_f, args.precision)
        net_c, net_f, args.precision = net_c3, net_f3, "bf16x3"
        out3 = step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            out3 = step()
        barrier()
        d3 = torch.tensor([time.perf_counter() - t1], device=device)
        if dist is not None:
            dist.all_reduce(d3, op=dist.ReduceOp.MAX)
        net_c, net_f, args.precision = saved
        alt = {"precision": "bf16x3 (hi/lo-split bf16 MFMA operands, f32 accumulate)", "value": n_total * args.steps / float(d3.item()),
               "unit": "rays/s", "ms_per_step": float(d3.item()) / args.steps * 1e3,
               "max_abs_rgb_vs_f32": float((out3["rgb_map"] - rgb_f32).abs().max())}

    res = None
    if rank == 0:
        ms = dt / args.steps * 1e3
        rays_s = n_total * args.steps / dt
        # dominant kernel: k_mlp_fwd over this rank's (hi-lo)*S samples, timed with HIP events on its launch stream
        flops_launch = F_MLP * (hi - lo) * (S + Ni)
        achieved = flops_launch / (mlp_ms * 1e-3)
        b3 = args.precision == "bf16x3"
        peak = PEAK_BF16_MFMA if b3 else PEAK_FP32_MFMA
        res = {
            "metric": "rays/sec", "value": rays_s, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "bf16x3 (hi/lo-split bf16 MFMA operands, f32 accumulate)" if b3 else "f32",
            "data": "synthetic",
            "config": {"workload": name, "rays_per_step": n_total, "samples_per_ray": S, "n_importance": Ni,
                       "parallelism": f"ray-sharded x{world}", "weights": "numpy-seeded random init (alpha bias +1)"},
            "roofline": {"bound": "mfma", "kernel": ("k_mlp_fwd_b3<7,4,0>" if b3 else "k_mlp_fwd<7,4,0,false,false,0>") +
                                                     (f" (fine pass, {S + Ni} samples/ray)" if Ni else ""),
                         "achieved": (3 if b3 else 1) * achieved / 1e12, "algorithmic": achieved / 1e12,
                         "peak": peak / 1e12, "unit": "TFLOP/s", "frac": (3 if b3 else 1) * achieved / peak,
                         "avg_launch_ms": mlp_ms, "launch_ms": step_stats(mlp_all), "flop_per_launch": flops_launch, "traffic": None},
            "step_ms": step_stats(step_all), "slow_steps": outliers(step_all),
            "allocator_in_timed_region": {k: alloc1[k] - alloc0[k] for k in alloc0},
            # shader clock / socket power DURING the timed steps (amdsmi, host thread); frac_at_observed_clock rescales the MFMA
            # peak (quoted at 2400 MHz, MI355X_MICROARCH.md) to the mean clock the kernel actually ran at
            "clocks": dict(clocks.summary(), hbm_copy_GBps=hbm_probe(device)),
        }
        ck = res["clocks"]
        if ck.get("sclk_mhz_mean"):
            res["roofline"]["frac_at_observed_clock"] = res["roofline"]["frac"] * 2400.0 / ck["sclk_mhz_mean"]
        attach_traffic(res, args.workload + ("_bf16x3" if b3 else ""), world)
        if alt is not None:
            res["alt_precision"] = alt
        if args.cpu_rays > 0 and world == 1:   # CPU baseline: rank 0 at N=1 only (bench contract)
            res["cpu_baseline"], res["parity"] = cpu_baseline(sc, S, Ni, Pc, Pf, out, lo, min(args.cpu_rays, hi - lo))
    return dist_record(res, dist, device, dt_local, args.steps, coll_ms, backend)


def bench_api_render(args, device, synth):
    """BASELINE config 2 through the drop-in route, with the inputs the reference's own render_path builds (run_nerf.py:62-88):
    `render(h, w, focal, rays=(rays_o, rays_d), chunk=4096, kp_batch / skts / cyls / bones = x.clone().expand(n_rays, ...)`
    (stride-0 views of the frame's pose), **render_kwargs_test)` -> batchify_rays -> RayCaster.forward per 4096-ray chunk ->
    torch.cat of the chunks' dicts.  What a maintainer who applies INTEGRATION.md's two-line patch gets per frame."""
    networks = importlib.import_module("a-nerf_amd.networks")
    raycaster = importlib.import_module("a-nerf_amd.raycaster")
    render_mod = importlib.import_module("a-nerf_amd.render")
    ops = importlib.import_module("a-nerf_amd.ops")
    H = W = 512; focal = 600.0; S, Ni, chunk = 64, 0, 4096
    dev = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32, device=device)
    kw = dict(D=8, W=256, input_ch=360, input_ch_bones=72, input_ch_views=648, use_viewdirs=True)
    net_c = networks.NeRF(**kw)
    net_c.load_state_dict({k: torch.tensor(v) for k, v in synth.make_net_params(11).items()})
    ck = {"cutoff": True, "cutoff_dist": 0.5, "cutoff_inputs": True, "cutoff_dim": 24}
    e_v, _ = networks.get_embedder(7, input_dims=24, cutoff_kwargs=dict(ck, dist_inputs=False))
    e_b, _ = networks.get_embedder(0, input_dims=72, cutoff_kwargs={"cutoff": False})
    e_d, _ = networks.get_embedder(4, input_dims=72, cutoff_kwargs=dict(ck, dist_inputs=True))
    caster = raycaster.RayCaster(net_c, e_v, e_b, e_d, network_fine=None).to(device).eval()
    caster.render_precision = args.precision
    sc = synth.make_scene(0, H, W, focal)
    n = len(sc["rays_o"])
    rays = (dev(sc["rays_o"]), dev(sc["rays_d"]))
    reuse = lambda x, *sh: dev(x)[None].clone().expand(n, *sh)          # run_nerf.py:62-72 reuse_input(x, expand)
    batch = dict(kp_batch=reuse(sc["pose"]["kp"], 24, 3), skts=reuse(sc["pose"]["skts"], 24, 4, 4), cyls=reuse(sc["cyl"], 5),
                 bones=reuse(sc["pose"]["bones"], 24, 3))
    rk = dict(ray_caster=caster, perturb=False, N_importance=Ni, N_samples=S, use_viewdirs=True, raw_noise_std=0., ray_noise_std=0.,
              ext_scale=0.001, preproc_kwargs={"density_scale": 1.0, "density_fn": torch.nn.functional.relu}, lindisp=False,
              nerf_type="nerf")
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def frame(i=None):
        if i is not None:
            ev[i][0].record()
        with torch.no_grad():
            ret = render_mod.render(H, W, focal, rays=rays, chunk=chunk, c2w=None, cams=None, subject_idxs=None, **batch, **rk)
        if i is not None:
            ev[i][1].record()
        return ret

    for _ in range(max(1, args.warmup)):
        out = frame()
    torch.cuda.synchronize()
    alloc0 = alloc_counters(device)
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = frame(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    alloc1 = alloc_counters(device)
    frame_all = [a.elapsed_time(b) for a, b in ev]
    med = float(np.median(frame_all))
    # agreement with ONE launch over the whole frame under the shared pose (the kernel-level bench's form): same kernels on the
    # same values -- identical unless a chunk-local NaN fallback (ray_utils.py:327-339) triggers, which this frame has none of
    rb = ops.make_ray_batch(*rays)
    which = 3 if args.precision == "bf16x3" else 0
    one = ops.forward(net_c.path_cfg, net_c.packed(which), None, rb, dev(sc["pose"]["skts"])[None], dev(sc["cyl"])[None], S, 0,
                      tau_v=e_v.get_tau(), tau_d=e_d.get_tau(), cut_v=e_v.cutoff_dist.detach(), cut_d=e_d.cutoff_dist.detach(),
                      precision=args.precision)
    max_diff = float((one["rgb_map"] - out["rgb_map"]).abs().max())
    b3 = args.precision == "bf16x3"
    peak = PEAK_BF16_MFMA if b3 else PEAK_FP32_MFMA
    flops = F_MLP * n * S
    achieved = flops / (med * 1e-3)
    n_chunks = (n + chunk - 1) // chunk
    return {"metric": "rays/sec", "value": n * args.steps / dt, "unit": "rays/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "bf16x3 (hi/lo-split bf16 MFMA operands, f32 accumulate)" if b3 else "f32", "data": "synthetic",
            "config": {"workload": "reference-shaped API route: render() -> batchify_rays(chunk 4096) -> RayCaster.forward with run_nerf.render_path's "
                                   "stride-0 expanded pose inputs, SURREAL-shaped 512x512 frame, 64 samples/ray (BASELINE config 2)",
                       "rays_per_step": n, "samples_per_ray": S, "n_importance": Ni, "chunk": chunk, "caster_calls_per_frame": n_chunks,
                       "parallelism": "single process"},
            # the WHOLE frame's HIP-event time (all chunks' k_mlp_fwd launches, the small kernels, the gaps and the final cat)
            # against the frame's algorithmic FLOPs: a lower bound of the kernel's own fraction
            "roofline": {"bound": "mfma", "kernel": f"{n_chunks} launches of " + ("k_mlp_fwd_b3<7,4,0>" if b3 else "k_mlp_fwd<7,4,0,false,false,0>") +
                                                    " per frame; priced on the whole frame's HIP-event time (median)",
                         "achieved": (3 if b3 else 1) * achieved / 1e12, "algorithmic": achieved / 1e12, "peak": peak / 1e12,
                         "unit": "TFLOP/s", "frac": (3 if b3 else 1) * achieved / peak, "avg_launch_ms": med,
                         "launch_ms": step_stats(frame_all), "flop_per_launch": flops, "traffic": None},
            "step_ms": step_stats(frame_all), "slow_steps": outliers(frame_all),
            "allocator_in_timed_region": {k: alloc1[k] - alloc0[k] for k in alloc0},
            "max_abs_rgb_vs_single_launch": max_diff}


def attach_traffic(res, key, world):
    """roofline.traffic = HBM bytes per launch / step of the dominant kernel(s) from the rocprofv3 --pmc passes committed under
    profiles/ (FETCH_SIZE x2 + WRITE_SIZE, separate passes, MI355X_MICROARCH.md HBM section).  profiles/pmc_traffic.json is
    regenerated from those passes by tools/make_pmc_traffic.py and records the git SHA of the binary they ran on."""
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(key)
    except (OSError, ValueError):
        return
    if tr and world == 1:
        res["roofline"]["traffic"] = tr["hbm_bytes"]
        res["roofline"]["traffic_note"] = (f"bytes per launch/step, FETCH_SIZE(x2)+WRITE_SIZE, {tr['source']} (build {tr.get('git_sha', '?')}); "
                                           f"algorithmic {tr['algorithmic_bytes']:.3g} B" + ("; " + tr["note"] if tr.get("note") else ""))


def live_traffic(argv_workload, kernel_sub, launches_per_step=1, timeout_s=120):
    """HBM bytes per launch of the dominant kernel measured in THIS run: rocprofv3 --kernel-trace --pmc <counter> (one counter group
    per pass, as MI355X_MICROARCH.md's HBM section prescribes) around a 2-step child run of the same workload; FETCH_SIZE and
    WRITE_SIZE are in KiB, gfx950 counts a wide coalesced read stream at half its bytes (x2).  Returns (dict, None) or (None, why)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process is itself running under a profiler (no nested rocprofv3 passes)"
    tmp = tempfile.mkdtemp(prefix="anerf_pmc_", dir="/tmp")
    vals = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "-d", out, "--", sys.executable, os.path.abspath(__file__)] + argv_workload + \
                  ["--steps", "2", "--warmup", "1", "--extra", "off", "--cpu-rays", "0", "--live-traffic", "off", "--graph", "off",
                   "--detail", os.path.join(tmp, "d.json")]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout_s)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {ctr} exited {r.returncode}: {r.stderr[-200:]}"
            dbs = glob.glob(os.path.join(out, "**", "*_results.db"), recursive=True)
            if not dbs:
                return None, f"no rocprofv3 result database for {ctr}"
            con = sqlite3.connect(dbs[0])
            row = con.execute("select avg(value), count(*) from counters_collection where counter_name = ? and kernel_name like ?",
                              (ctr, f"%{kernel_sub}%")).fetchone()
            con.close()
            if not row or row[0] is None:
                return None, f"no {ctr} rows for kernel {kernel_sub}"
            vals[ctr] = (float(row[0]), int(row[1]))
    except Exception as e:
        return None, f"{type(e).__name__}: {e}"[:300]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    fetch, write = vals["FETCH_SIZE"][0] * 1024 * 2, vals["WRITE_SIZE"][0] * 1024
    return {"hbm_bytes": launches_per_step * (fetch + write), "fetch_bytes_x2": fetch, "write_bytes": write,
            "dispatches_averaged": vals["FETCH_SIZE"][1]}, None


def bench_train(args, rank, world, device, dist, synth, mixamo=False, per_kernel=True):
    """BASELINE config 3: SURREAL training step, N_rand = 3072 rays (global), 64 + 16 samples, fwd + bwd + Adam,
    through the reference-shaped API (RayCaster mirror + render() + loss).  Strong scaling: each rank takes
    N_rand / world rays; gradients are averaged with one RCCL all-reduce of a flat bucket per step.
    mixamo=True is BASELINE config 4 (configs/mixamo/mixamo.txt:41-55): + per-frame codes (920-wide view layer), L1
    loss, and pose refinement: skts come from the FK layer (PoseOptLayer mirror) and the hot path's dskts flow back into
    the bone parameters, which live in the same flat FusedAdam bucket as the networks and are stepped every
    --opt-pose-step iterations (default 1: every step, the more expensive schedule; mixamo.txt:48 uses 20)."""
    networks = importlib.import_module("a-nerf_amd.networks")
    raycaster = importlib.import_module("a-nerf_amd.raycaster")
    render_mod = importlib.import_module("a-nerf_amd.render")
    parallel = importlib.import_module("a-nerf_amd.parallel")
    optim = importlib.import_module("a-nerf_amd.optim")
    N_rand, S, Ni = args.n_rand, 64, 16
    dev = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32, device=device)
    n_poses = 8
    kw = dict(D=8, W=256, input_ch=360, input_ch_bones=72, input_ch_views=648, use_viewdirs=True)
    mk = {}
    if mixamo:
        kw.update(use_framecode=True, framecode_ch=16, n_framecodes=n_poses)
        mk = dict(framecode_ch=16, n_codes=n_poses)
    net_c, net_f = networks.NeRF(**kw), networks.NeRF(**kw)
    net_c.load_state_dict({k: torch.tensor(v) for k, v in synth.make_net_params(11, **mk).items()})
    net_f.load_state_dict({k: torch.tensor(v) for k, v in synth.make_net_params(12, **mk).items()})
    ck = {"cutoff": True, "cutoff_dist": 0.5, "cutoff_inputs": True, "cutoff_dim": 24}
    e_v, _ = networks.get_embedder(7, input_dims=24, cutoff_kwargs=dict(ck, dist_inputs=False))
    e_b, _ = networks.get_embedder(0, input_dims=72, cutoff_kwargs={"cutoff": False})
    e_d, _ = networks.get_embedder(4, input_dims=72, cutoff_kwargs=dict(ck, dist_inputs=True))
    caster = raycaster.RayCaster(net_c, e_v, e_b, e_d, network_fine=net_f).to(device)
    caster.train()
    caster.train_precision = args.precision
    params = [p for p in caster.parameters() if p.requires_grad]
    fused = not args.torch_tail
    popt = popt_opt = None
    if mixamo:
        pose_opt = importlib.import_module("a-nerf_amd.pose_opt")
        poses = [synth.make_pose(k) for k in range(n_poses)]
        popt = pose_opt.PoseOptLayer(np.stack([q["kp"] for q in poses]), np.stack([q["bones"] for q in poses]),
                                     (synth.SMPL_REST_POSE * synth.SURREAL_SCALE)[None], use_rot6d=True).to(device)   # mixamo.txt:44
    if fused:
        # ONE flat bucket: group 0 = both networks (+ frame codes), group 1 = the pose layer's parameters, stepped every
        # opt_pose_step iterations (accumulating in between) -- so a step's all-reduce is one collective whatever is due
        groups = [{"params": params, "lr": 5e-4}]
        if mixamo:
            groups.append({"params": list(popt.parameters()), "lr": 5e-4, "step_every": args.opt_pose_step})
        opt = optim.FusedAdam(groups, betas=(0.9, 0.999))
        opt.attach(caster, pose_layer=popt)      # opt in: the backwards accumulate into the bucket in place
        if dist is not None and N_rand % world == 0:
            opt.enable_overlap()         # the fine network's all-reduce runs under the coarse half of the backward
    else:
        opt = torch.optim.Adam(params, lr=5e-4, betas=(0.9, 0.999))
        if mixamo:
            popt_opt = torch.optim.Adam(popt.parameters(), lr=5e-4)
    bucket = None if fused else parallel.GradBucket(params + (list(popt.parameters()) if mixamo else []))
    # per-ray replicated pose batch as the reference's collate produces it (dataset.py:813-820), 8 poses
    ro, rd, kp, skts, bones, cyls, pidx = synth.scene_batch(N_rand, list(range(n_poses)), H=512, W=512, focal=600.0, ray_seed=3,
                                                            per_ray_pose=True)
    lo, hi = parallel.shard_rays(N_rand, rank, world)
    sl = slice(lo, hi)
    rays = (dev(ro[sl]), dev(rd[sl]))
    batch = dict(kp_batch=dev(kp[sl]), skts=dev(skts[sl]), cyls=dev(cyls[sl]), bones=dev(bones[sl]))
    target = dev(np.random.default_rng(1).random((N_rand, 3))[sl])
    cams = None
    if mixamo:
        pose_idx_host = np.asarray(pidx)[sl]
        # popt_anchors (pose_opt.py:60-75) of the batch's DISTINCT poses + their share of the rays: the regulariser
        # (_compute_kp_loss) over them equals the reference's mean over the per-ray replicated batch
        uniq, counts = np.unique(pose_idx_host, return_counts=True)
        anchor_u = popt.bones.detach().clone()[torch.tensor(uniq, device=device)].contiguous()
        w_u = torch.tensor(counts / float(len(pose_idx_host)), dtype=torch.float32, device=device)
        cams = torch.tensor(pose_idx_host, device=device).to(torch.float32)
    pk = {"density_scale": 1.0, "density_fn": torch.nn.functional.relu}
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    ev_c = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    ev_e = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]        # end of the step (after the optimiser kernel)
    host_t = np.zeros(args.steps + 1)                                               # host clock at the start of each timed step
    backend = os.environ.get("ANERF_BENCH_BACKEND", "nccl")

    it = [0]                     # global iteration counter (the reference's `i`, trainer.py:451)

    def iteration(k, mark=None):
        """one training iteration (the reference's loop body, trainer.py:228-277) with iteration counter k; `mark(name)` lets the
        eager path record HIP events between its phases (a captured graph has no such points)"""
        b = batch
        if mixamo:   # per-ray pose indices, as the reference calls its layer (kp_idx of the batch): FK once per distinct pose
            kp_r, bones_r, skts_r, _, rots_r = popt(pose_idx_host)
            b = dict(batch, kp_batch=kp_r, skts=skts_r, bones=bones_r)
        out = render_mod.render(512, 512, 600.0, chunk=4096, rays=rays, use_viewdirs=True, ray_caster=caster, cams=cams,
                                subject_idxs=None, N_samples=S, N_importance=Ni, perturb=1.0, raw_noise_std=1.0,
                                preproc_kwargs=pk, **b)
        loss, _ = (optim.fused_nerf_loss if fused else render_mod.nerf_loss)(out, target, bgs=1.0,
                                                                              loss_fn="L1" if mixamo else "MSE")
        if mixamo:   # _compute_kp_loss (trainer.py:382-403; opt_pose_tol 0.01, opt_pose_coef 2.0, mixamo.txt:45,54), one launch
            loss = pose_opt.kp_loss(popt.last_unique["rots"], anchor_u, w_u, True, 0.01, 2.0, add_to=loss)[1] if fused else \
                loss + pose_opt.kp_loss(popt.last_unique["rots"], anchor_u, w_u, True, 0.01, 2.0)
        (optim.backward if fused else torch.autograd.backward)(loss)      # fused tail: cached unit seed, no `grad * 1` launches
        if mark:
            mark("bwd_done")
        if fused:
            opt.all_reduce_grads(i=k)     # ONE RCCL all-reduce over whatever is due (networks [+ pose]); 1/world folded into Adam
        else:
            bucket.all_reduce_mean()
        if mark:
            mark("reduced")
        if fused:
            opt.step(zero_grad=True, i=k)
        else:
            opt.step()
            opt.zero_grad()
            if mixamo:
                popt_opt.step()
                popt_opt.zero_grad()
        return {"loss": loss}

    # --graph: the iteration as ONE hipGraph launch (a-nerf_amd/graph_step.py; per-step scalars in a device-resident block, ABI
    # revision 6).  auto = on with the fused tail, for ANY number of ranks (round 6): with more than one rank the gradient
    # collectives -- both networks' all-reduces on the optimiser's side stream, the pose group's on its iteration -- are captured
    # with the step (bit-identical to the eager step on a one-rank RCCL communicator: tests/test_graph_step.py; 3.01 -> 2.91 ms at
    # the Mixamo shard, host 2.0 -> 0.05 ms).  After the captures every rank reports success through ONE 1-element all-reduce
    # OUTSIDE any capture (GraphedTrainStep.agree); if any rank's capture failed, ALL ranks time the eager step and the record
    # says so (`graph`: {"error": ...}, config.graph = false).
    # (gloo -- the control-flow tests with ranks sharing one GPU -- cannot be captured and is not attempted: GraphedTrainStep
    # reports why, the ranks agree, the record says graph = false)
    use_graph = fused and args.graph != "off"
    gs = None
    if use_graph:
        graph_step = importlib.import_module("a-nerf_amd.graph_step")
        gs = graph_step.GraphedTrainStep(iteration, caster, opt, eager_steps=0)

    def step(i=None, eager=False):
        if i is not None:
            host_t[i] = time.perf_counter()
            ev[i][0].record()
        it[0] += 1
        if gs is not None and not eager:
            loss = gs.step(it[0])["loss"]
            if i is not None:
                ev[i][1].record()         # (a captured step has no point between backward and optimiser to put an event at)
        else:
            def mark(name):
                if i is not None:
                    if name == "bwd_done":
                        ev[i][1].record()
                        if dist is not None:
                            ev_c[i][0].record()
                    elif dist is not None:
                        ev_c[i][1].record()
            loss = iteration(it[0], mark)["loss"]
        if i is not None:
            ev_e[i].record()
        return loss

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # the W warm-up steps, preceded by a few untimed priming steps: the first ~10 steps of a fresh process still hit
    # one-off runtime stalls (allocator pool growth, lazy code-object loads; a single 40 ms hiccup was seen as late as
    # step 6), which would dominate a short timed window of 3-20 ms steps
    for _ in range(8):
        step(eager=True)
    graph_error = None
    if gs is not None:           # capture both cadence phases now (nothing runs): no capture inside the timed steps
        try:
            gs.prepare(1)
            if mixamo and args.opt_pose_step > 1:
                gs.prepare(args.opt_pose_step)
        except Exception as e:   # a stack that cannot capture the step: say so in the record and time the eager step instead
            gs.eager_only[("prepare",)] = f"{type(e).__name__}: {e}"[:300]
        if os.environ.get("ANERF_BENCH_FAIL_CAPTURE_RANK") == str(rank):       # test hook: this rank's capture "failed"
            gs.eager_only[("test hook",)] = "ANERF_BENCH_FAIL_CAPTURE_RANK"
        mine_ok = not gs.eager_only and gs._capturable() is None
        if not gs.agree(dist):                      # one tiny all-reduce outside the capture: every rank runs the same mode
            graph_error = ("; ".join(gs.eager_only.values()) or gs._capturable()) if not mine_ok else "another rank's capture failed (agreed over all ranks)"
            sys.stderr.write(f"bench.py: rank {rank}: hipGraph capture not usable ({graph_error}); all ranks fall back to the eager step\n")
            gs = None
            torch.cuda.synchronize()
            opt.zero_grad()
    for _ in range(args.warmup):
        step()
    import gc
    gc.collect()
    gc.freeze()
    barrier()
    alloc0 = alloc_counters(device)
    gc0 = [g["collections"] for g in gc.get_stats()]
    trace = [] if os.environ.get("ANERF_BENCH_ALLOC_TRACE") == "1" else None     # diagnostic: pool growth per step (costs host time)
    clocks = ClockSampler(device.index or 0, period_s=0.1)
    with clocks:
        t0 = time.perf_counter()
        for i in range(args.steps):
            loss = step(i)
            if trace is not None:
                st = torch.cuda.memory_stats(device)
                trace.append((st["num_device_alloc"], st["reserved_bytes.all.current"], st["allocated_bytes.all.peak"]))
        host_t[args.steps] = time.perf_counter()
        barrier()
        dt = dt_local = time.perf_counter() - t0
    alloc1 = alloc_counters(device)
    gc1 = [g["collections"] for g in gc.get_stats()]
    tt = torch.tensor([dt], device=device)
    if dist is not None:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    fb_all = [a.elapsed_time(b) for a, b in ev]
    # a step on the GPU timeline: from its first launch to the end of its optimiser kernel; its period (start to next start)
    # also holds whatever idle time the host left in front of the next step
    step_all = [ev[i][0].elapsed_time(ev_e[i]) for i in range(args.steps)]
    period_all = [ev[i][0].elapsed_time(ev[i + 1][0]) for i in range(args.steps - 1)]
    host_all = list(np.diff(host_t) * 1e3)
    # the roofline prices the MEDIAN step (mean / p95 / max next to it): 5-step means once hid a one-off stall in config 4
    fb_ms = float(np.median(fb_all))
    # HIP-event time of the gradient all-reduce as the main stream sees it (with overlap: what is left of it after the coarse
    # half of the backward), plus the same collective on an idle GPU: the xGMI time of the 6.9 MB bucket itself
    coll_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_c])) if (dist is not None and args.steps and gs is None) else 0.0   # (no event
    coll_alone_ms = None                                                                                     # points inside a captured step)
    if dist is not None and fused:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        e0.record()
        for _ in range(10):
            dist.all_reduce(opt.flat_grad)
        e1.record()
        barrier()
        coll_alone_ms = e0.elapsed_time(e1) / 10
        opt.zero_grad()

    if os.environ.get("ANERF_BENCH_TORCH_PROFILE") == "1" and rank == 0:
        # developer aid (tools/): which torch ops still launch small kernels in a step, with the Python lines that issue them
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
            for _ in range(3):
                step(eager=True)
            torch.cuda.synchronize()
        sys.stderr.write(prof.key_averages(group_by_stack_n=8).table(sort_by="self_cuda_time_total", row_limit=60,
                                                                      max_name_column_width=60, max_src_column_width=110) + "\n")
        seen = set()
        for e in prof.events():          # who issues the small fills / copies: first Python frames of each distinct call site
            if e.name in ("aten::zeros", "aten::full", "aten::ones_like", "aten::mul", "aten::add", "aten::copy_", "aten::clone",
                          "aten::contiguous", "aten::index_select", "aten::index_add_", "aten::zeros_like", "aten::fill_"):
                st = tuple(f for f in (e.stack or []) if ".py" in f)[:4]
                if (e.name, st) not in seen:
                    seen.add((e.name, st))
                    sys.stderr.write(f"{e.name}: " + " <- ".join(st) + "\n")

    # Per-kernel HIP-event times (AnerfProfile: events recorded by the library around each MFMA kernel of both passes) over
    # a few extra, untimed steps -- the executed-FLOP roofline of every training kernel.  Executed FLOPs per sample: forward
    # 2 x 861 824 MAC, backward-data 2 x 557 696 (W^T of the trunk, feature and view layers only), weight-gradient GEMM
    # 2 x 861 824, input gradients (pose / frame-code configurations) 2 x (2 x 256 x 432 + 128 x u_width).
    ops = importlib.import_module("a-nerf_amd.ops")
    uw = 648 + (16 if mixamo else 0)
    F_FWD, F_BWD, F_GEMM, F_IN = F_MLP, 2 * 557696, F_MLP, 2 * (2 * 256 * 432 + 128 * uw)
    kernels = None
    if args.steps > 0 and per_kernel:
        prof = ops.Profile()
        acc = {}
        n_prof = 8
        with ops.profiling(prof):
            for _ in range(n_prof):
                prof.reset()             # fresh events: a pair this step does not record reads as None, not as last step's value
                step(eager=True)         # per-kernel events are recorded by the library at enqueue time: the eager form of the step
                torch.cuda.synchronize()
                for kind in ("fwd", "bwd", "gemm") + (("bwd_in",) if mixamo else ()):
                    for ps in (0, 1):
                        t = prof.ms(kind, ps)
                        if t is not None:
                            acc.setdefault((kind, ps), []).append(t)
        names = {"fwd": "k_mlp_fwd<train>", "bwd": "k_mlp_bwd", "gemm": "k_gemm_tn + k_reduce_dw", "bwd_in": "k_mlp_bwd_in_enc (input gradients + the encoding's backward as its epilogue)"}
        flops = {"fwd": F_FWD, "bwd": F_BWD, "gemm": F_GEMM, "bwd_in": F_IN}
        b3k = args.precision == "bf16x3"
        kernels = []
        for (kind, ps), ts in sorted(acc.items()):
            ms_k = float(np.median(ts[1:])) if len(ts) > 1 else float(ts[0])
            fl = flops[kind] * (hi - lo) * (S + Ni if ps else S)
            tf = fl / (ms_k * 1e-3) / 1e12
            kernels.append({"kernel": names[kind] + ("_b3" if b3k else ""), "pass": "fine" if ps else "coarse", "ms": ms_k, "flop": fl,
                            "tflops": tf, "frac": (3 * tf / (PEAK_BF16_MFMA / 1e12)) if b3k else tf / (PEAK_FP32_MFMA / 1e12)})
    res = None
    if rank == 0:
        flop_step_rank = 3 * F_MLP * (hi - lo) * (S + S + Ni)        # SURVEY 8(d)'s convention: fwd + 2x bwd
        exec_step_rank = (F_FWD + F_BWD + F_GEMM + (F_IN if mixamo else 0)) * (hi - lo) * (S + S + Ni)   # what the kernels execute
        achieved = exec_step_rank / (fb_ms * 1e-3)
        b3 = args.precision == "bf16x3"
        res = {"metric": "rays/sec", "value": N_rand * args.steps / dt, "unit": "rays/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None,
               "dtype": "f32" if not b3 else "f32 backward + bf16x3 forward (split bf16 operands, f32 accumulate)",
               "data": "synthetic",
               "config": {"workload": (f"Mixamo-shaped training step (frame codes, pose refinement through the FK layer, L1), N_rand={N_rand}, "
                                       "64+16 samples, fwd+bwd+Adam (BASELINE config 4)") if mixamo else
                                      f"SURREAL-shaped training step, N_rand={N_rand}, 64+16 samples, fwd+bwd+Adam (BASELINE config 3)",
                          "rays_per_step": N_rand, "samples_per_ray": S, "n_importance": Ni,
                          "parallelism": f"ray-sharded x{world}, 1 all-reduce/step", "loss": float(loss.detach()),
                          "tail": "fused loss + FusedAdam (anerf_loss / anerf_adam_step)" if fused else "torch loss + torch.optim.Adam"},
               # `frac` prices the FLOPs the kernels EXECUTE (forward 1.724 + backward-data 1.115 + weight-gradient GEMM 1.724
               # [+ input gradients] MFLOP per sample) against the HIP-event time of forward + backward; `survey_3x` keeps SURVEY
               # 8(d)'s "training = 3 x forward" convention next to it (it over-counts the backward-data kernel)
               "roofline": {"bound": "mfma", "kernel": "k_mlp_fwd<train> + k_mlp_bwd + k_gemm_tn" + (" + k_mlp_bwd_in_enc" if mixamo else "") +
                                                       (" (both nets), HIP-event time of the whole captured step (hipGraph: loss + optimiser included)"
                                                        if gs is not None else " (both nets), HIP-event time of fwd+bwd"),
                            # bf16x3: every algorithmic FLOP is issued as 3 bf16 MFMA FLOPs, priced against the bf16 peak
                            "achieved": (3 if b3 else 1) * achieved / 1e12, "algorithmic": achieved / 1e12,
                            "peak": (PEAK_BF16_MFMA if b3 else PEAK_FP32_MFMA) / 1e12, "unit": "TFLOP/s",
                            "frac": (3 * achieved / PEAK_BF16_MFMA) if b3 else achieved / PEAK_FP32_MFMA,
                            "avg_launch_ms": fb_ms, "launch_ms": step_stats(fb_all), "flop_per_launch": exec_step_rank,
                            "survey_3x": {"flop_per_step": flop_step_rank,
                                          "frac": (3 if b3 else 1) * flop_step_rank / (fb_ms * 1e-3) / (PEAK_BF16_MFMA if b3 else PEAK_FP32_MFMA)},
                            "kernels": kernels, "traffic": None},
               # per-step HIP-event statistics of the timed steps: the step itself (first launch .. optimiser kernel), the period
               # between step starts (adds host-side gaps), the host's enqueue time per step; steps slower than 1.5 x the median
               # are listed with both clocks
               "step_ms": step_stats(step_all), "period_ms": step_stats(period_all), "host_enqueue_ms": step_stats(host_all),
               "slow_steps": outliers(step_all, host_all),
               "allocator_in_timed_region": {k: alloc1[k] - alloc0[k] for k in alloc0},
               "gc_collections_in_timed_region": [b - a for a, b in zip(gc0, gc1)], "clocks": clocks.summary(),
               # True: every timed step was ONE hipGraphLaunch (+ the step-block write); captures / replays counted by the wrapper
               "graph": ({"error": graph_error} if graph_error else False) if gs is None else
                        {"replays": gs.replays, "captures": gs.captures, "eager_calls": gs.eager_calls, "graphs": [list(k) for k in gs.graphs]}}
        res["config"]["graph"] = gs is not None
        if trace is not None:
            res["alloc_trace"] = [{"step": i, "new_segments": trace[i][0] - (trace[i - 1][0] if i else alloc0["num_device_alloc"]),
                                   "reserved_MB": trace[i][1] / 2 ** 20, "grew_MB": (trace[i][1] - trace[i - 1][1]) / 2 ** 20 if i else None,
                                   "step_ms": step_all[i], "host_ms": host_all[i]} for i in range(len(trace))]
        if fused and dist is not None:
            res["overlap"] = dict(opt.overlap_stats, enabled=bool(opt.overlap))
        if coll_alone_ms is not None:
            res["collective_alone_ms"] = coll_alone_ms
            res["collective_bytes"] = int(opt.flat_grad.numel() * 4)
        key = ("train_mixamo" if mixamo else "train") + ("" if N_rand == 3072 else str(N_rand)) + ("_bf16x3" if b3 else "")
        attach_traffic(res, key, world)
        if mixamo:
            res["config"]["opt_pose_step"] = args.opt_pose_step
        if args.cpu_rays > 0 and world == 1 and not mixamo:
            res["cpu_baseline"] = cpu_train_baseline(synth, ro, rd, skts, cyls, S, Ni, min(args.cpu_rays, 512, N_rand))
    return dist_record(res, dist, device, dt_local, args.steps, coll_ms, backend)


def cpu_train_baseline(synth, ro, rd, skts, cyls, S, Ni, n_cpu):
    """One training step (forward, MSE loss on both heads, backward; no optimiser) of the torch-CPU oracle on the first
    n_cpu rays of the same batch: the reference's op sequence with autograd, best of a few thread counts."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    oracle = importlib.import_module("anerf_oracle")
    t = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32)
    ocfg = oracle.OracleConfig()
    P = oracle.params_from_numpy(synth.make_net_params(11))
    PF = oracle.params_from_numpy(synth.make_net_params(12))
    for q in list(P.values()) + list(PF.values()):
        q.requires_grad_(True)
    rb = oracle.make_ray_batch(t(ro[:n_cpu]), t(rd[:n_cpu]))
    sk, cy = t(skts[:n_cpu]), t(cyls[:n_cpu])
    target = t(np.random.default_rng(1).random((n_cpu, 3)))
    gen = torch.Generator().manual_seed(0)

    def one():
        out = oracle.render_rays(ocfg, P, PF, rb, sk, cy, S, Ni, t_rand=torch.rand(n_cpu, S, generator=gen),
                                 u_imp=torch.rand(n_cpu, Ni, generator=gen), noise=torch.randn(n_cpu, S, generator=gen),
                                 noise_fine=torch.randn(n_cpu, S + Ni, generator=gen))
        loss, _ = oracle.nerf_loss(out, target, 1.0)
        loss.backward()
    ncpu = os.cpu_count() or 1
    best, best_dt = 1, float("inf")
    for th in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(th)
        t0 = time.perf_counter()
        one()
        d = time.perf_counter() - t0
        if d < best_dt:
            best, best_dt = th, d
    torch.set_num_threads(best)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        one()
    dt = (time.perf_counter() - t0) / reps
    return {"value": n_cpu / dt, "unit": "rays/s", "cores": best, "kind": "port",
            "sample": f"first {n_cpu} rays of the same batch, {S}+{Ni} samples, forward + loss + backward (no optimiser), "
                      f"{dt:.1f} s per step; best of 8/16/32/64 torch threads on this {ncpu}-thread host"}


def cpu_baseline(sc, S, Ni, Pc, Pf, gpu_out, lo, n_cpu):
    """Time the torch-CPU oracle (port of the reference op sequence) on the first n_cpu rays of the same frame,
    all host cores, chunk 4096 as in the reference configs; also report GPU-vs-oracle parity on those rays."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    oracle = importlib.import_module("anerf_oracle")
    t = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32)
    ocfg = oracle.OracleConfig()
    P, PF = oracle.params_from_numpy(Pc), oracle.params_from_numpy(Pf)
    rb = oracle.make_ray_batch(t(sc["rays_o"][:n_cpu]), t(sc["rays_d"][:n_cpu]))
    skts = t(sc["pose"]["skts"])[None]
    cyls = t(sc["cyl"])[None].expand(n_cpu, -1)
    kw = dict(cfg=ocfg, P=P, P_fine=PF if Ni else None, n_samples=S, n_importance=Ni)
    with torch.no_grad():
        # torch-CPU does not scale to every hardware thread of a large host: probe a few thread counts on a
        # 1024-ray slice (this is also the warm-up) and time the sample with the fastest one.
        ncpu = os.cpu_count() or 1
        best, best_dt = 1, float("inf")
        for th in sorted({min(ncpu, c) for c in (8, 16, 32, 64, ncpu)}):
            torch.set_num_threads(th)
            oracle.render_chunked(4096, rb[:256], skts, cyls[:256], **kw)
            t0 = time.perf_counter()
            oracle.render_chunked(4096, rb[:1024], skts, cyls[:1024], **kw)
            d = time.perf_counter() - t0
            if d < best_dt:
                best, best_dt = th, d
        torch.set_num_threads(best)
        t0 = time.perf_counter()
        ref = oracle.render_chunked(4096, rb, skts, cyls, **kw)
        dt = time.perf_counter() - t0
    base = {"value": n_cpu / dt, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"first {n_cpu} rays of the same frame, chunk 4096, {dt:.1f} s of CPU work; best of "
                      f"8/16/32/64/{ncpu} torch threads on this {ncpu}-thread host"}
    parity = None
    if lo == 0:
        g = gpu_out["rgb_map"][:n_cpu].cpu()
        target = t(np.random.default_rng(7).random((n_cpu, 3)))
        parity = {"max_abs_rgb": float((g - ref["rgb_map"]).abs().max()),
                  "psnr_gpu_db": oracle.psnr(g, target), "psnr_oracle_db": oracle.psnr(ref["rgb_map"], target)}
    return base, parity


if __name__ == "__main__":
    main()
