"""The training iteration as ONE hipGraph launch.

The C ABI never allocates or synchronises, so an iteration of the reference's loop (`Trainer.train_batch`,
core/trainer.py:205-277: pose layer -> render -> losses -> backward -> optimiser) is capturable: ~45 kernel launches,
1.0 ms of host time per 2.9 ms step at the 8-GPU shard size (384 rays), become one graph launch.  What changes between
iterations besides buffer CONTENTS are a few scalars that are kernel arguments in the eager calls -- the Philox offset of the
random inputs, the gate temperatures tau, Adam's step count / learning rate, the 1/world gradient scale.  They live in a
device-resident `AnerfStepBlock` (ABI revision 6, ops.StepBlock); `GraphedTrainStep.step()` is then

    block.write()          one launch; the new values travel as its kernel arguments (computed by the same host code as eager)
    graph.replay()         hipGraphLaunch -- no node update

PyTorch does the capture (`torch.cuda.CUDAGraph` = hipStreamBeginCapture / hipGraphInstantiate on ROCm; its caching allocator
serves the captured region from a private pool, so every tensor the step allocates keeps its address across replays).
Bit-identical to the eager step: same kernels, same arithmetic, same order (tests/test_graph_step.py).

Contract of `step_fn(i)` (the caller's iteration; i = the reference's iteration counter):
  * static shapes and static INPUT ADDRESSES: feed it from tensors that live across iterations (`StaticBatch.load` copies a
    loader's batch into such tensors, stream-ordered, outside the graph);
  * no host read of device data (`.item()`, `.cpu()`), no data-dependent host control flow;
  * the optimiser is a `FusedAdam` stepped with `i=i`; groups with `step_every > 1` (the pose cadence, trainer.py:476-478)
    get one graph per set of due groups;
  * it returns a dict of tensors (loss, statistics): they live in the graph's pool and hold the last replay's values.
Not capturable here: `--freq_schedule` (its band factors are re-uploaded per iteration) -- `step()` falls back to the eager call
for it, loudly once.

Robustness (round 6):
  * every NEW (due groups, key) variant runs eagerly once before it is captured (a variant first seen mid-run -- the
    opt_pose_stop transition, a batch with another number of distinct poses -- may do lazy one-off work on its code path: a
    pageable upload, a first-use kernel attribute);
  * a capture that fails (anything not permitted under stream capture) is undone -- current stream, allocator routing, host
    counters, the optimiser's in-flight collective handles -- the variant is marked eager-only with ONE warning and the
    iteration runs eagerly: a long run does not die at a transition;
  * more than one rank: the captured step holds the gradient collectives (RCCL all-reduces on the side stream, captured like
    any other launch; tests/test_graph_step.py::test_rccl_collectives_inside_the_captured_step).  Ranks may mix replayed and eager
    iterations freely -- both enqueue the same collectives in the same order -- but a timed run wants one mode: `agree(dist)`
    (one 1-element all-reduce, OUTSIDE any capture, at a point every rank reaches) turns the graphs off on EVERY rank if any
    rank's prepared capture failed.
"""
import torch

from . import ops


def collectives_capturable(group=None):
    """(ok, why not).  With more than one rank the captured step holds the gradient collectives: that works with RCCL (backend
    "nccl": the all-reduce is a kernel launch on the optimiser's side stream) and not with gloo, which moves the bucket through the
    host and synchronises the stream -- under capture that is not an error one can undo but a crash inside the transport (seen as
    SIGSEGV with two gloo ranks on one GPU), so it is not attempted."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) <= 1 and not ops_force_collectives():
        return True, None
    backend = str(dist.get_backend(group))
    if backend != "nccl":
        return False, f"process-group backend {backend!r}: only RCCL (\"nccl\") collectives can be captured with the step"
    return True, None


def _rccl_group_is_up():
    import torch.distributed as dist
    try:
        return bool(dist.is_available() and dist.is_initialized() and "nccl" in str(dist.get_backend()))
    except Exception:
        return False


# ---- a capture beside ProcessGroupNCCL's watchdog thread (round 6) -----------------------------------------------------------------
# The process group runs a WATCHDOG THREAD that polls every work object still on its list (`WorkNCCL::isCompleted` -> hipEventQuery,
# every 100 ms; a finished eager collective stays listed until the next poll after it finished), and an exception on that thread is
# std::terminate: the rank is gone with "Process group watchdog thread terminated with exception: HIP error: operation not permitted
# when stream is capturing".  Seen in 2 of 5 runs of the GPU suite (profiles/r06_watchdog_vs_global_capture.txt).  What the runtime
# refuses while a capture is open, measured (tools/diag/capture_event_query_rules.py, profiles/r06_capture_event_query_rules.txt):
#   * GLOBAL-mode capture: EVERY event query of EVERY other thread (whose own mode is global -- the watchdog's), whatever stream
#     the event belongs to;
#   * THREAD-LOCAL capture: other threads keep their API, EXCEPT for events whose stream is part of the capture now -- also events
#     recorded long BEFORE the capture on a side stream that has joined it since ("operation not permitted on an event last recorded
#     in a capturing stream").  The end event of the previous eager step's all-reduce sits on exactly such a stream (the group's
#     internal RCCL stream / the optimiser's side stream), and it is still on the watchdog's list for up to 100 ms after the step.
# (The torch build in this image has no wait for listed work in CUDAGraph::capture_begin -- older ones spun there -- and the watchdog
# queries without a relaxed-mode guard.)  Hence TWO measures whenever an RCCL group is up:
#   1. the capture is thread-local whatever the caller asked for (capture_mode_beside_a_process_group): queries of unrelated events
#      -- a pin-memory thread's, the allocator's in another thread, the watchdog's on another communicator -- stay legal;
#   2. the watchdog's list is DRAINED before the capture begins (drain_process_group_watchdog): with the device synchronised every
#      listed work is complete, and the next poll -- at most one period away -- retires it; works issued INSIDE a capture are never
#      listed (ProcessGroupNCCL checks the capture status), so nothing is left for the thread to ask about.
WATCHDOG_PERIOD_S = 0.1          # kWatchdogThreadSleepMillis
WATCHDOG_DRAIN_S = 0.3           # three periods: a poll that was asleep when the device went idle, the loop's own run time, slack


def capture_mode_beside_a_process_group(requested="global"):
    """the hipStreamCaptureMode of a capture: thread-local whenever an RCCL process group is up (measure 1 above)"""
    return "thread_local" if requested == "global" and _rccl_group_is_up() else requested


def drain_process_group_watchdog():
    """measure 2 above; called with nothing in flight that the caller has not synchronised.  Costs 0.3 s per CAPTURE (a run has a
    handful: one per step variant), nothing per step."""
    if _rccl_group_is_up():
        import time
        torch.cuda.synchronize()
        time.sleep(WATCHDOG_DRAIN_S)


def _clear_sticky_hip_error():
    """hipGetLastError() until it reports success: a failed stream capture leaves the runtime's per-thread sticky error set, and the
    library's launch checks (hipGetLastError after every launch) would blame the next kernel for it."""
    try:
        hip = ops.Profile._hip_runtime()
        for _ in range(8):
            if hip.hipGetLastError() == 0:
                break
    except Exception:
        pass


def ops_force_collectives():
    import os
    return os.environ.get("ANERF_FORCE_COLLECTIVES") == "1"


class StaticBatch:
    """Persistent device tensors for a loader's batches: `load(batch)` copies the values in (non-blocking, stream-ordered) and
    returns the dict of persistent tensors -- the static input addresses a captured step reads."""

    def __init__(self, device):
        self.device, self.buf = torch.device(device), {}

    def load(self, batch):
        """one persistent tensor per (key, shape, dtype): a batch of another size gets its own set (and its own graph)"""
        out = {}
        for k, v in batch.items():
            if not torch.is_tensor(v):
                out[k] = v
                continue
            slot = (k, tuple(v.shape), v.dtype)
            b = self.buf.get(slot)
            if b is None:
                b = self.buf[slot] = torch.empty(v.shape, dtype=v.dtype, device=self.device)
            if v.data_ptr() != b.data_ptr():
                b.copy_(v, non_blocking=True)
            out[k] = b
        return out

    @staticmethod
    def identity(sb):
        """what a captured graph has frozen of a loaded batch: (key, address) of every tensor entry + which entries are None.
        Part of the graph key in Trainer._train_batch_graphed: an entry that changes shape / dtype or flips between tensor and
        None at the same n_rays gets ANOTHER graph instead of a replay that reads the old buffer."""
        return tuple(sorted((k, v.data_ptr() if torch.is_tensor(v) else None) for k, v in sb.items()
                            if torch.is_tensor(v) or v is None))


class GraphedTrainStep:
    def __init__(self, step_fn, caster, optimizer, eager_steps=3, enabled=True, capture_error_mode="thread_local", warm_each_key=True):
        """step_fn(i) -> dict of tensors; caster: the RayCaster (its DeviceRng and embedders supply seed / offset / tau);
        optimizer: the FusedAdam of the step.  The first `eager_steps` calls run step_fn eagerly (lazy one-off initialisation
        -- kernel attributes, allocator pools, index caches -- must not fall inside a capture).
        capture_error_mode: torch.cuda.graph's (hipStreamCaptureMode).  "thread_local" (the default since late round 6): only the
        capturing thread's own allocation- / query-class HIP calls are errors during the few milliseconds of a capture; "global"
        makes them errors in EVERY thread -- a DataLoader's pin-memory thread (run_nerf.py trains from one), a process group's
        watchdog -- and a refused call in such a thread usually ends it.  The autograd worker's launches into the capturing stream
        are captured in either mode.  With an RCCL process group up the capture is thread-local whatever is asked for here, and the
        group's watchdog thread is given time to retire the eager collectives it still lists (see the text above
        capture_mode_beside_a_process_group).
        warm_each_key: every new (due groups, key) variant runs eagerly once before it is captured (see the module text)."""
        self.capture_error_mode = capture_error_mode
        self.warm_each_key = bool(warm_each_key)       # False: capture a new variant at first sight (the caller vouches for warm caches)
        self.step_fn, self.caster, self.opt = step_fn, getattr(caster, "module", caster), optimizer
        self.eager_left, self.enabled = int(eager_steps), bool(enabled)
        self.block = None
        self.graphs = {}            # (due-groups tuple, caller's key) -> (CUDAGraph, outputs, fills per step)
        self.pool = None
        self.replays = self.captures = self.eager_calls = self.failed_captures = 0
        self.why_eager = None
        self.warmed = set()         # graph keys whose code path has run eagerly (prepare() vouches for its own)
        self.eager_only = {}        # graph key -> why its capture failed (runs eagerly from then on)

    # ---- host-side bookkeeping the eager path does inside FusedAdam / DeviceRng -------------------------------------
    def _capturable(self):
        c = self.caster
        if getattr(c.embed_fn, "freq_schedule", False) or getattr(c.embeddirs_fn, "freq_schedule", False):
            return "--freq_schedule re-uploads its band factors every iteration"
        if len(self.opt.param_groups) > 4:
            return "more than 4 optimiser groups"
        import torch.distributed as dist
        up = dist.is_available() and dist.is_initialized()         # (the answer can only change when a process group comes up)
        if self.__dict__.get("_cc", (None,))[0] != up:
            self._cc = (up, collectives_capturable())
        ok, why = self._cc[1]
        return None if ok else why

    def _fill_block(self, i, due):
        """the values iteration i runs with, exactly as the eager calls would pass them as kernel arguments"""
        rng = self.caster.rng()
        rng.follow_torch_seed()
        self.block.set_rng(rng.seed, rng.offset)
        self.block.set_tau(*self.caster._taus())
        opt = self.opt
        self.block.set_adam([(g["lr"], g["betas"][0], g["betas"][1], (opt._steps[gi] + 1) if gi in due else 0, opt._grad_scale[gi])
                             for gi, g in enumerate(opt.param_groups)])

    def _after_replay(self, due, fills):
        """what step_fn's host code did at capture time and a replay does not repeat"""
        opt = self.opt
        self.caster.rng().offset += fills
        stepped = []
        for gi in due:
            opt._steps[gi] += 1
            opt._grad_scale[gi] = 1.0
            stepped += opt.param_groups[gi]["params"]
        # parameters changed behind torch's back: eager consumers (weight-image caches of an eval render, checkpoints) key on versions
        torch.autograd.graph.increment_version(stepped)

    def _capture(self, i, due):
        opt, rng = self.opt, self.caster.rng()
        torch.cuda.synchronize()
        snap = (rng.offset, list(opt._steps), list(opt._grad_scale))
        # the images of the previous optimiser step must be re-gathered INSIDE the graph, whatever the caches say right now
        torch.autograd.graph.increment_version([p for g in opt.param_groups for p in g["params"]])
        g = torch.cuda.CUDAGraph()
        self.block.fills = 0
        drain_process_group_watchdog()         # (no-op without an RCCL process group)
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()      # ONE pool for every graph of this stepper (they never run concurrently);
        # Python's cycle collector must not run INSIDE the capture: it would free whatever cyclic garbage earlier iterations left
        # behind (autograd contexts holding workspaces, pinned staging buffers), and releasing device / pinned memory records and
        # queries events on streams -- not permitted while one of them is capturing (seen as a hard abort: a collection triggered by
        # the step's own allocations, in the middle of the capture).  Collect now, keep the collector off until the capture ends.
        import gc
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        dev = self.block.buf.device
        ctx = torch.cuda.graph(g, pool=self.pool, capture_error_mode=capture_mode_beside_a_process_group(self.capture_error_mode))
        entered = False
        try:
            with ops.step_block(self.block):
                ctx.__enter__()
                entered = True
                out = self.step_fn(i)
                ctx.__exit__(None, None, None)
                entered = False
        except BaseException:
            # torch.cuda.graph.__exit__ is not exception-safe.  Two cases:
            #  (a) the iteration raised a Python exception, the HIP capture itself is intact: capture_end() below succeeds and tidies up;
            #  (b) the capture was INVALIDATED (a synchronising call, a host read, a pageable copy, an unjoined side stream):
            #      hipStreamEndCapture fails inside capture_end() BEFORE torch (1) stops routing this stream's allocations into the graph
            #      pool, (2) takes the default generator's state out of its "capturing" stage -- every later capture_begin() would then
            #      throw "Cannot register the state during capturing stage" and the half-built CUDAGraph's destructor would terminate
            #      the process ("The graph should be registered to the state") -- and (3) with the runtime's sticky error still set: the
            #      next launch of the eager fallback would report "operation failed due to a previous error during capture".
            #  (c) capture_begin() itself raised after the stream had started capturing (torch.cuda.graph.__enter__ switches to the
            #      capture stream first): the stream would stay in capture mode and stay current -- ended and left like (a) / (b).
            ended = False
            begun = entered
            if not entered:
                try:
                    begun = bool(torch.cuda.is_current_stream_capturing())
                except Exception:
                    begun = True          # (an invalidated capture may make the query itself fail)
            if begun:
                try:
                    g.capture_end()
                    ended = True
                except Exception:
                    pass
            try:
                ctx.stream_ctx.__exit__(None, None, None)        # (a no-op error when __enter__ failed before switching streams)
            except Exception:
                pass
            if not ended:
                idx = dev.index if dev.index is not None else torch.cuda.current_device()
                try:
                    torch._C._cuda_endAllocateToPool(idx, self.pool)                       # (1)
                except Exception:
                    pass
                try:
                    gen = torch.cuda.default_generators[idx]
                    gen.graphsafe_set_state(gen.clone_state())                              # (2) a fresh state object (same seed / offset)
                except Exception:
                    pass
            _clear_sticky_hip_error()                                                       # (3)
            if hasattr(opt, "abort_step"):
                opt.abort_step()          # collective handles created under the dead capture
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
            _clear_sticky_hip_error()
            self.failed_captures += 1
            raise
        finally:
            if gc_was_on:
                gc.enable()
            # the capture recorded the launches without running them (or failed half way): put the host-side counters back
            fills = self.block.fills
            rng.offset, opt._steps[:], opt._grad_scale[:] = snap[0], snap[1], snap[2]
        self.captures += 1
        return g, out, fills

    def _mark_eager_only(self, gk, err):
        why = f"{type(err).__name__}: {err}"[:300]
        self.eager_only[gk] = why
        import warnings
        warnings.warn(f"GraphedTrainStep: capture of variant {gk!r} failed ({why}); this variant runs eagerly from now on")

    def agree(self, dist, group=None):
        """More than one rank, at a point EVERY rank reaches (after prepare(), before a timed stretch): one 1-element all-reduce
        (MIN) of "all my captures so far succeeded", outside any capture.  If any rank had a failure, every rank switches its
        graphs off -- the whole job then runs the eager step.  Returns True when the graphs stay on."""
        ok = 1.0 if (self.enabled and not self.eager_only and self._capturable() is None) else 0.0
        flag = torch.tensor([ok], dtype=torch.float32, device=self.block.buf.device if self.block is not None else "cuda")
        if dist is not None and dist.is_available() and dist.is_initialized():
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if float(flag.item()) < 1.0:
            if self.enabled:
                self.why_eager = "a rank's capture failed (agreed over all ranks)" if ok else (self.why_eager or "this rank's capture failed")
            self.enabled = False
        return self.enabled

    def prepare(self, i, key=None, due=None):
        """capture the graph iteration i would replay, now (nothing runs, no counter moves): keeps the capture -- milliseconds of
        host time -- out of a timed or latency-sensitive stretch.  The step must have run eagerly before (warm caches)."""
        why = self._capturable()
        if not self.enabled or why is not None:
            return False
        if self.block is None:
            self.block = ops.StepBlock(next(iter(self.opt.params)).device)
        due = tuple(self.opt._due(i)) if due is None else tuple(due)
        gk = due if key is None else (due, key)
        if gk in self.eager_only:
            return False
        if gk not in self.graphs:
            self.warmed.add(gk)
            try:
                self.graphs[gk] = self._capture(i, due)
            except Exception as e:
                self._mark_eager_only(gk, e)
                return False
        return True

    def step(self, i, key=None, due=None):
        """iteration i.  key: anything hashable that selects among captured variants of step_fn (batch size, pose layout, a mode
        flag): each (due groups, key) pair gets its own graph.  due: the optimiser groups this iteration steps (default: the
        optimiser's own cadence, FusedAdam._due(i))."""
        why = None if self.enabled else "disabled"
        if why is None and self.eager_left > 0:
            self.eager_left -= 1
            why = "warm-up"
            d0 = tuple(self.opt._due(i)) if due is None else tuple(due)
            self.warmed.add(d0 if key is None else (d0, key))          # this variant's code path has now run eagerly
        if why is None:
            why = self._capturable()
            if why is not None and self.why_eager != why:
                self.why_eager = why
                import warnings
                warnings.warn(f"GraphedTrainStep: running eagerly ({why})")
        if why is not None:
            self.eager_calls += 1
            return self.step_fn(i)
        if self.block is None:
            self.block = ops.StepBlock(next(iter(self.opt.params)).device)
        due = tuple(self.opt._due(i)) if due is None else tuple(due)
        gk = due if key is None else (due, key)
        hit = self.graphs.get(gk)
        if hit is None:
            if gk in self.eager_only:
                self.eager_calls += 1
                return self.step_fn(i)
            if self.warm_each_key and gk not in self.warmed:
                # first sight of this variant: its code path runs eagerly once (lazy one-off work must not fall inside a capture)
                self.warmed.add(gk)
                self.eager_calls += 1
                return self.step_fn(i)
            try:
                hit = self.graphs[gk] = self._capture(i, due)
            except Exception as e:
                self._mark_eager_only(gk, e)
                self.eager_calls += 1
                return self.step_fn(i)
        g, out, fills = hit
        self._fill_block(i, due)
        self.block.write()
        g.replay()
        self._after_replay(due, fills)
        self.replays += 1
        return out
