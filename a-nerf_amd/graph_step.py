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
Not capturable here: `--freq_schedule` (its band factors are re-uploaded per iteration), a batch whose pose layout (number of
distinct poses) changes -- `step()` falls back to the eager call for those, loudly once.
"""
import torch

from . import ops


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


class GraphedTrainStep:
    def __init__(self, step_fn, caster, optimizer, eager_steps=3, enabled=True, capture_error_mode="global"):
        """step_fn(i) -> dict of tensors; caster: the RayCaster (its DeviceRng and embedders supply seed / offset / tau);
        optimizer: the FusedAdam of the step.  The first `eager_steps` calls run step_fn eagerly (lazy one-off initialisation
        -- kernel attributes, allocator pools, index caches -- must not fall inside a capture).
        capture_error_mode: torch.cuda.graph's (hipStreamCaptureMode).  "global" (the default, the tested one) makes ANY thread's
        allocation-class HIP call during the few milliseconds of a capture an error -- e.g. a DataLoader's pin-memory thread; pass
        "thread_local" when such threads run beside the trainer (the autograd worker's launches into the capturing stream are
        captured in either mode)."""
        self.capture_error_mode = capture_error_mode
        self.step_fn, self.caster, self.opt = step_fn, getattr(caster, "module", caster), optimizer
        self.eager_left, self.enabled = int(eager_steps), bool(enabled)
        self.block = None
        self.graphs = {}            # (due-groups tuple, caller's key) -> (CUDAGraph, outputs, fills per step)
        self.pool = None
        self.replays = self.captures = self.eager_calls = 0
        self.why_eager = None

    # ---- host-side bookkeeping the eager path does inside FusedAdam / DeviceRng -------------------------------------
    def _capturable(self):
        c = self.caster
        if getattr(c.embed_fn, "freq_schedule", False) or getattr(c.embeddirs_fn, "freq_schedule", False):
            return "--freq_schedule re-uploads its band factors every iteration"
        if len(self.opt.param_groups) > 4:
            return "more than 4 optimiser groups"
        return None

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
        # Python's cycle collector must not run INSIDE the capture: it would free whatever cyclic garbage earlier iterations left
        # behind (autograd contexts holding workspaces, pinned staging buffers), and releasing device / pinned memory records and
        # queries events on streams -- not permitted while one of them is capturing (seen as a hard abort: a collection triggered by
        # the step's own allocations, in the middle of the capture).  Collect now, keep the collector off until the capture ends.
        import gc
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            with ops.step_block(self.block):
                with torch.cuda.graph(g, pool=self.pool, capture_error_mode=self.capture_error_mode):
                    out = self.step_fn(i)
        finally:
            if gc_was_on:
                gc.enable()
            # the capture recorded the launches without running them (or failed half way): put the host-side counters back
            fills = self.block.fills
            rng.offset, opt._steps[:], opt._grad_scale[:] = snap[0], snap[1], snap[2]
        if self.pool is None:
            self.pool = g.pool()          # later graphs (other cadence phases) share it: they never run concurrently
        self.captures += 1
        return g, out, fills

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
        if gk not in self.graphs:
            self.graphs[gk] = self._capture(i, due)
        return True

    def step(self, i, key=None, due=None):
        """iteration i.  key: anything hashable that selects among captured variants of step_fn (batch size, pose layout, a mode
        flag): each (due groups, key) pair gets its own graph.  due: the optimiser groups this iteration steps (default: the
        optimiser's own cadence, FusedAdam._due(i))."""
        why = None if self.enabled else "disabled"
        if why is None and self.eager_left > 0:
            self.eager_left -= 1
            why = "warm-up"
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
            hit = self.graphs[gk] = self._capture(i, due)
        g, out, fills = hit
        self._fill_block(i, due)
        self.block.write()
        g.replay()
        self._after_replay(due, fills)
        self.replays += 1
        return out
