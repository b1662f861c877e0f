"""Loss + optimiser step of the training loop as three kernel launches.

Mirrors, with the same names and meaning where the reference has them:
  * `_compute_nerf_loss` (core/trainer.py:353-380) -> `fused_nerf_loss`: background composite + MSE/L1/Huber of the
    fine and coarse heads + the PSNR numerator + the gradients w.r.t. the rendered maps, one kernel;
  * `torch.optim.Adam(params=grad_vars, lr=lrate, betas=(0.9, 0.999))` (trainer.py:173-183) -> `FusedAdam`: all
    parameters live in ONE flat fp32 buffer (each `p.data` / `p.grad` is a view of it), one kernel per step,
    fused with `zero_grad()` and with `get_gradnorm` (trainer.py:192-203, 48 `.item()` syncs in the reference);
  * `decay_optimizer_lrate` (trainer.py:173-183) works unchanged on `FusedAdam.param_groups` / `.state`.
The flat gradient buffer is also the DP bucket: `all_reduce_grads()` is one RCCL all-reduce on it, the 1/world
scale is folded into the Adam kernel.  No CPU fallback: the kernels come from libanerf_hip.so.
"""
import os
import weakref

import torch
import torch.distributed as dist


def _force_collectives():
    """ANERF_FORCE_COLLECTIVES=1: issue the collectives even in a one-rank process group (exercises the RCCL code path --
    communicator, side stream, async handle -- on a single-GPU box; sums over one rank are the identity)."""
    return os.environ.get("ANERF_FORCE_COLLECTIVES") == "1"

from . import ops


class FusedAdam:
    """Adam over ONE flat fp32 buffer that holds every parameter of every group.

    Interface of torch.optim.Adam (`param_groups`, `state`, `step()`, `zero_grad()`, `add_param_group()`,
    `state_dict()` / `load_state_dict()` in torch's format, so the reference's checkpoint keys `optimizer_state_dict`
    / `pose_optimizer_state_dict`, trainer.py:498-505, round-trip), plus what the ray-sharded DP step needs:

    * groups are consecutive SEGMENTS of the flat parameter / gradient / moment buffers (each padded to 16 bytes), so the
      gradient bucket of a step is one contiguous range and `all_reduce_grads()` is ONE collective whatever is in it;
    * a group may carry `step_every = k`: it is stepped (and its gradients all-reduced and cleared) only on iterations
      with `i % k == 0` and accumulates local gradients in between -- the reference's pose-optimiser cadence
      (`if i % args.opt_pose_step == 0: pose_optimizer.step()`, trainer.py:476-478; opt_pose_step = 20 in mixamo.txt:48).
      Summing the ranks' ACCUMULATED gradients once per k iterations equals all-reducing them every iteration (linearity);
    * `group_optimizer(g)` is a view with the torch optimiser surface over one group -- what a trainer holds as its
      `pose_optimizer` (own lr decay through `param_groups`, own `state_dict`).

    Difference from torch.optim.Adam: a parameter whose `.grad` is None at `step()` is treated as having a zero gradient
    (its moments decay and it moves along its first moment) -- torch skips it and keeps its own step count; one flat
    buffer has one step count per group.  No shipped training path leaves a trainable parameter without a gradient."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        params = list(params)
        self.param_groups = []
        self._steps = []                  # Adam step count per group
        self._grad_scale = []             # 1/world after an all-reduce, consumed by the next step of that group
        self.flat = self.flat_grad = self.exp_avg = self.exp_avg_sq = None
        self.norms = None
        self._pending = None              # optimizer state loaded before the buffers exist
        self.overlap = False              # see enable_overlap()
        self.split_adam = True            # overlap: Adam of the first-reduced network runs while the second one's collective is in flight
        self.overlap_stats = {"early_collectives": 0, "main_collectives": 0, "split_adam_steps": 0}     # counters (tests, bench record)
        self._async = []                  # [(work handle, lo, hi)]: all-reduces started during the backward, in launch order
        self._early = []                  # bucket ranges whose Adam may run before the rest is reduced (see all_reduce_grads)
        self._side = None
        self._defaults = {"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": 0, "amsgrad": False, "step_every": 1}
        if params and isinstance(params[0], dict):
            for g in params:
                self.add_param_group(g)
        else:
            self.add_param_group({"params": params})

    @classmethod
    def from_torch(cls, optimizer, pose_optimizer=None, pose_step_every=1):
        """One FusedAdam over the parameters of the torch Adam(s) run_nerf.py builds (create_raycaster's `optimizer`,
        run_nerf.py:518; create_popt's `pose_optimizer`, :523), taking over their hyper-parameters and -- when they were restored
        from a checkpoint -- their state.  The pose optimiser's groups come after the network's and get
        `step_every = pose_step_every` (args.opt_pose_step).  The parameters must be on the GPU already; the torch optimisers are
        not used afterwards: hand the Trainer `fused.group_optimizer(0)` / `fused.group_optimizer(1)` in their place."""
        groups, state, base = [], {}, 0
        for opt, every in ((optimizer, 1), (pose_optimizer, int(pose_step_every))):
            if opt is None:
                continue
            if not isinstance(opt, torch.optim.Adam) or any(g.get("amsgrad") or g.get("weight_decay") or g.get("maximize") for g in opt.param_groups):
                raise TypeError("FusedAdam.from_torch: plain torch.optim.Adam only (no amsgrad / weight decay / maximize)")
            sd = opt.state_dict()
            for g in opt.param_groups:
                groups.append({"params": list(g["params"]), "lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "step_every": every})
            state.update({base + int(k): v for k, v in sd["state"].items()})
            base += sum(len(g["params"]) for g in opt.param_groups)
        fused = cls(groups)
        if state:
            saved = fused.state_dict()
            fused.load_state_dict({"state": state, "param_groups": saved["param_groups"]})
        return fused

    def add_param_group(self, group):
        """torch.optim.Optimizer.add_param_group; extra key `step_every` (default 1).  Call before the first step."""
        if self.flat is not None:
            raise RuntimeError("FusedAdam.add_param_group: the flat buffers already exist; add groups before the first step")
        g = dict(self._defaults)
        g.update({k: v for k, v in group.items() if k != "params"})
        g["betas"] = tuple(g["betas"])
        g["params"] = [p for p in group["params"] if p.requires_grad]
        if not g["params"]:
            raise ValueError("FusedAdam: no parameters")
        if g["weight_decay"] or g["amsgrad"]:
            raise NotImplementedError("FusedAdam implements plain Adam (weight_decay 0, no amsgrad), as the reference uses it")
        self.param_groups.append(g)
        self._steps.append(0)
        self._grad_scale.append(1.0)

    @property
    def params(self):
        return [p for g in self.param_groups for p in g["params"]]

    # ---- flat storage ---------------------------------------------------------------------------------------
    def _segments(self):
        """[(offset, padded length)] of every group inside the flat buffers"""
        out, o = [], 0
        for g in self.param_groups:
            n = sum(p.numel() for p in g["params"])
            npad = (n + 3) // 4 * 4
            out.append((o, npad))
            o += npad
        return out

    def _views(self, flat, group=None):
        out = []
        for gi, (g, (o, _)) in enumerate(zip(self.param_groups, self._segments())):
            for p in g["params"]:
                if group is None or group == gi:
                    out.append(flat[o:o + p.numel()].view(p.shape))
                o += p.numel()
        return out

    def materialize(self):
        """Move every parameter (and its gradient) into the flat buffers; idempotent.  Call after `.to(device)`."""
        dev = self.params[0].device
        if self.flat is not None and self.flat.device == dev:
            return
        if dev.type != "cuda":
            raise RuntimeError("FusedAdam runs on the GPU only (no CPU fallback); move the model first")
        seg = self._segments()
        self.numel = sum(p.numel() for p in self.params)
        ntot = seg[-1][0] + seg[-1][1]
        self.flat = torch.zeros(ntot, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(ntot, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(ntot, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(ntot, dtype=torch.float32, device=dev)
        self.norms = torch.zeros(2, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, v, g in zip(self.params, self._views(self.flat), self._views(self.flat_grad)):
                v.copy_(p.data)
                if p.grad is not None:
                    g.copy_(p.grad)
                p.data = v
                p.grad = g
        if self._pending is not None:
            sd, self._pending = self._pending, None
            self.load_state_dict(sd)

    # ---- in-place gradient accumulation (opt-in) --------------------------------------------------------------
    def attach(self, caster, pose_layer=None):
        """Opt in to in-place gradient accumulation: the caster's one-call backward adds the parameter gradients straight
        into this optimiser's flat bucket and reports none to autograd (no 48 AccumulateGrad launches per step).  While
        attached, `torch.autograd.grad(loss, params)` on these parameters yields None -- call `detach()` (or never attach)
        when the gradients are wanted as autograd results.  `caster` may be the RayParallel wrapper.
        pose_layer: a PoseOptLayer whose `pelvis` / `bones` are parameters of this optimiser: its fused backward then adds
        their gradients in place too (same contract)."""
        caster = getattr(caster, "module", caster)
        caster._anerf_grad_sink = weakref.ref(self)
        self._attached = weakref.ref(caster)
        if pose_layer is not None:
            pose_layer._anerf_grad_sink = weakref.ref(self)
            self._attached_pose = weakref.ref(pose_layer)
        if self.params[0].is_cuda:
            self.materialize()           # the bucket must exist before the first backward for that one to land in it
        return self

    def detach(self):
        caster = self._attached() if getattr(self, "_attached", None) is not None else None
        if caster is not None and getattr(caster, "_anerf_grad_sink", None) is not None and caster._anerf_grad_sink() is self:
            caster._anerf_grad_sink = None
        self._attached = None
        layer = self._attached_pose() if getattr(self, "_attached_pose", None) is not None else None
        if layer is not None and getattr(layer, "_anerf_grad_sink", None) is not None and layer._anerf_grad_sink() is self:
            layer._anerf_grad_sink = None
        self._attached_pose = None

    def __del__(self):
        try:
            self.detach()
        except Exception:
            pass

    def owns_grads(self, params, device):
        """True when every parameter's .grad is (still) a contiguous fp32 view into this optimiser's flat gradient bucket."""
        fg = self.flat_grad
        if fg is None or fg.device != device:
            return False
        lo, hi = fg.data_ptr(), fg.data_ptr() + fg.numel() * 4
        for p in params:
            g = p.grad
            if g is None or not p.requires_grad or g.dtype != torch.float32 or not g.is_contiguous() or g.device != device:
                return False
            if not (lo <= g.data_ptr() and g.data_ptr() + g.numel() * 4 <= hi):
                return False
        return True

    @property
    def state(self):
        """torch-style per-parameter state (read-only views; `state[p]['step']` is what decay_optimizer_lrate reads)."""
        return self._state_of(None)

    def _state_of(self, group):
        if self.flat is None:
            return {}
        out = {}
        for gi, g in enumerate(self.param_groups):
            if (group is not None and group != gi) or self._steps[gi] == 0:
                continue
            ea, es = self._views(self.exp_avg, gi), self._views(self.exp_avg_sq, gi)
            out.update({p: {"step": self._steps[gi], "exp_avg": a, "exp_avg_sq": s} for p, a, s in zip(g["params"], ea, es)})
        return out

    # ---- step -----------------------------------------------------------------------------------------------
    def _due(self, i, only=None):
        """indices of the groups that step on iteration `i` (None: all of them)"""
        return [gi for gi, g in enumerate(self.param_groups)
                if (only is None or gi == only) and (i is None or g["step_every"] <= 1 or i % g["step_every"] == 0)]

    # ---- overlapping the all-reduce with the backward (data parallel, opt-in) --------------------------------------
    def enable_overlap(self, on=True):
        """With an attached caster (attach()) and more than one rank: the all-reduce of the FINE network's gradients is
        started in the middle of the backward -- they are complete once the fine pass is enqueued -- on a side stream,
        and runs under the coarse pass; all_reduce_grads() then reduces the rest and joins.  Same sums as the single
        collective (a split all-reduce adds the same numbers); needs an even ray split (no per-rank `weight`).
        Contract: exactly ONE backward per all_reduce_grads() / step().  A second backward (gradient accumulation over
        chunks, a retained graph) would add local gradients into a range that is already reduced -- possibly while the
        collective is still in flight -- and the ranks would diverge silently; it raises instead (`check_one_backward`).
        Accumulate without overlap, or all-reduce between the backwards."""
        self.overlap = bool(on)
        return self

    def check_one_backward(self):
        """Called by the caster's backward BEFORE it enqueues anything: with overlap on, an early all-reduce that has not been
        consumed by all_reduce_grads() means this is a second backward of the same step."""
        if self.overlap and self._async:
            raise RuntimeError("FusedAdam overlap: a second backward arrived before all_reduce_grads() consumed the early "
                               "all-reduce of the first; its gradients would be added to an already-reduced bucket range. "
                               "Use enable_overlap(False) for gradient accumulation, or call all_reduce_grads() between backwards.")

    def _drop_async(self):
        """forget early all-reduces nobody consumed (all_reduce_grads() was skipped after a backward): join them first so that
        nothing is in flight on the bucket, then clear the handles -- they must not be mistaken for the next step's."""
        if self._async:
            pend, self._async = self._async, []
            for work, _, _ in pend:
                work.wait()
            if self._side is not None and self.flat_grad is not None:
                torch.cuda.current_stream(self.flat_grad.device).wait_stream(self._side)
        self._join_early()

    def abort_step(self):
        """Forget every in-flight collective handle WITHOUT waiting on it: for handles created under a stream capture that was
        then abandoned (graph_step.GraphedTrainStep: nothing was launched, there is nothing to join, and `work.wait()` on such a
        handle would touch events of a dead capture)."""
        self._async, self._early = [], []

    def _join_early(self):
        """join the collectives all_reduce_grads() left in flight for step()'s split Adam (see there)"""
        early, self._early = self._early, []
        for _, _, rest in early:
            for work in rest:
                work.wait()
        if early and self._side is not None and self.flat_grad is not None:
            torch.cuda.current_stream(self.flat_grad.device).wait_stream(self._side)

    def begin_async_all_reduce(self, params, group=None):
        """Called by the caster's backward at the points where a network's parameter gradients are complete (after the fine pass;
        after the parameter part of the coarse pass): all-reduce (sum) the flat-bucket range that holds `params` (they must be
        contiguous in the bucket) asynchronously on the side stream.  No-op without a process group."""
        if not (dist.is_available() and dist.is_initialized()):
            return
        if dist.get_world_size(group) <= 1 and not _force_collectives():
            return
        fg = self.flat_grad
        base = fg.data_ptr()
        # the bucket range of this parameter list: computed once per list (48 data_ptr / numel calls per hook otherwise) and checked
        # against the gradients' current addresses through the first and last tensor
        rk = (id(params[0]), id(params[-1]), len(params), base, params[0].grad.data_ptr(), params[-1].grad.data_ptr())
        rng = self.__dict__.setdefault("_range_cache", {}).get(rk)
        if rng is None:
            lo = min(p.grad.data_ptr() for p in params) - base
            hi = max(p.grad.data_ptr() + p.grad.numel() * 4 for p in params) - base
            ok = not (lo < 0 or hi > fg.numel() * 4 or (hi - lo) != 4 * sum(p.numel() for p in params))
            if len(self._range_cache) > 64:
                self._range_cache.clear()
            rng = self._range_cache[rk] = (lo // 4, hi // 4, ok)
        lo, hi, ok = rng
        if not ok:
            return                                   # not one contiguous run of the bucket: leave it to the main collective
        if any(not (hi <= alo or lo >= ahi) for _, alo, ahi in self._async):
            return                                   # overlaps a range already in flight (same hook twice): the main collective's job
        if self._side is None:
            self._side = torch.cuda.Stream(device=fg.device)
        self._side.wait_stream(torch.cuda.current_stream(fg.device))      # everything enqueued so far produced these gradients
        with torch.cuda.stream(self._side):
            work = dist.all_reduce(fg[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=True)
        self._async.append((work, lo, hi))
        self.overlap_stats["early_collectives"] += 1

    def all_reduce_grads(self, group=None, i=None, weight=None, only_group=None):
        """Sum the gradient bucket over ranks; the 1/world scale is applied inside step().  ONE collective over the
        contiguous range of the groups due on iteration `i` (see class doc; i=None: every group).
        weight: multiply the local gradients first -- `parallel.shard_weight()` for ragged ray shards, so that the reduced
        gradient is the global-batch mean gradient even when the ranks own different numbers of rays."""
        self.materialize()
        world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        if world <= 1 and not (_force_collectives() and dist.is_available() and dist.is_initialized()):
            return
        due = self._due(i, only_group)                # only_group: just that parameter group (the pose group has stopped)
        seg = self._segments()
        runs = []                                     # maximal runs of consecutive due groups -> one collective each
        for gi in due:
            if runs and runs[-1][-1] == gi - 1:
                runs[-1].append(gi)
            else:
                runs.append([gi])
        pieces = [(seg[run[0]][0], seg[run[-1]][0] + seg[run[-1]][1]) for run in runs]
        self._join_early()                            # (a previous call's leftovers, had step() not consumed them: joined, not dropped)
        if self._async:                               # ranges already (being) reduced under the backward: reduce around them
            pend, self._async = self._async, []
            if weight is not None and float(weight) != 1.0:
                raise RuntimeError("FusedAdam: overlap needs an even ray split (the early all-reduces were not weighted)")
            for _, alo, ahi in pend:
                cut = []
                for lo, hi in pieces:
                    if ahi <= lo or alo >= hi:
                        cut.append((lo, hi))
                    else:
                        if lo < alo:
                            cut.append((lo, alo))
                        if ahi < hi:
                            cut.append((ahi, hi))
                pieces = cut
            main = torch.cuda.current_stream(self.flat_grad.device)
            if len(pend) > 1 and not pieces and self.split_adam:
                # nothing left to reduce here and more than one collective in flight: join only the FIRST now -- step() runs the
                # Adam of its range while the later ones are still on the wire, and joins those before it touches their ranges
                work, alo, ahi = pend[0]
                work.wait()
                self._early = [(alo, ahi, [w for w, _, _ in pend[1:]])]
            else:
                for work, _, _ in pend:
                    work.wait()                       # the current stream now waits for the early collectives
                if self._side is not None:
                    main.wait_stream(self._side)
        for lo, hi in pieces:
            buf = self.flat_grad[lo:hi]
            if weight is not None and float(weight) != 1.0:
                buf.mul_(float(weight))
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
            self.overlap_stats["main_collectives"] += 1
        for gi in due:
            self._grad_scale[gi] = 1.0 / world

    @torch.no_grad()
    def step(self, zero_grad=False, want_norms=False, i=None, only_group=None):
        """One Adam update of every group due on iteration `i` (None: all groups).  zero_grad: also clear the gradients of
        the groups that stepped (= the reference's `_optim_step`; groups that are not due keep accumulating).
        want_norms: returns a [2] device tensor (total_norm, avg_norm) of group 0's gradients at this step
        (`get_gradnorm(ray_caster)`, trainer.py:192-203) -- read it when convenient."""
        self.materialize()
        early, self._early = self._early, []      # keep it from _drop_async: step() joins these itself, behind the first Adam
        self._drop_async()                # only set here if all_reduce_grads() was skipped after an overlapped backward
        seg = self._segments()
        for gi in self._due(i, only_group):
            grp = self.param_groups[gi]
            # the group's gradient views of the flat bucket: built once (48 slice + view calls per step were 0.17 ms of host time)
            gv = self.__dict__.setdefault("_grad_views", {})
            if gi not in gv or gv[gi][0] != self.flat_grad.data_ptr():
                gv[gi] = (self.flat_grad.data_ptr(), self._views(self.flat_grad, gi))
            for p, g in zip(grp["params"], gv[gi][1]):
                if p.grad is None:
                    g.zero_()                                    # someone called zero_grad(set_to_none=True) elsewhere:
                    p.grad = g                                   # no gradient this step; the view goes back for the next
                elif p.grad.data_ptr() != g.data_ptr():
                    g.copy_(p.grad)                              # foreign gradient tensor: adopt its value
                    p.grad = g
            self._steps[gi] += 1
            o, n = seg[gi]
            norms = self.norms if (want_norms and gi == 0) else None

            def adam(lo, hi, nm=None):
                ops.adam_step(self.flat[lo:hi], self.flat_grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi],
                              grp["lr"], grp["betas"][0], grp["betas"][1], grp["eps"], self._steps[gi], self._grad_scale[gi],
                              zero_grad, len(grp["params"]), nm, group=gi)
            if early and gi == 0:
                # data-parallel overlap: the first-reduced network's range is final, the other network's collective is still in
                # flight on the side stream -- update the former now, join, update the rest.  Adam is element-wise: the same
                # numbers as one launch over the group.  (Not with want_norms -- one fixed-order sum over the group -- nor when a
                # range boundary is not 16-byte aligned: then join first and take the single launch.)
                alo, ahi, rest = early[0]
                ok = norms is None and o <= alo and ahi <= o + n and alo % 4 == 0 and ahi % 4 == 0
                if ok:
                    adam(alo, ahi)
                    self.overlap_stats["split_adam_steps"] += 1
                self._early, early = early, []
                self._join_early()
                if ok:
                    for lo, hi in ((o, alo), (ahi, o + n)):
                        if hi > lo:
                            adam(lo, hi)
                else:
                    adam(o, o + n, norms)
            else:
                adam(o, o + n, norms)
            self._grad_scale[gi] = 1.0
            for p in grp["params"]:
                torch.autograd.graph.increment_version(p)        # parameters changed behind torch's back
        if early:                          # (group 0 did not step: nothing consumed them)
            self._early = early
            self._join_early()
        return self.norms if want_norms else None

    def zero_grad(self, set_to_none=False, only_group=None):
        self._drop_async()
        if self.flat_grad is not None:
            if only_group is None:
                self.flat_grad.zero_()
            else:
                o, n = self._segments()[only_group]
                self.flat_grad[o:o + n].zero_()
        else:
            for gi, g in enumerate(self.param_groups):
                if only_group is None or only_group == gi:
                    for p in g["params"]:
                        p.grad = None

    def group_optimizer(self, gi):
        """torch-optimiser-shaped view of ONE group (a trainer's `pose_optimizer`)."""
        return _GroupView(self, gi)

    # ---- checkpoint (torch.optim.Adam format) ------------------------------------------------------------------
    def state_dict(self, group=None):
        """torch.optim.Adam.state_dict() layout.  group=None: all groups (parameter indices run on across groups, as torch
        numbers them); group=g: that group alone, numbered from 0 -- what a separate torch optimiser over it would save."""
        st, groups, base = {}, [], 0
        for gi, g in enumerate(self.param_groups):
            if group is not None and group != gi:
                continue
            q = {k: v for k, v in g.items() if k != "params"}
            q["params"] = list(range(base, base + len(g["params"])))
            groups.append(q)
            if self.flat is not None and self._steps[gi] > 0:
                for j, (a, s) in enumerate(zip(self._views(self.exp_avg, gi), self._views(self.exp_avg_sq, gi))):
                    st[base + j] = {"step": torch.tensor(float(self._steps[gi])), "exp_avg": a.clone(), "exp_avg_sq": s.clone()}
            base += len(g["params"])
        if self._pending is not None:      # a loaded state that has not reached the device buffers yet (model still on the host)
            pst = {int(k): v for k, v in self._pending["state"].items()}
            if group is None:
                st = pst
            else:                          # that group's slice, renumbered from 0 like a separate torch optimiser over it
                lo = sum(len(g["params"]) for g in self.param_groups[:group])
                n = len(self.param_groups[group]["params"])
                st = {j - lo: v for j, v in pst.items() if lo <= j < lo + n}
        return {"state": st, "param_groups": groups}

    def load_state_dict(self, sd, group=None):
        """Inverse of state_dict(group); accepts what torch.optim.Adam saved for the same parameter lists."""
        targets = list(range(len(self.param_groups))) if group is None else [group]
        if len(sd["param_groups"]) != len(targets):
            raise ValueError(f"FusedAdam.load_state_dict: {len(sd['param_groups'])} saved groups for {len(targets)} groups")
        for gi, grp in zip(targets, sd["param_groups"]):
            for k in ("lr", "betas", "eps", "step_every"):
                if k in grp:
                    self.param_groups[gi][k] = tuple(grp[k]) if k == "betas" else grp[k]
        if self.flat is None and self.params[0].is_cuda:
            self.materialize()
        if self.flat is None:                      # model still on the host: keep the state until materialize()
            if group is not None:
                raise RuntimeError("FusedAdam.load_state_dict(group=...) needs the model on the GPU first")
            self._pending = sd
            return
        state = {int(k): v for k, v in sd["state"].items()}
        base = 0
        with torch.no_grad():
            for gi in targets:
                n = len(self.param_groups[gi]["params"])
                steps = {int(state[j]["step"]) for j in range(base, base + n) if j in state}
                if len(steps) > 1:
                    raise ValueError("FusedAdam: per-parameter step counts differ inside a group; a flat segment has one step count")
                self._steps[gi] = steps.pop() if steps else 0
                for j, (a, s) in enumerate(zip(self._views(self.exp_avg, gi), self._views(self.exp_avg_sq, gi))):
                    if base + j in state:
                        a.copy_(state[base + j]["exp_avg"])
                        s.copy_(state[base + j]["exp_avg_sq"])
                base += n


class _GroupView:
    """One group of a FusedAdam behind torch.optim.Optimizer's surface (param_groups / state / step / zero_grad /
    state_dict / load_state_dict): the trainer's `pose_optimizer` when the pose parameters share the flat DP bucket."""

    def __init__(self, opt, gi):
        self.opt, self.gi = opt, gi

    @property
    def param_groups(self):
        return [self.opt.param_groups[self.gi]]

    @property
    def state(self):
        return self.opt._state_of(self.gi)

    def step(self, zero_grad=False):
        return self.opt.step(zero_grad=zero_grad, only_group=self.gi)

    def zero_grad(self, set_to_none=False):
        self.opt.zero_grad(only_group=self.gi)

    def state_dict(self):
        return self.opt.state_dict(group=self.gi)

    def load_state_dict(self, sd):
        self.opt.load_state_dict(sd, group=self.gi)


_UNIT_SEED = {}


def unit_seed(device):
    """the cached scalar 1.0 `backward()` below seeds the graph with, one per device"""
    k = str(device)
    if k not in _UNIT_SEED:
        _UNIT_SEED[k] = torch.ones((), dtype=torch.float32, device=device)
    return _UNIT_SEED[k]


def is_unit_seed(go):
    """True when an upstream gradient IS that cached 1.0 (same storage): the fused losses then hand their gradients on unscaled"""
    s = _UNIT_SEED.get(str(go.device))
    return s is not None and go.dim() == 0 and go.data_ptr() == s.data_ptr()


def backward(loss):
    """`loss.backward()` for a scalar loss built from fused_nerf_loss (+ pose_opt.kp_loss): seeds the graph with a CACHED 1.0 instead
    of a fresh ones_like (one fill launch per step), and the fused losses recognise that seed and skip their `gradient * 1.0`
    launches -- three small launches per training step (four with the pose regulariser).  Same gradients, bit for bit."""
    torch.autograd.backward(loss, grad_tensors=unit_seed(loss.device))


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb, acc, rgb0, acc0, target, bgs, loss_type, coarse_weight, beta):
        need = any(t is not None and t.requires_grad for t in (rgb, acc, rgb0, acc0))
        out, g = ops.loss(rgb, acc, target, rgb0, acc0, bgs, loss_type, coarse_weight, want_grads=need, beta=beta)
        ctx.g = g
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)      # (else autograd zero-fills a gradient for the statistics output every step: one launch)
        return out[0], out

    @staticmethod
    def backward(ctx, go, _):
        g = ctx.g
        if go is None or g is None:
            return (None,) * 9
        flat = g["flat"]
        scaled = flat if is_unit_seed(go) else flat * go    # ONE launch for the four maps (they are views of one buffer); none when
                                                            # the graph was seeded by optim.backward()
        base = flat.data_ptr()

        def sc(t):
            if t is None:
                return None
            o = (t.data_ptr() - base) // 4
            return scaled[o:o + t.numel()].view(t.shape)
        return sc(g["rgb"]), sc(g["acc"]), sc(g["rgb0"]), sc(g["acc0"]), None, None, None, None, None


_BG_CACHE = {}


def fused_nerf_loss(preds, target, bgs=1.0, loss_fn="MSE", coarse_weight=1.0, use_background=True, beta=0.1):
    """render.nerf_loss (= _compute_nerf_loss, trainer.py:353-380) as one kernel.
    Returns (loss, stats) with stats = [total, fine, coarse, fine_mse] on the device (PSNR = mse2psnr(stats[3]))."""
    kinds = {"MSE": 0, "L1": 1, "Huber": 2}          # get_loss_fn, trainer.py:146-156
    if loss_fn not in kinds:
        raise NotImplementedError(loss_fn)
    rgb = preds["rgb_map"]
    if not use_background:
        bg = None
    elif torch.is_tensor(bgs):
        bg = bgs.to(rgb.device, torch.float32)
        if bg.numel() == 1:
            bg = bg.reshape(1).expand(3).contiguous()
    else:
        key = (rgb.device, float(bgs))               # a constant background colour: one fill per (device, value), not one per step
        bg = _BG_CACHE.get(key)
        if bg is None:
            bg = _BG_CACHE[key] = torch.full((3,), float(bgs), dtype=torch.float32, device=rgb.device)
    loss, stats = _LossFn.apply(rgb, preds["acc_map"], preds.get("rgb0"), preds.get("acc0"), target, bg,
                                kinds[loss_fn], coarse_weight, beta)
    return loss, stats
