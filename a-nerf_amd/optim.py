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
import torch
import torch.distributed as dist

from . import ops


class FusedAdam:
    """Adam over a flat parameter buffer.  Interface subset of torch.optim.Adam: `param_groups`, `state`,
    `step()`, `zero_grad()`, `state_dict()` / `load_state_dict()` in torch's format (so the reference's
    checkpoint key `optimizer_state_dict`, trainer.py:498-505, round-trips)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FusedAdam: no parameters")
        self.param_groups = [{"params": self.params, "lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": 0,
                              "amsgrad": False}]
        self._step = 0
        self._grad_scale = 1.0
        self.flat = self.flat_grad = self.exp_avg = self.exp_avg_sq = None
        self.norms = None
        self._pending = None          # optimizer state loaded before the buffers exist

    # ---- flat storage ---------------------------------------------------------------------------------------
    def _views(self, flat):
        out, o = [], 0
        for p in self.params:
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
        n = sum(p.numel() for p in self.params)
        self.numel = n
        npad = (n + 3) // 4 * 4
        self.flat = torch.zeros(npad, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(npad, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(npad, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(npad, dtype=torch.float32, device=dev)
        self.norms = torch.zeros(2, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, v, g in zip(self.params, self._views(self.flat), self._views(self.flat_grad)):
                v.copy_(p.data)
                if p.grad is not None:
                    g.copy_(p.grad)
                p.data = v
                p.grad = g
                p._anerf_flat_grad = True        # autograd_path: gradients may be accumulated into the bucket in place
        if self._pending is not None:
            sd, self._pending = self._pending, None
            self.load_state_dict(sd)

    @property
    def state(self):
        """torch-style per-parameter state (read-only views; `state[p]['step']` is what decay_optimizer_lrate reads)."""
        if self.flat is None or self._step == 0:
            return {}
        ea, es = self._views(self.exp_avg), self._views(self.exp_avg_sq)
        return {p: {"step": self._step, "exp_avg": a, "exp_avg_sq": s} for p, a, s in zip(self.params, ea, es)}

    # ---- step -----------------------------------------------------------------------------------------------
    def all_reduce_grads(self, group=None):
        """Sum the flat gradient bucket over ranks (one collective); the 1/world scale is applied inside step()."""
        self.materialize()
        world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        if world > 1:
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=group)
            self._grad_scale = 1.0 / world

    @torch.no_grad()
    def step(self, zero_grad=False, want_norms=False):
        """One Adam update.  zero_grad: also clear the gradients (= the reference's `_optim_step`).  want_norms:
        returns a [2] device tensor (total_norm, avg_norm) of this step's gradients -- read it when convenient."""
        self.materialize()
        for p, g in zip(self.params, self._views(self.flat_grad)):
            if p.grad is None:
                g.zero_()                                    # someone called zero_grad(set_to_none=True) elsewhere:
                p.grad = g                                   # no gradient this step; the view goes back for the next
            elif p.grad.data_ptr() != g.data_ptr():
                g.copy_(p.grad)                              # foreign gradient tensor: adopt its value
                p.grad = g
        grp = self.param_groups[0]
        self._step += 1
        ops.adam_step(self.flat, self.flat_grad, self.exp_avg, self.exp_avg_sq, grp["lr"], grp["betas"][0], grp["betas"][1],
                      grp["eps"], self._step, self._grad_scale, zero_grad, len(self.params),
                      self.norms if want_norms else None)
        self._grad_scale = 1.0
        for p in self.params:
            torch.autograd.graph.increment_version(p)        # parameters changed behind torch's back
        return self.norms if want_norms else None

    def zero_grad(self, set_to_none=False):
        if self.flat_grad is not None:
            self.flat_grad.zero_()
        else:
            for p in self.params:
                p.grad = None

    # ---- checkpoint (torch.optim.Adam format) ------------------------------------------------------------------
    def state_dict(self):
        grp = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        grp["params"] = list(range(len(self.params)))
        st = {}
        if self.flat is not None and self._step > 0:
            for i, (a, s) in enumerate(zip(self._views(self.exp_avg), self._views(self.exp_avg_sq))):
                st[i] = {"step": torch.tensor(float(self._step)), "exp_avg": a.clone(), "exp_avg_sq": s.clone()}
        return {"state": st, "param_groups": [grp]}

    def load_state_dict(self, sd):
        if self.flat is None:
            self._pending = sd
        grp = sd["param_groups"][0]
        for k in ("lr", "betas", "eps"):
            if k in grp:
                self.param_groups[0][k] = tuple(grp[k]) if k == "betas" else grp[k]
        if self.flat is None:
            return
        steps = {int(v["step"]) for v in sd["state"].values()}
        if len(steps) > 1:
            raise ValueError("FusedAdam: per-parameter step counts differ; one flat buffer has one step count")
        self._step = steps.pop() if steps else 0
        with torch.no_grad():
            for i, (a, s) in enumerate(zip(self._views(self.exp_avg), self._views(self.exp_avg_sq))):
                if i in sd["state"]:
                    a.copy_(sd["state"][i]["exp_avg"])
                    s.copy_(sd["state"][i]["exp_avg_sq"])


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb, acc, rgb0, acc0, target, bgs, loss_type, coarse_weight, beta):
        need = any(t is not None and t.requires_grad for t in (rgb, acc, rgb0, acc0))
        out, g = ops.loss(rgb, acc, target, rgb0, acc0, bgs, loss_type, coarse_weight, want_grads=need, beta=beta)
        ctx.g = g
        ctx.mark_non_differentiable(out)
        return out[0], out

    @staticmethod
    def backward(ctx, go, _):
        g = ctx.g
        sc = lambda t: None if t is None else t * go
        return sc(g["rgb"]), sc(g["acc"]), sc(g["rgb0"]), sc(g["acc0"]), None, None, None, None, None


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
        bg = torch.full((3,), float(bgs), dtype=torch.float32, device=rgb.device)
    loss, stats = _LossFn.apply(rgb, preds["acc_map"], preds.get("rgb0"), preds.get("acc0"), target, bg,
                                kinds[loss_fn], coarse_weight, beta)
    return loss, stats
