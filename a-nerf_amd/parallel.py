"""Ray-batch data parallelism: one process per GPU, one RCCL all-reduce per step.

The reference wraps the caster in single-process nn.DataParallel (core/raycasters.py:157): per call it
re-broadcasts both networks, scatters every per-ray tensor, gathers the output dict and reduce-adds gradients
to device 0.  Here each rank owns N_rand / world rays of the step (rays are independent; the loss is a mean
over rays), runs the fused forward/backward locally and the gradients of all parameters are summed with ONE
all-reduce over a single flat fp32 bucket (2 x 864 260 floats = 6.9 MB; xGMI ring time << one step), then
scaled by 1/world -- identical to the gradient of the global-batch mean loss.
Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
import torch
import torch.distributed as dist


class GradBucket:
    """Flat fp32 gradient bucket over a fixed parameter list."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        self.flat = None

    def _ensure(self, device):
        if self.flat is None or self.flat.device != device:
            self.flat = torch.zeros(self.numel, dtype=torch.float32, device=device)
            self.views, o = [], 0
            for p in self.params:
                self.views.append(self.flat[o:o + p.numel()].view_as(p))
                o += p.numel()

    def all_reduce_mean(self, group=None, weight=None):
        """Sum gradients over ranks, divide by world size, write back into p.grad.  Returns the flat bucket.
        weight: factor applied to this rank's gradients first (`shard_weight()` when the ray shards are ragged)."""
        if not self.params:
            return None
        world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        if world == 1:
            return None                      # single process: gradients are already the global-batch gradients
        self._ensure(self.params[0].device)
        for p in self.params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        grads = [p.grad for p in self.params]
        torch._foreach_copy_(self.views, grads)          # one multi-tensor launch each way
        if weight is not None and float(weight) != 1.0:
            self.flat.mul_(float(weight))
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        self.flat.div_(world)
        torch._foreach_copy_(grads, self.views)
        return self.flat


def shard_rays(n_rays, rank=None, world=None):
    """Contiguous slice [lo, hi) of a ray batch owned by `rank` (last rank may be short)."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    per = (n_rays + world - 1) // world
    return min(n_rays, rank * per), min(n_rays, (rank + 1) * per)


def shard_weight(n_rays, rank=None, world=None):
    """Factor that turns "mean over ranks of per-rank MEAN-loss gradients" into the global-batch mean gradient when the
    shards are ragged: local_n * world / n_rays (1.0 for an even split).  Each rank's loss is a mean over ITS rays
    (trainer.py:21), so a short last shard would otherwise be over-weighted by the plain 1/world average."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    lo, hi = shard_rays(n_rays, rank, world)
    return (hi - lo) * world / float(n_rays)


def gather_rays(local, n_total, group=None):
    """All-gather per-ray outputs of a sharded render into [n_total, ...] (frame assembly on every rank)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    per = (n_total + world - 1) // world
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return out[:n_total]
