"""Host-side mirror of the reference's training iteration, `Trainer.train_batch` (core/trainer.py:205-277): the caller of the
hot path on the training side.  Same constructor, same `train_batch(batch, i, global_step)` ->
`(loss_dict, stats)`, same order of work per iteration:

    pose data (batch, or the PoseOptLayer)  ->  render()  ->  loss (+ pose regulariser)  ->  backward  ->  optimiser step(s)
    ->  decay_optimizer_lrate  ->  RayCaster.update_embed_fns

Every numeric step is a kernel of libanerf_hip.so (RayCaster, optim.fused_nerf_loss, pose_opt.kp_loss, FusedAdam); this file
is sequencing only.  Two optimiser forms are accepted, as the reference's trainer gets them from run_nerf.py:518-535:
  * torch optimisers (`optimizer`, `pose_optimizer`): stepped as the reference steps them;
  * a `FusedAdam` whose group 0 holds the networks and whose group 1 (with `step_every = opt_pose_step`) holds the pose
    layer: pass `optimizer = fused.group_optimizer(0)`, `pose_optimizer = fused.group_optimizer(1)` (or the FusedAdam itself as
    `optimizer` and None); one `step(zero_grad=True, i=i)` then does both on the reference's cadence, and with
    `torch.distributed` initialised the gradient bucket is all-reduced first (one collective over whatever is due).

Differences from the reference, all in what is REPORTED, none in what is computed:
  * `stats` values are 0-dim device tensors (the reference calls `.item()` on each: eight host syncs per iteration);
    `float(v)` them when logging;
  * `total_norm` / `avg_norm` are the gradient norms of the step in both branches; the reference's pose branch calls
    get_gradnorm after `zero_grad()` (trainer.py:466-472), i.e. reports zeros under torch 1.x and divides by zero under
    torch >= 2.0.
"""
import torch

from . import optim, pose_opt
from .render import img2mse, mse2psnr, nerf_loss, render


def decay_optimizer_lrate(lrate, lrate_decay, decay_rate, optimizer, global_step=None, decay_unit=1000):
    """core/trainer.py:173-183: lr = lrate * decay_rate ^ ((Adam step count // decay_unit) / lrate_decay), written to every
    param group of `optimizer` (a torch optimiser, a FusedAdam, or one group of a FusedAdam).  The count is the optimiser's
    own, not `global_step`, exactly as in the reference."""
    groups = optimizer.param_groups
    st = optimizer.state
    first = groups[0]["params"][0]
    step = st[first]["step"] if first in st else 0
    step = float(step.item()) if torch.is_tensor(step) else float(step)
    new_lrate = lrate * (decay_rate ** ((step // decay_unit) / lrate_decay))
    for g in groups:
        g["lr"] = new_lrate
    return new_lrate, None


@torch.no_grad()
def get_gradnorm(module):
    """core/trainer.py:192-203 without its per-tensor `.item()`: (total_norm, avg_norm) as device scalars"""
    sq = [p.grad.detach().float().pow(2).sum() for p in module.parameters() if p.grad is not None]
    if not sq:
        z = torch.zeros(())
        return z, z
    tot = torch.stack(sq).sum()
    return tot.sqrt(), (tot / len(sq)).sqrt()


class Trainer:
    def __init__(self, args, data_attrs, optimizer, pose_optimizer, render_kwargs_train, render_kwargs_test, popt_kwargs=None,
                 device=None):
        self.args, self.optimizer, self.pose_optimizer = args, optimizer, pose_optimizer
        self.render_kwargs_train, self.render_kwargs_test = render_kwargs_train, render_kwargs_test
        self.popt_kwargs, self.device = popt_kwargs, device
        self.hwf, self.data_attrs = data_attrs["hwf"], data_attrs
        self._fused = self._fused_of(optimizer)
        self._anchor_cache = {}
        self._gs = self._static = None          # enable_graph()

    @staticmethod
    def _fused_of(optimizer):
        if isinstance(optimizer, optim.FusedAdam):
            return optimizer
        if isinstance(optimizer, optim._GroupView):
            return optimizer.opt
        return None

    # ---- the iteration as one hipGraph launch (opt-in) ------------------------------------------------------------------
    def enable_graph(self, on=True, eager_steps=2, capture_error_mode="thread_local", warm_each_key=True):
        """Replay the device side of train_batch -- pose layer, render, losses, backward, optimiser -- from a captured hipGraph
        (graph_step.GraphedTrainStep): one launch per iteration instead of ~45, same kernels and bit-identical results.  Needs the
        fused tail (a FusedAdam).  The loader's batch is copied into persistent device tensors before each replay; the pose
        layer's grouping of the rays is staged the same way; learning rate, tau, Adam's step count and the random offset travel
        through the device-resident step block.  Each (batch size, number of distinct poses, pose-optimisation phase) gets its
        own graph on its SECOND use (the first runs eagerly: warm caches); a variant whose capture fails runs eagerly from then on,
        with one warning; `--freq_schedule` runs stay eager (with a warning).  With more than one rank the gradient collectives
        are captured with the step.
        ALIASING: in graph mode the tensors train_batch returns (loss_dict, stats: 0-dim device tensors) are views into the graph's
        private pool and the NEXT replay overwrites them in place -- read them (`.item()`, `float()`) or `.clone()` them before the
        next train_batch call if they are to be kept, e.g. to average over a logging interval.  The eager path returns fresh
        tensors every call.  (Cloning them here would put seven more launches behind every replay.)"""
        if not on:
            self._gs = self._static = None
            return self
        if self._fused is None:
            raise ValueError("Trainer.enable_graph needs the fused tail: pass a FusedAdam (or its group_optimizer views) as the optimizers")
        from . import graph_step
        caster = self.render_kwargs_train["ray_caster"]
        self._static = graph_step.StaticBatch(self.device if self.device is not None else next(caster.parameters()).device)
        self._gs = graph_step.GraphedTrainStep(self._graph_body, caster, self._fused, eager_steps=eager_steps,
                                               capture_error_mode=capture_error_mode,      # "thread_local": other threads (pin-memory, watchdogs) keep their HIP API
                                               warm_each_key=warm_each_key)
        return self

    def _graph_body(self, i):
        """the capturable part of train_batch (no host read of device data, static shapes, inputs at static addresses)"""
        sb, kp_idx, popt_detach, no_pose = self._cur
        args = self.args
        H, W, focal = self.hwf
        kp_args, extra_args = self.get_kp_args(dict(sb, kp_idx=kp_idx), detach=popt_detach)
        preds = render(H, W, focal, chunk=args.chunk, verbose=False, retraw=False, **kp_args, **self.get_fwd_args(sb),
                       **self.render_kwargs_train)
        loss_dict, stats = self.compute_loss(sb, preds, kp_opts={**kp_args, **extra_args}, popt_detach=no_pose)
        optim_stats = self.optimize(loss_dict["total_loss"], i, no_pose)
        return {"loss_dict": loss_dict, "stats": stats, "optim": optim_stats, "alpha": preds["acc_map"].detach().mean()}

    def _train_batch_graphed(self, batch, i, global_step):
        args, f = self.args, self._fused
        layer = None if self.popt_kwargs is None else self.popt_kwargs.get("popt_layer")
        kp_idx = batch.get("kp_idx") if layer is not None else None      # only the pose layer reads it (on the host)
        if torch.is_tensor(kp_idx):
            kp_idx = kp_idx.detach().cpu().numpy()
        sb = self._static.load({k: v for k, v in batch.items() if k != "kp_idx"})
        popt_detach = not (args.opt_pose_stop is None or i < args.opt_pose_stop)
        no_pose = popt_detach or not args.opt_pose
        n_rays = int(batch["target_s"].shape[0])
        # the key holds everything a captured graph has frozen: sizes, phase flags, the pose layout, and the ADDRESSES of the
        # static inputs (an entry that changes shape / dtype or flips tensor <-> None gets another graph, not a stale buffer)
        key = (n_rays, popt_detach, no_pose) + (layer.stage_batch(kp_idx) if layer is not None else ()) + (self._static.identity(sb),)
        only = 0 if (no_pose and len(f.param_groups) > 1) else None
        self._cur = (sb, kp_idx, popt_detach, no_pose)
        out = self._gs.step(i, key=key, due=f._due(i, only))
        self._cur = None
        return self._finish_iteration(out["loss_dict"], out["stats"], out["optim"], out["alpha"], global_step)

    def _finish_iteration(self, loss_dict, stats, optim_stats, alpha, global_step):
        """the host-side tail of an iteration (trainer.py:262-277): learning-rate decay, tau / alpha schedules, the stats dict"""
        args = self.args
        net_opt = self._fused.group_optimizer(0) if self._fused is not None else self.optimizer
        new_lrate, _ = decay_optimizer_lrate(args.lrate, args.lrate_decay, decay_rate=args.lrate_decay_rate, optimizer=net_opt,
                                             global_step=global_step, decay_unit=args.decay_unit)
        caster = self.render_kwargs_train["ray_caster"]
        caster = getattr(caster, "module", caster)
        if not args.finetune:
            caster.update_embed_fns(global_step, args)
        stats = {"lrate": new_lrate, "alpha": alpha, "cutoff": caster.embed_fn.get_tau(), **stats, **optim_stats}
        return loss_dict, stats

    # ---- step 1: pose data of the batch (trainer.py:290-317) ---------------------------------------------------------
    def get_kp_args(self, batch, detach=False):
        layer = None if self.popt_kwargs is None else self.popt_kwargs.get("popt_layer")
        if layer is None:
            return dict(kp_batch=batch["kp3d"], skts=batch["skts"], bones=batch["bones"], cyls=batch["cyls"]), {}
        kp_idx = batch["kp_idx"]
        if torch.is_tensor(kp_idx):
            # the layer groups rays by pose on the host (np.unique) and caches that grouping by CONTENT; the host copy itself is
            # taken afresh every iteration, as the reference does (trainer.py:299 `kp_idx.cpu().numpy()`): a cache keyed on the
            # tensor's address would hand back a previous batch's indices when the allocator reuses the block.  train_batch
            # leaves a loader's host-side kp_idx on the host, so this is normally no transfer at all.
            kp_idx = kp_idx.detach().cpu().numpy()
        kps, bones, skts, _, rots = layer(kp_idx)
        kp_args, extras = dict(kp_batch=kps, skts=skts, bones=bones, cyls=batch["cyls"]), {"rots": rots}
        if detach:
            kp_args = {k: (v.detach() if v is not None else None) for k, v in kp_args.items()}
            extras = {k: v.detach() for k, v in extras.items()}
        return kp_args, extras

    def get_fwd_args(self, batch):
        return {"rays": batch["rays"], "cams": batch["cam_idxs"] if self.args.opt_framecode else None,
                "subject_idxs": batch.get("subject_idxs")}

    # ---- step 3: losses (trainer.py:325-403) -------------------------------------------------------------------------
    def compute_loss(self, batch, preds, kp_opts=None, popt_detach=False):
        args = self.args
        if getattr(args, "reg_fn", None) is not None:
            raise NotImplementedError("reg_fn (acc_map regulariser) is used by no shipped config")
        bgs = batch["bgs"] if "bgs" in batch else 1.0
        kw = dict(bgs=bgs, loss_fn=args.loss_fn, coarse_weight=args.coarse_weight, use_background=args.use_background,
                  beta=args.loss_beta)
        loss_dict, stats = {}, {}
        if preds["rgb_map"].is_cuda:
            total, st = optim.fused_nerf_loss(preds, batch["target_s"], **kw)
            loss_dict["rgb_loss"] = st[1]
            stats["psnr"] = mse2psnr(st[3])
            if "rgb0" in preds:
                loss_dict["rgb_loss0"] = st[2]
        else:
            total, _ = nerf_loss(preds, batch["target_s"], **kw)
        if "rgb0" in preds:
            with torch.no_grad():        # PSNR of the coarse head: a statistic only (trainer.py:370)
                comp = preds["rgb0"] + ((1. - preds["acc0"])[..., None] * bgs if args.use_background else 0.)
                stats["psnr0"] = mse2psnr(img2mse(comp, batch["target_s"]))
        if not popt_detach:
            # (on the device the regulariser's launch also forms `total + kp_loss`: no separate add kernel)
            kp_l, kp_stats = self._compute_kp_loss(batch, kp_opts, add_to=total if total.is_cuda else None)
            total = kp_l.pop("_total", None) if "_total" in kp_l else total + kp_l["kp_loss"]
            loss_dict.update(kp_l)
            stats.update(kp_stats)
        loss_dict["total_loss"] = total
        return loss_dict, stats

    def _compute_kp_loss(self, batch, kp_opts, add_to=None):
        """trainer.py:382-403 over the batch's DISTINCT poses (pose_opt.kp_loss: one launch each way); equals the reference's
        mean over the per-ray replicated batch.  `use_temp_loss` is used by no shipped config."""
        args = self.args
        if getattr(args, "use_temp_loss", False):
            raise NotImplementedError("use_temp_loss is used by no shipped config")
        layer, anchors = self.popt_kwargs["popt_layer"], self.popt_kwargs["popt_anchors"]
        lu = layer.last_unique
        dev = lu["rots"].device
        if "idx_dev" in lu:
            # staged batch (PoseOptLayer.stage_batch, the captured step): the distinct pose rows and their ray shares are device
            # tensors at static addresses -- gather the anchors with them on the device (the same rows, the same values)
            full = self.__dict__.get("_anchors_dev")
            if full is None or full[0] != (id(anchors), str(dev), bool(args.opt_rot6d)):
                a = anchors["rots"].to(dev)[..., :3, :2].flatten(start_dim=-2) if args.opt_rot6d else anchors["bones"].to(dev)
                full = self._anchors_dev = ((id(anchors), str(dev), bool(args.opt_rot6d)), a.contiguous(), anchors["kps"].to(dev).contiguous())
            anc, w, anc_kps = full[1].index_select(0, lu["idx_dev"]), lu["w_dev"], full[2].index_select(0, lu["idx_dev"])
            values = lu["rots"] if args.opt_rot6d else lu["bones"]
            loss = pose_opt.kp_loss(values, anc, w, bool(args.opt_rot6d), args.opt_pose_tol, args.opt_pose_coef, add_to=add_to)
            with torch.no_grad():
                pj = (anc_kps - lu["kp"].detach()).pow(2.).sum(-1).pow(0.5)
                mpjpc = (pj.mean(-1) * w).sum() / args.ext_scale
            return ({"kp_loss": loss} if add_to is None else {"kp_loss": loss[0], "_total": loss[1]}), {"MPJPC": mpjpc}
        key = (lu["idxs"].tobytes(), lu["counts"].tobytes(), bool(args.opt_rot6d))
        hit = self._anchor_cache.get(key)
        if hit is None:
            if len(self._anchor_cache) > 64:
                self._anchor_cache.clear()
            sel = torch.as_tensor(lu["idxs"], device=dev)
            if args.opt_rot6d:
                anc = anchors["rots"].to(dev)[sel][..., :3, :2].flatten(start_dim=-2).contiguous()
            else:
                anc = anchors["bones"].to(dev)[sel].contiguous()
            w = torch.tensor(lu["counts"] / float(lu["counts"].sum()), dtype=torch.float32, device=dev)
            hit = self._anchor_cache[key] = (anc, w, anchors["kps"].to(dev)[sel].contiguous())
        anc, w, anc_kps = hit
        values = lu["rots"] if args.opt_rot6d else lu["bones"]
        loss = pose_opt.kp_loss(values, anc, w, bool(args.opt_rot6d), args.opt_pose_tol, args.opt_pose_coef, add_to=add_to)
        with torch.no_grad():            # mean per-joint position change, in mm (trainer.py:400-401)
            pj = (anc_kps - lu["kp"].detach()).pow(2.).sum(-1).pow(0.5)
            mpjpc = (pj.mean(-1) * w).sum() / args.ext_scale
        return ({"kp_loss": loss} if add_to is None else {"kp_loss": loss[0], "_total": loss[1]}), {"MPJPC": mpjpc}

    # ---- step 3b: backward + optimiser steps (trainer.py:441-483) ------------------------------------------------------
    def optimize(self, loss, i, popt_detach=False):
        args = self.args
        caster = self.render_kwargs_train["ray_caster"]
        if loss.is_cuda:
            optim.backward(loss)          # = loss.backward() seeded with a cached 1.0 (no fill, no `grad * 1` launches)
        else:
            loss.backward()
        if self._fused is not None:
            f = self._fused
            # after opt_pose_stop (or without opt_pose) the reference never touches pose_optimizer again (trainer.py:476: the
            # `not popt_detach` guard): only the network group is reduced and stepped -- Adam with zero gradients would still
            # move bones / pelvis along their remaining first moments
            only = 0 if (popt_detach and len(f.param_groups) > 1) else None
            f.all_reduce_grads(i=i, only_group=only)      # no-op in a single process; ONE collective over what is due otherwise
            pose_due = only is None and len(f.param_groups) > 1 and 1 in f._due(i)
            norms = f.step(zero_grad=True, want_norms=True, i=i, only_group=only)
            if pose_due and getattr(args, "opt_pose_cache", False) and self.popt_kwargs is not None:
                self.popt_kwargs["popt_layer"].update_cache()             # trainer.py:479-480
            return {"total_norm": norms[0], "avg_norm": norms[1]}
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            # one process per GPU: average the ranks' gradients with ONE all-reduce of a flat bucket (RayParallel.sync_gradients;
            # the pose layer's parameters ride in a second bucket) -- what nn.DataParallel's reduce-add did inside one process
            if hasattr(caster, "sync_gradients"):
                caster.sync_gradients()
            if not popt_detach and self.popt_kwargs is not None and self.popt_kwargs.get("popt_layer") is not None:
                if getattr(self, "_pose_bucket", None) is None:
                    from .parallel import GradBucket
                    self._pose_bucket = GradBucket(list(self.popt_kwargs["popt_layer"].parameters()))
                self._pose_bucket.all_reduce_mean()
        total_norm, avg_norm = get_gradnorm(caster)
        self.optimizer.step()
        self.optimizer.zero_grad()
        if not popt_detach and self.pose_optimizer is not None and i % args.opt_pose_step == 0:
            self.pose_optimizer.step()
            self.pose_optimizer.zero_grad()
            if getattr(args, "opt_pose_cache", False):
                self.popt_kwargs["popt_layer"].update_cache()
        return {"total_norm": total_norm, "avg_norm": avg_norm}

    # ---- one iteration (trainer.py:228-277) --------------------------------------------------------------------------
    def train_batch(self, batch, i=0, global_step=0):
        """one iteration (trainer.py:228-277) -> (loss_dict, stats).  After enable_graph() the returned tensors alias the graph's
        pool and hold the LATEST replay's values: see enable_graph."""
        args = self.args
        H, W, focal = self.hwf
        if self._gs is not None and self.device is not None and torch.device(self.device).type == "cuda":
            layer = None if self.popt_kwargs is None else self.popt_kwargs.get("popt_layer")
            unstaged = layer is not None and (layer.use_cache or layer.kp_map is not None or len(layer.rest_pose) != 1 or
                                              not getattr(layer, "fused_batch", True))
            # more than one rank: the gradient collectives are captured with the step (RCCL all-reduces on the optimiser's side
            # stream; tests/test_graph_step.py::test_rccl_collectives_inside_the_captured_step).  A transport that cannot be
            # captured (gloo moves the bucket through the host) is not attempted: GraphedTrainStep runs eagerly and says why.  Ranks
            # may mix replayed and eager iterations -- both enqueue the same collectives in the same order.
            why = ("render_kwargs_train['pytest'] uploads host random numbers every call" if self.render_kwargs_train.get("pytest", False) else
                   "this pose layer (cache / multi-view / per-pose rest poses) uploads its index tensors every call" if unstaged else None)
            if why is None:
                return self._train_batch_graphed(batch, i, global_step)
            if self.__dict__.get("_warned_eager") != why:
                self._warned_eager = why
                import warnings
                warnings.warn(f"Trainer.enable_graph: running eagerly ({why})")
        # kp_idx is consumed on the host (the pose layer groups rays by pose there): a loader's host tensor stays where it is
        batch = {k: (v.to(self.device) if torch.is_tensor(v) and self.device is not None and k != "kp_idx" else v)
                 for k, v in batch.items()}
        popt_detach = not (args.opt_pose_stop is None or i < args.opt_pose_stop)
        kp_args, extra_args = self.get_kp_args(batch, detach=popt_detach)
        preds = render(H, W, focal, chunk=args.chunk, verbose=i < 10, retraw=False, **kp_args, **self.get_fwd_args(batch),
                       **self.render_kwargs_train)
        no_pose = popt_detach or not args.opt_pose
        loss_dict, stats = self.compute_loss(batch, preds, kp_opts={**kp_args, **extra_args}, popt_detach=no_pose)
        optim_stats = self.optimize(loss_dict["total_loss"], i, no_pose)
        return self._finish_iteration(loss_dict, stats, optim_stats, preds["acc_map"].detach().mean(), global_step)

    # ---- checkpoints (trainer.py:485-517) ------------------------------------------------------------------------------
    def save_nerf(self, path, global_step):
        from . import checkpoint
        caster = self.render_kwargs_train["ray_caster"]
        layer = None if self.popt_kwargs is None else self.popt_kwargs["popt_layer"]
        checkpoint.save_nerf(path, global_step, getattr(caster, "module", caster), self.optimizer, popt_layer=layer,
                             pose_optimizer=self.pose_optimizer,
                             popt_anchors=None if self.popt_kwargs is None else self.popt_kwargs["popt_anchors"])

    def save_popt(self, path, global_step):
        from . import checkpoint
        checkpoint.save_popt(path, global_step, self.popt_kwargs["popt_layer"], self.popt_kwargs["popt_anchors"])
