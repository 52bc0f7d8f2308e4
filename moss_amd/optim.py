"""Flat fused AdamW over the gradient bucket (C ABI ``moss_adamw_flat``, csrc/optim.hip) -- SURVEY.md section 8(f) n4.

Same update rule as ``torch.optim.AdamW`` (which MOSS uses per parameter group, scene/gaussian_model.py:215-226), applied by
ONE streaming kernel to all parameters: they are re-homed as views of one flat buffer, next to their flat gradients
(``moss_amd.dist.GradBucket``) and flat first/second moments."""
from __future__ import annotations

import ctypes as C

import torch

from ._lib import check, lib


class FlatAdamW:
    def __init__(self, param_groups, bucket, betas=(0.9, 0.999), eps=1e-15, weight_decay=0.01, capturable=False, shard=None):
        """capturable=True keeps the step counter on the device (``moss_adamw_flat_devstep``) so that a hipGraph capture of
        the training step replays with the right bias correction (the analogue of torch.optim.AdamW(capturable=True)).

        ``shard=(rank, world)``: this rank updates only its 1/world of the flat parameter buffer (the bucket's ``shard_layout``: the
        bucket must have been made with the same ``world``) and keeps moments for that shard alone (moments memory and update time
        / world); the caller reduce-scatters the gradient bucket into ``grad_shard`` before ``step()`` and all-gathers ``flat_params``
        after it -- ``moss_amd.dist.ShardedStep`` does both."""
        self.bucket = bucket
        params = bucket.params
        lr_of, pat_of = {}, {}
        for gidx, grp in enumerate(param_groups):
            for p in grp["params"]:
                lr_of[id(p)] = float(grp["lr"])
                # optional periodic pattern (period, split, lr_rest): first `split` of every `period` elements use lr, the rest lr_rest
                pat_of[id(p)] = grp.get("lr_pattern")
        total = sum(bucket.sizes)
        dev = params[0].device
        self.shard = None if shard is None else (int(shard[0]), int(shard[1]))
        if self.shard is not None and self.shard[1] != bucket.world:
            raise ValueError(f"shard=(rank, {self.shard[1]}) but the bucket was laid out for {bucket.world} shards")
        # (sharded: the parameter buffer mirrors the bucket's padded layout -- parameters, loss block, padding -- so that the
        # all-gather of the updated shards is in place)
        self.flat_params = torch.zeros(bucket.flat.numel() if self.shard is not None else total, dtype=torch.float32, device=dev)
        off = 0
        ends, lrs, periods, splits, lr2s = [], [], [], [], []
        for p, n in zip(params, bucket.sizes):
            self.flat_params[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat_params[off:off + n].view_as(p)              # the parameter now lives in the flat buffer
            off += n
            ends.append(off); lrs.append(lr_of[id(p)])
            pat = pat_of[id(p)]
            periods.append(int(pat[0]) if pat else 0); splits.append(int(pat[1]) if pat else 0); lr2s.append(float(pat[2]) if pat else 0.0)
        if len(ends) > 8:
            raise ValueError("FlatAdamW supports at most 8 learning-rate segments")
        self.n = total
        self.seg_end = (C.c_longlong * len(ends))(*ends)
        self.seg_lr = (C.c_float * len(lrs))(*lrs)
        self.seg_period = (C.c_int * len(ends))(*periods)
        self.seg_split = (C.c_int * len(ends))(*splits)
        self.seg_lr2 = (C.c_float * len(ends))(*lr2s)
        self.nseg = len(ends)
        if self.shard is None:
            self.first, self.count = 0, total
        else:
            per = bucket.shard_len
            first = min(self.shard[0] * per, total)
            self.first, self.count = first, min(first + per, total) - first          # (the loss block and the padding are no parameters)
            # the reduce-scattered gradients of this rank's shard land here (equal-sized on every rank: the collective needs that)
            self.grad_shard = torch.zeros(per, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(max(self.count, 1), dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(max(self.count, 1), dtype=torch.float32, device=dev)
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.t = 0
        # device-side step counter + completion counters, each on a cache line of its own: the LIBRARY says how large (csrc/optim.hip)
        self.step_state = torch.zeros(int(lib().moss_adamw_state_bytes()) // 4, dtype=torch.int32, device=dev) if capturable else None

    def snapshot(self):
        """Copies of everything a step changes (parameters, both moments, step counters)."""
        return (self.flat_params.clone(), self.exp_avg.clone(), self.exp_avg_sq.clone(), self.t,
                None if self.step_state is None else self.step_state.clone())

    def restore(self, snap):
        self.flat_params.copy_(snap[0]); self.exp_avg.copy_(snap[1]); self.exp_avg_sq.copy_(snap[2]); self.t = snap[3]
        if self.step_state is not None:
            self.step_state.copy_(snap[4])

    def permute_rows(self, perm: torch.Tensor):
        """Re-index the Gaussians: row i of every parameter whose leading dimension is len(perm) -- and of its two moments -- becomes
        the old row perm[i] (e.g. ``moss_amd.densify.spatial_order(xyz)``).  The step count is shared and stays.  Not capturable:
        call it between graph captures, where MOSS rebuilds its tensors anyway (densify / prune)."""
        if self.shard is not None:
            raise RuntimeError("permute_rows on a sharded FlatAdamW: the moments of a parameter row live on several ranks; gather, "
                               "permute and rebuild the optimizer instead")
        n_rows = int(perm.numel())
        perm = perm.to(self.flat_params.device)
        off = 0
        with torch.no_grad():
            for p, n in zip(self.bucket.params, self.bucket.sizes):
                if p.dim() >= 1 and p.shape[0] == n_rows:
                    for flat in (self.flat_params, self.exp_avg, self.exp_avg_sq):
                        v = flat[off:off + n].view_as(p)
                        v.copy_(v[perm].clone())
                off += n

    def step(self, skip_word=None, skip_mask=2):
        """One update.  ``skip_word`` (capturable optimizers only): a one-element int32 / float32 DEVICE tensor; if
        ``skip_word & skip_mask`` is non-zero when the kernel runs, the step is a no-op on the device -- parameters, moments and the
        step counter stay bit for bit (C ABI ``moss_adamw_flat_guarded``).  ``frame_status_word(img_buffer)`` of a rasterizer forward
        with the default mask 2 skips the step of a frame that overflowed its capacity and rendered nothing (inside a captured
        hipGraph nobody else can)."""
        dev = self.flat_params.device
        if skip_word is not None:
            if self.step_state is None:
                raise RuntimeError("a guarded step needs capturable=True: a host-side step count cannot know about the skipped step")
            first, count = (0, self.n) if self.shard is None else (self.first, self.count)
            if count == 0:
                return
            grads = self.bucket.flat if self.shard is None else self.grad_shard
            with torch.cuda.device(dev):
                rc = lib().moss_adamw_flat_guarded(first, count, self.flat_params[first:].data_ptr(), grads.data_ptr(),
                                                   self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.nseg, self.seg_end, self.seg_lr,
                                                   self.seg_period, self.seg_split, self.seg_lr2, float(self.betas[0]), float(self.betas[1]),
                                                   float(self.eps), float(self.weight_decay), self.step_state.data_ptr(),
                                                   skip_word.data_ptr(), int(skip_mask) & 0xffffffff,
                                                   torch.cuda.current_stream(dev).cuda_stream)
            check(rc, "adamw_flat_guarded")
            return
        self.t += 1
        if self.shard is not None:
            if self.count == 0:
                return
            with torch.cuda.device(dev):
                rc = lib().moss_adamw_flat_range(self.first, self.count, self.flat_params[self.first:].data_ptr(), self.grad_shard.data_ptr(),
                                                 self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.nseg, self.seg_end, self.seg_lr,
                                                 self.seg_period, self.seg_split, self.seg_lr2, float(self.betas[0]), float(self.betas[1]),
                                                 float(self.eps), float(self.weight_decay), self.t,
                                                 None if self.step_state is None else self.step_state.data_ptr(),
                                                 torch.cuda.current_stream(dev).cuda_stream)
            check(rc, "adamw_flat_range")
            return
        if self.step_state is not None:
            with torch.cuda.device(dev):
                rc = lib().moss_adamw_flat_devstep(self.n, self.flat_params.data_ptr(), self.bucket.flat.data_ptr(),
                                                   self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.nseg, self.seg_end,
                                                   self.seg_lr, self.seg_period, self.seg_split, self.seg_lr2, float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                                   float(self.weight_decay), self.step_state.data_ptr(),
                                                   torch.cuda.current_stream(dev).cuda_stream)
            check(rc, "adamw_flat_devstep")
            return
        with torch.cuda.device(dev):
            rc = lib().moss_adamw_flat(self.n, self.flat_params.data_ptr(), self.bucket.flat.data_ptr(), self.exp_avg.data_ptr(),
                                       self.exp_avg_sq.data_ptr(), self.nseg, self.seg_end, self.seg_lr,
                                       self.seg_period, self.seg_split, self.seg_lr2, float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.weight_decay),
                                       self.t, torch.cuda.current_stream(dev).cuda_stream)
        check(rc, "adamw_flat")
