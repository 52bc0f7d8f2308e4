"""Flat fused AdamW over the gradient bucket (C ABI ``moss_adamw_flat``, csrc/optim.hip) -- SURVEY.md section 8(f) n4.

Same update rule as ``torch.optim.AdamW`` (which MOSS uses per parameter group, scene/gaussian_model.py:215-226), applied by
ONE streaming kernel to all parameters: they are re-homed as views of one flat buffer, next to their flat gradients
(``moss_amd.dist.GradBucket``) and flat first/second moments."""
from __future__ import annotations

import ctypes as C

import torch

from ._lib import check, lib


class _FusedAdamW:
    """What ``FlatAdamW.fuse_into_backward`` leaves on a RasterContext: the host struct of the C ABI (kept alive here), the data
    pointers of the parameters it describes (the backward refuses other inputs: the update is applied IN PLACE to what the op is given)."""

    def __init__(self, struct, param_ptrs, owner):
        self.struct, self.param_ptrs, self.owner = struct, dict(param_ptrs), owner
        self.address = C.addressof(struct)

    def check_inputs(self, **tensors):
        for name, ptr in self.param_ptrs.items():
            t = tensors.get(name)
            if t is None or t.data_ptr() != ptr or not t.is_contiguous() or t.dtype != torch.float32:
                raise RuntimeError(f"fused AdamW update: the rasterizer's `{name}` input is not the parameter tensor the optimizer was "
                                   f"given (a copy, a cast or an activated value?) -- use pipe.raw_parameters_in_op, or take the step "
                                   f"out of the backward again with optimizer.unfuse() (clears the context AND re-arms optimizer.step())")


class FlatAdamW:
    def __init__(self, param_groups, bucket, betas=(0.9, 0.999), eps=1e-15, weight_decay=0.01, capturable=False, shard=None):
        """capturable=True keeps the step counter on the device (``moss_adamw_flat_devstep``) so that a hipGraph capture of
        the training step replays with the right bias correction (the analogue of torch.optim.AdamW(capturable=True)).

        ``shard=(rank, world)``: this rank updates only its 1/world of the flat parameter buffer (the bucket's ``shard_layout``: the
        bucket must have been made with the same ``world``) and keeps moments for that shard alone (moments memory and update time
        / world); the caller reduce-scatters the gradient bucket into ``grad_shard`` before ``step()`` and all-gathers ``flat_params``
        after it -- ``moss_amd.dist.ShardedStep`` does both."""
        self.bucket = bucket
        self._lr_of, self._pat_of = {}, {}
        for gidx, grp in enumerate(param_groups):
            for p in grp["params"]:
                self._lr_of[id(p)] = float(grp["lr"])
                # optional periodic pattern (period, split, lr_rest): first `split` of every `period` elements use lr, the rest lr_rest
                self._pat_of[id(p)] = grp.get("lr_pattern")
        if len(bucket.params) > 8:
            raise ValueError("FlatAdamW supports at most 8 learning-rate segments")
        self.shard = None if shard is None else (int(shard[0]), int(shard[1]))
        if self.shard is not None and self.shard[1] != bucket.world:
            raise ValueError(f"shard=(rank, {self.shard[1]}) but the bucket was laid out for {bucket.world} shards")
        self._adopt_layout([p.data for p in bucket.params], None, None)
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.t = 0
        self.fused = None                                    # set by fuse_into_backward
        # DEGREE-AWARE SH update (set_active_sh_degree): the SH tensor is the one whose learning rates follow MOSS's (48, 3) pattern --
        # features_dc / features_rest in one (P,16,3) parameter; 3 = every coefficient is active (nothing is skipped)
        self.sh_index = next((i for i, p in enumerate(bucket.params) if self._pat_of[id(p)] and tuple(self._pat_of[id(p)][:2]) == (48, 3)
                              and p.dim() == 3 and tuple(p.shape[1:]) == (16, 3)), None)
        self.sh_active_degree, self.sh_inactive_zero = 3, False
        # device-side step counter + completion counters, each on a cache line of its own: the LIBRARY says how large (csrc/optim.hip)
        dev = bucket.params[0].device
        self.step_state = torch.zeros(int(lib().moss_adamw_state_bytes()) // 4, dtype=torch.int32, device=dev) if capturable else None

    def _adopt_layout(self, values, exp_avg, exp_avg_sq):
        """(Re)build the flat buffers for the bucket's CURRENT layout: ``values[i]`` becomes parameter i (re-homed as a view of
        ``flat_params``), ``exp_avg[i]`` / ``exp_avg_sq[i]`` its moments (None: zeros -- a fresh optimizer).  The constructor and the
        row surgery (``prune_rows`` / ``append_rows``) end here; learning rates that ``set_learning_rates`` changed are kept."""
        bucket = self.bucket
        params = bucket.params
        total = bucket.n_params                              # (every tensor starts 16-byte aligned: GradBucket.offsets)
        dev = params[0].device
        # (sharded: the parameter buffer mirrors the bucket's padded layout -- parameters, loss block, padding -- so that the
        # all-gather of the updated shards is in place)
        self.flat_params = torch.zeros(bucket.flat.numel() if self.shard is not None else total, dtype=torch.float32, device=dev)
        old_lr = {i: (float(self.seg_lr[i]), float(self.seg_lr2[i])) for i in range(getattr(self, "nseg", 0))}
        ends, lrs, periods, splits, lr2s = [], [], [], [], []
        for i, (p, n, off) in enumerate(zip(params, bucket.sizes, bucket.offsets)):
            self.flat_params[off:off + n].copy_(values[i].reshape(-1))
            p.data = self.flat_params[off:off + n].view(values[i].shape)      # the parameter now lives in the flat buffer
            # a segment runs to the (aligned) start of the next tensor: the <= 3 floats of padding behind a tensor are zeros with zero
            # gradients, which the update leaves zero
            ends.append(bucket.offsets[i + 1] if i + 1 < len(params) else total); lrs.append(self._lr_of[id(p)])
            pat = self._pat_of[id(p)]
            periods.append(int(pat[0]) if pat else 0); splits.append(int(pat[1]) if pat else 0); lr2s.append(float(pat[2]) if pat else 0.0)
        for i, (lr, lr2) in old_lr.items():                  # (a schedule's current rates survive a re-layout)
            lrs[i] = lr
            if periods[i]:
                lr2s[i] = lr2
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
        if exp_avg is not None:                              # (row surgery: unsharded by construction)
            for m, v, n, off in zip(exp_avg, exp_avg_sq, bucket.sizes, bucket.offsets):
                self.exp_avg[off:off + n].copy_(m.reshape(-1)); self.exp_avg_sq[off:off + n].copy_(v.reshape(-1))

    # ---- the update applied by the rasterizer's backward kernel itself -----------------------------------------------------------
    def fuse_into_backward(self, context, means3D=None, sh=None, opacity=None, scales=None, rotations=None, local_only=False):
        """Hand the update of the named parameters to the per-Gaussian backward kernel of the rasterizer (C ABI
        ``moss_raster_backward_raw_adamw``): the kernel that produces a Gaussian's gradients applies its AdamW step on the spot --
        parameters in place, moments in this optimizer's buffers, same bits as ``step()`` would give -- and the gradients of those
        tensors never leave it (autograd sees ``None`` for them).  Valid when the rasterizer is the ONLY source of their gradients
        (MOSS: features, opacity, scaling, rotation -- every loss term of train_ZJU.py:111-131 goes through the image; the position only
        where it does not also feed the LBS network) and the op takes the RAW parameters (``pipe.raw_parameters_in_op``).

        ONE backward per forward: the parameters are updated IN PLACE by that backward (a second ``backward()`` over a retained
        graph would differentiate at the old parameters and step again), and nothing else may add to these tensors' gradients.

        Every parameter of the bucket must be named (give the rest to a second optimizer over its own bucket); needs
        ``capturable=True`` and no shard.  Afterwards ``step()`` is a no-op: the step is taken inside ``loss.backward()``, and a frame
        that overflowed its binning capacity takes none (the kernel reads the frame's status word itself).
        ``context``: the :class:`RasterContext` of the rasterizer whose backward does it (None: the default one).

        In a process group of more than one rank the fused step is LOCAL: the gradients never reach the bucket, so no exchange can
        average them and data-parallel replicas would drift apart without a word.  It is therefore refused there unless the caller
        says ``local_only=True`` -- every rank trains a model of its own (the loss-only exchange of BASELINE configs[3]).
        ``unfuse()`` takes the step out of the backward again."""
        import torch.distributed as tdist
        from ._lib import FusedAdamWStruct, OPT_BITS
        from .diff_gaussian_rasterization import _C
        if self.shard is not None or self.step_state is None:
            raise RuntimeError("fuse_into_backward needs capturable=True and an unsharded optimizer")
        if not local_only and tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1:
            raise RuntimeError("fuse_into_backward in a process group of %d ranks: the step inside the backward kernel is local to the rank "
                               "(its gradients never reach the bucket, so nothing averages them and replicas diverge).  Pass "
                               "local_only=True if every rank trains its OWN model (loss-only exchange); otherwise keep the flat "
                               "step() behind the gradient exchange" % tdist.get_world_size())
        named = {"means3D": means3D, "sh": sh, "opacity": opacity, "scales": scales, "rotations": rotations}
        given = {k: v for k, v in named.items() if v is not None}
        if {id(v) for v in given.values()} != {id(p) for p in self.bucket.params} or len(given) != len(self.bucket.params):
            raise ValueError("fuse_into_backward: name every parameter of the bucket exactly once (other parameters belong to a "
                             "second optimizer over their own bucket)")
        st = FusedAdamWStruct()
        for i in range(5):
            st.lr_segment[i] = -1
        offs = {id(p): (off, i) for i, (p, off) in enumerate(zip(self.bucket.params, self.bucket.offsets))}
        ptrs = {}
        for name, p in given.items():
            slot = OPT_BITS[name].bit_length() - 1
            off, seg = offs[id(p)]
            st.tensors |= OPT_BITS[name]
            st.exp_avg[slot] = self.exp_avg.data_ptr() + 4 * off
            st.exp_avg_sq[slot] = self.exp_avg_sq.data_ptr() + 4 * off
            st.lr[slot] = self.seg_lr[seg]
            st.lr_segment[slot] = seg
            if name == "sh":
                period, split = int(self.seg_period[seg]), int(self.seg_split[seg])
                if period == 0:
                    st.lr_sh_rest = self.seg_lr[seg]
                elif (period, split) == (48, 3):
                    st.lr_sh_rest = self.seg_lr2[seg]
                else:
                    raise ValueError("fuse_into_backward: the SH learning-rate pattern must be (48, 3, lr_rest) -- features_dc / features_rest")
            elif int(self.seg_period[seg]) != 0:
                raise ValueError(f"fuse_into_backward: a learning-rate pattern on {name} is not supported")
            ptrs[name] = p.data_ptr()
        st.beta1, st.beta2, st.eps, st.weight_decay = float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.weight_decay)
        st.step_state = self.step_state.data_ptr()
        st.sh_active_degree, st.sh_inactive_zero = int(self.sh_active_degree), int(self.sh_inactive_zero)
        self.fused = _FusedAdamW(st, ptrs, self)
        self._fused_context = context or _C.DEFAULT
        self._fused_context.fused_adamw = self.fused
        self._fused_args = (dict(given), bool(local_only))   # (row surgery re-fuses with the same names: the moments move)
        return self.fused

    def unfuse(self, context=None):
        """Take the step out of the backward kernel again: clears the context's ``fused_adamw`` AND this optimizer's own flag, so that
        the backward writes gradients to the bucket as before and ``step()`` applies them.  (Clearing only the context -- what round 4's
        error message suggested -- left ``step()`` a no-op: parameters silently stopped updating.)  ``context``: the one given to
        ``fuse_into_backward`` (default: that one).  A step captured in a hipGraph must be re-captured afterwards."""
        cx = context or getattr(self, "_fused_context", None)
        if cx is not None and cx.fused_adamw is self.fused:
            cx.fused_adamw = None
        self.fused = None
        self._fused_context = None
        self._fused_args = None

    # ---- degree-aware SH update --------------------------------------------------------------------------------------------------
    def set_active_sh_degree(self, degree: int):
        """Tell the optimizer the ACTIVE SH degree of its (P,16,3) SH parameter (MOSS: ``active_sh_degree`` starts at 0 and goes up every
        1000 iterations, train_ZJU.py:85-86, scene/gaussian_model.py:171-173 -- 2999 of 3000 iterations run below degree 3).  The
        optimizer keeps the HIGHEST degree it has been told since the coefficients above it were last seen to have zero moments: those
        coefficients have never received a gradient, so their AdamW step is the weight decay alone and their moments stay exactly zero --
        the update kernels (the flat one and the rasterizer backward that takes the step itself) then neither read nor write those
        moments, and when the parameters there are exactly zero as well (MOSS initialises ``features_rest`` with zeros,
        scene/gaussian_model.py:179-181: checked HERE, one device reduction and a host read per call) they are not touched at all.
        Results are bit-identical to the full update.  The degree is a launch argument: a captured step must be captured again
        (the rasterizer's own ``sh_degree`` is one too).  Returns the degree in force."""
        if self.sh_index is None:
            return 3
        degree = max(0, min(3, int(degree)))
        k_prev = (self.sh_active_degree + 1) ** 2
        m, v = self._moments_of(self.sh_index) if self.shard is None else (None, None)
        if self.shard is not None:
            degree = 3                                       # (a shard sees a slice of the tensor: everything stays active)
        elif degree < self.sh_active_degree:
            # lower than what is in force: only if the moments above it are (still) exactly zero -- a fresh optimizer, or one that
            # never trained above that degree
            k = (degree + 1) ** 2
            if bool(m[:, k:k_prev, :].any().item()) or bool(v[:, k:k_prev, :].any().item()):
                degree = self.sh_active_degree
        self.sh_active_degree = degree
        self._verify_sh_inactive()
        return self.sh_active_degree

    def _verify_sh_inactive(self):
        """Re-establish ``sh_inactive_zero`` (and that the moments above the active degree are zero: otherwise everything is active)."""
        if self.sh_index is None or self.sh_active_degree >= 3 or self.shard is not None:
            self.sh_active_degree = 3 if self.sh_index is None or self.shard is not None else self.sh_active_degree
            self.sh_inactive_zero = False
        else:
            k = (self.sh_active_degree + 1) ** 2
            p = self.bucket.params[self.sh_index]
            m, v = self._moments_of(self.sh_index)
            if bool(m[:, k:, :].any().item()) or bool(v[:, k:, :].any().item()):
                self.sh_active_degree, self.sh_inactive_zero = 3, False
            else:
                self.sh_inactive_zero = not bool(p.data[:, k:, :].any().item())
        if self.fused is not None:
            self.fused.struct.sh_active_degree = int(self.sh_active_degree)
            self.fused.struct.sh_inactive_zero = int(self.sh_inactive_zero)

    def _seg_active(self):
        act = (C.c_int * self.nseg)(*([0] * self.nseg))
        if self.sh_index is not None and self.sh_active_degree < 3:
            act[self.sh_index] = 3 * (self.sh_active_degree + 1) ** 2
        return act

    def set_learning_rates(self, rates):
        """A learning-rate schedule.  ``rates``: {parameter (or its index in the bucket): lr, or (lr, lr_rest) for a tensor with a
        periodic pattern}.  MOSS decays the position rate every iteration (``GaussianModel.update_learning_rate``,
        scene/gaussian_model.py:263-268, train_ZJU.py:82).

        With ``capturable=True`` the rates also go into the optimizer's device-side state block (one small asynchronous copy on the
        current stream) and every update kernel -- the flat ones and the rasterizer backward that takes the step itself -- reads them
        from there: call this BETWEEN replays of a captured step; no re-capture (the launch arguments baked into the graph are then
        ignored)."""
        index = {id(p): i for i, p in enumerate(self.bucket.params)}
        for key, val in rates.items():
            i = key if isinstance(key, int) else index[id(key)]
            lr, lr2 = (val if isinstance(val, (tuple, list)) else (val, None))
            self.seg_lr[i] = float(lr)
            if lr2 is not None:
                self.seg_lr2[i] = float(lr2)
        if self.fused is not None:
            st = self.fused.struct
            for slot in range(5):
                seg = st.lr_segment[slot]
                if seg >= 0:
                    st.lr[slot] = self.seg_lr[seg]
                    if slot == 1:
                        st.lr_sh_rest = self.seg_lr2[seg] if int(self.seg_period[seg]) else self.seg_lr[seg]
        if self.step_state is not None:
            # a small ring of pinned staging buffers: the copy is asynchronous and the device may be several steps behind the host, so a
            # buffer is rewritten only after the copy that last read it has completed (its event)
            if getattr(self, "_lr_ring", None) is None:
                self._lr_ring = [[torch.zeros(20, dtype=torch.float32).pin_memory(), None] for _ in range(8)]
                self._lr_next = 0
            slot = self._lr_ring[self._lr_next]
            self._lr_next = (self._lr_next + 1) % len(self._lr_ring)
            if slot[1] is not None:
                slot[1].synchronize()
            h = slot[0]
            hi = h.view(torch.int32)
            hi[0] = 1                                        # word 12: the table is valid; words 13-15 unused
            for s_ in range(8):
                h[4 + s_] = self.seg_lr[s_] if s_ < self.nseg else 0.0
                h[12 + s_] = (self.seg_lr2[s_] if int(self.seg_period[s_]) else self.seg_lr[s_]) if s_ < self.nseg else 0.0
            self.step_state.view(torch.float32)[12:32].copy_(h, non_blocking=True)
            slot[1] = torch.cuda.Event()
            slot[1].record(torch.cuda.current_stream(self.step_state.device))

    def step_count(self) -> int:
        """Steps taken so far (reads the device-side counter when there is one: it synchronises)."""
        return int(self.step_state[0].item()) if self.step_state is not None else self.t

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
        with torch.no_grad():
            for p, n, off in zip(self.bucket.params, self.bucket.sizes, self.bucket.offsets):
                if p.dim() >= 1 and p.shape[0] == n_rows:
                    for flat in (self.flat_params, self.exp_avg, self.exp_avg_sq):
                        v = flat[off:off + n].view_as(p)
                        v.copy_(v[perm].clone())

    # ---- row surgery: MOSS's densification rebuilds its tensors AND its optimizer state (scene/gaussian_model.py:362-454) ----------
    def _row_params(self, n_rows):
        return [p.dim() >= 1 and p.shape[0] == n_rows for p in self.bucket.params]

    def _relayout(self, values, exp_avg, exp_avg_sq):
        """New parameter values / moments (one tensor per bucket parameter, the Parameter OBJECTS stay) -> new bucket layout, new flat
        buffers, the fused step re-armed on the new moment addresses.  Everything that holds device addresses of the old buffers -- a
        captured hipGraph above all -- is stale afterwards: ``GraphedStep.recapture()``."""
        if self.shard is not None:
            raise RuntimeError("row surgery on a SHARDED FlatAdamW: a parameter row's moments live on several ranks; use the all-reduce "
                               "exchange while the set densifies (or gather, rebuild and re-shard)")
        fused = getattr(self, "_fused_args", None) if self.fused is not None else None
        cx = getattr(self, "_fused_context", None)
        for p, v in zip(self.bucket.params, values):
            p.data = v                                       # (the new shapes: what the bucket lays itself out from)
        self.bucket.relayout()
        self._adopt_layout(values, exp_avg, exp_avg_sq)
        if self.sh_active_degree < 3:
            self._verify_sh_inactive()                       # (appended rows may carry non-zero coefficients above the active degree)
        if fused is not None:
            self.fuse_into_backward(cx, local_only=fused[1], **fused[0])

    def _moments_of(self, i):
        off, n = self.bucket.offsets[i], self.bucket.sizes[i]
        shape = self.bucket.params[i].shape
        return self.exp_avg[off:off + n].view(shape), self.exp_avg_sq[off:off + n].view(shape)

    @torch.no_grad()
    def prune_rows(self, keep_mask: torch.Tensor):
        """``_prune_optimizer(mask)`` (scene/gaussian_model.py:377-394): every parameter whose leading dimension is len(keep_mask)
        keeps the rows where the mask is True, and so do its two moments; the shared step count stays (torch keeps ``state['step']``
        too).  The Parameter objects are the same afterwards (``.data`` re-homed in the new flat buffer, ``.grad`` dropped)."""
        keep = keep_mask.to(self.flat_params.device).bool()
        rows = self._row_params(int(keep.numel()))
        vals, ms, vs = [], [], []
        for i, (p, is_row) in enumerate(zip(self.bucket.params, rows)):
            m, v = self._moments_of(i)
            vals.append((p.data[keep] if is_row else p.data).clone()); ms.append((m[keep] if is_row else m).clone()); vs.append((v[keep] if is_row else v).clone())
        self._relayout(vals, ms, vs)

    @torch.no_grad()
    def append_rows(self, new_rows):
        """``cat_tensors_to_optimizer(tensors_dict)`` (scene/gaussian_model.py:413-434): ``new_rows`` maps a parameter (the object, or
        its index in the bucket) to the rows appended to it; their moments start at ZERO (``torch.zeros_like(extension_tensor)``,
        :422-423), the existing rows keep theirs, the shared step count stays.  Parameters not named keep their shape."""
        index = {id(p): i for i, p in enumerate(self.bucket.params)}
        ext = {}
        for key, t in new_rows.items():
            ext[key if isinstance(key, int) else index[id(key)]] = t
        vals, ms, vs = [], [], []
        for i, p in enumerate(self.bucket.params):
            m, v = self._moments_of(i)
            if i in ext:
                e = ext[i].detach().to(device=p.device, dtype=torch.float32)
                if tuple(e.shape[1:]) != tuple(p.shape[1:]):
                    raise ValueError(f"append_rows: rows of shape {tuple(e.shape)} do not extend a parameter of shape {tuple(p.shape)}")
                vals.append(torch.cat((p.data, e), dim=0)); ms.append(torch.cat((m, torch.zeros_like(e)), dim=0)); vs.append(torch.cat((v, torch.zeros_like(e)), dim=0))
            else:
                vals.append(p.data.clone()); ms.append(m.clone()); vs.append(v.clone())
        self._relayout(vals, ms, vs)

    @torch.no_grad()
    def reset_rows(self, param, values: torch.Tensor):
        """``replace_tensor_to_optimizer(tensor, name)`` (scene/gaussian_model.py:362-375; ``reset_opacity`` :314-317): the parameter
        takes ``values`` and BOTH its moments are zeroed.  Same shape, same addresses: in place, nothing is re-laid-out and a captured
        hipGraph stays valid (the copies run on the current stream, in order with the replays)."""
        i = param if isinstance(param, int) else {id(p): k for k, p in enumerate(self.bucket.params)}[id(param)]
        p = self.bucket.params[i]
        if tuple(values.shape) != tuple(p.shape):
            raise ValueError(f"reset_rows: values of shape {tuple(values.shape)} for a parameter of shape {tuple(p.shape)}")
        if self.shard is not None:
            raise RuntimeError("reset_rows on a sharded FlatAdamW: the parameter's moments live on several ranks")
        p.data.copy_(values.detach())
        m, v = self._moments_of(i)
        m.zero_(); v.zero_()

    def step(self, skip_word=None, skip_mask=2, extra_grads=None, grad_scale=1.0):
        """One update.  ``skip_word`` (capturable optimizers only): a one-element int32 / float32 DEVICE tensor; if
        ``skip_word & skip_mask`` is non-zero when the kernel runs, the step is a no-op on the device -- parameters, moments and the
        step counter stay bit for bit (C ABI ``moss_adamw_flat_guarded``).  Not with ``shard``: the skip is a per-rank decision.  ``frame_status_word(img_buffer)`` of a rasterizer forward
        with the default mask 2 skips the step of a frame that overflowed its capacity and rendered nothing (inside a captured
        hipGraph nobody else can).  ``extra_grads`` (up to three flat tensors laid out like the bucket) + ``grad_scale``: the step's
        gradient is ((bucket + extra[0]) + extra[1] ...) x grad_scale, formed inside the update kernel in that order -- B views per
        optimizer step on one device (``moss_amd.multiview``)."""
        dev = self.flat_params.device
        if getattr(self, "fused", None) is not None:
            cx = getattr(self, "_fused_context", None)
            if cx is not None and cx.fused_adamw is self.fused:
                return                                       # the rasterizer's backward kernel took the step (fuse_into_backward)
            # somebody cleared the context by hand: the backward wrote gradients to the bucket again, and a silent no-op here would
            # freeze the parameters
            raise RuntimeError("FlatAdamW.step(): this optimizer was fused into a rasterizer backward, but that context no longer "
                               "carries it (context.fused_adamw was cleared by hand?) -- call optimizer.unfuse() to go back to step()")
        from ._lib import AdamWFlatArgs
        if skip_word is not None:
            if self.step_state is None:
                raise RuntimeError("a guarded step needs capturable=True: a host-side step count cannot know about the skipped step")
            if self.shard is not None:
                # one rank's frame overflowed, its peers' did not: this rank would skip ITS shard's update while the others step theirs,
                # and the all-gather would then spread a half-stepped parameter vector
                raise RuntimeError("a guarded step (skip_word) on a SHARDED optimizer: the skip is a per-rank decision and would leave the "
                                   "shards at different steps; guard the frame before the gradient exchange instead")
        else:
            self.t += 1
        first, count = (0, self.n) if self.shard is None else (self.first, self.count)
        if count == 0:
            return
        grads = self.bucket.flat if self.shard is None else self.grad_shard
        # ONE entry point for every form (C ABI moss_adamw_flat_ex: host / device step count, shard range, frame guard) -- and the
        # degree-aware SH update (set_active_sh_degree)
        a = AdamWFlatArgs()
        a.first, a.count = first, count
        a.params, a.grads = self.flat_params[first:].data_ptr(), grads.data_ptr()
        a.exp_avg, a.exp_avg_sq = self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr()
        a.num_segments = self.nseg
        a.segment_end, a.segment_lr = C.addressof(self.seg_end), C.addressof(self.seg_lr)
        a.segment_period, a.segment_split, a.segment_lr2 = C.addressof(self.seg_period), C.addressof(self.seg_split), C.addressof(self.seg_lr2)
        act = self._seg_active()
        a.segment_active, a.inactive_zero = C.addressof(act), int(self.sh_inactive_zero)
        a.beta1, a.beta2, a.eps, a.weight_decay = float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.weight_decay)
        a.step = max(self.t, 1)
        a.step_state = None if self.step_state is None else self.step_state.data_ptr()
        a.skip_word = None if skip_word is None else skip_word.data_ptr()
        a.skip_mask = int(skip_mask) & 0xffffffff
        extra = list(extra_grads or [])
        if len(extra) > 3:
            raise ValueError("at most three extra gradient buffers (four views per step)")
        a.num_grads_extra, a.grad_scale = len(extra), float(grad_scale)
        for i, e in enumerate(extra):
            if e.numel() < grads.numel() or e.dtype != torch.float32 or e.device != grads.device or not e.is_contiguous():
                raise ValueError("an extra gradient buffer must be a contiguous float32 tensor laid out like the bucket")
            a.grads_extra[i] = e[first:].data_ptr() if self.shard is None else e.data_ptr()
        with torch.cuda.device(dev):
            rc = lib().moss_adamw_flat_ex(C.addressof(a), torch.cuda.current_stream(dev).cuda_stream)
        check(rc, "adamw_flat_ex")


class AdamW(torch.optim.Optimizer):
    """Drop-in for ``torch.optim.AdamW(params, lr, betas, eps, weight_decay)`` as MOSS builds it (scene/gaussian_model.py:226:
    eight parameter groups, ``lr=0.0, eps=1e-15``): the same update rule, ONE kernel for all the single-tensor groups (C ABI
    ``moss_adamw_multi``: up to eight tensors with their own state tensors and step counts per launch; rounds 4-5: one launch per tensor,
    ``moss_adamw_flat``, bit-identical) instead of torch's nine ``multi_tensor_apply`` launches per group -- with MOSS's six single-tensor
    Gaussian groups that is 54 launches of ~13 us per step, more than half of the patched call pattern's step (rocprofv3,
    ``profiles/r04_notes.md``).

    Everything MOSS does to its optimizer keeps working: ``param_groups`` with per-group ``lr`` rewritten every iteration
    (``update_learning_rate``), and the densification surgery on ``state[p]["exp_avg"] / ["exp_avg_sq"]`` (``cat_tensors_to_optimizer``,
    ``_prune_optimizer``, ``replace_tensor_to_optimizer``, scene/gaussian_model.py:362-430) -- the state keys and meanings are torch's
    (``step`` is kept as a Python int).  Parameters that are not contiguous float32 GPU tensors (or have no 16-byte aligned storage)
    take torch's own functional AdamW.  Not capturable in a hipGraph (the step count is a launch argument): the graph path uses
    :class:`FlatAdamW`."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        from ._lib import AdamWMultiArgs
        self._multi = AdamWMultiArgs()

    # A group of MANY tensors (MOSS's two network groups, scene/gaussian_model.py:222-223: the parameters of `auto_regression` and
    # `cross_attention_lbs`) is stepped with torch's multi-tensor primitives -- nine launches for the whole group, where one kernel per
    # tensor would be dozens; the single-tensor Gaussian groups are where one kernel replaces nine.
    FOREACH_ABOVE = 4

    def _foreach_group(self, group) -> bool:
        ps = [p for p in group["params"] if p.grad is not None]
        if not ps:
            return True
        for p in ps:
            st = self.state[p]
            if len(st) == 0:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        steps = {int(self.state[p]["step"]) for p in ps}
        if len(steps) != 1 or len({(p.device, p.dtype) for p in ps}) != 1:
            return False                                     # (tensors that joined later, mixed devices: one by one below)
        t = steps.pop() + 1
        beta1, beta2 = group["betas"]
        lr, wd, eps = group["lr"], group["weight_decay"], group["eps"]
        gs = [p.grad for p in ps]
        ms = [self.state[p]["exp_avg"] for p in ps]
        vs = [self.state[p]["exp_avg_sq"] for p in ps]
        torch._foreach_mul_(ps, 1 - lr * wd)
        torch._foreach_lerp_(ms, gs, 1 - beta1)
        torch._foreach_mul_(vs, beta2)
        torch._foreach_addcmul_(vs, gs, gs, 1 - beta2)
        denom = torch._foreach_sqrt(vs)
        torch._foreach_div_(denom, (1 - beta2 ** t) ** 0.5)
        torch._foreach_add_(denom, eps)
        torch._foreach_addcdiv_(ps, ms, denom, -lr / (1 - beta1 ** t))
        for p in ps:
            self.state[p]["step"] = t
        return True

    @staticmethod
    def _native_ok(*tensors):
        return all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0 for t in tensors)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # the single tensors of ALL groups that the kernel can take, batched by what a launch holds as scalars (device, betas, eps, weight
        # decay): MOSS's six Gaussian groups share them, so its step is ONE launch (C ABI moss_adamw_multi; up to eight tensors each)
        batches = {}
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            if len(group["params"]) > self.FOREACH_ABOVE and self._foreach_group(group):
                continue
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] = int(st["step"]) + 1
                g, m, v = p.grad, st["exp_avg"], st["exp_avg_sq"]
                n = p.numel()
                if n == 0:
                    continue
                if self._native_ok(p, g, m, v):
                    key = (p.device, float(beta1), float(beta2), float(group["eps"]), float(group["weight_decay"]))
                    batches.setdefault(key, []).append((p, g, m, v, float(group["lr"]), int(st["step"])))
                else:                                        # torch's expressions (CPU tensors, other dtypes, views with odd alignment)
                    p.mul_(1 - group["lr"] * group["weight_decay"])
                    m.lerp_(g, 1 - beta1)
                    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
                    bc1, bc2 = 1 - beta1 ** st["step"], 1 - beta2 ** st["step"]
                    p.addcdiv_(m, (v.sqrt() / (bc2 ** 0.5)).add_(group["eps"]), value=-group["lr"] / bc1)
        if batches:
            L = lib()
            a = self._multi
            for (dev, beta1, beta2, eps, wd), items in batches.items():
                a.beta1, a.beta2, a.eps, a.weight_decay = beta1, beta2, eps, wd
                with torch.cuda.device(dev):
                    stream = torch.cuda.current_stream(dev).cuda_stream
                    for i0 in range(0, len(items), 8):
                        part = items[i0:i0 + 8]
                        a.num_tensors = len(part)
                        for k, (p, g, m, v, lr, t) in enumerate(part):
                            a.numel[k], a.params[k], a.grads[k], a.exp_avg[k], a.exp_avg_sq[k] = p.numel(), p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
                            a.lr[k], a.step[k] = lr, t
                        check(L.moss_adamw_multi(C.addressof(a), stream), "adamw_multi")
        return loss
