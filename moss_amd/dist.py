"""Frame-parallel training across the GPUs of one node.

The rasterizer path shards by VIEW: every rank holds a full replica of the Gaussians (100k x 59 floats = 23.6 MB),
renders a different camera each step, and the replicas are kept identical by ONE all-reduce per step over a single
flat fp32 bucket holding every parameter gradient plus the scalar loss (RCCL over xGMI when the backend is "nccl";
gloo in the CPU tests).  There is no collective inside the rasterizer itself.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun).  Returns
    (rank, world, local_rank).  A single process (no WORLD_SIZE) stays un-initialised."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # MOSS_DIST_BACKEND=gloo: lets the N > 1 code path be exercised with several processes on ONE GPU (RCCL refuses two
            # ranks on the same device); the production backend on MI355X is "nccl" (= RCCL over xGMI)
            backend = os.environ.get("MOSS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


_AVG_OK = {}          # backend -> does the collective library average inside the collective (ncclAvg)?  Decided ONCE per process.


def avg_supported(device=None) -> bool:
    """Whether ``ReduceOp.AVG`` can be used (RCCL / NCCL >= 2.10: no separate divide pass over the bucket).  Decided once and
    IDENTICALLY on every rank -- the same backend name, the same environment switch, the same torch build answering the same
    one-element probe at the same point of the program (every rank's first exchange) -- and never revised afterwards: a rank that
    caught an error from a DATA collective and issued a different one while its peers did not would hang the job instead of
    reporting the failure.  The probe's refusal (an unsupported-op check in torch, raised before anything is launched)
    is the only exception that is caught."""
    backend = dist.get_backend()
    if backend not in _AVG_OK:
        ok = backend == "nccl" and os.environ.get("MOSS_ALLREDUCE_AVG", "1") != "0"
        if ok:
            try:
                probe = torch.ones(1, dtype=torch.float32, device=device if device is not None else torch.device("cuda", torch.cuda.current_device()))
                dist.all_reduce(probe, op=dist.ReduceOp.AVG)
            except (RuntimeError, ValueError):
                ok = False
        _AVG_OK[backend] = ok
    return _AVG_OK[backend]


def shard_layout(n: int, world: int):
    """(elements per shard, padded length) of a flat vector of `n` elements cut into `world` EQUAL shards: ceil(n / world) rounded up
    to a multiple of 4 elements (the update kernel works on float4 and every shard then starts 16-byte aligned; reduce-scatter /
    all-gather want equal counts).  The padding (< 4 * world elements) is exchanged with the rest and never read."""
    world = max(int(world), 1)
    per = -(-int(n) // world)
    per = (per + 3) // 4 * 4
    return per, per * world


def shard_views(num_views: int, rank: int, world: int):
    """View indices of this rank: r, r+n, r+2n, ... (independent units, no data-path exchange)."""
    return list(range(rank, num_views, world))


class GradBucket:
    """One flat fp32 buffer for all parameter gradients (+4 slots for the loss and its three terms); a single all-reduce
    averages it.  Layout: [gradients, in parameter order, EVERY tensor starting at a multiple of 4 floats (16-byte aligned: the
    kernels that read and write these slices -- SH staging, gradient sinks, the fused optimizer step -- work on float4; up to 3 unused
    floats after a tensor, zero and never read; ``offsets`` / ``pack``) | pad to a multiple of 4 | loss, L1, SSIM, mask L2 | pad]; ``world`` > 1
    pads the buffer to `world` equal shards (``shard_layout``) so that the same buffer serves the reduce-scatter of the sharded
    optimizer path (``ShardedStep``); the 4-float loss block is 4-aligned, so it never straddles two shards."""

    def __init__(self, params, world: int = 1):
        """COLLECTIVE when a process group of more than one rank is initialised: the constructor issues the ``ReduceOp.AVG`` probe
        all-reduce (``avg_supported``), so EVERY rank must construct its bucket(s), in the same order relative to other collectives
        (a rank-0-only evaluation bucket would hang the job; build such buckets before ``init_process_group`` or on every rank)."""
        self.params = [p for p in params if p.requires_grad]
        self.world = max(int(world), 1)
        self._layout()
        # whether the collective library averages inside the collective is decided HERE, once, on every rank alike -- not at the first
        # exchange, which may sit inside a hipGraph capture or behind a rank-dependent branch (needs the process group)
        dev = self.params[0].device
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            avg_supported(dev if dev.type == "cuda" else None)

    def _layout(self):
        """Offsets, the flat buffer and its views from the CURRENT shapes of ``self.params`` (constructor and ``relayout``)."""
        self.sizes = [p.numel() for p in self.params]
        self.offsets, off = [], 0
        for n in self.sizes:
            off = (off + 3) // 4 * 4
            self.offsets.append(off)
            off += n
        self.n_params = off                                # length of the parameter region (inner padding included)
        self.tail = (self.n_params + 3) // 4 * 4           # offset of the loss block
        self.n_exchange = self.tail + 4                    # what has to travel: gradients + loss block
        self.shard_len, padded = shard_layout(self.n_exchange, self.world)
        dev = self.params[0].device
        self.flat = torch.zeros(padded if self.world > 1 else self.n_exchange, dtype=torch.float32, device=dev)
        self.views = [self.flat[off:off + n].view_as(p) for p, n, off in zip(self.params, self.sizes, self.offsets)]
        off = self.tail
        self.loss_slot = self.flat[off:off + 1]
        self.loss_terms = self.flat[off:off + 4]          # [loss, L1, SSIM, mask L2]: moss_photometric_loss can write here directly
        self._offset = {id(p): off for p, off in zip(self.params, self.offsets)}
        self._handed_out = set()

    def relayout(self):
        """The parameters changed SHAPE (rows pruned or appended: MOSS's densification, scene/gaussian_model.py:377-454; the Parameter
        objects are the same): a new flat buffer with the new offsets -- all-zero, like a fresh bucket -- and new views; the loss block
        moves with the tail.  Views handed out before (``views``, ``loss_terms``, sinks, ``.grad``) belong to the OLD buffer: every
        parameter's ``.grad`` is dropped here, and a hipGraph that captured a step over the old buffer must be captured again
        (``FlatAdamW.prune_rows`` / ``append_rows`` call this; ``moss_amd.surgery.densification_event`` does the whole sequence).
        No collective: every replica calls it with the same shapes."""
        for p in self.params:
            p.grad = None
        self._layout()

    def pack(self, tensors):
        """The flat image (``n_params`` floats, zeros in the alignment gaps) of one tensor per parameter, in parameter order."""
        out = torch.zeros(self.n_params, dtype=torch.float32, device=tensors[0].device)
        for t, n, off in zip(tensors, self.sizes, self.offsets):
            out[off:off + n] = t.reshape(-1)
        return out

    def attach(self):
        """Make every parameter's .grad a view into the bucket, so backward ACCUMULATES straight into it (no pack copy)."""
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            p.grad = v

    # -- direct mode: the op that produces a parameter's gradient WRITES it into the bucket (no zero fill, no accumulate) ----
    def detach_grads(self):
        """Start of a step in direct mode: parameters carry no .grad, so autograd adopts whatever tensor backward returns."""
        for p in self.params:
            p.grad = None
        self._handed_out.clear()

    def sink_for(self, param):
        """A NEW view object of `param`'s slice of the bucket (or None if it is not in the bucket).  A gradient-producing op
        writes its result there and returns the view; autograd's AccumulateGrad adopts a fresh, exclusively-owned tensor as
        .grad without copying it.

        SINGLE USE per step: the producing kernels OVERWRITE the slice, so the second producer of the same parameter's gradient in
        one backward pass (two views rendered before one loss.backward(), gradient accumulation, a retained graph run again) gets
        None, writes into a fresh tensor of its own, and autograd ADDS that to the slice the first one filled -- instead of
        overwriting memory .grad already aliases.  ``detach_grads()`` (start of the next step) re-arms every slice."""
        off = self._offset.get(id(param))
        if off is None or id(param) in self._handed_out:
            return None
        self._handed_out.add(id(param))
        return self.flat[off:off + param.numel()].view_as(param)

    def collect(self):
        """End of backward in direct mode: make sure every parameter's gradient is where the bucket says it is.  Costs nothing
        when every gradient was written through a sink; otherwise copies (or zero-fills a parameter that got no gradient)."""
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
            p.grad = v

    def _mean_(self, t, world):
        if avg_supported(t.device):
            dist.all_reduce(t, op=dist.ReduceOp.AVG)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            t.div_(world)

    def all_reduce_mean(self, loss=None, world=None, sh_param=None, active_sh_degree=None):
        """One all-reduce (mean) of the whole bucket.  ``sh_param`` + ``active_sh_degree`` < its maximum: only the ACTIVE SH
        coefficients travel -- MOSS starts at degree 0 and raises it every 1000 iterations (train_ZJU.py:85-86), and the backward
        writes exact zeros above the active degree, so the mean of the rest is known: zero.  The SH record is coefficient-major PER
        GAUSSIAN ((P, K, 3): the active part is the first 3 (d+1)^2 of every 3 K floats, not a prefix of the bucket), so the active
        part is packed into a contiguous staging tensor (one copy kernel each way) and the bucket goes out as up to three pieces: what
        precedes the SH slice, the packed coefficients, what follows it (with the loss block).  59 -> 14 / 23 / 38 floats per Gaussian
        at degree 0 / 1 / 2 for the price of two more collective latencies: pays when the exchange is bandwidth-bound (DESIGN.md
        section 5: it is, at every N).

        The result is RANK-CONSISTENT (every rank ends with the same bits: what keeps replicas identical) and, with two ranks, also
        bit-identical to the one-piece all-reduce (a + b has one order).  With more ranks a ring or tree adds in an order that depends
        on how the buffer is chunked, so the three-piece form may differ from the one-piece form in the last bit -- never between ranks.
        It RELIES on the inactive coefficients' gradients being exactly zero on every rank (the rasterizer's backward writes them so);
        ``MOSS_DIST_DEBUG=1`` asserts it (one device reduction + a host read per exchange)."""
        if loss is not None and loss.data_ptr() != self.loss_slot.data_ptr():
            self.loss_slot.copy_(loss.detach().reshape(1))
        world = dist.get_world_size() if world is None and dist.is_initialized() else (world or 1)
        if world > 1 and sh_param is not None and active_sh_degree is not None and id(sh_param) in self._offset and sh_param.dim() == 3:
            K = int(sh_param.shape[1]); k = (int(active_sh_degree) + 1) ** 2
            if 0 < k < K:
                off, n = self._offset[id(sh_param)], sh_param.numel()
                sh = self.flat[off:off + n].view_as(sh_param)
                if os.environ.get("MOSS_DIST_DEBUG") == "1" and not (sh.is_cuda and torch.cuda.is_current_stream_capturing()):
                    assert not bool(sh[:, k:, :].any().item()), \
                        "active-degree SH exchange: a gradient above the active degree is not zero on this rank (something besides the rasterizer adds to it?)"
                packed = sh[:, :k, :].contiguous()
                if off > 0:
                    self._mean_(self.flat[:off], world)
                self._mean_(packed, world)
                self._mean_(self.flat[off + n:], world)      # (never empty: the loss block follows the parameters)
                sh[:, :k, :].copy_(packed)
                return self.loss_slot
        if world > 1:
            # RCCL averages inside the collective (ncclAvg): no separate pass over the bucket to divide by the world size.
            # gloo (CPU tests) has no AVG: sum, then divide.  Which of the two is decided once, for all ranks alike (avg_supported).
            if avg_supported(self.flat.device):
                dist.all_reduce(self.flat, op=dist.ReduceOp.AVG)
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
                self.flat.div_(world)
        return self.loss_slot

    def all_reduce_loss_only(self, world=None):
        """BASELINE configs[3] as written -- "frames of six subjects sharded across 8 GPUs, RCCL loss all-reduce": every rank trains
        its OWN model on its own frames (independent units, SURVEY 8e "pure task parallelism"), and the only thing that crosses xGMI
        is the 4-float loss block, averaged for logging.  16 bytes per step instead of the 23.6 MB gradient bucket."""
        world = dist.get_world_size() if world is None and dist.is_initialized() else (world or 1)
        if world > 1:
            if avg_supported(self.flat.device):
                dist.all_reduce(self.loss_terms, op=dist.ReduceOp.AVG)
            else:
                dist.all_reduce(self.loss_terms, op=dist.ReduceOp.SUM)
                self.loss_terms.div_(world)
        return self.loss_slot


def _reduce_scatter_mean(out: torch.Tensor, inp: torch.Tensor, world: int):
    """out (per) = this rank's shard of the MEAN over ranks of inp (world * per).  RCCL: one reduce-scatter with ncclAvg.  gloo (the
    CPU tests) implements no reduce-scatter: all-reduce and keep the shard -- the same values, which is all those tests need."""
    if dist.get_backend() == "nccl":
        if avg_supported(inp.device):
            dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.AVG)
        else:
            dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM)
            out.div_(world)
        return
    dist.all_reduce(inp, op=dist.ReduceOp.SUM)
    r = dist.get_rank()
    out.copy_(inp[r * out.numel():(r + 1) * out.numel()])
    out.div_(world)


class ShardedStep:
    """The exchange of an N > 1 step with the optimizer SHARDED over the ranks (the alternative to
    ``GradBucket.all_reduce_mean`` + a full optimizer step on every rank):

        reduce-scatter (mean) of the gradient bucket  ->  AdamW on this rank's 1/N of the flat parameters  ->  all-gather of the
        updated parameters (in place: every rank's shard is a slice of the same flat buffer).

    The bytes on the xGMI links equal those of the ring all-reduce it replaces (which IS a reduce-scatter followed by an all-gather);
    the update's time and the moments' memory divide by N.  The averaged loss block rides along: its owner copies it from its
    gradient shard into the parameter buffer's tail before the all-gather, after which ``loss_terms`` holds it on every rank.
    ``optimizer``: anything with ``flat_params`` (padded to world * shard_len), ``grad_shard`` (shard_len), ``first`` / ``count`` and
    ``step()`` -- ``FlatAdamW(shard=(rank, world))``; the CPU tests pass a torch restatement of the same update."""

    def __init__(self, bucket: GradBucket, optimizer, rank: int, world: int):
        if bucket.world != world:
            raise ValueError(f"the bucket was laid out for {bucket.world} shards, the group has {world} ranks: GradBucket(params, world={world})")
        self.bucket, self.opt, self.rank, self.world = bucket, optimizer, int(rank), int(world)
        per = bucket.shard_len
        if optimizer.flat_params.numel() != per * world or optimizer.grad_shard.numel() != per:
            raise ValueError("the optimizer's flat_params / grad_shard do not match the bucket's shard layout")
        self.tail_rank, self.tail_off = divmod(bucket.tail, per)              # who owns the loss block, and where in its shard
        self.loss_terms = optimizer.flat_params[bucket.tail:bucket.tail + 4]

    def step(self):
        b, o, per = self.bucket, self.opt, self.bucket.shard_len
        _reduce_scatter_mean(o.grad_shard, b.flat, self.world)
        o.step()
        if self.rank == self.tail_rank:
            self.loss_terms.copy_(o.grad_shard[self.tail_off:self.tail_off + 4])
        dist.all_gather_into_tensor(o.flat_params, o.flat_params[self.rank * per:(self.rank + 1) * per])
        return self.loss_terms[:1]
