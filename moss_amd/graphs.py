"""Capture a training step in a hipGraph and replay it (plumbing around ``torch.cuda.CUDAGraph``; no kernels of its own).

Everything in the MI355X step that is local to a rank -- the rasterizer forward in asynchronous mode
(``diff_gaussian_rasterization.set_async(True)``: no host read-back), the fused loss, the rasterizer backward, the flat AdamW
with its step counter on the device -- talks to the device only, so it can be captured once and replayed with ONE launch per
step: the eager step is launch-bound (0.46-1.0 ms depending on the host), the replayed one is the sum of its kernels (0.32 ms).

Rules the captured function must follow (``bench.py`` and tests/test_gpu_ops.py::test_step_captured_in_hipgraph... do):
* run the step eagerly a few times first (the first synchronous forward sizes the binning capacity; allocator warm-up);
* return only DETACHED tensors: an output that still has a ``grad_fn`` keeps the step's autograd graph alive into the next replay;
* no ``hipMemsetAsync`` inside (memset nodes did not re-execute on replay with ROCm 7.2: clear with a kernel), no host reads;
* call ``step.check()`` every few hundred steps (or ``diff_gaussian_rasterization.check_async_status()`` after the last replay):
  the binning capacity is baked into the graph, and a scene whose instance count grows needs a re-capture before it overflows.
  EVERY frame that overflowed between two checks is counted in ``step.dropped_frames`` (the library keeps a sticky counter in the
  context's frame state; the status words alone describe only the last frame) and the step is re-captured with the grown capacity,
  not raised.  Such a frame rendered nothing and left zero gradients.  Give the captured optimizer the frame's status word --
  ``opt.step(skip_word=frame_status_word(ctx.last_img_buffer))`` (``FlatAdamW``, C ABI ``moss_adamw_flat_guarded``) -- and a dropped
  frame is a NO-OP: parameters, moments and the device-side step counter stay bit for bit
  (tests/test_gpu_ops.py::test_dropped_frame_is_not_an_optimizer_step).  Without the guard the update kernel still runs on the
  zero gradients: a weight-decay-only step that also decays the moments (what rounds 2-3 did).
"""
from __future__ import annotations

import contextlib
import gc

import torch

__all__ = ["GraphedStep", "capturing"]


@contextlib.contextmanager
def capturing(graph, collect=False, empty_cache=True, pool=None, stream=None, capture_error_mode="global"):
    """``torch.cuda.graph(graph, pool=, stream=, capture_error_mode=)`` with two differences.

    * Python's cyclic garbage collector is held off for the duration of the capture.  A ``GraphedStep`` that was dropped together
      with the object whose bound method it captured is a reference CYCLE: it is freed by whichever allocation happens to trigger a
      collection -- and if that allocation is a tensor wrapper created inside a LATER capture, ``~CUDAGraph`` (hipGraphExecDestroy) runs
      while the stream is capturing: "operation not permitted when stream is capturing", raised from a destructor, i.e. ``terminate``.
      (torch >= 2.9 no longer calls ``gc.collect()`` on entering a capture.)  With the collector off the dead cycles simply wait for the
      first collection after the capture; ``collect=True`` frees them (and the device memory they hold) BEFORE it: one full collection,
      ~45 ms of host time in a process that has torch loaded -- a first capture can afford it, the re-capture of a densification event
      (5 ms in all) cannot.
    * ``empty_cache=False`` skips the ``torch.cuda.empty_cache()`` that ``torch.cuda.graph`` performs on entry.  It returns every cached
      segment to the driver (measured at a densification event of the bench frame: 9-11 hipFree and then 14-17 hipMalloc per event,
      3 of its 6 ms).  A first capture wants it (the private pool is carved out of what is free); a RE-capture into the pool of the
      graph it replaces does not."""
    was = gc.isenabled()
    if collect:
        gc.collect()
    gc.disable()
    try:
        if empty_cache:
            with torch.cuda.graph(graph, pool=pool, stream=stream, capture_error_mode=capture_error_mode):
                yield
        else:
            side = stream if stream is not None else torch.cuda.Stream()
            torch.cuda.synchronize(side.device)
            with torch.cuda.stream(side):
                graph.capture_begin(*(() if pool is None else (pool,)), capture_error_mode=capture_error_mode)
                try:
                    yield
                finally:
                    graph.capture_end()
    finally:
        if was:
            gc.enable()


class GraphedStep:
    """``step = GraphedStep(fn, warmup=3)``; ``out = step()`` replays the captured ``fn`` and returns the (static) outputs of the
    capture.  ``fn`` takes no arguments: it reads its inputs from tensors that are updated in place between replays."""

    def __init__(self, fn, warmup: int = 3, device=None, context=None, extra_contexts=()):
        self.fn = fn
        self.extra_contexts = list(extra_contexts)           # further RasterContexts whose forwards ``fn`` captures (B views per step)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.context = context                               # the RasterContext whose capacity is baked into the graph (None: the default one)
        self.recaptures = 0
        self.dropped_frames = 0                              # frames that overflowed the baked-in capacity (they rendered nothing)
        self.graph = None
        self.outputs = None
        self._capture(max(int(warmup), 1))

    def _capture(self, warmup):
        from .diff_gaussian_rasterization import _C
        fn, dev = self.fn, self.device
        self.captured_capacity = (self.context or _C.DEFAULT).capacity   # the binning capacity is a kernel argument: baked into the graph
        # A RE-capture goes into the memory pool of the graph it replaces, and that graph is destroyed FIRST: its blocks -- the
        # rasterizer's scratch buffers, ~250 MB at the bench frame -- are then free in the pool and serve the new capture instead of
        # hipMalloc (9-13 driver allocations per densification event while the old graph was kept alive until the new one existed:
        # a capture of 3 ms instead of 1).  A pool lives only as long as a graph uses it (torch asserts `use_count > 0`), so a
        # one-element ANCHOR graph captured into the same pool keeps it alive across the gap.
        first = self.graph is None
        pool = None if first else self._pool
        self.graph = None                                    # (the replaced graph and its static outputs go before the new capture)
        self.outputs = None
        self.graph = torch.cuda.CUDAGraph()
        # ONE capture stream for the life of the step: the allocator's free blocks belong to the stream that allocated them, also inside
        # a graph's private pool -- a new side stream per capture could not take a single block the replaced graph had left behind
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(dev)
        side = self._side
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(int(warmup)):                     # (0 on a re-capture: fn() has side effects -- it is a training step)
                fn()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        # thread_local: other threads (e.g. a data loader pinning memory) may make HIP calls during the capture
        with capturing(self.graph, collect=first, empty_cache=first, pool=pool, stream=side, capture_error_mode="thread_local"):
            self.outputs = fn()
        torch.cuda.synchronize(dev)
        if first:
            self._pool = self.graph.pool()
            self._anchor = torch.cuda.CUDAGraph()
            with capturing(self._anchor, empty_cache=False, pool=self._pool, stream=side, capture_error_mode="thread_local"):
                self._anchor_out = torch.zeros(1, device=dev)
            torch.cuda.synchronize(dev)
        # overflows of EAGER forwards on this context before (or during the warm-up of) this capture were raised to, or seen by, the
        # caller: only what the replays drop from here on is counted in ``dropped_frames``
        (self.context or _C.DEFAULT).read_dropped_frames(reset=True)
        # (the captured forwards' image buffers have not been written by anybody yet: a check before the first replay must not read them)
        for cx in [self.context or _C.DEFAULT] + self.extra_contexts:
            cx._clear_captured_status()

    def reserve_pool(self, nbytes: int) -> None:
        """Put ONE free block of ``nbytes`` into this step's graph memory pool (a throw-away capture allocates it; a private pool keeps
        what it has until it dies).  A re-capture after a densification event needs scratch a little larger than what the replaced
        graph freed -- the binning buffer follows the capacity, the gradient temporaries the number of Gaussians -- so without this
        every one of those is a hipMalloc inside the capture (5-13 per event at the bench frame: the first re-capture 7 ms instead of
        1.5).  Twice the scratch of the largest set expected is ample: ``RasterContext`` sizes ~0.4 KB per instance of capacity."""
        g = torch.cuda.CUDAGraph()
        with capturing(g, empty_cache=False, pool=self._pool, stream=self._side, capture_error_mode="thread_local"):
            t = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
            t[:16].zero_()                                   # (one node: torch warns about an empty graph)
        torch.cuda.synchronize(self.device)
        del t, g

    def __call__(self):
        self.graph.replay()
        return self.outputs

    def recapture(self, probe=None) -> None:
        """Capture ``fn`` again after something it bakes in has changed from OUTSIDE: the parameter / moment / bucket buffers
        (``FlatAdamW.prune_rows`` / ``append_rows``: MOSS densifies every 100 iterations, train_ZJU.py:171-186), the SH degree, a
        learning rate passed as a launch argument.  No eager warm-up run of ``fn`` (it is a training step and would be an uncounted
        one).  ``probe`` (optional): a side-effect-free callable run eagerly first -- e.g. a forward-only ``render()`` of the new set
        under ``torch.no_grad()`` after ``RasterContext.relearn_capacity()`` -- so that the capture sees the capacity the new set needs."""
        if probe is not None:
            probe()
            torch.cuda.synchronize(self.device)
        self._capture(warmup=0)
        self.recaptures += 1

    def check(self) -> bool:
        """Call every few hundred steps (it synchronises): verifies that the last replayed frame fitted its binning capacity
        (raises like ``check_async_status`` if it did not) and, when the instance count has drifted to within 25 % of the captured
        capacity, captures the step again with the grown one.  Returns True if it re-captured."""
        from .diff_gaussian_rasterization import _C
        cx = self.context or _C.DEFAULT
        last_overflowed = False
        try:
            cx.check_status()
        except _C.CapacityOverflow:
            last_overflowed = True                           # the frame rendered nothing; check_status already grew the capacity
        # every overflowed frame since the last check, not only the last one (the sticky counter of the frame state)
        self.dropped_frames += cx.read_dropped_frames() + (1 if last_overflowed else 0)   # (the raised one is not in the count)
        if cx.capacity != self.captured_capacity:
            # no eager warm-up run: the step was running a moment ago, and an eager fn() would be one more (uncounted) training step.
            # NOTE self.outputs is re-bound: callers must read step.outputs / the return value of step() afresh after a check().
            self._capture(warmup=0)
            self.recaptures += 1
            return True
        return False
