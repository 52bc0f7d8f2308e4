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
"""
from __future__ import annotations

import torch

__all__ = ["GraphedStep"]


class GraphedStep:
    """``step = GraphedStep(fn, warmup=3)``; ``out = step()`` replays the captured ``fn`` and returns the (static) outputs of the
    capture.  ``fn`` takes no arguments: it reads its inputs from tensors that are updated in place between replays."""

    def __init__(self, fn, warmup: int = 3, device=None):
        self.fn = fn
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._capture(warmup)

    def _capture(self, warmup):
        from .diff_gaussian_rasterization import _C
        fn, dev = self.fn, self.device
        self.captured_capacity = _C.ASYNC.capacity           # the binning capacity is a kernel argument: baked into the graph
        self.graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(int(warmup), 1)):
                fn()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        # thread_local: other threads (e.g. a data loader pinning memory) may make HIP calls during the capture
        with torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):
            self.outputs = fn()
        torch.cuda.synchronize(dev)

    def __call__(self):
        self.graph.replay()
        return self.outputs

    def check(self) -> bool:
        """Call every few hundred steps (it synchronises): verifies that the last replayed frame fitted its binning capacity
        (raises like ``check_async_status`` if it did not) and, when the instance count has drifted to within 25 % of the captured
        capacity, captures the step again with the grown one.  Returns True if it re-captured."""
        from .diff_gaussian_rasterization import _C
        _C.check_async_status()
        if _C.ASYNC.capacity != self.captured_capacity:
            self._capture(warmup=1)
            return True
        return False
