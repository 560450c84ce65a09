"""hipGraph capture of a whole training step (forward + gsp_wls_edge + backward).

The kernels are launched through a C ABI on the caller's stream with no host synchronisation, no
allocation inside the library and a cached graph structure, so a step can be captured into a
hipGraph with PyTorch's capture machinery and replayed: for launch-bound shapes (BASELINE config
C1: B = 64, H = 32: ~30 launches of a few microseconds each) replay is ~4x faster than eager
dispatch; for the GPU-bound shapes (C2, C3) it changes nothing.

Inputs must be static tensors (same storage every replay); parameter gradients land in the
``.grad`` tensors created during capture.
"""
from __future__ import annotations

from typing import Callable

import torch


class GraphedStep:
    """``step_fn()`` runs forward + loss + backward on static inputs and returns the loss tensor.
    Capture follows the canonical order: warm-up and capture on a side stream BEFORE any eager step has
    created AccumulateGrad nodes on the default stream."""

    def __init__(self, step_fn: Callable[[], torch.Tensor], warmup: int = 3, capture_error_mode: str = "global", stream=None):
        """capture_error_mode: passed to torch.cuda.graph.  "thread_local" lets other threads of the process (the
        process-group watchdog of a step that contains RCCL collectives) issue HIP calls during the capture."""
        # stream: capture on this (non-default) stream -- e.g. the one the caller has run its eager steps on, so that the
        # autograd graph's AccumulateGrad nodes already belong to it
        self.stream = stream if stream is not None else torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):
                step_fn()
        torch.cuda.current_stream().wait_stream(self.stream)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode=capture_error_mode):
            self.loss = step_fn()

    def replay(self) -> torch.Tensor:
        self.graph.replay()
        return self.loss


_PLAN_RECORDING = [False]


def plan_recording() -> bool:
    """True while a PlannedStep records: by-value seeds / step counts must come from device-side state, as under hipGraph capture."""
    return _PLAN_RECORDING[0]


class PlannedStep:
    """``step_fn()`` recorded ONCE as the library's own launch list (include/dss2_hip.h, "launch plans") and re-issued from one C
    call: for steps that cannot be captured into a hipGraph (or as a check of one that can).  Same contract as GraphedStep: static
    inputs; the step's tensors -- activations, gradient buffers, the ``.grad`` tensors created during the recording -- live in a
    private memory pool for the plan's lifetime; only launches of this library are replayed (a step with a torch kernel in it is not
    a candidate: compare ``replay()`` with an eager step once, as tests/test_gpu_plan.py does)."""

    def __init__(self, step_fn: Callable[[], torch.Tensor], warmup: int = 2, stream=None):
        from . import _lib
        import ctypes as C
        self._lib = _lib
        self.stream = stream if stream is not None else torch.cuda.current_stream()
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):
                step_fn()
            torch.cuda.synchronize()
            self.pool = torch.cuda.MemPool()
            handle = C.c_void_p()
            _PLAN_RECORDING[0] = True
            try:
                with torch.cuda.use_mem_pool(self.pool):
                    _lib.check(_lib.lib().dss2_plan_begin(C.byref(handle)), "dss2_plan_begin")
                    try:
                        self.loss = step_fn()
                    finally:
                        _lib.check(_lib.lib().dss2_plan_end(handle), "dss2_plan_end")
            finally:
                _PLAN_RECORDING[0] = False
            torch.cuda.synchronize()
        self.handle = handle
        self.n_launches = int(_lib.lib().dss2_plan_size(handle))

    def replay(self) -> torch.Tensor:
        self._lib.check(self._lib.lib().dss2_plan_run(self.handle, self.stream.cuda_stream), "dss2_plan_run")
        return self.loss

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self._lib.lib().dss2_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass
