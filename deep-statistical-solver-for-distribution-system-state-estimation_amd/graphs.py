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


import contextlib


@contextlib.contextmanager
def _all_threads_to_pool(pool, device):
    """Route EVERY thread's allocations on ``device`` to ``pool`` while a step is recorded.  torch.cuda.use_mem_pool routes the
    calling thread only, and backward() allocates from the autograd engine's device thread: weight-gradient slabs and intermediate
    gradients would then live in the general allocator, be handed to the next tensor anybody allocates after the recording, and
    every replay would write into that tensor (found by cloning gradients between two replays: tests/test_gpu_rccl.py, round 5)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    torch._C._cuda_beginAllocateToPool(idx, pool.id)
    try:
        yield
    finally:
        torch._C._cuda_endAllocateToPool(idx, pool.id)
        torch._C._cuda_releasePool(idx, pool.id)


_PLAN_RECORDER = [None]      # the PlannedStep that is recording (its collectives cut the record into segments), or None


def plan_collective(fn: Callable[[], object]):
    """The product's collective call sites (parallel.py: loss sums, gradient buckets, their joins) run ``fn`` through here.  Outside
    a recording it is just ``fn()``.  While a PlannedStep records, the collective ENDS the current plan segment, runs, and the next
    segment begins behind it: a replay then issues segment, collective, segment, ... -- the collectives are not launches of this
    library, so they are called again (same static tensors), from Python, between two ``dss2_plan_run`` calls."""
    rec = _PLAN_RECORDER[0]
    if rec is None:
        return fn()
    return rec._collective(fn)


class PlannedStep:
    """``step_fn()`` recorded ONCE as the library's own launch list (include/dss2_hip.h, "launch plans") and re-issued from one C
    call: for steps that cannot be captured into a hipGraph (or as a check of one that can).  Same contract as GraphedStep: static
    inputs; the step's tensors -- activations, gradient buffers, the ``.grad`` tensors created during the recording -- live in a
    private memory pool for the plan's lifetime; only launches of this library are replayed (a step with a torch kernel in it is not
    a candidate: compare ``replay()`` with an eager step once, as tests/test_gpu_plan.py does).  A data-parallel step's collectives
    (``plan_collective``) cut the record into segments: one C call per segment, the collectives between them."""

    def __init__(self, step_fn: Callable[[], torch.Tensor], warmup: int = 2, stream=None, verify: Callable[[], list] = None):
        """``verify``: a callable returning tensors the step produces (the loss, parameter gradients).  After the recording they are
        cloned, the plan is replayed ONCE and every one must come out bit for bit the same -- a step that contains a launch the plan
        does not carry (a torch kernel: a fill, an add, ``.to()``) is caught at construction instead of replaying silently without it.
        Only for steps that are functions of their static inputs (no optimizer step, no dropout inside): the check replays the step."""
        from . import _lib
        import ctypes as C
        self._lib, self._C = _lib, C
        self.stream = stream if stream is not None else torch.cuda.current_stream()
        self.segments = []           # [(plan handle, collective or None)]: run the plan, then the collective
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):
                step_fn()
            torch.cuda.synchronize()
            self.pool = torch.cuda.MemPool()
            _PLAN_RECORDING[0] = True
            _PLAN_RECORDER[0] = self
            try:
                with _all_threads_to_pool(self.pool, self.stream.device):
                    self._cur = self._begin()
                    try:
                        self.loss = step_fn()
                    finally:
                        self._end(None)
            finally:
                _PLAN_RECORDING[0] = False
                _PLAN_RECORDER[0] = None
            torch.cuda.synchronize()
        # The tensors of the recorded step that no recorded launch rewrites (the root gradient autograd hands to the loss node, anything a
        # torch kernel produced) stay valid only while the autograd graph of the recording keeps them alive: a step_fn that returns a
        # DETACHED loss lets them go back to the pool during the recording, where a later allocation of the same step may reuse them and
        # every replay would read overwritten memory.  So the attached loss is part of the contract.
        if torch.is_tensor(self.loss) and torch.is_grad_enabled() and self.loss.grad_fn is None and any(
                int(_lib.lib().dss2_plan_size(h)) for h, _ in self.segments):
            for h, _ in self.segments:
                _lib.lib().dss2_plan_destroy(h)
            self.segments = []
            raise ValueError("PlannedStep: step_fn must return the ATTACHED loss tensor (its autograd graph keeps the recorded step's "
                             "tensors alive in the plan's pool); it returned a tensor without grad_fn")
        self.handle = self.segments[0][0]      # (single-segment plans: the handle itself, as before)
        self.n_launches = sum(int(_lib.lib().dss2_plan_size(h)) for h, _ in self.segments)
        self.n_collectives = sum(1 for _, fn in self.segments if fn is not None)
        if verify is not None:
            with torch.cuda.stream(self.stream):
                outs = list(verify())
                want = [t.detach().clone() for t in outs]
                for t in outs:
                    if t.is_floating_point():
                        t.detach().fill_(float("nan"))      # (poisoned: the replay must rewrite every one)
                self.replay()
                torch.cuda.synchronize()
                bad = [i for i, (t, w) in enumerate(zip(verify(), want)) if not torch.equal(t.detach(), w)]
            if bad:
                raise RuntimeError(f"PlannedStep: a replay does not reproduce the recorded step (outputs {bad} of verify() differ): the step "
                                   "contains work the plan does not carry (only launches of libdss2_hip are recorded)")

    def _begin(self):
        h = self._C.c_void_p()
        self._lib.check(self._lib.lib().dss2_plan_begin(self._C.byref(h)), "dss2_plan_begin")
        return h

    def _end(self, fn):
        self._lib.check(self._lib.lib().dss2_plan_end(self._cur), "dss2_plan_end")
        self.segments.append((self._cur, fn))
        self._cur = None

    def _collective(self, fn):
        self._end(fn)
        try:
            return fn()
        finally:
            self._cur = self._begin()

    def replay(self) -> torch.Tensor:
        lib, sp = self._lib.lib(), self.stream.cuda_stream
        if len(self.segments) == 1:
            self._lib.check(lib.dss2_plan_run(self.segments[0][0], sp), "dss2_plan_run")
            return self.loss
        with torch.cuda.stream(self.stream):
            for h, fn in self.segments:
                self._lib.check(lib.dss2_plan_run(h, sp), "dss2_plan_run")
                if fn is not None:
                    fn()
        return self.loss

    def __del__(self):
        try:
            for h, _ in getattr(self, "segments", []):
                if h:
                    self._lib.lib().dss2_plan_destroy(h)
            self.segments = []
            self.handle = None
        except Exception:
            pass
