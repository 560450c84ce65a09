"""Fused Adamax: drop-in for ``torch.optim.Adamax`` as the reference driver uses it
(/root/reference/dss2_run.py:91-92 ``getattr(optim, 'Adamax')(model.parameters(), lr=3e-3)``, stepped
at :143), as ONE HIP launch over all parameter tensors instead of ~6 ATen launches per tensor.
State keys (``exp_avg``, ``exp_inf``, ``step``) match torch's, so optimizer checkpoints
(``optimizer_state_dict`` in dss2_run.py:240-247) load either way.  No CPU fallback."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib



class FusedAdamax(torch.optim.Optimizer):
    def __init__(self, params, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capturable: bool = False):
        """capturable=True (as in torch's optimizers): the step count lives on the device, so ``step()`` may be captured
        into a hipGraph together with forward + loss + backward (graphs.GraphedStep) and every replay advances it."""
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.capturable = bool(capturable)
        self._table = {}
        self._flat_table = {}
        self.table_builds = 0      # diagnostics: how often the descriptor table had to be rebuilt

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._table = {}           # the state tensors were replaced
        self._flat_table = {}
        for g in self.param_groups:
            g.pop("_step", None)   # re-derived from the loaded per-parameter steps

    def state_dict(self):
        sd = super().state_dict()
        for g in sd["param_groups"]:
            g.pop("_step", None)   # internal: torch's format keeps the count per parameter (state[...]["step"])
        # Internally every parameter of a group shares ONE step tensor.  Exported as is, pickle / deepcopy would keep the
        # aliasing, and torch.optim.Adamax loading such a checkpoint would advance the shared tensor once per PARAMETER
        # per iteration (wrong bias correction).  torch's format is one independent tensor per parameter: export clones.
        state = {}
        for k, st in sd["state"].items():
            st = dict(st)
            if torch.is_tensor(st.get("step")):
                st["step"] = st["step"].detach().clone()
            state[k] = st
        sd["state"] = state
        return sd

    @torch.no_grad()
    def init_state(self) -> None:
        """Create the optimizer state of every parameter now (normally done lazily by the first step).  Needed before
        capturing ``step()`` into a hipGraph without warm-up steps: state created INSIDE a capture would be re-zeroed by
        every replay."""
        for group in self.param_groups:
            ps = list(group["params"])
            if not ps:
                continue
            if group.get("_step") is None:
                group["_step"] = (torch.zeros((), dtype=torch.float32, device=ps[0].device) if self.capturable else torch.tensor(0.0))
            for p in ps:
                st = self.state[p]
                if "exp_avg" not in st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_inf"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] = group["_step"]

    def _step_flat(self, gi, group, ps, shared, step, dev) -> bool:
        """ONE launch for all tensors when every gradient is a contiguous view of one flat bucket (what the backward of
        this library hands to autograd): the descriptor table lives on the device and holds the gradients' offsets inside
        the bucket, which do not change from step to step; only the bucket's address travels with the launch."""
        g0 = ps[0].grad
        base = g0.untyped_storage().data_ptr()
        for p in ps:
            g = p.grad
            if not g.is_contiguous() or g.dtype != torch.float32 or g.untyped_storage().data_ptr() != base:
                return False
        offs = tuple(p.grad.storage_offset() for p in ps)
        key = (tuple(p.data_ptr() for p in ps), offs)
        cached = self._flat_table.get(gi)
        if cached is None or cached[0] != key:
            if torch.cuda.is_current_stream_capturing():
                return False      # (a table upload cannot be captured: the by-value path below serves this step)
            import numpy as np
            arr = np.zeros((len(ps), 5), dtype=np.int64)
            for i, p in enumerate(ps):
                stp = self.state[p]
                arr[i] = (p.data_ptr(), offs[i], stp["exp_avg"].data_ptr(), stp["exp_inf"].data_ptr(), p.numel())
            tab = torch.from_numpy(arr).to(dev)
            cached = self._flat_table[gi] = (key, tab, max(p.numel() for p in ps),
                                             torch.zeros(1, dtype=torch.int32, device=dev))
            self.table_builds += 1
        _, tab, max_n, counter = cached
        b1, b2 = group["betas"]
        _lib.check(_lib.lib().dss2_adamax_step_flat(tab.data_ptr(), len(ps), max_n, base, float(group["lr"]), float(b1), float(b2),
                                                    float(group["eps"]), float(group["weight_decay"]), int(step),
                                                    (shared.data_ptr() if self.capturable else None), counter.data_ptr(),
                                                    _lib.stream_ptr(dev)), "dss2_adamax_step_flat")
        return True

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi in range(len(self.param_groups)):
            self._step_group(gi)
        return loss

    @torch.no_grad()
    def step_group(self, gi: int) -> None:
        """Step ONE parameter group (parallel.step_overlapped: the groups follow the chunks of the gradient bucket)."""
        self._step_group(gi)

    def _step_group(self, gi: int) -> None:
        group = self.param_groups[gi]
        ps = [p for p in group["params"] if p.grad is not None]
        if not ps:
            return
        dev = ps[0].device
        for p in ps:
            if not p.is_cuda or p.dtype != torch.float32:
                raise RuntimeError("FusedAdamax needs fp32 GPU parameters (there is no CPU fallback)")
        # one step tensor shared by every parameter of the group (torch keeps one per parameter with the same value):
        # advanced ONCE per step -- on the host, or by the kernel itself when capturable
        shared = group.get("_step")
        if shared is None:
            known = [self.state[p]["step"] for p in ps if "step" in self.state[p]]
            first = float(known[0]) if known else 0.0
            shared = group["_step"] = (torch.tensor(first, dtype=torch.float32, device=dev) if self.capturable
                                       else torch.tensor(first))
        for p in ps:
            st = self.state[p]
            if "exp_avg" not in st:
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_inf"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                self._table.pop(gi, None)
                self._flat_table.pop(gi, None)
            if st.get("step") is not shared:
                st["step"] = shared
        if not self.capturable:
            shared += 1.0
            step = int(shared)
        if self._step_flat(gi, group, ps, shared, step if not self.capturable else 0, dev):
            return
        grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]
        # host-side descriptor table, passed to the kernels BY VALUE (no device copy to keep alive, capture-safe);
        # parameter and state addresses are written once, the gradient addresses every step (the flat gradient
        # buckets of the backward move)
        cached = self._table.get(gi)
        pkey = tuple(p.data_ptr() for p in ps)
        if cached is None or cached[1] != pkey:
            tab = (_lib.AdamaxDesc * len(ps))()
            for d, p in zip(tab, ps):
                stp = self.state[p]
                d.param, d.exp_avg, d.exp_inf, d.n = p.data_ptr(), stp["exp_avg"].data_ptr(), stp["exp_inf"].data_ptr(), p.numel()
            cached = self._table[gi] = (tab, pkey)
            self.table_builds += 1
        tab = cached[0]
        for d, g in zip(tab, grads):
            d.grad = g.data_ptr()
        b1, b2 = group["betas"]
        st = _lib.stream_ptr(dev)
        if self.capturable:
            _lib.check(_lib.lib().dss2_adamax_step_dev(C.addressof(tab), len(ps), float(group["lr"]), float(b1),
                                                       float(b2), float(group["eps"]), float(group["weight_decay"]),
                                                       shared.data_ptr(), st), "dss2_adamax_step_dev")
        else:
            _lib.check(_lib.lib().dss2_adamax_step(C.addressof(tab), len(ps), float(group["lr"]), float(b1),
                                                   float(b2), float(group["eps"]), float(group["weight_decay"]), step, st),
                       "dss2_adamax_step")
