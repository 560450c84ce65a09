"""Fused Adamax: drop-in for ``torch.optim.Adamax`` as the reference driver uses it
(/root/reference/dss2_run.py:91-92 ``getattr(optim, 'Adamax')(model.parameters(), lr=3e-3)``, stepped
at :143), as ONE HIP launch over all parameter tensors instead of ~6 ATen launches per tensor.
State keys (``exp_avg``, ``exp_inf``, ``step``) match torch's, so optimizer checkpoints
(``optimizer_state_dict`` in dss2_run.py:240-247) load either way.  No CPU fallback."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib

_DESC = np.dtype([("param", "<u8"), ("grad", "<u8"), ("exp_avg", "<u8"), ("exp_inf", "<u8"), ("n", "<i8")])


class FusedAdamax(torch.optim.Optimizer):
    def __init__(self, params, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._table = {}
        self.table_builds = 0      # diagnostics: how often the descriptor table had to be rebuilt

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._table = {}           # the state tensors were replaced

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                if not p.is_cuda or p.dtype != torch.float32:
                    raise RuntimeError("FusedAdamax needs fp32 GPU parameters (there is no CPU fallback)")
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_inf"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            step = int(self.state[ps[0]]["step"]) + 1
            for p in ps:
                self.state[p]["step"] = torch.tensor(float(step))
            grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]
            # the descriptor table is keyed by every address it holds; the gradient tensors themselves are NOT kept
            # (holding them would pin the previous flat gradient buffer, the next backward would allocate elsewhere
            # and the table would be rebuilt every step)
            key = tuple((p.data_ptr(), g.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_inf"].data_ptr())
                        for p, g in zip(ps, grads))
            cached = self._table.get(gi)
            if cached is None or cached[0] != key:
                arr = np.array([k + (p.numel(),) for k, p in zip(key, ps)], dtype=_DESC)
                host = torch.from_numpy(arr.view(np.uint8).copy())
                dev_tab = cached[1] if cached is not None and cached[1].numel() == host.numel() else \
                    torch.empty(host.numel(), dtype=torch.uint8, device=ps[0].device)
                dev_tab.copy_(host.pin_memory(), non_blocking=True)
                cached = (key, dev_tab, max(p.numel() for p in ps))
                self._table[gi] = cached
                self.table_builds += 1
            b1, b2 = group["betas"]
            st = torch.cuda.current_stream(ps[0].device).cuda_stream
            _lib.check(_lib.lib().dss2_adamax_step(cached[1].data_ptr(), len(ps), cached[2], float(group["lr"]), float(b1),
                                                   float(b2), float(group["eps"]), float(group["weight_decay"]), step, st),
                       "dss2_adamax_step")
        return loss
