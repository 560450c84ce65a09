"""Drop-in for the hot-path functions of the reference's ``data.py`` on hand-written HIP kernels:

    gsp_wls_edge(input, edge_input, output, x_mean, x_std, edge_mean, edge_std, edge_index,
                 reg_coefs, num_samples, node_param, edge_param)     /root/reference/data.py:393-459
    get_pflow(y, edge_index, node_param, edge_param, phase_shift=True)  /root/reference/data.py:328-390

Same keyword names as the call sites /root/reference/dss2_run.py:140,193-194.  Preserved
behaviour: masks are ``!= 0`` on the normalised inputs; ``output[:, 1]`` is zeroed IN PLACE at
slack buses (data.py:413); ``phase_shift=True`` means shift = 0; trafo flag = ceil(phase shift);
batch-global V_hv / V_lv; penalties are squares of batch means; ``mu_v``, ``mu_theta`` and
``num_samples`` are accepted and unused.  NOT reproduced: the unused dense Laplacian of
data.py:422-423 (O(N^2) memory, no effect on the result).

Data-parallel use: ``gsp_wls_edge(..., group=pg)`` all-reduces the five batch sums and the node /
edge counts over the process group between the two kernel phases, which gives exactly the
single-process loss and gradient of the GLOBAL batch.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch


from . import _lib
from . import flags as FL
from .networks import _F32, _require_gpu, _rows, _stream
from .topology import Topology, get_topology


def _static_replay() -> bool:
    """True while the step is being captured into a hipGraph or recorded into a launch plan (graphs.PlannedStep): by-value seeds /
    cached batch constants must not be baked in -- the device-side state is used instead."""
    from .graphs import plan_recording
    return plan_recording() or torch.cuda.is_current_stream_capturing()


_N_PARTIAL_BLOCKS = 5


def _vec(t: torch.Tensor, n: int, device) -> torch.Tensor:
    t = t.to(device=device, dtype=_F32).contiguous()
    if t.numel() < n:
        raise ValueError(f"expected at least {n} statistics, got {t.numel()}")
    return t


def _fill_args(topo: Topology, input, edge_input, output, node_param, edge_param, x_mean, x_std, edge_mean, edge_std,
               reg_coefs, bufs) -> "_lib.WlsArgs":
    a = _lib.WlsArgs()
    a.input, a.ld_in = input[0].data_ptr(), input[1]
    a.edge_input, a.ld_ein = edge_input[0].data_ptr(), edge_input[1]
    a.output, a.ld_out = output[0].data_ptr(), output[1]
    a.node_param, a.ld_np = node_param[0].data_ptr(), node_param[1]
    a.edge_param, a.ld_ep = edge_param[0].data_ptr(), edge_param[1]
    a.x_mean, a.x_std, a.edge_mean, a.edge_std = x_mean.data_ptr(), x_std.data_ptr(), edge_mean.data_ptr(), edge_std.data_ptr()
    a.efrom, a.eto = topo.efrom.data_ptr(), topo.eto.data_ptr()
    a.inc_rowptr, a.inc_ent = topo.inc_rowptr.data_ptr(), topo.inc_ent.data_ptr()
    a.n_nodes, a.n_edges = topo.N, topo.E
    a.lam_v, a.lam_p, a.lam_pf, a.lam_reg = (float(reg_coefs[k]) for k in ("lam_v", "lam_p", "lam_pf", "lam_reg"))
    a.sums, a.partials, a.vminmax = bufs["sums"].data_ptr(), bufs["partials"].data_ptr(), bufs["vminmax"].data_ptr()
    a.apq, a.loss = bufs["apq"].data_ptr(), bufs["loss"].data_ptr()
    a.grad_output = bufs["grad"].data_ptr() if bufs.get("grad") is not None else None
    a.pflow = bufs["pflow"].data_ptr() if bufs.get("pflow") is not None else None
    return a


_VMM_CACHE = {}        # id(base tensor) -> (weakref, version, data_ptr, shape, strides, vminmax scratch)
_COUNTERS = {}         # device -> zeroed int32 word of the fused finish (the kernel re-zeroes it after every use)


def _vminmax_scratch(node_param: torch.Tensor):
    """(scratch [130], valid): the per-workgroup (min, max) pairs of ``node_param[:, 0]`` (vn_kv) are a constant of the
    batch tensor, so they are computed once per tensor and not once per step: the cache is keyed by the identity of the
    tensor that owns the storage (``x`` for the usual ``x[:, 8:]`` view), its version counter, address and layout, and an
    entry dies with its tensor.  (The version counter sees torch's own writes; the library's collate kernels write through raw
    pointers into tensors they have just allocated, which are new cache keys, and ``invalidate_vminmax`` is there for callers
    that refill a tensor through the C ABI.)"""
    import weakref
    if _static_replay():
        # a capture executes nothing and its replays run on whatever the static input holds by then: the captured step
        # always carries its own vminmax launch and never reads or fills the cache
        return torch.empty(130, dtype=_F32, device=node_param.device), False
    base = node_param._base if node_param._base is not None else node_param
    key = id(base)
    sig = (base._version, node_param.data_ptr(), tuple(node_param.shape), tuple(node_param.stride()))
    hit = _VMM_CACHE.get(key)
    if hit is not None and hit[0]() is base and hit[1] == sig:
        return hit[2], True
    vmm = torch.empty(130, dtype=_F32, device=node_param.device)
    if len(_VMM_CACHE) >= 64:
        _VMM_CACHE.pop(next(iter(_VMM_CACHE)))

    def _drop(ref, key=key):
        h = _VMM_CACHE.get(key)
        if h is not None and h[0] is ref:
            _VMM_CACHE.pop(key, None)
    _VMM_CACHE[key] = (weakref.ref(base, _drop), sig, vmm)
    return vmm, False


def invalidate_vminmax(node_param: Optional[torch.Tensor] = None) -> None:
    """Forget the cached V_hv / V_lv of one tensor (or of all): for callers that rewrite a batch tensor through raw pointers
    (C-ABI kernels), which torch's version counter does not see."""
    if node_param is None:
        _VMM_CACHE.clear()
        return
    base = node_param._base if node_param._base is not None else node_param
    _VMM_CACHE.pop(id(base), None)


_UNIT: Dict[str, torch.Tensor] = {}


def unit_grad(loss: torch.Tensor) -> torch.Tensor:
    """A cached scalar 1 on the loss's device: ``loss.backward(unit_grad(loss))`` is ``loss.backward()`` without the
    ``ones_like`` fill kernel autograd launches for the root gradient (4-5 us per step; the only non-HIP kernel left on the path)."""
    key = f"{loss.device}/{loss.dtype}"
    t = _UNIT.get(key)
    if t is None:
        t = _UNIT[key] = torch.ones((), device=loss.device, dtype=loss.dtype)
    return t


def _counter(dev) -> torch.Tensor:
    c = _COUNTERS.get(dev)
    if c is None:
        c = _COUNTERS[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
    return c


class _WlsFn(torch.autograd.Function):
    """loss, output = f(output): `output` is modified in place (theta masked) and marked dirty.

    Forward = `dss2_wls_loss_partials` (flows, injections, residual coefficients, the five batch sums; small batches: its
    last workgroup sums the partials in a fixed order and writes the loss -- ONE launch; bigger ones: a one-workgroup finish
    launch does); backward = ONE launch (`dss2_wls_loss_grad`, which takes autograd's upstream gradient as a device scalar).  Data-parallel: the sums are all-reduced after the forward
    launch and a one-thread launch re-evaluates the loss from the global sums."""

    @staticmethod
    def forward(ctx, output, topo, tensors, reg_coefs, group, pflow_out=None, node_param_orig=None):
        input, edge_input, node_param, edge_param, x_mean, x_std, edge_mean, edge_std = tensors
        ctx.set_materialize_grads(False)
        dev = output.device
        N = topo.N
        if output.dim() != 2 or output.size(1) != 2 or output.stride(1) != 1:
            raise ValueError("output must be [N, 2] with unit column stride")
        nb = (N + 255) // 256
        vmm, vmm_valid = _vminmax_scratch(node_param_orig if node_param_orig is not None else node_param[0])
        bufs = {
            "sums": torch.empty(8, dtype=torch.float64, device=dev),
            "partials": torch.empty(nb * 5, dtype=torch.float64, device=dev),
            "vminmax": vmm,
            "apq": torch.empty(N, 2, dtype=_F32, device=dev),
            "loss": torch.empty(1, dtype=_F32, device=dev),
            "grad": None,
            "pflow": pflow_out,
        }
        a = _fill_args(topo, input, edge_input, (output, output.stride(0)), node_param, edge_param,
                       x_mean, x_std, edge_mean, edge_std, reg_coefs, bufs)
        # (up to 16 workgroups the last one of the partials launch finishes the sums and writes the loss; bigger batches run the
        #  one-workgroup finish launch, which costs what 240 arrivals on one counter word cost -- FL.WLS_FUSED_FINISH: fused at any size)
        a.flags = (_lib.WLS_FUSED_FINISH if ((nb <= 16 and FL.WLS_FUSED_FINISH_SMALL) or FL.WLS_FUSED_FINISH) else 0) | (_lib.WLS_VMM_CACHED if vmm_valid else 0)
        a.counter = _counter(dev).data_ptr()
        st = _stream(output)
        L = _lib.lib()
        _lib.check(L.dss2_wls_loss_partials(C.byref(a), st), "dss2_wls_loss_partials")
        if group is not None:   # global-batch sums and counts (exact data-parallel loss)
            from . import parallel
            parallel.allreduce_loss_sums(bufs["sums"], group)
            _lib.check(L.dss2_wls_loss_value(C.byref(a), st), "dss2_wls_loss_value")
        ctx.mark_dirty(output)
        ctx.save_for_backward(output)        # (the backward launch reads the masked output: autograd checks it is unmodified)
        ctx.keep = (bufs, tensors, topo, dict(reg_coefs))   # scratch and operands of the backward launch
        return bufs["loss"].reshape(()), output

    @staticmethod
    def backward(ctx, gloss, gout_unused):
        bufs, tensors, topo, reg_coefs = ctx.keep
        (output,) = ctx.saved_tensors
        input, edge_input, node_param, edge_param, x_mean, x_std, edge_mean, edge_std = tensors
        dev = output.device
        if gloss is None:
            g = torch.zeros(topo.N, 2, dtype=_F32, device=dev)
        else:
            g = bufs["grad"] = torch.empty(topo.N, 2, dtype=_F32, device=dev)
            gl = gloss if (gloss.dtype == _F32 and gloss.is_cuda) else gloss.to(device=dev, dtype=_F32)
            a = _fill_args(topo, input, edge_input, (output, output.stride(0)), node_param, edge_param,
                           x_mean, x_std, edge_mean, edge_std, reg_coefs, bufs)
            a.flags = _lib.WLS_NO_LOSS_WRITE
            a.gscale = gl.data_ptr()
            _lib.check(_lib.lib().dss2_wls_loss_grad(C.byref(a), _stream(output)), "dss2_wls_loss_grad")
            ctx.keep_g = gl
        if gout_unused is not None:
            # other consumers of the MASKED output: theta_out = theta_in * (1 - slack) (data.py:413), so their gradient
            # reaches the pre-mask theta scaled by (1 - slack), i.e. not at all at slack buses
            g = g + gout_unused
            npar = node_param[0]
            g[:, 1] = g[:, 1] - gout_unused[:, 1] * npar[:, 1]
        return g, None, None, None, None, None, None


def gsp_wls_edge(input, edge_input, output, x_mean, x_std, edge_mean, edge_std, edge_index, reg_coefs,
                 num_samples=None, node_param=None, edge_param=None, group=None, pflow_out=None):
    """/root/reference/data.py:393-459.  Returns a 0-dim loss attached to autograd; zeroes
    ``output[:, 1]`` at slack buses in place like the reference.  Beyond the reference's arguments: ``group`` (data-
    parallel loss, see the module docstring) and ``pflow_out``, an optional contiguous [E, 8] fp32 buffer that receives
    the eight get_pflow quantities exactly as the loss kernel computed them (diagnostics / tests)."""
    _require_gpu(input, edge_input, output, node_param, edge_param, edge_index)
    dev = output.device
    topo = get_topology(edge_index, input.size(0))
    tensors = (_rows(input), _rows(edge_input), _rows(node_param), _rows(edge_param),
               _vec(x_mean, 8, dev), _vec(x_std, 8, dev), _vec(edge_mean, 4, dev), _vec(edge_std, 4, dev))
    for k in ("lam_v", "lam_p", "lam_pf", "lam_reg"):
        if k not in reg_coefs:
            raise KeyError(f"reg_coefs['{k}'] missing")
    if pflow_out is not None and (pflow_out.dtype != _F32 or tuple(pflow_out.shape) != (topo.E, 8) or not pflow_out.is_contiguous()
                                  or pflow_out.device != dev):
        raise ValueError("pflow_out must be a contiguous [E, 8] fp32 tensor on the output's device")
    loss, _ = _WlsFn.apply(output, topo, tensors, dict(reg_coefs), group, pflow_out, node_param)
    return loss


def get_pflow(y, edge_index, node_param, edge_param, phase_shift=True):
    """/root/reference/data.py:328-390: (loading_lines, loading_trafo, P_from, Q_from, P_to, Q_to,
    I_from, I_to) per stored edge.  Forward only (the reference uses it under no_grad for the
    evaluation metrics, dss2_run.py:193-194; the training path differentiates it inside
    gsp_wls_edge).  ``phase_shift`` keeps the reference's inverted sense: True (default) means shift = 0,
    False subtracts edge_param[:, 5] from the angle difference (data.py:362-365)."""
    _require_gpu(y, edge_index, node_param, edge_param)
    y2, ldy = _rows(y.detach())
    npar, ld_np = _rows(node_param)
    epar, ld_ep = _rows(edge_param)
    topo = get_topology(edge_index, y.size(0))
    dev = y.device
    vmm = torch.empty(130, dtype=_F32, device=dev)
    pf = torch.empty(topo.E, 8, dtype=_F32, device=dev)
    _lib.check(_lib.lib().dss2_get_pflow(y2.data_ptr(), ldy, npar.data_ptr(), ld_np, epar.data_ptr(), ld_ep,
                                         topo.efrom.data_ptr(), topo.eto.data_ptr(), topo.N, topo.E, vmm.data_ptr(),
                                         pf.data_ptr(), int(not phase_shift), _stream(y)), "dss2_get_pflow")
    return tuple(pf[:, k] for k in range(8))


EVAL_METRICS = ("rmse_v", "mae_v", "rmse_th", "mae_th", "rmse_loading", "mae_loading", "rmse_loading_trafos",
                "mae_loading_trafos", "prop_std_v", "prop_std_th")


def eval_batch(out, y, x, edge_index, edge_attr, x_mean, x_std, acc: torch.Tensor) -> torch.Tensor:
    """/root/reference/dss2_run.py:178-208 for one test batch without leaving the device: ``acc`` (10 doubles,
    EVAL_METRICS order) += the batch's RMSE / MAE of V and theta, of the line and trafo loadings, and the
    std ratios.  ``x`` is the full [N, 11] node tensor (features + vn_kv, slack, zero_inj), ``edge_attr`` the
    full [E, 13] one.  Returns yhat [N, 2] (de-normalised V, slack-masked theta)."""
    _require_gpu(out, y, x, edge_index, edge_attr)
    if acc.dtype != torch.float64 or acc.numel() < 10 or not acc.is_contiguous():
        raise ValueError("acc must be a contiguous float64 tensor of 10 elements")
    o2, ldo = _rows(out.detach())
    y2, ldy = _rows(y)
    npar, ld_np = _rows(x[:, 8:])
    epar, ld_ep = _rows(edge_attr[:, 6:])
    dev = out.device
    topo = get_topology(edge_index, out.size(0))
    L = _lib.lib()
    xm, xs = _vec(x_mean, 1, dev), _vec(x_std, 1, dev)
    yhat = torch.empty(topo.N, 2, dtype=_F32, device=dev)
    pft = torch.empty(topo.E, 8, dtype=_F32, device=dev)
    pfo = torch.empty(topo.E, 8, dtype=_F32, device=dev)
    vmm = torch.empty(130, dtype=_F32, device=dev)
    scratch = torch.empty(int(L.dss2_eval_scratch_doubles()), dtype=torch.float64, device=dev)
    _lib.check(L.dss2_eval_batch(o2.data_ptr(), ldo, y2.data_ptr(), ldy, npar.data_ptr(), ld_np, epar.data_ptr(), ld_ep,
                                 topo.efrom.data_ptr(), topo.eto.data_ptr(), topo.N, topo.E, xm.data_ptr(), xs.data_ptr(),
                                 yhat.data_ptr(), pft.data_ptr(), pfo.data_ptr(), vmm.data_ptr(), scratch.data_ptr(),
                                 acc.data_ptr(), _stream(out)), "dss2_eval_batch")
    return yhat
