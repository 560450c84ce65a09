"""Drop-in for the hot-path classes of the reference's ``networks.py``, on hand-written HIP kernels.

Same names, constructor arguments, ``forward(x, edge_index, edge_attr)`` signatures and
``state_dict`` keys as /root/reference/networks.py:159-388:

    EdgeAggregation(dim_featn, dim_feate, dim_hid, dim_out)            networks.py:159-209
    MPN / SkipMPN(dim_featn, dim_feate, dim_out, dim_hid, n_gnn_layers, K, dropout_rate)
                                                                       networks.py:212-338
    PFN / SkipPFN(..., L)                                              networks.py:340-388
    TAGConv(in_channels, out_channels, K)       PyG's operator as used at networks.py:230-234

All compute runs in libdss2_hip.so (include/dss2_hip.h) through ctypes; tensors must be fp32 on a
gfx950 device.  There is no CPU or eager-PyTorch fallback: CPU tensors raise.

Preserved reference behaviour: concat order [x_i | x_j | edge_attr]; first-edge-only
``is_directed``; reverse edges with edge_attr columns 0 and 2 negated; no self loops; dropout
applied in ``eval()`` too (a fresh nn.Dropout is built inside forward, networks.py:268); ReLU after
dropout; no activation after the last conv; SkipMPN's residual.
"""
from __future__ import annotations

import ctypes as C
import os as _os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .topology import Topology, get_topology

_F32 = torch.float32

from . import flags as FL
from .ops import (_DROP_PARAMS, _dropout_params, _ncg, _ptr, _reduce, _require_gpu, _round16, _round8, _rows, _stream, _wgrad_per_cu, _wgrad_tiles, chain16_supported, chain_f16_supported, chain_gate_words, chain_head_supported, chain_head_wgrad_supported, chain_supported, csr_axpy, dropout_mask, dropout_snapshot, gather_rows, gemm16_supported, gemm_prop, gemm_prop_chain, is_narrow, finish_weights, prep_weights, reduce_pending, segment_sum, wgrad, wgrad_batched)
from .plans import (_DESC_DTYPE, _FoldPlan, _MatView, _PackPlan, _SG_DTYPE, _as_view, _pack_table, _sg, _sg_table, _small_gemm)


# ------------------------------------------------------------------------------------------
# functional pieces (raw tensors in, raw tensors out); used by the autograd Functions below
# ------------------------------------------------------------------------------------------
def _edge_aggr_forward(topo, x, ldx, ea, ldea, W1, b1, b2, pack_w2_fwd, hid, hout, fn, fe, second_linear=True, need_dx=False):
    """need_dx: the backward will be asked for the gradient w.r.t. x (the library then picks the forward whose gates that backward recomputes exactly)."""
    N = topo.N
    S = torch.empty(N, hid, dtype=_F32, device=W1.device)
    if topo.ell_ent_tiles is not None and FL.EDGE_TILE_KERNELS:
        _lib.check(_lib.lib().dss2_edge_tile_fwd_paired(x.data_ptr(), ldx, ea.data_ptr(), ldea, W1.data_ptr(), b1.data_ptr(),
                                                        topo.tile_start.data_ptr(), topo.ell_ent_tiles.data_ptr(), topo.ell,
                                                        topo.nrb, topo.ntiles, S.data_ptr(), hid, fn, fe, int(bool(need_dx)), _stream(S)),
                   "dss2_edge_tile_fwd_paired")
    else:   # general graphs (hub nodes beyond the ELL width): row-per-wave kernel on the CSR
        _lib.check(_lib.lib().dss2_edge_hidden_fwd(x.data_ptr(), ldx, ea.data_ptr(), ldea, W1.data_ptr(), b1.data_ptr(),
                                                   topo.rowptr.data_ptr(), topo.col.data_ptr(), topo.ent.data_ptr(),
                                                   S.data_ptr(), N, hid, fn, fe, _stream(S)), "dss2_edge_hidden_fwd")
    if not second_linear:   # folded into the consumer (see _FoldPlan)
        return S, None
    x0 = torch.empty(N, hout, dtype=_F32, device=W1.device)
    # second Linear of the edge MLP after the (linear) aggregation: sum_e (W2 h_e + b2) = W2 S + deg b2
    gemm_prop(topo, S, hid, hid, pack_w2_fwd, 1, hout, x0, bias=b2, rowscale=topo.deg)
    return S, x0


def _edge_aggr_backward(topo, gx0, x, ldx, ea, ldea, W1, b1, S, pack_w2_bwd, hid, hout, fn, fe, g_w1, g_w2, need_dx,
                        pack_dx=None, dS=None, pending=None, dx_add=None):
    """g_w1: flat [hid*(2fn+fe) + hid] <- dW1, db1;  g_w2: flat [hout*hid + hout] <- dW2, db2.
    With ``dS`` given (second Linear folded into the consumer) gx0 / g_w2 are not used.
    pack_dx = (W1[:, :fn] packed, W1[:, fn:2fn] packed[, both stacked along k]); dx_add [N, fn] is added to the result
    (the skip connection's gradient).  Returns dx [N, fn] or None."""
    N = topo.N
    if dS is None:
        wgrad(topo, gx0, hout, S, hid, 1, g_w2, rowscale=topo.deg, pending=pending)
        dS = torch.empty(N, hid, dtype=_F32, device=gx0.device)
        gemm_prop(topo, gx0, gx0.stride(0), hout, pack_w2_bwd, 1, hid, dS)
    gx0 = dS
    dev = dS.device
    stride = hid * (2 * fn + fe) + hid
    tiled = topo.ell_ent_tiles is not None and topo.ellT_ent_tiles is not None and FL.EDGE_TILE_KERNELS
    n_slabs = min(topo.ntiles, 512) if tiled else int(min(512, max(1, (N + 15) // 16)))
    slab = torch.empty(n_slabs * stride, dtype=_F32, device=dev)
    L = _lib.lib()
    # U0 = sum of dz over incoming edges (x enters as x_i), U1 over outgoing edges (as x_j).  Side by side in one [N, 2 hid]
    # buffer when the K = 2 hid tile fits LDS: dx is then ONE GEMM [U0 | U1] [W1[:, :fn] ; W1[:, fn:2fn]]
    merged = bool(need_dx and pack_dx is not None and len(pack_dx) > 2 and pack_dx[2] is not None and FL.DX_MERGE and
                  L.dss2_gemm_prop_lds_bytes(topo.nrb, 1, _round8(2 * hid), 1, 0, 0) <= 160 * 1024)
    if not need_dx:
        U = u0 = u1 = None
        ldu = hid
    elif merged:
        U = torch.empty(N, 2 * hid, dtype=_F32, device=dev)
        u0, u1, ldu = U, U[:, hid:], 2 * hid
    else:
        U = torch.empty(2, N, hid, dtype=_F32, device=dev)
        u0, u1, ldu = U[0], U[1], hid
    st = _stream(gx0)
    if tiled:
        _lib.check(L.dss2_edge_tile_bwd(x.data_ptr(), ldx, ea.data_ptr(), ldea, W1.data_ptr(), b1.data_ptr(), dS.data_ptr(),
                                        topo.tile_start.data_ptr(), topo.ell_ent_tiles.data_ptr(), topo.ell, topo.nrb,
                                        topo.ntiles, slab.data_ptr(), n_slabs, _ptr(u0), ldu, hid, fn, fe, 0, st),
                   "dss2_edge_tile_bwd")
    else:
        _lib.check(L.dss2_edge_hidden_bwd(x.data_ptr(), ldx, ea.data_ptr(), ldea, W1.data_ptr(), b1.data_ptr(), dS.data_ptr(),
                                          topo.rowptr.data_ptr(), topo.col.data_ptr(), topo.ent.data_ptr(), slab.data_ptr(),
                                          n_slabs, _ptr(u0), ldu, N, hid, fn, fe, 0, st), "dss2_edge_hidden_bwd")
    _reduce(slab, 0, n_slabs, stride, g_w1, stride, pending)
    if not need_dx:
        return None
    if tiled:
        _lib.check(L.dss2_edge_tile_bwd(x.data_ptr(), ldx, ea.data_ptr(), ldea, W1.data_ptr(), b1.data_ptr(), dS.data_ptr(),
                                        topo.tile_start.data_ptr(), topo.ellT_ent_tiles.data_ptr(), topo.ellT, topo.nrb,
                                        topo.ntiles, None, n_slabs, u1.data_ptr(), ldu, hid, fn, fe, 1, st),
                   "dss2_edge_tile_bwd")
    else:
        _lib.check(L.dss2_edge_hidden_bwd(x.data_ptr(), ldx, ea.data_ptr(), ldea, W1.data_ptr(), b1.data_ptr(), dS.data_ptr(),
                                          topo.rowptrT.data_ptr(), topo.colT.data_ptr(), topo.entT.data_ptr(), None,
                                          n_slabs, u1.data_ptr(), ldu, N, hid, fn, fe, 1, st), "dss2_edge_hidden_bwd")
    dx = torch.empty(N, fn, dtype=_F32, device=dev)
    if merged:
        gemm_prop(topo, U, 2 * hid, 2 * hid, pack_dx[2], 1, fn, dx, add_src=dx_add,
                  add_ld=(dx_add.stride(0) if dx_add is not None else 0))
        return dx
    # two K = hid -> fn GEMMs on the tile kernel (the two W1 blocks are packed in place by the module's pack launch), the
    # second adding the first through the residual epilogue
    dx0 = torch.empty(N, fn, dtype=_F32, device=dev)
    gemm_prop(topo, u0, hid, hid, pack_dx[0], 1, fn, dx0)
    gemm_prop(topo, u1, hid, hid, pack_dx[1], 1, fn, dx, add_src=dx0, add_ld=fn)
    return dx if dx_add is None else dx + dx_add


def _dx_views(W1, hid, fn, fe):
    """W1[:, :fn] (x enters as x_i) and W1[:, fn:2fn] (as x_j) as in-place blocks for the pack kernel: each alone, and
    both as one group (packed stacked along k for the merged dx GEMM, see _edge_aggr_backward)."""
    ld = 2 * fn + fe
    a, b = _MatView(W1, hid, fn, ld, 0), _MatView(W1, hid, fn, ld, fn)
    return [[a], [b], [a, b]]


def _tagconv_forward(topo, h, pack_fwd, bias, nmat, hin, hout, dmask=None, relu=False, add_src=None, add_ld=0,
                     prebias=None, pre_rowscale=None, drop=None, b_format=0):
    out = torch.empty(topo.N, hout, dtype=_F32, device=h.device)
    narrow = is_narrow(nmat, hout) and prebias is None and (drop is None or drop[2] == 0)
    gemm_prop(topo, h, h.stride(0), hin, pack_fwd, nmat, hout, out, bias=bias, dmask=dmask, relu=relu,
              add_src=add_src, add_ld=add_ld, narrow_h=(hout if narrow else 0),
              prebias=prebias, pre_rowscale=pre_rowscale, drop=drop, b_format=b_format)
    return out


def _tagconv_forward_global(topo, h, pack_fwd, bias, nmat, hin, hout, relu=False, add_src=None, add_ld=0, drop=None):
    """TAGConv for graphs beyond the LDS-resident tiles (> 192 nodes): ONE plain tile GEMM X [W_0|..|W_K]^T, then K
    propagation hops in global memory in Horner order, the last one carrying the epilogue.  ``pack_fwd`` is the stacked
    layout of _PackPlan(stacked=True)."""
    N, dev = topo.N, h.device
    out = torch.empty(N, hout, dtype=_F32, device=dev)
    if nmat == 1:
        gemm_prop(topo, h, h.stride(0), hin, pack_fwd, 1, hout, out, bias=bias, relu=relu, add_src=add_src, add_ld=add_ld, drop=drop)
        return out
    wcat = nmat * hout
    Gc = torch.empty(N, (wcat + 3) // 4 * 4, dtype=_F32, device=dev)
    gemm_prop(topo, h, h.stride(0), hin, pack_fwd, 1, wcat, Gc)
    T = Gc[:, (nmat - 1) * hout:]
    for m in range(nmat - 2, -1, -1):
        blk = Gc[:, m * hout:(m + 1) * hout]
        if m > 0:
            csr_axpy(topo, T, blk, hout, add=blk)          # in place: T_m = G_m + A_hat T_{m+1}
            T = blk
        else:
            csr_axpy(topo, T, out, hout, add=blk, bias=bias, relu=relu, add_src=add_src, add_ld=add_ld, drop=drop)
    return out


def _tagconv_backward_global(topo, g, h, pack_bwd, nmat, hin, hout, g_flat, relu_src=None, need_dh=True, pending=None, drop=None):
    """Backward of _tagconv_forward_global.  Z = [g | A^T g | .. | (A^T)^K g] (K hops in global memory); then
    dW_m = Z_m^T h and db as ONE weight-gradient launch on the stacked Z, dh = Z [W_0; ..; W_K] as ONE plain GEMM."""
    N, dev = topo.N, g.device
    if nmat == 1:
        Z, wcat = g, hout
    else:
        wcat = nmat * hout
        Z = torch.empty(N, (wcat + 3) // 4 * 4, dtype=_F32, device=dev)
        if Z.size(1) != wcat:
            Z[:, wcat:].zero_()
        Z[:, :hout].copy_(g)
        for m in range(1, nmat):
            csr_axpy(topo, Z[:, (m - 1) * hout:], Z[:, m * hout:], hout, transposed=True)
    wgrad(topo, Z, wcat, h, hin, 1, g_flat, pending=pending, out_len=wcat * hin + hout)
    if not need_dh:
        return None
    dh = torch.empty(N, hin, dtype=_F32, device=dev)
    gemm_prop(topo, Z, Z.stride(0), wcat, pack_bwd, 1, hin, dh, relu_src=relu_src, drop=drop)
    return dh


def _tagconv_backward(topo, g, h, pack_bwd, nmat, hin, hout, g_flat, relu_src=None, dmask=None, need_dh=True,
                      rowscale2=None, defer_wgrad=False, pending=None, drop=None, b_format=0):
    """g: gradient w.r.t. the conv's pre-activation output [N, hout] (already masked).
    g_flat <- [dW_0..dW_K, db]; returns dh (masked by relu_src / dmask of the PREVIOUS layer)."""
    if not defer_wgrad:      # (deferred: the caller batches this layer's weight gradient with its siblings, wgrad_batched)
        wgrad(topo, g, hout, h, hin, nmat, g_flat, rowscale2=rowscale2, pending=pending)
    if not need_dh:
        return None
    dh = torch.empty(topo.N, hin, dtype=_F32, device=g.device)
    if is_narrow(nmat, hout) and rowscale2 is None:   # narrow gradient rows: propagate g on the input side, one stacked GEMM
        gemm_prop(topo, g, g.stride(0), nmat * hout, pack_bwd, 1, hin, dh, relu_src=relu_src, dmask=dmask,
                  transposed=True, prop_in=nmat - 1, drop=drop)
    else:
        gemm_prop(topo, g, g.stride(0), hout, pack_bwd, nmat, hin, dh, relu_src=relu_src, dmask=dmask, transposed=True,
                  drop=drop, b_format=b_format)
    return dh


# ------------------------------------------------------------------------------------------
# modules
# ------------------------------------------------------------------------------------------
class _GatherFn(torch.autograd.Function):
    """x_j = x[edge_index[0]] (by_source) / x_i = x[edge_index[1]] of PyG's propagate; the backward of a gather is the
    segmented sum over the CSR grouped by that end."""

    @staticmethod
    def forward(ctx, x, topo, by_source):
        ctx.topo, ctx.by_source, ctx.n = topo, by_source, x.size(0)
        return gather_rows(x, topo.efrom if by_source else topo.eto)

    @staticmethod
    def backward(ctx, g):
        t = ctx.topo
        rp, perm = (t.rowptrT, t.permT) if ctx.by_source else (t.rowptr, t.perm)
        return segment_sum(g.contiguous(), rp, perm, ctx.n), None, None


class _SegmentSumFn(torch.autograd.Function):
    """aggr='add': out[i] = sum of the messages of the edges whose target is i; backward = gather by target."""

    @staticmethod
    def forward(ctx, msg, topo):
        ctx.topo = topo
        return segment_sum(msg, topo.rowptr, topo.perm, topo.N)

    @staticmethod
    def backward(ctx, g):
        return gather_rows(g.contiguous(), ctx.topo.eto), None


class MessagePassing(nn.Module):
    """PyG's ``MessagePassing(aggr='add')`` base class as the reference uses it (networks.py:7,159,164,206), on the HIP
    gather / segmented-sum kernels: ``propagate(edge_index, **kwargs)`` gathers every ``foo_j`` / ``foo_i`` parameter of
    ``message()`` from ``kwargs['foo']`` at the source / target end of each edge (flow='source_to_target'), passes other
    parameters through by name, ignores keyword arguments ``message()`` does not name (which is why the ``norm`` of
    networks.py:206 is dead), sums the messages per target node (``dim_size`` = rows of the gathered tensor) and calls
    ``update()``.  Differentiable.  ``edge_index`` is used exactly as given.  ``EdgeAggregation.forward`` does not come
    through here -- it runs the fused edge-MLP kernels -- but ``EdgeAggregation.propagate`` works and gives the same
    numbers (tests), as does any user subclass with its own ``message``."""

    def __init__(self, aggr: str = "add", flow: str = "source_to_target", node_dim: int = 0):
        super().__init__()
        if aggr != "add":
            raise NotImplementedError("only aggr='add' is built")
        if flow != "source_to_target" or node_dim != 0:
            raise NotImplementedError("only flow='source_to_target', node_dim=0 are built (the reference's defaults)")
        self.aggr, self.flow, self.node_dim = aggr, flow, node_dim

    def message(self, x_j):
        return x_j

    def update(self, aggr_out):
        return aggr_out

    def propagate(self, edge_index, size=None, **kwargs):
        import inspect
        _require_gpu(edge_index)
        params = inspect.signature(self.message).parameters
        n_nodes = None if size is None else int(size if isinstance(size, int) else size[1])
        for name in params:
            if name.endswith(("_i", "_j")) and torch.is_tensor(kwargs.get(name[:-2])):
                n_nodes = int(kwargs[name[:-2]].size(0)) if n_nodes is None else n_nodes
        if n_nodes is None:
            raise ValueError("propagate: cannot infer the number of nodes (pass size= or a *_i / *_j message argument)")
        topo = get_topology_asis(edge_index, n_nodes)
        args = {}
        for name, prm in params.items():
            base = name[:-2] if name.endswith(("_i", "_j")) else None
            if base is not None and base in kwargs:
                t = kwargs[base]
                if not torch.is_tensor(t):
                    args[name] = t
                    continue
                _require_gpu(t)
                flat = t if t.dim() == 2 else t.reshape(t.size(0), -1)
                if t.is_floating_point():
                    g = _GatherFn.apply(flat, topo, name.endswith("_j"))
                else:       # index / flag tensors: plain device gather, nothing to differentiate
                    g = flat.index_select(0, (topo.efrom if name.endswith("_j") else topo.eto).long())
                args[name] = g if t.dim() == 2 else g.reshape((topo.E,) + tuple(t.shape[1:]))
            elif name in kwargs:
                args[name] = kwargs[name]
            elif prm.default is inspect.Parameter.empty and prm.kind not in (prm.VAR_KEYWORD, prm.VAR_POSITIONAL):
                raise TypeError(f"propagate: message() needs '{name}' but it was not passed")
        msg = self.message(**args)
        _require_gpu(msg)
        m2 = msg if msg.dim() == 2 else msg.reshape(msg.size(0), -1)
        if m2.size(0) != topo.E:
            raise ValueError(f"message() returned {m2.size(0)} rows for {topo.E} edges")
        out = _SegmentSumFn.apply(m2.contiguous(), topo)
        return self.update(out if msg.dim() == 2 else out.reshape((n_nodes,) + tuple(msg.shape[1:])))


class TAGConv(nn.Module):
    """PyG TAGConv(in, out, K, bias=True, normalize=True) on the fused HIP kernel.
    state_dict keys: ``bias``, ``lins.k.weight`` (k = 0..K)."""

    def __init__(self, in_channels: int, out_channels: int, K: int = 3):
        super().__init__()
        if K < 0:
            raise ValueError("TAGConv: K >= 0")
        # K <= 3: the fused tile GEMM + Horner kernels (templated on K + 1 <= 4 matrices); K > 3: ONE plain tile GEMM
        # X [W_0 | .. | W_K]^T and K propagation hops in global memory (the path of graphs beyond the LDS-resident tiles)
        self.in_channels, self.out_channels, self.K = in_channels, out_channels, K
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self.lins = nn.ModuleList([nn.Linear(in_channels, out_channels, bias=False) for _ in range(K + 1)])
        self._plan = None

    def _weights(self):
        return [l.weight for l in self.lins]

    def forward(self, x: torch.Tensor, edge_index: torch.Tensor) -> torch.Tensor:
        _require_gpu(x, edge_index)
        topo = get_topology_asis(edge_index, x.size(0))
        return _TAGConvFn.apply(x, topo, self, self.bias, *self._weights())


class _TAGConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, topo, mod, bias, *ws):
        x = x.contiguous()
        hin, hout, nmat = mod.in_channels, mod.out_channels, mod.K + 1
        glob = use_global_path(topo, nmat)
        if mod._plan is None or mod._plan.device != x.device or mod._plan.stacked != glob:
            mod._plan = _PackPlan([list(ws)], x.device, stacked=glob)
        plan = mod._plan
        ctx.ver = plan.refresh()
        if not glob:
            topo.lds_check(nmat, _round8(hin), _ncg(hout))
        out = (_tagconv_forward_global if glob else _tagconv_forward)(topo, x, plan.fwd[0], bias, nmat, hin, hout)
        ctx.save_for_backward(x)
        ctx.topo, ctx.mod, ctx.glob = topo, mod, glob
        return out

    @staticmethod
    def backward(ctx, gout):
        (x,) = ctx.saved_tensors
        mod, topo, plan = ctx.mod, ctx.topo, ctx.mod._plan
        if plan.version != ctx.ver:
            plan.refresh()
        hin, hout, nmat = mod.in_channels, mod.out_channels, mod.K + 1
        g = gout.contiguous()
        flat = torch.empty(nmat * hout * hin + hout, dtype=_F32, device=g.device)
        if plan.stacked != ctx.glob:
            raise RuntimeError("the module's weight layouts changed between forward and backward")
        dh = (_tagconv_backward_global if ctx.glob else _tagconv_backward)(
            topo, g, x, plan.bwd[0], nmat, hin, hout, flat, need_dh=ctx.needs_input_grad[0])
        gw = [flat[m * hout * hin:(m + 1) * hout * hin].view(hout, hin) for m in range(nmat)]
        gb = flat[nmat * hout * hin:]
        return (dh, None, None, gb, *gw)


class EdgeAggregation(MessagePassing):
    """/root/reference/networks.py:159-209 on HIP.  ``forward`` takes the edge list AS GIVEN
    (the reference's MPN hands it the already doubled list)."""

    def __init__(self, dim_featn, dim_feate, dim_hid, dim_out):
        super().__init__(aggr="add")
        self.dim_featn, self.dim_feate, self.dim_hid, self.dim_out = dim_featn, dim_feate, dim_hid, dim_out
        self.edge_aggr = nn.Sequential(nn.Linear(dim_featn * 2 + dim_feate, dim_hid), nn.ReLU(),
                                       nn.Linear(dim_hid, dim_out))
        self._plan = None
        self._gplan, self._post = None, None      # (state of the general path, multi._EdgeAggrGeneralFn)

    def fused_dims(self) -> bool:
        """The fused edge-MLP kernels are built for the reference's data (8 node / 6 edge features, networks.py:170) and
        dim_hid <= 256; every other width runs the general path: AB = X [W1a; W1b]^T as one tile GEMM, then per edge
        relu(A[i] + B[src] + W1c ea + b1) summed per target (dss2_edge_combine_*), then the second Linear."""
        return self.dim_featn == 8 and self.dim_feate == 6 and self.dim_hid <= 256

    def message(self, x_i, x_j, edge_attr):
        """networks.py:176-181, the reference expression.  ``forward`` does not call it (the fused kernels evaluate the same
        arithmetic with the second Linear moved behind the aggregation); it serves ``propagate`` and subclasses."""
        return self.edge_aggr(torch.cat([x_i, x_j, edge_attr], dim=-1))

    def forward(self, x, edge_index, edge_attr):
        _require_gpu(x, edge_index, edge_attr)
        _no_edge_attr_grad(edge_attr)
        topo = get_topology_asis(edge_index, x.size(0))
        lin1, lin2 = self.edge_aggr[0], self.edge_aggr[2]
        if not self.fused_dims():
            from .multi import _EdgeAggrGeneralFn, _check_general_dims
            _check_general_dims(self)
            return _EdgeAggrGeneralFn.apply(x, edge_attr, topo, self, lin1.weight, lin1.bias, lin2.weight, lin2.bias)
        return _EdgeAggrFn.apply(x, edge_attr, topo, self, lin1.weight, lin1.bias, lin2.weight, lin2.bias)


class _EdgeAggrFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ea, topo, mod, W1, b1, W2, b2):
        x, ldx = _rows(x)
        ea, ldea = _rows(ea)
        if mod._plan is None or mod._plan.device != x.device:
            mod._plan = _PackPlan([[W2]] + _dx_views(W1, mod.dim_hid, mod.dim_featn, mod.dim_feate), x.device, stacked_groups=(3,))
        ctx.ver = mod._plan.refresh()
        topo.lds_check(1, _round8(mod.dim_hid), _ncg(mod.dim_out))
        S, x0 = _edge_aggr_forward(topo, x, ldx, ea, ldea, W1, b1, b2, mod._plan.fwd[0], mod.dim_hid, mod.dim_out,
                                   mod.dim_featn, mod.dim_feate, need_dx=ctx.needs_input_grad[0])
        ctx.save_for_backward(x, ea, S, W1, b1)
        ctx.topo, ctx.mod, ctx.ld = topo, mod, (ldx, ldea)
        return x0

    @staticmethod
    def backward(ctx, g):
        x, ea, S, W1, b1 = ctx.saved_tensors
        mod, topo = ctx.mod, ctx.topo
        if mod._plan.version != ctx.ver:
            mod._plan.refresh()
        hid, hout, fn, fe = mod.dim_hid, mod.dim_out, mod.dim_featn, mod.dim_feate
        g = g.contiguous()
        g1 = torch.empty(hid * (2 * fn + fe) + hid, dtype=_F32, device=g.device)
        g2 = torch.empty(hout * hid + hout, dtype=_F32, device=g.device)
        dx = _edge_aggr_backward(topo, g, x, ctx.ld[0], ea, ctx.ld[1], W1, b1, S, mod._plan.bwd[0], hid, hout, fn, fe,
                                 g1, g2, ctx.needs_input_grad[0], pack_dx=tuple(mod._plan.bwd[1:4]))
        nc = 2 * fn + fe
        return (dx, None, None, None, g1[:hid * nc].view(hid, nc), g1[hid * nc:], g2[:hout * hid].view(hout, hid),
                g2[hout * hid:])


def _no_edge_attr_grad(edge_attr: torch.Tensor) -> None:
    """The kernels produce gradients for the parameters and (when asked) for x, never for edge_attr (the reference's edge
    features are data: dss2_run.py:138).  Asking for one must not silently return None."""
    if edge_attr.requires_grad and torch.is_grad_enabled():
        raise NotImplementedError("the DSS2 HIP path does not differentiate with respect to edge_attr (edge features are inputs "
                                  "of the reference's training loop, dss2_run.py:138); detach() it")


def use_global_path(topo: Topology, nmat: int) -> bool:
    """A TAGConv runs as ONE plain tile GEMM + K propagation hops in global memory when the graphs exceed the LDS-resident
    tiles (> 192 nodes) or when K > 3 (the fused tile kernels are instantiated for K + 1 <= 4 matrices)."""
    return bool(topo.global_only or nmat > 4)


def get_topology_asis(edge_index: torch.Tensor, num_nodes: int) -> Topology:
    """Topology of an edge list used exactly as given (standalone EdgeAggregation / TAGConv / propagate)."""
    return get_topology(edge_index, num_nodes, double=False)


class MPN(nn.Module):
    """/root/reference/networks.py:212-273 on HIP (one autograd node for the whole block)."""

    skip = False

    def __init__(self, dim_featn, dim_feate, dim_out, dim_hid, n_gnn_layers, K, dropout_rate):
        super().__init__()
        self.dim_featn, self.dim_feate, self.dim_out, self.dim_hid = dim_featn, dim_feate, dim_out, dim_hid
        self.n_gnn_layers, self.K, self.dropout_rate = n_gnn_layers, K, dropout_rate
        if self.skip and dim_out != dim_featn:
            raise ValueError("SkipMPN needs dim_out == dim_featn (networks.py:336)")
        self.edge_aggr = EdgeAggregation(dim_featn, dim_feate, dim_hid, dim_hid)
        self.convs = nn.ModuleList()
        for l in range(n_gnn_layers):
            self.convs.append(TAGConv(dim_hid, dim_out if l == n_gnn_layers - 1 else dim_hid, K=K))
        self._plan = None
        self._fold = None

    # -- reference helpers kept for API parity (networks.py:236-258); not used by forward()
    def is_directed(self, edge_index):
        from .topology import reference_is_directed
        return reference_is_directed(edge_index)

    def undirect_graph(self, edge_index, edge_attr):
        if self.is_directed(edge_index):
            edge_index = torch.cat([edge_index, torch.stack([edge_index[1, :], edge_index[0, :]], dim=0)], dim=1)
            edge_attr = torch.cat([edge_attr, torch.cat([-edge_attr[:, 0:1], edge_attr[:, 1:2], -edge_attr[:, 2:3],
                                                         edge_attr[:, 3:]], dim=1)], dim=0)
        return edge_index, edge_attr

    def _params(self) -> List[torch.Tensor]:
        lin1, lin2 = self.edge_aggr.edge_aggr[0], self.edge_aggr.edge_aggr[2]
        ps = [lin1.weight, lin1.bias, lin2.weight, lin2.bias]
        for c in self.convs:
            ps.append(c.bias)
            ps.extend(l.weight for l in c.lins)
        return ps

    def _flat_offsets(self):
        """Element offsets of [W1|b1], [W2|b2], [conv l: W_0..W_K | bias] in the flat gradient buffer (a list of ints;
        the model's dimensions are fixed at construction, so it is computed once)."""
        cached = self.__dict__.get("_flat_offs")
        if cached is not None:
            return cached
        hid, fn, fe, nmat, L = self.dim_hid, self.dim_featn, self.dim_feate, self.K + 1, self.n_gnn_layers
        sizes = [hid * (2 * fn + fe) + hid, hid * hid + hid]
        for l in range(L):
            hout = self.dim_out if l == L - 1 else hid
            sizes.append(nmat * hout * hid + hout)
        offs = [0]
        for s_ in sizes:
            offs.append(offs[-1] + s_)
        self.__dict__["_flat_offs"] = offs
        return offs

    def forward(self, x, edge_index, edge_attr):
        _require_gpu(x, edge_index, edge_attr)
        _no_edge_attr_grad(edge_attr)
        topo = get_topology(edge_index, x.size(0))
        if not self.edge_aggr.fused_dims():
            return self._forward_general(x, edge_attr, topo)
        from . import stack as _stack
        dims = _stack.route(self, [self], topo)
        if dims is not None:          # dim_hid 32, K 2: the whole-stack kernels (one launch forward, one backward)
            return _stack.run(self, [self], dims, topo, x, edge_attr)
        return _MPNFn.apply(x, edge_attr, topo, self, *self._params())

    def _forward_general(self, x, edge_attr, topo):
        """networks.py:260-273 for input widths other than (8, 6) (or dim_hid > 256): the same layer loop out of the
        per-layer autograd nodes of multi.py -- general edge aggregation, TAGConv with the fused "dropout, then ReLU"
        epilogue -- instead of the fused block.  Correct for every width the kernels take; not the tuned path."""
        from .multi import _EdgeAggrGeneralFn, _check_general_dims, _run_tagconv
        ea_mod = self.edge_aggr
        _check_general_dims(ea_mod)
        lin1, lin2 = ea_mod.edge_aggr[0], ea_mod.edge_aggr[2]
        ea_mod._post = None
        h = _EdgeAggrGeneralFn.apply(x, edge_attr, topo, ea_mod, lin1.weight, lin1.bias, lin2.weight, lin2.bias)
        last = len(self.convs) - 1
        for i, conv in enumerate(self.convs):
            conv._post = None if i == last else (float(self.dropout_rate) if self.dropout_rate > 0 else True)
            h = _run_tagconv(conv, h, topo)
        return x + h if self.skip else h


class SkipMPN(MPN):
    """/root/reference/networks.py:275-338: MPN + ``input_x + x``."""

    skip = True


def _ensure_plans(mod, topo, dev, ps):
    """(pack plan, fold plan or None, global-memory mode) of an MPN block, created on first use / when the mode changes."""
    L, nmat, hid = mod.n_gnn_layers, mod.K + 1, mod.dim_hid
    W1, b1, W2, b2 = ps[0:4]
    conv_ps = [ps[4 + l * (nmat + 1): 4 + (l + 1) * (nmat + 1)] for l in range(L)]   # (bias, W_0..W_K)
    hout0 = mod.dim_out if L == 1 else hid
    glob = use_global_path(topo, nmat)       # graphs beyond the LDS-resident tiles, or K > 3: plain GEMMs + propagation hops in global memory
    fold_on = FL.FOLD_W2 and not is_narrow(nmat, hout0) and not glob
    b16 = tuple(range(1, L)) if (FL.CHAIN_BF16 and not glob and hid % 4 == 0 and hid <= 256 and not is_narrow(nmat, hid) and L >= 2
                                 and (L >= 3 or gemm16_supported(topo, nmat, hid, False))) else ()
    # ... as f16x3 where both chains of the block have the form (64-row tiles; csrc/dss2_gemm_chain_sp.hip MS = 2) and the backward takes
    # the chained route
    f16 = bool(b16 and L >= 3 and FL.WGRAD_BATCH and chain_f16_supported(topo, nmat, hid))
    if (mod._plan is None or mod._plan.device != dev or (mod._fold is not None) != fold_on or mod._plan.stacked != glob
            or tuple(sorted(mod._plan.fwd16)) != b16 or mod._plan.f16 != f16):
        offs = mod._flat_offsets()
        mod._fold = _FoldPlan(W2, b2, conv_ps[0][1:], dev, int(offs[1]), int(offs[2])) if fold_on else None
        conv_groups = [list(cp[1:]) for cp in conv_ps]
        if fold_on:   # conv 0 is packed from the folded weights
            conv_groups[0] = [_MatView(mod._fold.Wf[m], hout0, hid, hid, 0, dep=True) for m in range(nmat)]
        mod._plan = _PackPlan([[W2]] + conv_groups + _dx_views(W1, hid, mod.dim_featn, mod.dim_feate), dev, stacked=glob,
                              stacked_groups=(L + 3,), bf16_groups=b16, f16=f16)
    if mod._fold is not None:
        mod._fold.params = (W2, b2, list(conv_ps[0][1:]))
    return mod._plan, mod._fold, glob


def _mpn_forward(mod, topo, x, ea, ps, stack=None, need_dx=False):
    """One MPN block forward on raw tensors.  ``stack`` (a _StackRun, PFN / SkipPFN): the weight packing, the fold and the
    dropout snapshot were already done for all blocks of the stack in one launch each.
    Returns (out, tensors to keep for backward, meta)."""
    x, ldx = _rows(x)
    ea, ldea = _rows(ea)
    dev = x.device
    L, nmat, hid = mod.n_gnn_layers, mod.K + 1, mod.dim_hid
    W1, b1, W2, b2 = ps[0:4]
    conv_ps = [ps[4 + l * (nmat + 1): 4 + (l + 1) * (nmat + 1)] for l in range(L)]   # (bias, W_0..W_K)
    plan, fold, glob = _ensure_plans(mod, topo, dev, ps)
    if stack is None:
        ver = plan.refresh(fold)      # (fold + packing: one launch)
    else:
        ver = plan.version
    if not glob:
        topo.lds_check(nmat, _round8(hid), _ncg(hid))
    # the hid -> hid layers 0 .. L-2 as ONE chained launch (activation tile stays in LDS between layers)
    n_chain = L - 1 if (L - 1 >= 2 and chain_supported(topo, nmat, hid, False, bool(plan.fwd16))) else 0
    use16 = bool(n_chain) and bool(plan.fwd16) and chain16_supported(topo, nmat, hid, False)
    gw = chain_gate_words(topo, nmat, hid) if use16 else 0      # (inside autograd.Function.forward grad mode is off: always written; 1/32 of a layer output)
    S, h = _edge_aggr_forward(topo, x, ldx, ea, ldea, W1, b1, b2, plan.fwd[0], hid, hid, mod.dim_featn, mod.dim_feate,
                              second_linear=fold is None, need_dx=need_dx)
    if fold is not None:
        h = S            # conv 0 consumes the aggregated hidden directly
    acts = [h]
    act_bits = {}              # index into acts -> sign-bit words of that activation (chain_gate_words)
    p = float(mod.dropout_rate)
    # dropout is active regardless of .training (a fresh nn.Dropout is built inside forward, networks.py:268).  The
    # masks are not tensors: the epilogues regenerate them from (snapshot, layer id) in forward and backward.
    if p <= 0.0:
        snap, base = None, 0
    elif stack is not None:
        snap, base = stack.snapshot, stack.drop_base(mod)
    else:
        snap, base = dropout_snapshot(mod, dev), 0
    mod._last_dropout, mod._drop_base = (snap, p), base

    def drop_id(l):            # mask applied to conv l's output (none after the last conv)
        return base + l + 1 if (snap is not None and l < L - 1) else 0

    if n_chain:
        layers = []
        # the chain also writes the sign bits of its outputs; the data-gradient chain reads those instead of the
        # activations for its ReLU gates (dss2_chain_layer.y_bits / gate_bits)
        for l in range(n_chain):
            out_l = torch.empty(topo.N, hid, dtype=_F32, device=dev)
            layers.append(dict(Bp=(plan.fwd16[1 + l] if use16 else plan.fwd[1 + l]), Y=out_l, bias=conv_ps[l][0], relu=True,
                               drop_id=drop_id(l), prebias=(fold.bf if (fold is not None and l == 0) else None)))
            if gw:
                act_bits[len(acts)] = layers[-1]["y_bits"] = torch.empty(topo.ntiles * gw, dtype=torch.int64, device=dev)
            acts.append(out_l)
        # the narrow last layer inside the same launch (the tile is still in the waves' registers): dss2_gemm_prop_chain_head
        head_fused = (FL.CHAIN_HEAD_FWD and use16 and n_chain == L - 1 and n_chain <= FL.CHAIN_MAX and not glob and is_narrow(nmat, mod.dim_out)
                      and chain_head_supported(topo, nmat, hid, mod.dim_out, False))
        head = None
        if head_fused:
            y_head = torch.empty(topo.N, mod.dim_out, dtype=_F32, device=dev)
            head = dict(W=list(conv_ps[L - 1][1:1 + nmat]), nout=mod.dim_out, Y=y_head, bias=conv_ps[L - 1][0],
                        add_src=(x if mod.skip else None), add_ld=ldx)
        gemm_prop_chain(topo, h, hid, nmat, layers, pre_rowscale=(topo.deg_pows if fold is not None else None),
                        drop=((snap, p) if snap is not None else None), b_format=(2 if (use16 and plan.f16) else int(use16)), head=head)
        h = acts[-1]
        if head_fused:
            h, n_chain = y_head, L
    for l in range(n_chain, L):
        last = l == L - 1
        hout = mod.dim_out if last else hid
        if glob:
            h = _tagconv_forward_global(topo, h, plan.fwd[1 + l], conv_ps[l][0], nmat, hid, hout, relu=not last,
                                        add_src=(x if (last and mod.skip) else None), add_ld=ldx,
                                        drop=((snap, p, drop_id(l)) if snap is not None else None))
            if not last:
                acts.append(h)
            continue
        pre = (fold.bf, topo.deg_pows) if (fold is not None and l == 0) else (None, None)
        g16 = (not last) and (1 + l) in plan.fwd16 and not plan.f16 and gemm16_supported(topo, nmat, hid, False)      # tall tiles: bf16x6 per layer
        h = _tagconv_forward(topo, h, (plan.fwd16[1 + l] if g16 else plan.fwd[1 + l]), conv_ps[l][0], nmat, hid, hout, relu=not last,
                             add_src=(x if (last and mod.skip) else None), add_ld=ldx,
                             prebias=pre[0], pre_rowscale=pre[1],
                             drop=((snap, p, drop_id(l)) if snap is not None else None), b_format=int(g16))
        if not last:
            acts.append(h)
    meta = (ldx, ldea, len(acts), (snap, p, base), fold is not None, glob, ver, act_bits)
    return h, [x, ea, S] + acts, meta


def _mpn_backward(mod, topo, saved, ps, meta, gout, need_dx, flat=None, pending=None):
    """Backward of _mpn_forward.  Standalone (flat is None): allocates the block's flat gradient buffer, runs its slab
    reductions and the fold's chain rule, calls the all-reduce hook.  Inside a stack: ``flat`` is the block's slice of the
    stack's buffer and every slab reduction is only recorded in ``pending``; the caller runs them (and the chain rule of
    all folds) in one launch each after the last block.  Returns (dx, parameter gradients as views into flat, fold_late)."""
    ldx, ldea, n_acts, (snap, p_drop, base), folded, glob, ver, act_bits = meta
    x, ea, S = saved[0:3]
    acts = list(saved[3:3 + n_acts])
    in_stack = flat is not None

    def drop_of(l):            # the mask that was applied to conv l's output: (snapshot, p, id) or None
        return (snap, p_drop, base + l + 1) if snap is not None else None
    plan, fold = mod._plan, mod._fold
    if folded != (fold is not None) or plan.stacked != glob:
        raise RuntimeError("DSS2_FOLD_W2 / the module's plan changed between forward and backward")
    if plan.version != ver:
        plan.refresh(fold)   # weights are checked unchanged by autograd's saved-tensor versioning
    dev = gout.device
    L, nmat, hid, fn, fe = mod.n_gnn_layers, mod.K + 1, mod.dim_hid, mod.dim_featn, mod.dim_feate
    W1, b1 = ps[0], ps[1]
    # one flat gradient buffer; parameter gradients are returned as views into it
    offs = mod._flat_offsets()
    if flat is None:
        flat = torch.empty(int(offs[-1]), dtype=_F32, device=dev)
    g = gout.contiguous()
    deferred = []
    l_start = L - 1
    dS = None
    # slab reductions recorded in ``pending`` run in ONE launch at the end (chained path; always inside a stack)
    fold_late = False
    if L >= 3 and FL.WGRAD_BATCH and chain_supported(topo, nmat, hid, True, bool(plan.bwd16)):
        # last layer on its own; then the data-gradients of layers L-2 .. 0 as ONE chained launch
        if pending is None:
            pending = []
        l = L - 1
        use16 = bool(plan.bwd16) and chain16_supported(topo, nmat, hid, True)
        # the head's data gradient inside the chained launch (its input tile is computed from the dim_out-wide upstream gradient):
        # dss2_gemm_prop_chain_head, mode 2; only the head's weight gradient keeps a launch of its own
        # (tall tiles: only the direction-specialised data-gradient chain has the head form -- its layers gate with bit words)
        head_fused = (use16 and L - 1 <= FL.CHAIN_MAX and is_narrow(nmat, mod.dim_out)
                      and chain_head_supported(topo, nmat, hid, mod.dim_out, True)
                      and (topo.nrb <= 2 or all(act_bits.get(l_) is not None for l_ in range(1, L - 1))))
        head = None
        head_wg = None
        if head_fused:
            # ... and the head's weight gradient from the same staging (it holds the hop results and the head's input rows): one slab
            # per tile, summed with the step's other slabs; elsewhere the narrow weight-gradient launch re-reads the activation
            hw_fused = chain_head_wgrad_supported(topo, nmat, hid, mod.dim_out)
            if hw_fused:
                hw_len = nmat * mod.dim_out * hid + mod.dim_out
                hw_stride = (hw_len + 3) & ~3      # (16-byte lanes in the reduction)
                head_wg = (torch.empty(topo.ntiles * hw_stride, dtype=_F32, device=dev), hw_len, flat[offs[2 + l]:offs[3 + l]], hw_stride)
            else:
                _tagconv_backward(topo, g, acts[l], plan.bwd[1 + l], nmat, hid, mod.dim_out, flat[offs[2 + l]:offs[3 + l]],
                                  need_dh=False, pending=pending)
            g_in = torch.empty(topo.N, hid, dtype=_F32, device=dev)
            dr = drop_of(l - 1)
            head = dict(W=list(ps[4 + l * (nmat + 1) + 1:4 + (l + 1) * (nmat + 1)]), nout=mod.dim_out, G=g, gate=acts[l], Xout=g_in,
                        drop_id=(dr[2] if dr is not None else 0), wg_slab=(head_wg[0] if head_wg is not None else None),
                        wg_stride=(head_wg[3] if head_wg is not None else 0))
            g = g_in
        else:
            g = _tagconv_backward(topo, g, acts[l], plan.bwd[1 + l], nmat, hid, mod.dim_out, flat[offs[2 + l]:offs[3 + l]],
                                  relu_src=acts[l], drop=drop_of(l - 1), pending=pending)
        gl = [None] * (L - 1)                   # gl[l]: gradient w.r.t. layer l's pre-activation output
        gl[L - 2] = g
        layers = []
        for l in range(L - 2, -1, -1):
            out_l = torch.empty(topo.N, hid, dtype=_F32, device=dev)
            layers.append(dict(Bp=(plan.bwd16[1 + l] if use16 else plan.bwd[1 + l]), Y=out_l, relu_src=(acts[l] if l > 0 else None),
                               gate_bits=(act_bits.get(l) if (l > 0 and use16) else None),
                               drop_id=(base + l if (l > 0 and snap is not None) else 0)))      # mask of conv l-1: id (l-1)+1
            if l > 0:
                gl[l - 1] = out_l
        gemm_prop_chain(topo, (None if head_fused else g), hid, nmat, layers, transposed=True,
                        drop=((snap, p_drop) if snap is not None else None), b_format=(2 if (use16 and plan.f16) else int(use16)), head=head)
        if head_wg is not None:
            _reduce(head_wg[0], 0, topo.ntiles, head_wg[3], head_wg[2], head_wg[1], pending)
        d_in = layers[-1]["Y"]                  # gradient w.r.t. conv 0's input: dS (folded) or dx0
        # The folded conv 0 joins the batched launch of the plain layers (round 4; FL.WGRAD_JOIN_FOLDED=False: its own launch).
        # Round 3 kept it apart because three layers x 85 workgroups leave a 13-vs-12-tile tail at C2; measured now, the
        # joined launch is 141 us against 93 + 57, and -- what matters more -- the step writes and re-reads half the slabs
        # (255 x 197 KB instead of 128 x 2 + 256): reduction 24.7 -> 17.8 us, C2 step 0.537 -> 0.509 ms on one box.
        join = True if FL.WGRAD_JOIN_FOLDED is None else bool(FL.WGRAD_JOIN_FOLDED)
        if fold is not None and L - 1 <= 8 and join:
            # the folded conv 0 (input S, extra scaled bias sums) and the plain layers 1 .. L-2 in ONE launch
            wgrad_batched(topo, gl, hid, [S] + acts[1:L - 1], hid, nmat, flat[offs[3]:offs[2 + L - 1]],
                          first_rowscale2=topo.deg_pows, first_out=fold.gfold, pending=pending)
            fold_late = True        # the chain rule of the fold needs the reduced gfold
            dS, g = d_in, None
        else:
            if fold is not None:
                wgrad(topo, gl[0], hid, S, hid, nmat, fold.gfold, rowscale2=topo.deg_pows, pending=pending)
                fold_late = True
                dS, g = d_in, None
            else:
                g = d_in
            deferred = [(l, gl[l], acts[l]) for l in range(L - 2, (0 if fold is not None else -1), -1)]
        l_start = -1
    for l in range(l_start, -1, -1):
        hout = mod.dim_out if l == L - 1 else hid
        seg = flat[offs[2 + l]:offs[3 + l]]
        if l == 0 and fold is not None:
            # folded conv 0: weight gradient w.r.t. Wf / bf into the plan's buffer, data gradient is dS;
            # one small-GEMM launch then writes dW_m, conv0.bias, dW2, db2 into the flat buffer
            wgrad(topo, g, hout, S, hid, nmat, fold.gfold, rowscale2=topo.deg_pows, pending=pending)
            dS = torch.empty(topo.N, hid, dtype=_F32, device=dev)
            g16 = hout == hid and 1 in plan.bwd16 and not plan.f16 and gemm16_supported(topo, nmat, hid, True)
            gemm_prop(topo, g, g.stride(0), hout, (plan.bwd16[1] if g16 else plan.bwd[1]), nmat, hid, dS, transposed=True, b_format=int(g16))
            fold_late = True
            g = None
            break
        # dgrad epilogue applies the ReLU / dropout mask of the layer BELOW (its output is acts[l])
        if glob:
            g = _tagconv_backward_global(topo, g, acts[l], plan.bwd[1 + l], nmat, hid, hout, seg,
                                         relu_src=(acts[l] if l > 0 else None),
                                         drop=(drop_of(l - 1) if l > 0 else None), pending=pending)
            continue
        defer = FL.WGRAD_BATCH and hout == hid and not is_narrow(nmat, hout)
        if defer:
            deferred.append((l, g, acts[l]))
        g16 = hout == hid and (1 + l) in plan.bwd16 and not plan.f16 and gemm16_supported(topo, nmat, hid, True)
        g = _tagconv_backward(topo, g, acts[l], (plan.bwd16[1 + l] if g16 else plan.bwd[1 + l]), nmat, hid, hout, seg,
                              relu_src=(acts[l] if l > 0 else None), drop=(drop_of(l - 1) if l > 0 else None),
                              defer_wgrad=defer, pending=pending, b_format=int(g16))
    # weight gradients of the hid -> hid layers: independent of each other, so one launch (and one slab
    # reduction) covers up to 8 consecutive layers; their segments in the flat buffer are contiguous
    deferred.reverse()
    for c0 in range(0, len(deferred), 8):
        chunk = deferred[c0:c0 + 8]
        l0, l1 = chunk[0][0], chunk[-1][0]
        out = flat[offs[2 + l0]:offs[3 + l1]]
        if len(chunk) == 1:
            wgrad(topo, chunk[0][1], hid, chunk[0][2], hid, nmat, out, pending=pending)
        else:
            wgrad_batched(topo, [c[1] for c in chunk], hid, [c[2] for c in chunk], hid, nmat, out, pending=pending)
    dx = _edge_aggr_backward(topo, g, x, ldx, ea, ldea, W1, b1, S, plan.bwd[0], hid, hid, fn, fe,
                             flat[offs[0]:offs[1]], flat[offs[1]:offs[2]], need_dx,
                             pack_dx=tuple(plan.bwd[1 + L:4 + L]), dS=(dS if fold is not None else None),
                             pending=pending, dx_add=(gout if (need_dx and mod.skip) else None))
    if not in_stack:
        # all slab reductions of the block in one launch, then the chain rule of the fold (it needs the reduced gfold)
        if fold_late:
            fold._check()
        finish_weights(pending if pending is not None else [], (fold.bwd_tab if fold_late else None), flat, dev)
        hook = getattr(mod, "_grad_bucket_hook", None)
        if hook is not None:      # data-parallel: all-reduce the flat bucket once (parallel.py)
            hook(flat)
    # parameter gradients = views into the flat buffer, in the order of MPN._params(): one split call for all pieces
    lay = mod.__dict__.get("_grad_layout")
    if lay is None:
        nc = 2 * fn + fe
        pieces = [(hid * nc, (hid, nc)), (hid, None), (hid * hid, (hid, hid)), (hid, None)]      # W1, b1, W2, b2
        order = [0, 1, 2, 3]
        for l in range(L):
            hout = mod.dim_out if l == L - 1 else hid
            base = len(pieces)
            pieces += [(hout * hid, (hout, hid))] * nmat + [(hout, None)]                       # stored [W_0..W_K | bias]
            order += [base + nmat] + list(range(base, base + nmat))                              # returned bias first
        lay = mod.__dict__["_grad_layout"] = ([sz for sz, _ in pieces], [sh for _, sh in pieces], order)
    parts = flat.split(lay[0])
    grads = [parts[i] if lay[1][i] is None else parts[i].view(lay[1][i]) for i in lay[2]]
    return dx, grads, fold_late


class _MPNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ea, topo, mod, *ps):
        out, saved, meta = _mpn_forward(mod, topo, x, ea, ps, need_dx=ctx.needs_input_grad[0])
        ctx.save_for_backward(*saved, *ps)
        ctx.meta = (topo, mod, meta, len(saved))
        return out

    @staticmethod
    def backward(ctx, gout):
        topo, mod, meta, n_saved = ctx.meta
        saved = ctx.saved_tensors
        dx, grads, _ = _mpn_backward(mod, topo, saved[:n_saved], saved[n_saved:], meta, gout, ctx.needs_input_grad[0])
        return (dx, None, None, None, *grads)


class _StackPlan:
    """What the blocks of a PFN / SkipPFN stack share per step: ONE small-GEMM launch folding every block's second Linear,
    ONE pack launch for every block's weights, ONE dropout snapshot (layer ids offset per block), ONE flat gradient buffer
    (block b at base[b]), ONE slab-reduction launch and ONE chain-rule launch for all folds.  The merged descriptor tables
    are rebuilt only when a parameter or plan buffer moves."""

    DROP_STRIDE = 64      # dropout mask ids of block b: b * DROP_STRIDE + layer + 1

    def __init__(self, blocks, device):
        self.blocks, self.device = list(blocks), device
        sizes = [int(m._flat_offsets()[-1]) for m in self.blocks]
        self.base = [0]
        for s in sizes:
            self.base.append(self.base[-1] + s)
        self.total = self.base[-1]
        self.key = None
        self.fold_fwd = self.fold_bwd = self.pack_tab = None
        self.table_builds = 0

    def refresh(self, topo, params):
        plans = [_ensure_plans(m, topo, self.device, ps) for m, ps in zip(self.blocks, params)]
        key = tuple((id(p), id(f), p.pointers(), f.pointers() if f is not None else None) for p, f, _ in plans)
        if key != self.key:
            fwd, bwd, recs, max_elems = [], [], [], 0
            for b, (p, f, _) in enumerate(plans):
                if f is not None:
                    f_, b_ = f.records(self.base[b])
                    fwd += f_
                    bwd += b_
                recs += p.records()
                max_elems = max(max_elems, p.max_elems)
            self.fold_fwd = _sg_table(fwd, self.device) if fwd else None
            self.fold_bwd = _sg_table(bwd, self.device) if bwd else None
            t, cnt, n_dep = _pack_table(recs, self.device)
            self.pack_tab = (t, cnt, max_elems, n_dep)
            self.key = key
            self.table_builds += 1
        prep_weights(self.fold_fwd, self.pack_tab, self.device)      # every block's fold, then every block's packing: one launch each
        for p, _, _ in plans:
            p.version += 1


class _StackRun:
    """Per-forward state of a stack (what one call's blocks and its backward share)."""

    def __init__(self, plan: _StackPlan, snapshot):
        self.plan, self.snapshot = plan, snapshot

    def drop_base(self, mod) -> int:
        return self.plan.blocks.index(mod) * _StackPlan.DROP_STRIDE


class _PFNFn(torch.autograd.Function):
    """All blocks of a PFN / SkipPFN as ONE autograd node: the per-block housekeeping launches (fold, pack, dropout state,
    slab reductions, chain rule of the folds, gradient all-reduce) run once per stack, and every gradient of the stack is
    complete when the node returns -- with blocking collectives.  (parallel.attach_grad_allreduce(async_op=True) returns views of
    a bucket whose all-reduce is still in flight: sound only while every .grad is None, so its hook falls back to the blocking
    collective when gradients are already in place.)"""

    @staticmethod
    def forward(ctx, x, ea, topo, pfn, n_per_block, *ps):
        blocks = list(pfn.mpns)
        dev = x.device
        bounds = np.concatenate([[0], np.cumsum(n_per_block)])
        params = [ps[bounds[b]:bounds[b + 1]] for b in range(len(blocks))]
        sp = pfn._stack_plan
        if sp is None or sp.device != dev:
            sp = pfn._stack_plan = _StackPlan(blocks, dev)
        sp.refresh(topo, params)
        snap = dropout_snapshot(pfn, dev) if float(pfn.dropout_rate) > 0.0 else None
        run = _StackRun(sp, snap)
        saved, metas, counts = [], [], []
        for bi, (m, bp) in enumerate(zip(blocks, params)):
            x, sv, meta = _mpn_forward(m, topo, x, ea, bp, stack=run, need_dx=(bi > 0 or ctx.needs_input_grad[0]))
            saved += sv
            metas.append(meta)
            counts.append(len(sv))
        ctx.save_for_backward(*saved, *ps)
        ctx.meta = (topo, pfn, metas, counts, bounds)
        return x

    @staticmethod
    def backward(ctx, gout):
        topo, pfn, metas, counts, bounds = ctx.meta
        blocks = list(pfn.mpns)
        sp = pfn._stack_plan
        st = ctx.saved_tensors
        n_saved = sum(counts)
        saved, ps = st[:n_saved], st[n_saved:]
        dev = gout.device
        flat = torch.empty(sp.total, dtype=_F32, device=dev)
        pending, grads_all, any_fold = [], [None] * len(blocks), False
        sbounds = np.concatenate([[0], np.cumsum(counts)])
        g = gout
        for b in range(len(blocks) - 1, -1, -1):
            m = blocks[b]
            need_dx = b > 0 or ctx.needs_input_grad[0]
            g, grads_all[b], fl = _mpn_backward(m, topo, saved[sbounds[b]:sbounds[b + 1]], ps[bounds[b]:bounds[b + 1]], metas[b],
                                                g, need_dx, flat=flat[sp.base[b]:sp.base[b + 1]],
                                                pending=pending)
            any_fold = any_fold or fl
        if any_fold and sp.fold_bwd is None:
            raise RuntimeError("the stack's fold tables are missing")
        finish_weights(pending, (sp.fold_bwd if any_fold else None), flat, dev)
        hook = getattr(blocks[0], "_grad_bucket_hook", None)
        if hook is not None:      # data-parallel: ONE all-reduce for the whole stack's bucket
            hook(flat)
        return (g, None, None, None, None, *[t_ for gr in grads_all for t_ in gr])


class PFN(nn.Module):
    """/root/reference/networks.py:340-363: L chained MPN blocks on the same edge inputs."""

    inner = MPN

    def __init__(self, dim_featn, dim_feate, dim_out, dim_hid, n_gnn_layers, K, dropout_rate, L):
        super().__init__()
        self.dim_featn, self.dim_feate, self.dim_out, self.dim_hid = dim_featn, dim_feate, dim_out, dim_hid
        self.n_gnn_layers, self.K, self.dropout_rate, self.L = n_gnn_layers, K, dropout_rate, L
        self.mpns = nn.ModuleList()
        for l in range(L):
            if l == L - 1:
                self.mpns.append(MPN(dim_featn, dim_feate, dim_out, dim_hid, n_gnn_layers, K, dropout_rate))
            else:
                self.mpns.append(self.inner(dim_featn, dim_feate, dim_featn, dim_hid, n_gnn_layers, K, dropout_rate))
        self._stack_plan = None

    def forward(self, x, edge_index, edge_attr):
        _require_gpu(x, edge_index, edge_attr)
        _no_edge_attr_grad(edge_attr)
        if not FL.STACK_NODE or self.n_gnn_layers + 1 >= _StackPlan.DROP_STRIDE or not self.mpns[0].edge_aggr.fused_dims():
            for m in self.mpns:
                x = m(x, edge_index, edge_attr)
            return x
        topo = get_topology(edge_index, x.size(0))
        from . import stack as _stack
        blocks = list(self.mpns)
        dims = _stack.route(self, blocks, topo)
        if dims is not None:          # the reference driver's own model line: ONE launch for all blocks of the stack
            return _stack.run(self, blocks, dims, topo, x, edge_attr)
        params = [m._params() for m in self.mpns]
        return _PFNFn.apply(x, edge_attr, topo, self, tuple(len(p) for p in params), *[t_ for p in params for t_ in p])


class SkipPFN(PFN):
    """/root/reference/networks.py:365-388: SkipMPN blocks, last one a plain MPN."""

    inner = SkipMPN
