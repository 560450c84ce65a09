"""The reference's per-layer-interleaved variants on the HIP kernels:

    MaskEmbdMPN(dim_featn, dim_feate, dim_out, dim_hid, n_gnn_layers, K, dropout_rate)            networks.py:390-470
    MultiMPN(...)                                                                                  networks.py:473-549
    MaskEmbdMultiMPN(...)                                                                          networks.py:552-644
    MaskEmbdMultiMPN_NoMP(...)                                                                     networks.py:647-735

Same constructor arguments, ``forward(data)`` taking an object with ``.x / .edge_index / .edge_attr`` and the same
``state_dict`` keys (``layers.i.*`` / ``edge_aggr.*`` / ``convs.l.*`` / ``mask_embd.{0,2}.*``).  Differences from
MPN that these classes have in the reference and that are kept: ``undirect_graph`` duplicates ``edge_attr`` for the
reverse edges WITHOUT sign flips (networks.py:440-444); ``data.x`` carries 4 node-type columns, the ``dim_featn``
features and their mask (networks.py:452-455); MultiMPN applies EdgeAggregation to the hidden activation, i.e. with
node features of width ``dim_hid``.  No CPU fallback.

Building blocks (each one autograd node over the C ABI): ``_DenseFn`` (tile GEMM + bias / activation epilogue),
``EdgeAggregationGeneral`` (AB = X [W1a; W1b]^T as one tile GEMM, ``dss2_edge_combine_*`` per edge, second Linear after
the aggregation), ``TAGConv`` with a fused dropout + ReLU epilogue.  ``dropout, then ReLU`` after a layer is fused into
that layer's last kernel; its gradient is one ``dss2_gate_grad`` launch.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .networks import (_F32, _DESC_DTYPE, _MPNFn, _PackPlan, _MatView, _ncg, _require_gpu, _round8, _rows, _stream,
                       MPN, TAGConv, dropout_snapshot, gemm_prop, is_narrow, wgrad, _reduce, _tagconv_forward,
                       _tagconv_backward, _tagconv_forward_global, _tagconv_backward_global, use_global_path)
from .topology import Topology, get_topology


def _post_spec(mod, dev):
    """(relu, snapshot or None, p) of the 'dropout then ReLU' that follows a layer inside a Multi* stack."""
    post = getattr(mod, "_post", None)
    if not post:
        return False, None, 0.0
    p = float(post)
    if post is True:
        p = 0.0
    snap = dropout_snapshot(mod, dev) if p > 0.0 else None
    mod._last_dropout = (snap, p)
    return True, snap, p


def _gate(g: torch.Tensor, y: torch.Tensor, snap, p: float) -> torch.Tensor:
    out = torch.empty_like(y)
    g = g.contiguous()
    _lib.check(_lib.lib().dss2_gate_grad(g.data_ptr(), y.data_ptr(), out.data_ptr(), y.size(0), y.size(1),
                                         (snap.data_ptr() if snap is not None else None), 1, float(p), 1, _stream(y)),
               "dss2_gate_grad")
    return out


class _StackedPack:
    """Fragment-packed [W[:, :d] ; W[:, d:2d]] (2h rows, d columns) of a first-Linear weight W [h, 2d + fe]: forward layout
    for AB = X Wab^T and data-gradient layout for dX = dAB Wab; one dss2_pack_weights launch."""

    def __init__(self, W: torch.Tensor, h: int, d: int, device):
        self.h, self.d, self.device = h, d, device
        kf, cf, kb, cb = _round8(d), _ncg(2 * h), _round8(2 * h), _ncg(d)
        self.fwd = torch.zeros(cf * (kf // 8) * 256, dtype=_F32, device=device)
        self.bwd = torch.zeros(cb * (kb // 8) * 256, dtype=_F32, device=device)
        self.geom = (kf, cf, kb, cb)
        self.ptr = None
        self.max_elems = max((cf + 1) * (kf // 8 + 1) * 64, (cb + 1) * (kb // 8 + 1) * 64)

    def refresh(self, W: torch.Tensor):
        if W.data_ptr() != self.ptr:
            kf, cf, kb, cb = self.geom
            h, d, ld = self.h, self.d, W.stride(0)
            recs = []
            for blk in (0, 1):
                src = W.data_ptr() + 4 * blk * d
                recs.append((src, self.fwd.data_ptr(), h, d, ld, 1, 0, kf, cf, blk * h))     # B[k][blk*h + j] = W[j][blk*d + k]
                recs.append((src, self.bwd.data_ptr(), h, d, ld, 0, blk * h, kb, cb, 0))     # B[blk*h + j][k] = W[j][blk*d + k]
            self.table = torch.from_numpy(np.array(recs, dtype=_DESC_DTYPE).view(np.uint8).copy()).to(self.device)
            self.ptr = W.data_ptr()
        _lib.check(_lib.lib().dss2_pack_weights(self.table.data_ptr(), 4, self.max_elems,
                                                _lib.stream_ptr(self.device)), "dss2_pack_weights")


class _EdgeAggrGeneralFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ea, topo, mod, W1, b1, W2, b2):
        x, ldx = _rows(x)
        ea, ldea = _rows(ea)
        dev = x.device
        d, fe, h, ho = mod.dim_featn, mod.dim_feate, mod.dim_hid, mod.dim_out
        if mod._gplan is None or mod._gplan[0].device != dev:
            mod._gplan = (_StackedPack(W1, h, d, dev), _PackPlan([[W2]], dev))
        sp, p2 = mod._gplan
        sp.refresh(W1)
        ctx.ver = p2.refresh()
        topo.lds_check(1, max(_round8(d), _round8(h), _round8(2 * h)), max(_ncg(2 * h), _ncg(ho)))
        L = _lib.lib()
        N = topo.N
        AB = torch.empty(N, 2 * h, dtype=_F32, device=dev)
        gemm_prop(topo, x, ldx, d, sp.fwd, 1, 2 * h, AB)
        S = torch.empty(N, h, dtype=_F32, device=dev)
        w1c = W1.data_ptr() + 4 * 2 * d
        _lib.check(L.dss2_edge_combine_fwd(AB.data_ptr(), 2 * h, ea.data_ptr(), ldea, w1c, W1.stride(0), b1.data_ptr(),
                                           topo.rowptr.data_ptr(), topo.col.data_ptr(), topo.ent.data_ptr(), S.data_ptr(), N, h, fe,
                                           _stream(S)), "dss2_edge_combine_fwd")
        relu, snap, p = _post_spec(mod, dev)
        y = torch.empty(N, ho, dtype=_F32, device=dev)
        # second Linear after the (linear) aggregation: sum_e (W2 h_e + b2) = W2 S + deg b2, then dropout + ReLU if inside a stack
        gemm_prop(topo, S, h, h, p2.fwd[0], 1, ho, y, bias=b2, rowscale=topo.deg, relu=relu,
                  drop=((snap, p, 1) if snap is not None else None))
        # (y is only needed for the gate; the last layer's output is not saved: gsp_wls_edge masks the model output IN PLACE,
        #  data.py:413, and a saved tensor must not change under autograd)
        ctx.save_for_backward(x, ea, AB, S, y if relu else None, W1, b1)
        ctx.meta = (topo, mod, ldx, ldea, relu, snap, p)
        return y

    @staticmethod
    def backward(ctx, g):
        x, ea, AB, S, y, W1, b1 = ctx.saved_tensors
        topo, mod, ldx, ldea, relu, snap, p = ctx.meta
        sp, p2 = mod._gplan
        if p2.version != ctx.ver:
            sp.refresh(W1)
            p2.refresh()
        d, fe, h, ho = mod.dim_featn, mod.dim_feate, mod.dim_hid, mod.dim_out
        dev, N, L = g.device, topo.N, _lib.lib()
        g = _gate(g, y, snap, p) if relu else g.contiguous()
        g2 = torch.empty(ho * h + ho, dtype=_F32, device=dev)
        wgrad(topo, g, ho, S, h, 1, g2, rowscale=topo.deg)
        dS = torch.empty(N, h, dtype=_F32, device=dev)
        gemm_prop(topo, g, ho, ho, p2.bwd[0], 1, h, dS)
        dAB = torch.empty(N, 2 * h, dtype=_F32, device=dev)
        n_slabs = int(min(512, max(1, (N + 15) // 16)))
        stride = h * fe + h
        slab = torch.empty(n_slabs * stride, dtype=_F32, device=dev)
        w1c = W1.data_ptr() + 4 * 2 * d
        st = _stream(g)
        _lib.check(L.dss2_edge_combine_bwd(AB.data_ptr(), 2 * h, ea.data_ptr(), ldea, w1c, W1.stride(0), b1.data_ptr(), dS.data_ptr(),
                                           topo.rowptr.data_ptr(), topo.col.data_ptr(), topo.ent.data_ptr(), dAB.data_ptr(),
                                           slab.data_ptr(), n_slabs, N, h, fe, 0, st), "dss2_edge_combine_bwd")
        _lib.check(L.dss2_edge_combine_bwd(AB.data_ptr(), 2 * h, ea.data_ptr(), ldea, w1c, W1.stride(0), b1.data_ptr(), dS.data_ptr(),
                                           topo.rowptrT.data_ptr(), topo.colT.data_ptr(), topo.entT.data_ptr(), dAB.data_ptr(),
                                           None, n_slabs, N, h, fe, 1, st), "dss2_edge_combine_bwd")
        g1c = torch.empty(stride, dtype=_F32, device=dev)
        _reduce(slab, 0, n_slabs, stride, g1c, stride, None)
        gab = torch.empty(2 * h * d + 2 * h, dtype=_F32, device=dev)          # [2h, d] dW1ab, then column sums (unused)
        wgrad(topo, dAB, 2 * h, x, d, 1, gab)
        dW1 = torch.cat([gab[:h * d].view(h, d), gab[h * d:2 * h * d].view(h, d), g1c[:h * fe].view(h, fe)], dim=1)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(N, d, dtype=_F32, device=dev)
            gemm_prop(topo, dAB, 2 * h, 2 * h, sp.bwd, 1, d, dx)
        return dx, None, None, None, dW1, g1c[h * fe:], g2[:ho * h].view(ho, h), g2[ho * h:]


def _check_general_dims(mod) -> None:
    if mod.dim_feate > 32:
        raise NotImplementedError("EdgeAggregation on HIP: dim_feate <= 32 (the per-edge kernel holds a unit's edge-feature "
                                  "weights in registers: 8 / 16 / 32 wide instantiations)")


class EdgeAggregationGeneral(nn.Module):
    """/root/reference/networks.py:159-209 for any ``dim_featn`` (MultiMPN applies it to the hidden activation) and
    ``dim_feate <= 32``; same parameters and state_dict keys as ``networks.EdgeAggregation``.  ``forward`` takes the graph
    structure of the caller (the Multi* stacks double the graph once for all their layers)."""

    def __init__(self, dim_featn, dim_feate, dim_hid, dim_out):
        super().__init__()
        if dim_feate > 32:
            raise NotImplementedError("EdgeAggregation on HIP: dim_feate <= 32")
        self.dim_featn, self.dim_feate, self.dim_hid, self.dim_out = dim_featn, dim_feate, dim_hid, dim_out
        self.edge_aggr = nn.Sequential(nn.Linear(dim_featn * 2 + dim_feate, dim_hid), nn.ReLU(), nn.Linear(dim_hid, dim_out))
        self._gplan = None
        self._post = None      # set by the enclosing stack: True / dropout rate = "dropout, then ReLU" after this layer

    def run(self, x, edge_attr, topo: Topology):
        lin1, lin2 = self.edge_aggr[0], self.edge_aggr[2]
        return _EdgeAggrGeneralFn.apply(x, edge_attr, topo, self, lin1.weight, lin1.bias, lin2.weight, lin2.bias)

    def forward(self, x, edge_index, edge_attr):
        _require_gpu(x, edge_index, edge_attr)
        return self.run(x, edge_attr, get_topology(edge_index, x.size(0), double=False))


class _TAGConvPostFn(torch.autograd.Function):
    """TAGConv followed (inside a Multi* stack) by dropout + ReLU, fused into the layer's epilogue."""

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
        relu, snap, p = _post_spec(mod, x.device)
        out = (_tagconv_forward_global if glob else _tagconv_forward)(topo, x, plan.fwd[0], bias, nmat, hin, hout, relu=relu,
                               drop=((snap, p, 1) if snap is not None else None))
        ctx.save_for_backward(x, out if relu else None)       # (see _EdgeAggrGeneralFn: an un-gated output may be modified in place)
        ctx.meta = (topo, mod, relu, snap, p, glob)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, y = ctx.saved_tensors
        topo, mod, relu, snap, p, glob = ctx.meta
        plan = mod._plan
        if plan.version != ctx.ver:
            plan.refresh()
        hin, hout, nmat = mod.in_channels, mod.out_channels, mod.K + 1
        g = _gate(gout, y, snap, p) if relu else gout.contiguous()
        flat = torch.empty(nmat * hout * hin + hout, dtype=_F32, device=g.device)
        if plan.stacked != glob:
            raise RuntimeError("the module's weight layouts changed between forward and backward")
        dh = (_tagconv_backward_global if glob else _tagconv_backward)(
            topo, g, x, plan.bwd[0], nmat, hin, hout, flat, need_dh=ctx.needs_input_grad[0])
        gw = [flat[m * hout * hin:(m + 1) * hout * hin].view(hout, hin) for m in range(nmat)]
        return (dh, None, None, flat[nmat * hout * hin:], *gw)


def _run_tagconv(conv: TAGConv, x, topo):
    return _TAGConvPostFn.apply(x, topo, conv, conv.bias, *[l.weight for l in conv.lins])


class _MaskEmbdFn(torch.autograd.Function):
    """x + Linear(dim_hid -> dim_featn)(ReLU(Linear(dim_featn -> dim_hid)(mask)))   (networks.py:418-422,457)"""

    @staticmethod
    def forward(ctx, mask, x, topo, mod, W1, b1, W2, b2):
        mask, ldm = _rows(mask)
        x, ldx = _rows(x)
        dev = x.device
        fn, h = W1.shape[1], W1.shape[0]
        if getattr(mod, "_me_plan", None) is None or mod._me_plan.device != dev:
            mod._me_plan = _PackPlan([[W1], [W2]], dev)
        plan = mod._me_plan
        ctx.ver = plan.refresh()
        topo.lds_check(1, max(_round8(fn), _round8(h)), max(_ncg(h), _ncg(fn)))
        h1 = torch.empty(topo.N, h, dtype=_F32, device=dev)
        gemm_prop(topo, mask, ldm, fn, plan.fwd[0], 1, h, h1, bias=b1, relu=True)
        y = torch.empty(topo.N, fn, dtype=_F32, device=dev)
        gemm_prop(topo, h1, h, h, plan.fwd[1], 1, fn, y, bias=b2, add_src=x, add_ld=ldx)
        ctx.save_for_backward(mask, h1)
        ctx.meta = (topo, mod, fn, h)
        return y

    @staticmethod
    def backward(ctx, g):
        mask, h1 = ctx.saved_tensors
        topo, mod, fn, h = ctx.meta
        plan = mod._me_plan
        if plan.version != ctx.ver:
            plan.refresh()
        g = g.contiguous()
        dev = g.device
        g2 = torch.empty(fn * h + fn, dtype=_F32, device=dev)
        wgrad(topo, g, fn, h1, h, 1, g2)
        dh1 = torch.empty(topo.N, h, dtype=_F32, device=dev)
        gemm_prop(topo, g, fn, fn, plan.bwd[1], 1, h, dh1, relu_src=h1)
        g1 = torch.empty(h * fn + h, dtype=_F32, device=dev)
        wgrad(topo, dh1, h, mask, fn, 1, g1)
        dmask = None
        if ctx.needs_input_grad[0]:          # data.x requires grad: the mask columns receive dh1 W1, the features g itself
            dmask = torch.empty(topo.N, fn, dtype=_F32, device=dev)
            gemm_prop(topo, dh1, h, h, plan.bwd[0], 1, fn, dmask)
        dx = g if ctx.needs_input_grad[1] else None
        return dmask, dx, None, None, g1[:h * fn].view(h, fn), g1[h * fn:], g2[:fn * h].view(fn, h), g2[fn * h:]


def _mask_embd(mod: nn.Module, data_x: torch.Tensor, topo: Topology):
    """networks.py:452-457: columns [4, 4 + fn) are the features, the last fn columns their mask."""
    fn = mod.dim_featn
    if data_x.shape[-1] != fn * 2 + 4:
        raise AssertionError("data.x must hold 4 node-type columns, dim_featn features and their dim_featn mask columns")
    x, mask = data_x[:, 4:4 + fn], data_x[:, -fn:]
    l1, l2 = mod.mask_embd[0], mod.mask_embd[2]
    return _MaskEmbdFn.apply(mask, x, topo, mod, l1.weight, l1.bias, l2.weight, l2.bias)


class _ReferenceHelpers:
    def is_directed(self, edge_index):
        from .topology import reference_is_directed
        if edge_index.shape[1] == 0:
            return False
        return reference_is_directed(edge_index)

    def undirect_graph(self, edge_index, edge_attr):
        """networks.py:432-449: reverse edges appended with the SAME edge_attr (API parity; forward() does not call it)."""
        if self.is_directed(edge_index):
            edge_index = torch.cat([edge_index, torch.stack([edge_index[1, :], edge_index[0, :]], dim=0)], dim=1)
            edge_attr = torch.cat([edge_attr, edge_attr], dim=0)
        return edge_index, edge_attr


class MaskEmbdMPN(MPN, _ReferenceHelpers):
    """/root/reference/networks.py:390-470: mask embedding, then the MPN layer loop on a graph doubled without sign flips;
    runs on MPN's fused block (folded first layer, layer chain, batched weight gradients)."""

    def __init__(self, dim_featn, dim_feate, dim_out, dim_hid, n_gnn_layers, K, dropout_rate):
        if n_gnn_layers < 2:
            # the reference builds TWO dim_hid -> dim_out convs for n_gnn_layers == 1 (networks.py:408-416), which only runs for
            # dim_out == dim_hid (its forward fails on the second conv otherwise): that is a two-conv stack of equal widths
            if dim_out != dim_hid:
                raise ValueError("MaskEmbdMPN(n_gnn_layers=1) is two dim_hid -> dim_out TAGConvs in the reference (networks.py:408-416): "
                                 "its forward only runs for dim_out == dim_hid")
            n_gnn_layers = 2
        super().__init__(dim_featn, dim_feate, dim_out, dim_hid, n_gnn_layers, K, dropout_rate)
        self.mask_embd = nn.Sequential(nn.Linear(dim_featn, dim_hid), nn.ReLU(), nn.Linear(dim_hid, dim_featn))
        self._me_plan = None

    is_directed = _ReferenceHelpers.is_directed
    undirect_graph = _ReferenceHelpers.undirect_graph

    def forward(self, data):
        _require_gpu(data.x, data.edge_index, data.edge_attr)
        topo = get_topology(data.edge_index, data.x.size(0), flip=False)
        x = _mask_embd(self, data.x, topo)
        if not self.edge_aggr.fused_dims():
            return self._forward_general(x, data.edge_attr, topo)
        return _MPNFn.apply(x, data.edge_attr, topo, self, *self._params())


class MultiMPN(nn.Module, _ReferenceHelpers):
    """/root/reference/networks.py:473-549: EdgeAggregation and TAGConv interleaved per layer, dropout + ReLU after every
    layer but the last, the last layer an EdgeAggregation(dim_hid -> dim_out)."""

    mask_embedding = False
    first_edge_aggr = True

    def __init__(self, dim_featn, dim_feate, dim_out, dim_hid, n_gnn_layers, K, dropout_rate):
        super().__init__()
        self.dim_featn, self.dim_feate, self.dim_out, self.dim_hid = dim_featn, dim_feate, dim_out, dim_hid
        self.n_gnn_layers, self.K, self.dropout_rate = n_gnn_layers, K, dropout_rate
        self.layers = nn.ModuleList()
        if self.first_edge_aggr:
            self.layers.append(EdgeAggregationGeneral(dim_featn, dim_feate, dim_hid, dim_hid))
        self.layers.append(TAGConv(dim_hid, dim_out if n_gnn_layers == 1 else dim_hid, K=K))
        for _ in range(n_gnn_layers - 2):
            if self.first_edge_aggr:
                self.layers.append(EdgeAggregationGeneral(dim_hid, dim_feate, dim_hid, dim_hid))
            self.layers.append(TAGConv(dim_hid, dim_hid, K=K))
        self.layers.append(EdgeAggregationGeneral(dim_hid, dim_feate, dim_hid, dim_out))
        if self.mask_embedding:
            self.mask_embd = nn.Sequential(nn.Linear(dim_featn, dim_hid), nn.ReLU(), nn.Linear(dim_hid, dim_featn))
            self._me_plan = None

    def forward(self, data):
        _require_gpu(data.x, data.edge_index, data.edge_attr)
        topo = get_topology(data.edge_index, data.x.size(0), flip=False)
        x = _mask_embd(self, data.x, topo) if self.mask_embedding else data.x
        ea = data.edge_attr
        last = len(self.layers) - 1
        for i, layer in enumerate(self.layers):
            # a fresh nn.Dropout per call => active in eval() as well (networks.py:539); p = 0 is plain ReLU
            layer._post = None if i == last else (float(self.dropout_rate) if self.dropout_rate > 0 else True)
            x = layer.run(x, ea, topo) if isinstance(layer, EdgeAggregationGeneral) else _run_tagconv(layer, x, topo)
        return x


class MaskEmbdMultiMPN(MultiMPN):
    """/root/reference/networks.py:552-644: MultiMPN on mask-embedded features."""
    mask_embedding = True


class MaskEmbdMultiMPN_NoMP(MultiMPN):
    """/root/reference/networks.py:647-735: TAGConv layers only, one EdgeAggregation(dim_hid -> dim_out) at the end.  The first
    TAGConv expects dim_hid input columns, so (as in the reference) it only runs with dim_featn == dim_hid."""
    mask_embedding = True
    first_edge_aggr = False
