"""Raw wrappers of the C ABI (libdss2_hip.so through ctypes) for the hot path: tile GEMM + propagation (single layer and
layer chain), weight gradients and their slab reductions, the standalone scatter-add / gather, propagation hops in global
memory, dropout state.  No autograd, no modules: ``networks.py`` builds the reference's classes on these."""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib
from . import flags as FL
from .topology import Topology


def _static_replay() -> bool:
    """True while the step is being captured into a hipGraph or recorded into a launch plan (graphs.PlannedStep): by-value seeds /
    cached batch constants must not be baked in -- the device-side state is used instead."""
    from .graphs import plan_recording
    return plan_recording() or torch.cuda.is_current_stream_capturing()


_F32 = torch.float32


def _stream(t: torch.Tensor) -> int:
    return _lib.stream_ptr(t.device)


def _require_gpu(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("DSS2 HIP path needs GPU tensors (gfx950); there is no CPU fallback")
        if t.is_floating_point() and t.dtype != _F32:
            raise TypeError("DSS2 HIP path computes in fp32; got " + str(t.dtype))


def _rows(t: torch.Tensor) -> Tuple[torch.Tensor, int]:
    """A 2-D fp32 tensor usable with an explicit leading dimension (column slices of a row-major
    matrix are fine); anything else is made contiguous."""
    if t.dim() != 2:
        raise ValueError("expected a 2-D tensor")
    if t.stride(1) != 1 or (t.size(0) > 1 and t.stride(0) < t.size(1)):
        t = t.contiguous()
    return t, (t.stride(0) if t.size(0) > 1 else t.size(1))


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _round8(k: int) -> int:
    return (k + 7) // 8 * 8


def _round16(k: int) -> int:
    return (k + 15) // 16 * 16


def _ncg(j: int) -> int:
    return (j + 31) // 32


# ------------------------------------------------------------------------------------------
# thin op wrappers over the C ABI
# ------------------------------------------------------------------------------------------
def gemm_prop(topo: Topology, X: torch.Tensor, ldx: int, kreal: int, Bp: torch.Tensor, nmat: int, hout: int,
              Y: torch.Tensor, bias=None, rowscale=None, relu_src=None, dmask=None, add_src=None, add_ld=0,
              relu: bool = False, transposed: bool = False, prop_in: int = 0, narrow_h: int = 0,
              prebias=None, pre_rowscale=None, drop=None, b_format: int = 0) -> None:
    """drop = (snapshot, p, drop_id): in-kernel dropout mask of layer drop_id (see dropout_snapshot).
    b_format = 2: Bp holds the f16x2 group buffers of _PackPlan(f16_groups=...) and the tile GEMM runs as f16x3 (chain_f16_supported);
    b_format = 1: Bp holds bf16x3 fragments (_PackPlan.fwd16 / bwd16) and the tile GEMM runs as bf16x6 -- the tall-tile
    shapes of gemm16_supported only."""
    if topo.global_only and (nmat > 1 or prop_in > 0):
        raise NotImplementedError(f"largest connected component has {topo.max_segment} nodes: the fused GEMM + propagation "
                                  "kernels hold a whole graph in LDS (<= 192 nodes); use the MPN / TAGConv modules, which "
                                  "switch to the global-memory propagation path")
    a = _lib.GemmPropArgs()
    if drop is not None and drop[2] > 0:
        a.drop_state, a.drop_id = drop[0].data_ptr(), int(drop[2])
        a.drop_thr, a.drop_scale = _dropout_params(drop[1])
    a.prebias, a.pre_rowscale = _ptr(prebias), _ptr(pre_rowscale)
    a.prop_in, a.narrow_h = prop_in, narrow_h
    a.X, a.ldx, a.kreal, a.kpad = X.data_ptr(), ldx, kreal, (_round16(kreal) if b_format == 1 else _round8(kreal))
    a.b_format = b_format
    a.Bp, a.bias, a.rowscale = Bp.data_ptr(), _ptr(bias), _ptr(rowscale)
    a.relu_src, a.ld_relu = _ptr(relu_src), (relu_src.stride(0) if relu_src is not None else 0)
    a.dmask, a.ld_dmask = _ptr(dmask), (dmask.stride(0) if dmask is not None else 0)
    a.add_src, a.ld_add = _ptr(add_src), add_ld
    a.Y, a.ldy, a.hout, a.ncg = Y.data_ptr(), Y.stride(0), hout, (1 if narrow_h else _ncg(hout))
    a.relu, a.nmat, a.nrb, a.ntiles = int(relu), nmat, topo.nrb, topo.ntiles
    a.tile_start = topo.tile_start.data_ptr()
    if transposed:
        a.rowptr, a.col, a.w, a.max_nnz = topo.rowptrT.data_ptr(), topo.colT.data_ptr(), topo.wT.data_ptr(), topo.max_nnzT
        a.ell_width, a.ell_tiles = topo.ellT, _ptr(topo.ellT_tiles)
    else:
        a.rowptr, a.col, a.w, a.max_nnz = topo.rowptr.data_ptr(), topo.col.data_ptr(), topo.w.data_ptr(), topo.max_nnz
        a.ell_width, a.ell_tiles = topo.ell, _ptr(topo.ell_tiles)
    _lib.check(_lib.lib().dss2_gemm_prop(C.byref(a), _stream(Y)), "dss2_gemm_prop")




def chain_supported(topo: Topology, nmat: int, hid: int, transposed: bool, have16: bool = False) -> bool:
    """True when n >= 2 consecutive hid -> hid layers can run as one chained launch (dss2_gemm_prop_chain).  ``have16``: the
    caller holds bf16x6 weight packs, so shapes that only the split-plane form covers (192-row tiles) count too."""
    ell, tiles = (topo.ellT, topo.ellT_tiles) if transposed else (topo.ell, topo.ell_tiles)
    if not FL.CHAIN_LAYERS or tiles is None:
        return False
    return bool(_lib.lib().dss2_gemm_prop_chain_supported(topo.nrb, nmat, hid, hid, ell)) or (
        have16 and chain16_supported(topo, nmat, hid, transposed))


def _single_group_tall_veto(topo: Topology, hid: int) -> bool:
    """96- / 192-row tiles with ONE column group (hid <= 32): the split-plane chain runs them as single-wave workgroups, which pays from
    dss2_chain_sp6_single_group_min_tiles() tiles on (the library applies the same count at launch); below it the block keeps the
    multi-wave chain of that shape with its bf16x3 weights and fp32 gates, i.e. every sp6-only capability is declined here."""
    if hid > 32 or topo.nrb not in (3, 6):
        return False
    m = int(_lib.lib().dss2_chain_sp6_single_group_min_tiles())
    return m < 0 or topo.ntiles < m


def chain_gate_words(topo: Topology, nmat: int, hid: int) -> int:
    """64-bit words per tile of a layer's sign-bit buffer (``y_bits`` of a forward chain -> ``gate_bits`` of the data-gradient
    chain over the same tiles); 0 where the chain kernel of this shape has no bit form (or either direction is not chained)."""
    if not (FL.CHAIN_LAYERS and FL.CHAIN_BF16 and FL.CHAIN_GATE_BITS) or topo.ell_tiles is None or topo.ellT_tiles is None:
        return 0
    if _single_group_tall_veto(topo, hid):
        return 0
    cache = topo.__dict__.setdefault("_gate_words", {})      # (asked once per forward: keep the two library calls off the step)
    gw = cache.get((nmat, hid))
    if gw is None:
        L = _lib.lib()
        gw = cache[(nmat, hid)] = min(int(L.dss2_gemm_prop_chain_gate_words(topo.nrb, nmat, hid, hid, topo.ell)),
                                      int(L.dss2_gemm_prop_chain_gate_words(topo.nrb, nmat, hid, hid, topo.ellT)))
    return gw


def chain16_supported(topo: Topology, nmat: int, hid: int, transposed: bool) -> bool:
    """True when the chain can run its tile GEMM on the bf16 matrix pipe (bf16x6, fp32-accurate; dss2_gemm_chain16.hip)."""
    ell, tiles = (topo.ellT, topo.ellT_tiles) if transposed else (topo.ell, topo.ell_tiles)
    return FL.CHAIN_BF16 and tiles is not None and bool(_lib.lib().dss2_gemm_prop_chain16_supported(topo.nrb, nmat, hid, hid, ell))


def chain_f16_supported(topo: Topology, nmat: int, hid: int) -> bool:
    """True when BOTH chains of a block (forward, data gradients) can run their tile GEMM as f16x3 (b_format 2: weights as two fp16
    planes with scale exponents, _PackPlan f16_groups): the split-plane chain of 64-row tiles with bit-word ReLU gates
    (csrc/dss2_gemm_chain_sp.hip, MS = 2).  flags.CHAIN_F16 = False: bf16x6."""
    if not (FL.CHAIN_F16 and FL.CHAIN_BF16 and topo.ell_tiles is not None and topo.ellT_tiles is not None and chain_gate_words(topo, nmat, hid) > 0):
        return False
    L = _lib.lib()
    return bool(L.dss2_gemm_prop_chain_f16_supported(topo.nrb, nmat, hid, hid, topo.ell)) and bool(L.dss2_gemm_prop_chain_f16_supported(topo.nrb, nmat, hid, hid, topo.ellT))


def chain_head_supported(topo: Topology, nmat: int, hid: int, nout: int, transposed: bool) -> bool:
    """True when the narrow head TAGConv (hid -> nout) can ride inside the chained launch of the hid -> hid layers
    (dss2_gemm_prop_chain_head: forward = the head after the last chained layer, transposed = the chain's input computed from
    the head's upstream gradient); DSS2_CHAIN_HEAD=0 switches it off."""
    ell, tiles = (topo.ellT, topo.ellT_tiles) if transposed else (topo.ell, topo.ell_tiles)
    if _single_group_tall_veto(topo, hid):
        return False
    return bool(FL.CHAIN_HEAD) and FL.CHAIN_BF16 and tiles is not None and bool(      # (a mask of modes: bit 0 forward, bit 1 backward)
        _lib.lib().dss2_gemm_prop_chain_head_supported(topo.nrb, nmat, hid, hid, ell, nout) & (2 if transposed else 1))


def chain_head_wgrad_supported(topo: Topology, nmat: int, hid: int, nout: int) -> bool:
    """True when the data-gradient chain with the fused head (mode 2) can also form the head's weight gradient in its staging
    (dss2_chain_head.wg_slab, round 5): 64-, 96- and 192-row tiles, nout <= 2.  flags.CHAIN_HEAD_WGRAD = False: the narrow weight-gradient launch."""
    return bool(FL.CHAIN_HEAD_WGRAD) and topo.ellT_tiles is not None and bool(
        _lib.lib().dss2_gemm_prop_chain_head_wgrad_supported(topo.nrb, nmat, hid, hid, topo.ellT, nout))


def gemm16_supported(topo: Topology, nmat: int, hid: int, transposed: bool) -> bool:
    """True when a single hid -> hid layer (dss2_gemm_prop) can take bf16x3 weights: the tall tiles (128 / 192 rows) that
    run matrix-sequentially with K-halved staging and therefore have no layer chain."""
    ell, tiles, nnz = (topo.ellT, topo.ellT_tiles, topo.max_nnzT) if transposed else (topo.ell, topo.ell_tiles, topo.max_nnz)
    return FL.CHAIN_BF16 and tiles is not None and bool(_lib.lib().dss2_gemm_prop16_supported(topo.nrb, nmat, hid, hid, nnz, ell))


def gemm_prop_chain(topo: Topology, X: Optional[torch.Tensor], hid: int, nmat: int, layers: Sequence[dict], transposed: bool = False,
                    pre_rowscale=None, drop=None, b_format: int = 0, head: Optional[dict] = None) -> None:
    """layers: dicts with Bp, Y and optionally bias, relu, relu_src, dmask, prebias; every tensor is [N, hid]
    contiguous.  Layer i reads layer i-1's output from LDS; every Y is written once.  The library chains at most
    FL.CHAIN_MAX layers per launch; deeper stacks run as consecutive launches (the next one reads the previous one's last Y)."""
    if head is not None and len(layers) > FL.CHAIN_MAX:
        raise ValueError("gemm_prop_chain: a fused head needs the whole chain in one launch")
    if len(layers) > FL.CHAIN_MAX:
        for c0 in range(0, len(layers), FL.CHAIN_MAX):
            gemm_prop_chain(topo, X if c0 == 0 else layers[c0 - 1]["Y"], hid, nmat, layers[c0:c0 + FL.CHAIN_MAX],
                            transposed=transposed, pre_rowscale=pre_rowscale, drop=drop, b_format=b_format)
        return
    a = _lib.GemmPropArgs()
    a.b_format = b_format          # 1: every layer's Bp holds bf16x3 fragments (_PackPlan.fwd16 / bwd16)
    if drop is not None:            # (snapshot, p); the layers name their masks with "drop_id"
        a.drop_state = drop[0].data_ptr()
        a.drop_thr, a.drop_scale = _dropout_params(drop[1])
    dev_t = X if X is not None else layers[-1]["Y"]
    a.X, a.ldx = (X.data_ptr(), X.stride(0)) if X is not None else (0, hid)
    a.kreal, a.kpad = hid, (_round16(hid) if b_format >= 1 else _round8(hid))
    a.hout, a.ncg, a.ldy, a.ld_relu, a.ld_dmask, a.ld_add = hid, _ncg(hid), hid, hid, hid, hid
    a.nmat, a.nrb, a.ntiles = nmat, topo.nrb, topo.ntiles
    a.max_tile_rows = int(getattr(topo, "max_tile_rows", 0) or 0)
    a.tile_start = topo.tile_start.data_ptr()
    a.pre_rowscale = _ptr(pre_rowscale)
    if transposed:
        a.rowptr, a.col, a.w, a.max_nnz = topo.rowptrT.data_ptr(), topo.colT.data_ptr(), topo.wT.data_ptr(), topo.max_nnzT
        a.ell_width, a.ell_tiles = topo.ellT, _ptr(topo.ellT_tiles)
    else:
        a.rowptr, a.col, a.w, a.max_nnz = topo.rowptr.data_ptr(), topo.col.data_ptr(), topo.w.data_ptr(), topo.max_nnz
        a.ell_width, a.ell_tiles = topo.ell, _ptr(topo.ell_tiles)
    tab = (_lib.ChainLayer * len(layers))()
    for d, ly in zip(tab, layers):
        for t_ in (ly.get("Y"), ly.get("relu_src"), ly.get("dmask")):
            if t_ is not None and (t_.stride(0) != hid or t_.stride(1) != 1):
                raise ValueError("gemm_prop_chain: [N, hid] contiguous tensors expected")
        d.Bp, d.Y, d.bias = ly["Bp"].data_ptr(), _ptr(ly.get("Y")), _ptr(ly.get("bias"))
        d.relu_src, d.dmask, d.prebias = _ptr(ly.get("relu_src")), _ptr(ly.get("dmask")), _ptr(ly.get("prebias"))
        d.gate_bits, d.y_bits = _ptr(ly.get("gate_bits")), _ptr(ly.get("y_bits"))      # (only where chain_gate_words() > 0)
        d.relu = int(bool(ly.get("relu", False)))
        d.drop_id = int(ly.get("drop_id", 0)) if drop is not None else 0
    if head is None:
        _lib.check(_lib.lib().dss2_gemm_prop_chain(C.byref(a), C.addressof(tab), len(layers), _stream(dev_t)), "dss2_gemm_prop_chain")
        return
    # head: dict(W=[W_0..W_K] ([nout, hid] contiguous), nout, and forward: Y, bias, add_src / backward: G, gate, Xout, drop_id)
    hd = _lib.ChainHead()
    for m, w_ in enumerate(head["W"]):
        if w_.stride(0) != hid or w_.stride(1) != 1:
            raise ValueError("gemm_prop_chain: head weights [nout, hid] contiguous expected")
        hd.W[m] = w_.data_ptr()
    hd.nout = int(head["nout"])
    if not transposed:
        hd.mode = 1
        y_ = head["Y"]
        hd.Y, hd.ldy, hd.bias = y_.data_ptr(), y_.stride(0), _ptr(head.get("bias"))
        add_ = head.get("add_src")
        if add_ is not None:
            hd.add_src, hd.ld_add = add_.data_ptr(), int(head["add_ld"])
    else:
        hd.mode = 2
        g_, xo_ = head["G"], head["Xout"]
        hd.G, hd.ldg, hd.Xout, hd.ldxo = g_.data_ptr(), g_.stride(0), xo_.data_ptr(), xo_.stride(0)
        gate_ = head.get("gate")
        if gate_ is not None:
            hd.gate, hd.ld_gate = gate_.data_ptr(), gate_.stride(0)
        hd.drop_id = int(head.get("drop_id", 0)) if drop is not None else 0
        hd.wg_slab = _ptr(head.get("wg_slab"))      # (optional: the head's weight gradient, one slab per tile; chain_head_wgrad_supported)
        hd.pad = int(head.get("wg_stride", 0))      # (stride of those slabs in floats)
    _lib.check(_lib.lib().dss2_gemm_prop_chain_head(C.byref(a), C.addressof(tab), len(layers), C.byref(hd), _stream(dev_t)),
               "dss2_gemm_prop_chain_head")


def _wgrad_tiles(topo: Topology, nmat: int, hout: int, hin: int, b16: int):
    """The tile set a weight-gradient launch walks: the topology's own, or -- 64-row tilings under the bf16x6 kernel -- a
    32-row tiling of the same graphs (Topology.tiles_for(1)), on which two 4-wave workgroups share a CU and overlap each
    other's staging / propagation with their MFMA phases (csrc/dss2_wgrad16.hip)."""
    # (each X tile is staged by the two workgroups that own its 64-column output halves, at unrelated times: beyond the
    #  Infinity Cache that is a second trip to HBM.  As bf16x6 that lost to the 64-row kernel -- B = 32768: 3.52 ms against 3.43 --, so
    #  that route takes the 32-row form only while one layer's input stays well inside the cache; as f16x3 (round 5) the 32-row kernel
    #  wins at every size: B = 32768 2.75 -> 2.45 ms, B = 16384 1.44 -> 1.28 ms)
    if (FL.WGRAD_TM32 and b16 and topo.nrb == 2 and not topo.global_only and nmat in (2, 3) and hout > 32 and 1 <= topo.ellT <= 8
            and (FL.WGRAD_F16 or topo.N * hin * 4 <= FL.WGRAD_TM32_MAX_BYTES)):
        alt = topo.tiles_for(1)
        # ... and only where the library's bf16x6 kernel covers the shape on that tiling (its LDS query answers with the fp32 kernel's
        # size when it does not: hout % 4, hin % 4, hout > 32, K <= 2 -- the conditions live in ONE place, wgrad16_covers)
        if alt is not None and alt.ellT_tiles is not None and 1 <= alt.ellT <= 8:
            L_ = _lib.lib()
            if (L_.dss2_wgrad_lds_bytes_ex(1, nmat, hout, hin, alt.max_nnzT, alt.ellT, 1)
                    != L_.dss2_wgrad_lds_bytes_ex(1, nmat, hout, hin, alt.max_nnzT, alt.ellT, 0)):
                return alt
    return topo


def _wgrad_mode(ts, nmat: int, b16: int, hinted: bool = False) -> int:
    """args.mfma_bf16 of a weight-gradient launch on the tile set ``ts``: 0 fp32 MFMA, 1 bf16x6, or -- flags.WGRAD_F16 on 32-row tiles
    (csrc/dss2_wgrad16h.hip) and on 96- .. 192-row tiles (csrc/dss2_wgrad16th.hip) with ELL slices -- 2 | hb << 8: the f16x3 kernels
    with hb headroom bits for the gain of the propagation hops, ceil(log2(max row sum of |P^T| ^ K)), read from the ELL slices once per
    tile set (one device-to-host copy, cached; ``hinted``: from the ELL width alone, no copy).  Shapes the f16x3 kernels do not cover run bf16x6 on the same value (the library decides)."""
    if not (b16 and FL.WGRAD_F16 and ts.nrb in (1, 3, 4, 5, 6) and nmat in (2, 3) and ts.ellT_tiles is not None and 1 <= ts.ellT <= 8):
        return b16
    cache = ts.__dict__.setdefault("_f16_gain_bits", {})
    hb = cache.get(nmat)
    if hb is None:
        if hinted or torch.cuda.is_current_stream_capturing():
            # no copy to the host here (a topology built from a TopologyHint never reads anything back; neither may a capture): the
            # bound that needs no data -- gcn_norm weights are <= 1, a row of P^T has at most ellT entries -- costs the smallest
            # elements a bit or two of their 22 and is kept for the life of the tile set (same bits every step)
            gain = float(max(int(ts.ellT), 1))
        else:
            w = ts.ellT_tiles[..., 1].contiguous().view(torch.float32)      # [tiles][width][rows]: the entries' weights
            gain = max(float(w.abs().sum(dim=1).max()), 1.0)
        hb = cache[nmat] = max(0, int(math.ceil(math.log2(gain) * (nmat - 1) - 1e-6)))
    return (2 | (hb << 8)) if hb <= 10 else b16


def wgrad(topo: Topology, G: torch.Tensor, hout: int, X: torch.Tensor, hin: int, nmat: int, out_flat: torch.Tensor,
          rowscale=None, rowscale2=None, pending=None, out_len: Optional[int] = None) -> None:
    """out_flat[nmat*hout*hin + hout] <- [dW_0 .. dW_{nmat-1}, db] (deterministic two-pass sum); with
    rowscale2 additionally [nmat*hout] scaled column sums of P^m G (one block per matrix).  ``out_len``: reduce only
    the first out_len elements of the result."""
    if topo.global_only and nmat > 1:
        raise NotImplementedError("wgrad with propagation needs LDS-resident graph tiles (graphs of <= 192 nodes)")
    narrow = nmat > 1 and nmat * hout <= 32 and rowscale2 is None
    b16 = int(FL.WGRAD_BF16 and rowscale is None and not narrow)
    ts = _wgrad_tiles(topo, nmat, hout, hin, b16)
    lds = _lib.lib().dss2_wgrad_lds_bytes_ex(ts.nrb, nmat, hout, hin, ts.max_nnzT, ts.ellT, b16)
    per_cu = _wgrad_per_cu(int(lds))
    ys = _lib.lib().dss2_wgrad_y_slices(ts.nrb, nmat, hout, hin, ts.ellT, b16, int(rowscale2 is not None))
    n_split = min(ts.ntiles, max(1, (256 * per_cu) // ys))
    stride = nmat * hout * hin + hout + (nmat * hout if rowscale2 is not None else 0)
    slab = torch.empty(n_split * stride, dtype=_F32, device=G.device)
    a = _lib.WgradArgs()
    a.G, a.ldg, a.hout = G.data_ptr(), G.stride(0), hout
    a.X, a.ldx, a.hin = X.data_ptr(), X.stride(0), hin
    a.rowscale, a.rowscale2 = _ptr(rowscale), _ptr(rowscale2)
    a.slab, a.n_split, a.nmat, a.nrb, a.ntiles = slab.data_ptr(), n_split, nmat, ts.nrb, ts.ntiles
    a.tile_start = ts.tile_start.data_ptr()
    a.rowptrT, a.colT, a.wT, a.max_nnz = topo.rowptrT.data_ptr(), topo.colT.data_ptr(), topo.wT.data_ptr(), ts.max_nnzT
    a.ell_width, a.ell_tiles = ts.ellT, _ptr(ts.ellT_tiles)
    a.narrow, a.mfma_bf16 = int(narrow), (_wgrad_mode(ts, nmat, b16, getattr(topo, "hint", None) is not None) if (not narrow and rowscale is None) else b16)
    st = _stream(G)
    _lib.check(_lib.lib().dss2_wgrad(C.byref(a), st), "dss2_wgrad")
    _reduce(slab, 0, n_split, stride, out_flat, stride if out_len is None else out_len, pending)


def wgrad_batched(topo: Topology, Gs: Sequence[torch.Tensor], hout: int, Xs: Sequence[torch.Tensor], hin: int, nmat: int,
                  out_flat: torch.Tensor, first_rowscale2=None, first_out=None, pending=None) -> None:
    """len(Gs) layers of identical shape in one launch:
    out_flat[j * (nmat*hout*hin + hout) + ...] <- [dW_0 .. dW_{nmat-1}, db] of the plain layers, in order.
    With ``first_rowscale2`` layer 0 is a folded layer (see ``wgrad``): its result, with the extra nmat*hout
    scaled sums, goes to ``first_out`` and the remaining layers to ``out_flat`` (two slab reductions)."""
    nl = len(Gs)
    ts = _wgrad_tiles(topo, nmat, hout, hin, int(FL.WGRAD_BF16))
    lds = _lib.lib().dss2_wgrad_lds_bytes_ex(ts.nrb, nmat, hout, hin, ts.max_nnzT, ts.ellT, int(FL.WGRAD_BF16))
    per_cu = _wgrad_per_cu(int(lds))
    ys = _lib.lib().dss2_wgrad_y_slices(ts.nrb, nmat, hout, hin, ts.ellT, int(FL.WGRAD_BF16), int(first_rowscale2 is not None))
    mode = _wgrad_mode(ts, nmat, int(FL.WGRAD_BF16), getattr(topo, "hint", None) is not None)
    nz = int(_lib.lib().dss2_wgrad_batched_groups(ts.nrb, hout, hin, mode, nl))      # workgroup groups along z (two 32-column layers may share one)
    n_split = min(ts.ntiles, max(1, (256 * per_cu) // (nz * ys)))       # the layers share the chip
    stride = nmat * hout * hin + hout
    lens = [stride + (nmat * hout if (first_rowscale2 is not None and l == 0) else 0) for l in range(nl)]
    total = sum(lens)
    slab = torch.empty(n_split * total, dtype=_F32, device=Gs[0].device)
    a = _lib.WgradArgs()
    a.ldg, a.hout, a.ldx, a.hin = Gs[0].stride(0), hout, Xs[0].stride(0), hin
    for g_, x_ in zip(Gs, Xs):
        if g_.stride(0) != a.ldg or x_.stride(0) != a.ldx or g_.shape != Gs[0].shape or x_.shape != Xs[0].shape:
            raise ValueError("wgrad_batched: layers must share shapes and leading dimensions")
    a.n_split, a.nmat, a.nrb, a.ntiles = n_split, nmat, ts.nrb, ts.ntiles
    a.mfma_bf16 = mode
    a.tile_start = ts.tile_start.data_ptr()
    a.rowptrT, a.colT, a.wT, a.max_nnz = topo.rowptrT.data_ptr(), topo.colT.data_ptr(), topo.wT.data_ptr(), ts.max_nnzT
    a.ell_width, a.ell_tiles = ts.ellT, _ptr(ts.ellT_tiles)
    PtrArr = C.c_void_p * nl
    gs, xs = PtrArr(*[g_.data_ptr() for g_ in Gs]), PtrArr(*[x_.data_ptr() for x_ in Xs])
    offs = [sum(lens[:l]) for l in range(nl)]
    sl = PtrArr(*[slab.data_ptr() + 4 * o for o in offs])
    rs = PtrArr(*[(first_rowscale2.data_ptr() if (first_rowscale2 is not None and l == 0) else None) for l in range(nl)])
    st = _stream(Gs[0])
    L_ = _lib.lib()
    _lib.check(L_.dss2_wgrad_batched(C.byref(a), gs, xs, sl, rs, total, nl, st), "dss2_wgrad_batched")
    if first_rowscale2 is not None:
        _reduce(slab, 0, n_split, total, first_out, lens[0], pending)
        if nl > 1:
            _reduce(slab, lens[0], n_split, total, out_flat, total - lens[0], pending)
    else:
        _reduce(slab, 0, n_split, total, out_flat, total, pending)


def _reduce(slab: torch.Tensor, slab_off: int, n_slabs: int, stride: int, out: torch.Tensor, length: int, pending) -> None:
    """out[:length] <- fixed-order sum of the slabs; with ``pending`` (a list) the reduction is only recorded, and
    ``reduce_pending`` later runs all recorded ones in one launch (the slab tensors are kept alive by the list)."""
    if pending is not None:
        pending.append((slab, slab.data_ptr() + 4 * slab_off, n_slabs, stride, out, length))
        return
    _lib.check(_lib.lib().dss2_reduce_slabs(slab.data_ptr() + 4 * slab_off, n_slabs, stride, out.data_ptr(), length,
                                            _stream(slab)), "dss2_reduce_slabs")


def _reduce_descs(chunk):
    descs = (_lib.ReduceDesc * len(chunk))()
    for d, (slab, ptr, n_slabs, stride, out, length) in zip(descs, chunk):
        d.slab, d.out, d.stride, d.len, d.n_slabs = ptr, out.data_ptr(), stride, length, n_slabs
    return descs


def reduce_pending(pending) -> None:
    for c0 in range(0, len(pending), 32):
        chunk = pending[c0:c0 + 32]
        descs = _reduce_descs(chunk)
        _lib.check(_lib.lib().dss2_reduce_slabs_multi(C.addressof(descs), len(chunk), _stream(chunk[0][0])),
                   "dss2_reduce_slabs_multi")
    pending.clear()


def prep_weights(fold_tab, pack_tab, device) -> None:
    """Start of a step in weight space: the fold of the edge MLP's second Linear (``fold_tab``: small-GEMM table or None), then the
    packing of every weight (``pack_tab`` = (device table, descriptors, max elements, leading descriptors that read folded weights)):
    two launches (one launch with in-kernel hand-offs measured slower: HISTORY round 5)."""
    t, cnt, mx, _n_dep = pack_tab
    st = _lib.stream_ptr(device)
    if fold_tab is not None:
        ft, fcnt, fmx = fold_tab
        _lib.check(_lib.lib().dss2_small_gemm(ft.data_ptr(), fcnt, fmx, None, st), "dss2_small_gemm")
    _lib.check(_lib.lib().dss2_pack_weights(t.data_ptr(), cnt, mx, st), "dss2_pack_weights")


def finish_weights(pending, rule_tab, base: Optional[torch.Tensor], device) -> None:
    """End of a step in weight space: the recorded slab reductions in one launch and, with ``rule_tab`` (small-GEMM table writing
    into ``base``), the chain rule of the fold behind them."""
    if pending:
        reduce_pending(pending)
    if rule_tab is None:
        return
    rt, rcnt, rmx = rule_tab
    _lib.check(_lib.lib().dss2_small_gemm(rt.data_ptr(), rcnt, rmx, base.data_ptr(), _lib.stream_ptr(device)), "dss2_small_gemm")


def segment_sum(msg: torch.Tensor, rowptr: torch.Tensor, ent: torch.Tensor, n_rows: int) -> torch.Tensor:
    """K6: out[i] = sum of msg rows listed in CSR row i (the scatter-add of aggr='add')."""
    _require_gpu(msg)
    msg, ldm = _rows(msg)
    if ent.dtype != torch.int32:
        ent = ent.to(torch.int32)
    out = torch.empty(n_rows, msg.size(1), dtype=_F32, device=msg.device)
    _lib.check(_lib.lib().dss2_segment_sum(msg.data_ptr(), ldm, rowptr.data_ptr(), ent.data_ptr(), out.data_ptr(),
                                           out.stride(0), n_rows, msg.size(1), _stream(msg)), "dss2_segment_sum")
    return out



def _wgrad_per_cu(lds: int) -> int:
    """Persistent weight-gradient workgroups per CU: what LDS allows, capped at 2 -- every extra workgroup is another
    slab to write and reduce (measured at H = 32, where LDS would allow 4: caps 1 / 2 / 3 / 4 -> 2.73 / 2.15 / 2.32 /
    2.27 ms per SkipPFN step)."""
    return max(1, min(FL.WGRAD_PER_CU, (160 * 1024) // max(lds, 1)))


def is_narrow(nmat: int, hout: int) -> bool:
    """Layers whose nmat*hout output columns fit one 32-wide MFMA block use the packed layouts:
    forward = matrices side by side in one column group (output-side Horner across column blocks),
    data-gradient = matrices stacked along k (input-side propagation)."""
    return nmat > 1 and nmat * hout <= 32


def csr_axpy(topo: Topology, T: torch.Tensor, out: torch.Tensor, h: int, add=None, transposed: bool = False, bias=None,
             relu: bool = False, relu_src=None, add_src=None, add_ld: int = 0, drop=None) -> None:
    """out[:, :h] = epi(add[:, :h] + A_hat T[:, :h]) -- one propagation hop in global memory (dss2_csr_axpy); T, add and
    out may be column blocks of wider row-major buffers (their stride(0) is the leading dimension)."""
    a = _lib.CsrAxpyArgs()
    if transposed:
        a.rowptr, a.col, a.w = topo.rowptrT.data_ptr(), topo.colT.data_ptr(), topo.wT.data_ptr()
    else:
        a.rowptr, a.col, a.w = topo.rowptr.data_ptr(), topo.col.data_ptr(), topo.w.data_ptr()
    a.T, a.ldt, a.out, a.ldo = T.data_ptr(), T.stride(0), out.data_ptr(), out.stride(0)
    a.add, a.ld_add = _ptr(add), (add.stride(0) if add is not None else 0)
    a.bias, a.relu = _ptr(bias), int(relu)
    a.relu_src, a.ld_relu = _ptr(relu_src), (relu_src.stride(0) if relu_src is not None else 0)
    a.add_src, a.ld_src = _ptr(add_src), add_ld
    if drop is not None and drop[2] > 0:
        a.drop_state, a.drop_id = drop[0].data_ptr(), int(drop[2])
        a.drop_thr, a.drop_scale = _dropout_params(drop[1])
    a.n_rows, a.h = topo.N, h
    _lib.check(_lib.lib().dss2_csr_axpy(C.byref(a), _stream(out)), "dss2_csr_axpy")


_DROP_PARAMS = {}


def _dropout_params(p: float):
    """(threshold, scale) of a dropout rate: keep iff Philox uint32 >= threshold (dss2_dropout_params)."""
    v = _DROP_PARAMS.get(p)
    if v is None:
        thr, sc = C.c_uint32(), C.c_float()
        _lib.lib().dss2_dropout_params(float(p), C.byref(thr), C.byref(sc))
        v = _DROP_PARAMS[p] = (thr.value, sc.value)
    return v


def dropout_snapshot(mod: nn.Module, device) -> torch.Tensor:
    """The {seed, offset} pair (device int64[2]) the kernels of ONE forward call and its backward read to regenerate
    their dropout masks (nn.Dropout, networks.py:268, without storing [N, H] masks).  Eager: the seed is drawn from
    torch's CPU generator (as nn.Dropout consumes torch's generator in the reference), so ``torch.manual_seed``
    reproduces a run.  Inside a hipGraph capture a by-value seed would be frozen into the graph, so the module's
    device-side state is used and advanced by the captured kernel itself: every replay sees fresh masks."""
    host_seed = int(torch.empty((), dtype=torch.int64).random_().item())       # exactly ONE generator draw per forward call
    st = getattr(mod, "_rng_state", None)
    if st is None or st.device != device:
        st = mod._rng_state = torch.tensor([host_seed ^ 0x5DEECE66D, 0], dtype=torch.int64).to(device)
    snap = torch.empty(2, dtype=torch.int64, device=device)
    capturing = _static_replay()
    _lib.check(_lib.lib().dss2_rng_next(st.data_ptr(), snap.data_ptr(), host_seed, int(not capturing),
                                        _lib.stream_ptr(device)), "dss2_rng_next")
    return snap


def dropout_mask(snapshot: torch.Tensor, p: float, drop_id: int, n_rows: int, h: int) -> torch.Tensor:
    """The [n_rows, h] multipliers (0 or 1/(1-p)) the kernels apply for layer mask ``drop_id`` of that forward call
    (dss2_dropout_mask): lets a test hand the very same masks to the CPU oracle."""
    out = torch.empty(n_rows, h, dtype=_F32, device=snapshot.device)
    _lib.check(_lib.lib().dss2_dropout_mask(snapshot.data_ptr(), drop_id, float(p), n_rows, h, out.data_ptr(), h,
                                            _lib.stream_ptr(snapshot.device)), "dss2_dropout_mask")
    return out


def gather_rows(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """out[r] = src[idx[r]] (idx: int32 device tensor): dss2_gather_rows."""
    src, lds = _rows(src)
    out = torch.empty(idx.numel(), src.size(1), dtype=_F32, device=src.device)
    _lib.check(_lib.lib().dss2_gather_rows(src.data_ptr(), lds, idx.data_ptr(), out.data_ptr(), out.stride(0) if out.size(0) > 1 else out.size(1),
                                           idx.numel(), src.size(1), _stream(src)), "dss2_gather_rows")
    return out
