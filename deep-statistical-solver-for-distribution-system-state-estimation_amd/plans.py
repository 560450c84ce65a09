"""Weight-space plans of an MPN block: packing of the weights into MFMA fragment order (``_PackPlan``), the edge MLP's second
Linear folded into the first TAGConv (``_FoldPlan``) and the small-GEMM descriptor tables they launch."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from . import flags as FL
from .ops import _F32, _ncg, _round8, _round16, is_narrow, prep_weights

# ------------------------------------------------------------------------------------------
# weight packing plans
# ------------------------------------------------------------------------------------------
_DESC_DTYPE = np.dtype([("src", "<u8"), ("dst", "<u8"), ("rows", "<i4"), ("cols", "<i4"), ("ld", "<i4"),
                        ("transpose", "<i4"), ("koff", "<i4"), ("kpad", "<i4"), ("ncg", "<i4"), ("joff", "<i4")])


_SG_DTYPE = np.dtype([("A", "<u8", (4,)), ("B", "<u8", (4,)), ("C", "<u8"), ("u", "<u8"), ("v", "<u8"), ("c_off", "<i8"),
                      ("M", "<i4"), ("N", "<i4"), ("K", "<i4"), ("lda", "<i4"), ("ldb", "<i4"), ("ldc", "<i4"),
                      ("transA", "<i4"), ("transB", "<i4"), ("nbatch", "<i4"), ("accumulate", "<i4")])


def _sg(A, B, M, N, K, lda, ldb, ldc, C=0, c_off=-1, tA=0, tB=0, u=0, v=0):
    A = list(A) + [0] * (4 - len(A))
    B = list(B) + [0] * (4 - len(B))
    nb = sum(1 for a in A if a)
    return (A, B, C, u, v, c_off, M, N, K, lda, ldb, ldc, tA, tB, nb, 0)


def _sg_table(recs, device):
    """(device table, record count, largest output in 32x32 blocks) of a list of small-GEMM records."""
    arr = np.zeros(len(recs), dtype=_SG_DTYPE)
    for i, r in enumerate(recs):
        arr[i] = tuple(r)
    return (torch.from_numpy(arr.view(np.uint8).copy()).to(device), len(recs),
            max(((r[6] + 31) // 32) * ((r[7] + 31) // 32) for r in recs))


def _pack_table(recs, device):
    """(device table, descriptors, leading descriptors whose source the fold of the same step writes) of packing records: the
    dependent ones (transpose bit 2, _MatView.dep) first."""
    recs = sorted(recs, key=lambda r: 0 if (r[5] & 4) else 1)      # (stable)
    arr = np.array(recs, dtype=_DESC_DTYPE)
    return torch.from_numpy(arr.view(np.uint8).copy()).to(device), len(recs), sum(1 for r in recs if r[5] & 4)


def _small_gemm(tab, base, device) -> None:
    t, cnt, mx = tab
    _lib.check(_lib.lib().dss2_small_gemm(t.data_ptr(), cnt, mx, base, _lib.stream_ptr(device)), "dss2_small_gemm")


class _FoldPlan:
    """The edge MLP's second Linear folded into the first TAGConv (same mathematics):
        conv0(S W2^T + deg b2^T) = sum_m A^m (S (W_m W2)^T) + sum_m (A^m deg) (W_m b2)^T + bias
    so conv 0 runs directly on the aggregated hidden S with folded weights Wf_m = W_m W2 and a
    rank-(K+1) bias term bf_m = W_m b2 scaled by the topology constants A^m deg in the epilogue: the
    H x H GEMM of the Linear, its data-gradient and its weight-gradient (3 launches + a slab reduction
    per step) disappear.  Chain rule back, in
    weight space (one batched small-GEMM launch):
        dW_m = dWf_m W2^T + dbf_m (x) b2 ;  dW2 = sum_m W_m^T dWf_m ;  db2 = sum_m W_m^T dbf_m."""

    def __init__(self, W2, b2, ws, device, off_w2: int, off_conv0: int):
        nm, (ho, hid) = len(ws), ws[0].shape
        self.nm, self.ho, self.hid, self.device = nm, ho, hid, device
        self.Wf = torch.zeros(nm, ho, hid, dtype=_F32, device=device)
        self.bf = torch.zeros(nm, ho, dtype=_F32, device=device)
        # wgrad of the folded conv writes here: [nm*ho*hid dWf][ho db][nm*ho dbf]
        self.gfold = torch.zeros(nm * ho * hid + ho + nm * ho, dtype=_F32, device=device)
        self.one = torch.ones(1, dtype=_F32, device=device)
        self.params = (W2, b2, list(ws))
        self.off_w2, self.off_conv0 = off_w2, off_conv0
        self.ptrs = None

    def records(self, base_off: int = 0):
        """(forward, backward) small-GEMM records; backward outputs are element offsets from the flat gradient buffer
        handed to the launch, shifted by ``base_off`` (a stack's blocks share ONE buffer and ONE launch)."""
        W2, b2, ws = self.params
        nm, ho, hid = self.nm, self.ho, self.hid
        f4 = 4
        fwd, bwd = [], []
        off_conv0, off_w2 = self.off_conv0 + base_off, self.off_w2 + base_off
        for m, w in enumerate(ws):
            fwd.append(_sg([w.data_ptr()], [W2.data_ptr()], ho, hid, hid, hid, hid, hid, C=self.Wf[m].data_ptr()))
            fwd.append(_sg([w.data_ptr()], [b2.data_ptr()], ho, 1, hid, hid, 1, 1, C=self.bf[m].data_ptr()))
        g = self.gfold.data_ptr()
        dWf = [g + f4 * m * ho * hid for m in range(nm)]
        db = g + f4 * nm * ho * hid
        dbf = [db + f4 * ho + f4 * m * ho for m in range(nm)]
        for m, w in enumerate(ws):   # dW_m = dWf_m W2^T + dbf_m (x) b2
            bwd.append(_sg([dWf[m]], [W2.data_ptr()], ho, hid, hid, hid, hid, hid, c_off=off_conv0 + m * ho * hid,
                           tB=1, u=dbf[m], v=b2.data_ptr()))
        # conv0.bias gradient: plain copy of the unscaled column sums (K = 0, rank-1 term with v = 1)
        bwd.append(_sg([], [], ho, 1, 0, 1, 1, 1, c_off=off_conv0 + nm * ho * hid, u=db, v=self.one.data_ptr()))
        # dW2 = sum_m W_m^T dWf_m ; db2 = sum_m W_m^T dbf_m
        bwd.append(_sg([w.data_ptr() for w in ws], dWf, hid, hid, ho, hid, hid, hid, c_off=off_w2, tA=1))
        bwd.append(_sg([w.data_ptr() for w in ws], dbf, hid, 1, ho, hid, 1, 1, c_off=off_w2 + hid * hid, tA=1))
        return fwd, bwd

    def _tables(self):
        fwd, bwd = self.records()
        self.fwd_tab, self.bwd_tab = _sg_table(fwd, self.device), _sg_table(bwd, self.device)

    def pointers(self):
        W2, b2, ws = self.params
        return (W2.data_ptr(), b2.data_ptr()) + tuple(w.data_ptr() for w in ws)

    def _check(self):
        ptrs = self.pointers()
        if ptrs != self.ptrs:
            self._tables()
            self.ptrs = ptrs

    def refresh_forward(self):
        self._check()
        t, n, mx = self.fwd_tab
        st = _lib.stream_ptr(self.device)
        _lib.check(_lib.lib().dss2_small_gemm(t.data_ptr(), n, mx, None, st), "dss2_small_gemm")

    def backward(self, flat: torch.Tensor):
        self._check()
        t, n, mx = self.bwd_tab
        st = _lib.stream_ptr(self.device)
        _lib.check(_lib.lib().dss2_small_gemm(t.data_ptr(), n, mx, flat.data_ptr(), st), "dss2_small_gemm")


class _MatView:
    """A [rows, cols] block of a row-major parameter tensor (leading dimension ld, element offset
    off): lets the pack kernel read e.g. W1[:, :fn] in place, without a copy per step."""

    def __init__(self, t: torch.Tensor, rows: int, cols: int, ld: int, off: int = 0, dep: bool = False):
        self.t, self.shape, self.ld, self.off = t, (rows, cols), ld, off
        self.dep = dep      # written by the fold of the same step (transpose bit 2; the packing launch runs behind the fold's)

    def data_ptr(self) -> int:
        return self.t.data_ptr() + 4 * self.off

    def is_contiguous(self) -> bool:
        return self.t.is_contiguous()


def _as_view(w):
    return w if isinstance(w, _MatView) else _MatView(w, w.shape[0], w.shape[1], w.shape[1], 0)


class _PackPlan:
    """Fragment-packed copies (forward and data-gradient layouts) of a list of weight matrices,
    refreshed by ONE kernel launch per forward."""

    def __init__(self, groups: Sequence[Sequence[torch.Tensor]], device, stacked: bool = False, stacked_groups=(),
                 bf16_groups=(), f16: bool = False):
        # groups[g] = the nmat matrices [hout, hin] of one fused GEMM (TAGConv lins, or one Linear);
        # entries are Parameters or _MatView blocks of a Parameter.
        # stacked: every group uses the narrow layouts whatever its width -- forward = matrices side by side along the
        # output columns, data-gradient = stacked along k -- as ONE plain GEMM (the global-memory propagation path)
        self.groups = groups = [[_as_view(w) for w in mats] for mats in groups]
        self.device = device
        self.stacked = stacked
        self._stk = [bool(len(mats) > 1 and (stacked or g in stacked_groups)) for g, mats in enumerate(groups)]
        self.fwd, self.bwd, self.meta = [], [], []
        for g, mats in enumerate(groups):
            hout, hin = mats[0].shape
            nm = len(mats)
            if is_narrow(nm, hout) or self._stk[g]:
                kf, cf, kb, cb = _round8(hin), _ncg(nm * hout), _round8(nm * hout), _ncg(hin)
                self.fwd.append(torch.zeros(cf * (kf // 8) * 256, dtype=_F32, device=device))
                self.bwd.append(torch.zeros(cb * (kb // 8) * 256, dtype=_F32, device=device))
            else:
                kf, cf, kb, cb = _round8(hin), _ncg(hout), _round8(hout), _ncg(hin)
                self.fwd.append(torch.zeros(nm * cf * (kf // 8) * 256, dtype=_F32, device=device))
                self.bwd.append(torch.zeros(nm * cb * (kb // 8) * 256, dtype=_F32, device=device))
            self.meta.append((nm, hout, hin, kf, cf, kb, cb))
        # bf16_groups: additionally the bf16x3 fragment layout (fp32-accurate tile GEMM on the bf16 matrix pipe,
        # csrc/dss2_gemm_chain16.hip): [matrix][col group][k/16][3 planes][64 lanes][8 bf16]
        # f16: those groups in the f16x2 layout instead (f16x3 chains, b_format 2): [matrix][col group][k/16][2 planes][64 lanes][8 fp16],
        # then one int32 scale exponent per matrix and packed column (written by the pack kernel, which forms each column's maximum)
        self.fwd16, self.bwd16, self.f16 = {}, {}, bool(f16)
        for g in bf16_groups:
            nm, hout, hin = self.meta[g][0:3]
            if is_narrow(nm, hout) or self._stk[g]:
                raise ValueError("bf16x3 packing is for plain per-matrix layouts")
            kf, cf, kb, cb = _round16(hin), _ncg(hout), _round16(hout), _ncg(hin)
            per = 512 if self.f16 else 768      # floats per (column group, k-step of 16)
            self.fwd16[g] = torch.zeros(nm * cf * (kf // 16) * per + (nm * cf * 32 if self.f16 else 0), dtype=_F32, device=device)      # (+ one exponent
            self.bwd16[g] = torch.zeros(nm * cb * (kb // 16) * per + (nm * cb * 32 if self.f16 else 0), dtype=_F32, device=device)      #  per matrix and packed column)
        self.ptrs = None
        self.table = None
        self.max_elems = 0
        self.version = 0

    def pointers(self):
        return tuple(w.data_ptr() for mats in self.groups for w in mats)

    def records(self):
        """The pack kernel's descriptor records of this plan (a stack concatenates those of its blocks into one launch)."""
        recs = []
        for g, mats in enumerate(self.groups):
            nm, hout, hin, kf, cf, kb, cb = self.meta[g]
            narrow = is_narrow(nm, hout) or self._stk[g]
            for m, w in enumerate(mats):
                if not w.is_contiguous():
                    raise RuntimeError("weight matrices must be contiguous")
                # record = (src, dst, rows, cols, ld, transpose, koff, kpad, ncg, joff)
                dp = 4 if w.dep else 0
                if narrow:
                    recs.append((w.data_ptr(), self.fwd[g].data_ptr(), hout, hin, w.ld, 1 | dp, 0, kf, cf, m * hout))
                    recs.append((w.data_ptr(), self.bwd[g].data_ptr(), hout, hin, w.ld, 0 | dp, m * hout, kb, cb, 0))
                else:
                    recs.append((w.data_ptr(), self.fwd[g].data_ptr() + 4 * m * cf * (kf // 8) * 256, hout, hin, w.ld, 1 | dp, 0, kf, cf, 0))
                    recs.append((w.data_ptr(), self.bwd[g].data_ptr() + 4 * m * cb * (kb // 8) * 256, hout, hin, w.ld, 0 | dp, 0, kb, cb, 0))
                self.max_elems = max(self.max_elems, (cf + 1) * (kf // 8 + 1) * 64, (cb + 1) * (kb // 8 + 1) * 64)
                if g in self.fwd16 and self.f16:      # transpose | 8: f16x2 layout (dst = the group's buffer, koff = matrices, joff = this one)
                    k16, b16 = _round16(hin), _round16(hout)
                    recs.append((w.data_ptr(), self.fwd16[g].data_ptr(), hout, hin, w.ld, 9 | dp, nm, k16, cf, m))
                    recs.append((w.data_ptr(), self.bwd16[g].data_ptr(), hout, hin, w.ld, 8 | dp, nm, b16, cb, m))
                elif g in self.fwd16:      # transpose | 2: bf16x3 layout
                    k16, b16 = _round16(hin), _round16(hout)
                    recs.append((w.data_ptr(), self.fwd16[g].data_ptr() + 4 * m * cf * (k16 // 16) * 768, hout, hin, w.ld, 3 | dp, 0, k16, cf, 0))
                    recs.append((w.data_ptr(), self.bwd16[g].data_ptr() + 4 * m * cb * (b16 // 16) * 768, hout, hin, w.ld, 2 | dp, 0, b16, cb, 0))
        return recs

    def _build_table(self):
        self.table, self.n_desc, self.n_dep = _pack_table(self.records(), self.device)

    def refresh(self, fold: Optional["_FoldPlan"] = None):
        """Packs every matrix of the plan; with ``fold`` the fold's small GEMMs run first -- inside the same launch
        (ops.prep_weights)."""
        ptrs = self.pointers()
        if ptrs != self.ptrs:
            self._build_table()
            self.ptrs = ptrs
        fold_tab = None
        if fold is not None:
            fold._check()
            fold_tab = fold.fwd_tab
        prep_weights(fold_tab, (self.table, self.n_desc, self.max_elems, self.n_dep), self.device)
        self.version += 1
        return self.version
