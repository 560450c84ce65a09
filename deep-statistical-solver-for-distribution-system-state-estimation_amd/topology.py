"""Per-topology graph structure for the HIP kernels (built once per distinct edge_index, cached).

Replaces, per forward call of the reference: ``MPN.is_directed`` / ``undirect_graph``
(/root/reference/networks.py:236-258: host sync + 3 cats), PyG ``gcn_norm`` (degree, pow,
masked_fill, 2 gathers per TAGConv call) and PyG's per-call scatter index handling.

Built with torch index ops on the tensor's own device (sort / bincount / cumsum: plumbing, run
once per topology), then frozen as int32/fp32 device arrays in the layout include/dss2_hip.h
documents: CSR by target, CSR by source, incidence CSR of the stored edges, whole-graph tiles.
"""
from __future__ import annotations

import os
import weakref
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib

_FLIP = 1 << 31
_NRB_CHOICES = (2, 4, 3, 1, 6)   # preference order on utilisation ties (32*nrb rows per tile)
_LDS_LIMIT = 160 * 1024
_ELL_MAX = 8


def reference_is_directed(edge_index: torch.Tensor) -> bool:
    """/root/reference/networks.py:236-238: looks only at the first edge of the batch:
    is there NO edge (v0 -> u0) among the edges leaving v0?  (One host sync; cached per topology.)"""
    u0, v0 = edge_index[0, 0], edge_index[1, 0]
    cand = edge_index[1, edge_index[0, :] == v0]
    return not bool((cand == u0).any().item())


def _csr(key: torch.Tensor, n: int):
    """Stable sort permutation and int32 row pointer for grouping by `key` (values in [0, n))."""
    perm = torch.sort(key, stable=True).indices
    cnt = torch.bincount(key, minlength=n)
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=key.device)
    torch.cumsum(cnt, 0, out=rowptr[1:])
    return perm, rowptr.to(torch.int32), cnt


def _pack_tiles(bounds: np.ndarray, tm: int) -> np.ndarray:
    """Greedy: consecutive whole segments (bounds = sorted cut positions incl. 0 and N) per tile of
    at most tm rows.  Returns tile_start (len ntiles+1) or None if a segment exceeds tm."""
    seg = np.diff(bounds)
    if seg.max() > tm:
        return None
    if (seg == seg[0]).all():  # uniform graphs: closed form
        per = tm // int(seg[0])
        idx = np.arange(0, len(seg), per)
        return np.append(bounds[idx], bounds[-1]).astype(np.int32)
    out, i, nb = [int(bounds[0])], 0, len(bounds)
    while i < nb - 1:
        j = int(np.searchsorted(bounds, bounds[i] + tm, side="right")) - 1
        out.append(int(bounds[j]))
        i = j
    return np.asarray(out, dtype=np.int32)


class Topology:
    """Frozen device-side structure of one batched graph.  See include/dss2_hip.h."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, nrb: Optional[int] = None,
                 double: Optional[bool] = None):
        """double=None: the reference's rule (MPN.is_directed on the first edge); False: use the
        edge list exactly as given (standalone EdgeAggregation / TAGConv); True: always double."""
        if edge_index.dim() != 2 or edge_index.size(0) != 2 or edge_index.dtype != torch.int64:
            raise ValueError("edge_index must be an int64 tensor of shape [2, E]")
        dev = edge_index.device
        ei = edge_index
        N, E = int(num_nodes), int(ei.size(1))
        if E == 0 or N == 0:
            raise ValueError("empty graph batch")
        if N >= 2 ** 31 or 2 * E >= 2 ** 31:
            raise ValueError("graph too large for the int32 CSR")
        self.N, self.E, self.device = N, E, dev
        self.directed = reference_is_directed(ei) if double is None else bool(double)
        ar = torch.arange(E, device=dev)
        if self.directed:  # networks.py:242-254: append reversed edges, flag them for the sign flip
            src = torch.cat([ei[0], ei[1]])
            tgt = torch.cat([ei[1], ei[0]])
            eid = torch.cat([ar, ar - _FLIP])
        else:
            src, tgt, eid = ei[0], ei[1], ar
        self.E2 = int(src.numel())
        # ---- PyG gcn_norm(add_self_loops=False): in-degree on the (doubled) graph
        perm, self.rowptr, cnt = _csr(tgt, N)
        degf = cnt.to(torch.float32)
        dis = degf.pow(-0.5)
        dis = dis.masked_fill(dis == float("inf"), 0.0)
        w_d = dis[src] * dis[tgt]
        self.deg = degf.contiguous()
        self._deg_pows = None
        self.col = src[perm].to(torch.int32).contiguous()
        self.ent = eid[perm].to(torch.int32).contiguous()
        self.w = w_d[perm].contiguous()
        self.perm = perm  # directed-edge id of each CSR entry (for the standalone segment_sum)
        permT, self.rowptrT, _ = _csr(src, N)
        self.colT = tgt[permT].to(torch.int32).contiguous()
        self.entT = eid[permT].to(torch.int32).contiguous()
        self.wT = w_d[permT].contiguous()
        # ---- incidence CSR of the STORED edges (loss: bus injections, data.py:428-429)
        nodes = torch.cat([ei[0], ei[1]])
        inc = torch.cat([ar, ar - _FLIP])          # end 0 = from-end, end 1 (flag) = to-end
        permI, self.inc_rowptr, _ = _csr(nodes, N)
        self.inc_ent = inc[permI].to(torch.int32).contiguous()
        self.efrom = ei[0].to(torch.int32).contiguous()
        self.eto = ei[1].to(torch.int32).contiguous()
        # ---- whole-graph tiles: a cut before row p is legal iff no edge spans it
        lo, hi = torch.minimum(src, tgt), torch.maximum(src, tgt)
        cover = torch.zeros(N + 2, dtype=torch.int32, device=dev)
        one = torch.ones_like(lo, dtype=torch.int32)
        cover.index_add_(0, lo + 1, one)
        cover.index_add_(0, hi + 1, -one)
        cuts = (torch.cumsum(cover, 0)[: N + 1] == 0).nonzero().flatten().cpu().numpy().astype(np.int64)
        bounds = np.unique(np.concatenate([cuts, [0, N]]))
        rowptr_h = self.rowptr.cpu().numpy().astype(np.int64)
        rowptrT_h = self.rowptrT.cpu().numpy().astype(np.int64)
        self.max_segment = int(np.diff(bounds).max())
        env = os.environ.get("DSS2_NRB")
        choices = (int(nrb),) if nrb else ((int(env),) if env else _NRB_CHOICES)
        best = None
        for cand in choices:
            ts = _pack_tiles(bounds, 32 * cand)
            if ts is None:
                continue
            util = N / float((len(ts) - 1) * 32 * cand)
            if best is None or util > best[0] + 0.03:
                best = (util, cand, ts)
        if best is None:
            raise NotImplementedError(
                f"largest connected component has {self.max_segment} nodes; the LDS-resident tile kernels "
                f"support up to {32 * max(choices)} nodes per graph")
        self.utilisation, self.nrb, ts = best
        self.ntiles = len(ts) - 1
        self.max_nnz = int((rowptr_h[ts[1:]] - rowptr_h[ts[:-1]]).max())
        self.max_nnzT = int((rowptrT_h[ts[1:]] - rowptrT_h[ts[:-1]]).max())
        # ELL width for the in-LDS propagation (0 = use the CSR path: hubs would waste padded slots)
        md, mdT = int(np.diff(rowptr_h).max()), int(np.diff(rowptrT_h).max())
        self.ell = md if md <= _ELL_MAX else 0
        self.ellT = mdT if mdT <= _ELL_MAX else 0
        self.tile_start = torch.from_numpy(ts).to(dev)
        # per-tile ELL slices [ntiles, D, 32*nrb] of {local source row, weight bits}: what the kernels stage
        # in LDS for the in-tile propagation, precomputed here so that staging is one coalesced copy
        self.ell_tiles = self._ell_tiles(self.rowptr, self.col, self.w, self.ell)
        self.ellT_tiles = self._ell_tiles(self.rowptrT, self.colT, self.wT, self.ellT)
        # same slices carrying the stored edge id | flip instead of the weight (edge-MLP kernels); -1 = empty
        self.ell_ent_tiles = self._ell_tiles(self.rowptr, self.col, self.ent, self.ell, ids=True)
        self.ellT_ent_tiles = self._ell_tiles(self.rowptrT, self.colT, self.entT, self.ellT, ids=True)

    @property
    def deg_pows(self) -> torch.Tensor:
        """[N, 4] fp32, column m = A_hat^m deg (A_hat = the gcn_norm propagation matrix): the row scales of
        a bias folded through m propagations (networks._FoldPlan).  Built once per topology, in fp64."""
        if self._deg_pows is None:
            rp = self.rowptr.to(torch.int64)
            rows = torch.repeat_interleave(torch.arange(self.N, device=self.device), rp[1:] - rp[:-1])
            col, w = self.col.to(torch.int64), self.w.to(torch.float64)
            v = self.deg.to(torch.float64)
            cols = [v]
            for _ in range(3):
                v = torch.zeros(self.N, dtype=torch.float64, device=self.device).index_add_(0, rows, w * v[col])
                cols.append(v)
            self._deg_pows = torch.stack(cols, dim=1).to(torch.float32).contiguous()
        return self._deg_pows

    def _ell_tiles(self, rowptr, col, w, width, ids=False):
        if width <= 0:
            return None
        dev, tm, nt = self.device, 32 * self.nrb, self.ntiles
        ts = self.tile_start.to(torch.int64)
        rp = rowptr.to(torch.int64)
        deg = rp[1:] - rp[:-1]
        rows = torch.repeat_interleave(torch.arange(self.N, device=dev), deg)
        k = torch.arange(rows.numel(), device=dev) - rp[rows]
        tile = torch.searchsorted(ts, rows, right=True) - 1
        r = rows - ts[tile]
        out = torch.zeros(nt, width, tm, 2, dtype=torch.int32, device=dev)
        if ids:
            out[:, :, :, 1] = -1                                                    # padding: empty slot
        else:
            out[:, :, :, 0] = torch.arange(tm, dtype=torch.int32, device=dev)      # padding: {own row, weight 0}
        out[tile, k, r, 0] = (col.to(torch.int64) - ts[tile]).to(torch.int32)
        out[tile, k, r, 1] = w if ids else w.view(torch.int32)
        return out.contiguous()

    def lds_check(self, nmat: int, kpad: int, ncg: int) -> None:
        need = _lib.lib().dss2_gemm_prop_lds_bytes(self.nrb, nmat, kpad, ncg, max(self.max_nnz, self.max_nnzT),
                                                    min(self.ell, self.ellT))
        if need > _LDS_LIMIT:
            raise NotImplementedError(f"tile of {32 * self.nrb} rows x K={kpad} needs {need} B of LDS (> 160 KiB)")


# ------------------------------------------------------------------------------------------
# cache: identity fast path (same tensor object, unmodified) -> no sync; otherwise content hash
# ------------------------------------------------------------------------------------------
_by_hash: Dict[Tuple, Topology] = {}
_last: Dict[int, Tuple] = {}   # id(tensor) -> (weakref, version, data_ptr, num_nodes, topo)
_MAX_CACHE = 64


def content_hash(edge_index: torch.Tensor) -> int:
    """64-bit device-side hash of edge_index (one tiny kernel + one 8-byte D2H)."""
    ei = edge_index if edge_index.is_contiguous() else edge_index.contiguous()
    out = torch.zeros(1, dtype=torch.int64, device=ei.device)
    stream = torch.cuda.current_stream(ei.device).cuda_stream
    _lib.check(_lib.lib().dss2_topology_hash(ei.data_ptr(), ei.numel(), out.data_ptr(), stream), "dss2_topology_hash")
    return int(out.item())


def get_topology(edge_index: torch.Tensor, num_nodes: int) -> Topology:
    if not edge_index.is_cuda:
        raise RuntimeError("DSS2 HIP path: edge_index must live on the GPU (there is no CPU fallback)")
    key_id = id(edge_index)
    hit = _last.get(key_id)
    if hit is not None:
        ref, ver, ptr, nn, topo = hit
        if ref() is edge_index and ver == edge_index._version and ptr == edge_index.data_ptr() and nn == num_nodes:
            return topo
    h = content_hash(edge_index)
    key = (edge_index.device.index, int(num_nodes), int(edge_index.size(1)), h)
    topo = _by_hash.get(key)
    if topo is None:
        topo = Topology(edge_index, num_nodes)
        if len(_by_hash) >= _MAX_CACHE:
            _by_hash.pop(next(iter(_by_hash)))
        _by_hash[key] = topo
    if len(_last) >= _MAX_CACHE:
        _last.pop(next(iter(_last)))
    _last[key_id] = (weakref.ref(edge_index), edge_index._version, edge_index.data_ptr(), num_nodes, topo)
    return topo


def clear_cache() -> None:
    _by_hash.clear()
    _last.clear()
