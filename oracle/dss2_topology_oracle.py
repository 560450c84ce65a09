"""CPU oracle for the per-topology graph structure the HIP kernels consume (TEST INFRASTRUCTURE ONLY).

Nothing in the product package imports this file; only ``tests/`` do, as the checker of
``dss2_csr_build`` / ``dss2_tiles_*`` / ``dss2_ell_tiles_build`` / ``dss2_deg_pows`` (csrc/dss2_topology.hip), whose
outputs must equal these arrays BIT FOR BIT.

What it restates, with plain torch index ops on the CPU:

* ``MPN.is_directed`` / ``undirect_graph``   /root/reference/networks.py:236-258 (first-edge-only rule; reverse edges
  appended, flagged for the sign flip of edge_attr columns 0 and 2);
* PyG ``gcn_norm(add_self_loops=False)``     in-degree on the (doubled) list, ``deg.pow(-0.5)`` with inf -> 0,
  ``w = dis[src] * dis[tgt]`` (same ops as oracle/dss2_oracle.gcn_norm_no_self_loops);
* the row order of ``index_add_``            CSR rows list their entries in ascending directed edge id (stable sort);
* the layout of include/dss2_hip.h           CSR by target / by source, incidence CSR of the stored edges, whole-graph
  tiles (greedy packing of segments no edge spans), per-tile ELL slices, deg_pows.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

FLIP = 1 << 31
NRB_CHOICES = (2, 4, 3, 1, 6)   # preference order on utilisation ties (32*nrb rows per tile)
ELL_MAX = 8


def is_directed(edge_index: torch.Tensor) -> bool:
    """/root/reference/networks.py:236-238."""
    u0, v0 = edge_index[0, 0], edge_index[1, 0]
    cand = edge_index[1, edge_index[0, :] == v0]
    return not bool((cand == u0).any().item())


def _csr(key: torch.Tensor, n: int):
    perm = torch.sort(key, stable=True).indices
    cnt = torch.bincount(key, minlength=n)
    rowptr = torch.zeros(n + 1, dtype=torch.int64)
    torch.cumsum(cnt, 0, out=rowptr[1:])
    return perm, rowptr.to(torch.int32), cnt


def pack_tiles(bounds: np.ndarray, tm: int) -> Optional[np.ndarray]:
    """Greedy: consecutive whole segments per tile of at most tm rows; None if a segment exceeds tm."""
    seg = np.diff(bounds)
    if seg.max() > tm:
        return None
    out, i, nb = [int(bounds[0])], 0, len(bounds)
    while i < nb - 1:
        j = int(np.searchsorted(bounds, bounds[i] + tm, side="right")) - 1
        out.append(int(bounds[j]))
        i = j
    return np.asarray(out, dtype=np.int32)


class TopologyOracle:
    def __init__(self, edge_index: torch.Tensor, num_nodes: int, nrb: Optional[int] = None, double: Optional[bool] = None):
        ei = edge_index.cpu()
        N, E = int(num_nodes), int(ei.size(1))
        self.N, self.E = N, E
        self.directed = is_directed(ei) if double is None else bool(double)
        ar = torch.arange(E)
        if self.directed:
            src, tgt = torch.cat([ei[0], ei[1]]), torch.cat([ei[1], ei[0]])
            eid = torch.cat([ar, ar - FLIP])
        else:
            src, tgt, eid = ei[0], ei[1], ar
        self.E2 = int(src.numel())
        perm, self.rowptr, cnt = _csr(tgt, N)
        degf = cnt.to(torch.float32)
        dis = degf.pow(-0.5)
        dis = dis.masked_fill(dis == float("inf"), 0.0)
        w_d = dis[src] * dis[tgt]
        self.deg = degf
        self.col, self.ent, self.w, self.perm = src[perm].to(torch.int32), eid[perm].to(torch.int32), w_d[perm], perm.to(torch.int32)
        permT, self.rowptrT, _ = _csr(src, N)
        self.colT, self.entT, self.wT, self.permT = tgt[permT].to(torch.int32), eid[permT].to(torch.int32), w_d[permT], permT.to(torch.int32)
        nodes = torch.cat([ei[0], ei[1]])
        inc = torch.cat([ar, ar - FLIP])
        permI, self.inc_rowptr, _ = _csr(nodes, N)
        self.inc_ent = inc[permI].to(torch.int32)
        self.efrom, self.eto = ei[0].to(torch.int32), ei[1].to(torch.int32)
        lo, hi = torch.minimum(src, tgt), torch.maximum(src, tgt)
        cover = torch.zeros(N + 2, dtype=torch.int64)
        cover.index_add_(0, lo + 1, torch.ones_like(lo))
        cover.index_add_(0, hi + 1, -torch.ones_like(hi))
        cuts = (torch.cumsum(cover, 0)[: N + 1] == 0).nonzero().flatten().numpy().astype(np.int64)
        self.bounds = bounds = np.unique(np.concatenate([cuts, [0, N]]))
        self.max_segment = int(np.diff(bounds).max())
        rp, rpT = self.rowptr.numpy().astype(np.int64), self.rowptrT.numpy().astype(np.int64)
        self.max_deg, self.max_degT = int(np.diff(rp).max()), int(np.diff(rpT).max())
        best = None
        for cand in ((int(nrb),) if nrb else NRB_CHOICES):
            ts = pack_tiles(bounds, 32 * cand)
            if ts is None:
                continue
            util = N / float((len(ts) - 1) * 32 * cand)
            if best is None or util > best[0] + 0.03:
                best = (util, cand, ts)
        self.tiled = best is not None
        if not self.tiled:
            return
        self.utilisation, self.nrb, ts = best
        self.ntiles = len(ts) - 1
        self.tile_start = torch.from_numpy(ts)
        self.max_nnz = int((rp[ts[1:]] - rp[ts[:-1]]).max())
        self.max_nnzT = int((rpT[ts[1:]] - rpT[ts[:-1]]).max())
        self.ell = self.max_deg if self.max_deg <= ELL_MAX else 0
        self.ellT = self.max_degT if self.max_degT <= ELL_MAX else 0
        self.ell_tiles = self._ell_tiles(self.rowptr, self.col, self.w, self.ell)
        self.ellT_tiles = self._ell_tiles(self.rowptrT, self.colT, self.wT, self.ellT)
        self.ell_ent_tiles = self._ell_tiles(self.rowptr, self.col, self.ent, self.ell, ids=True)
        self.ellT_ent_tiles = self._ell_tiles(self.rowptrT, self.colT, self.entT, self.ellT, ids=True)

    @property
    def deg_pows(self) -> torch.Tensor:
        rp = self.rowptr.to(torch.int64)
        rows = torch.repeat_interleave(torch.arange(self.N), rp[1:] - rp[:-1])
        col, w = self.col.to(torch.int64), self.w.to(torch.float64)
        v = self.deg.to(torch.float64)
        cols = [v]
        for _ in range(3):
            v = torch.zeros(self.N, dtype=torch.float64).index_add_(0, rows, w * v[col])
            cols.append(v)
        return torch.stack(cols, dim=1).to(torch.float32).contiguous()

    def _ell_tiles(self, rowptr, col, w, width, ids=False):
        if width <= 0:
            return None
        tm, nt = 32 * self.nrb, self.ntiles
        ts = self.tile_start.to(torch.int64)
        rp = rowptr.to(torch.int64)
        deg = rp[1:] - rp[:-1]
        rows = torch.repeat_interleave(torch.arange(self.N), deg)
        k = torch.arange(rows.numel()) - rp[rows]
        tile = torch.searchsorted(ts, rows, right=True) - 1
        r = rows - ts[tile]
        out = torch.zeros(nt, width, tm, 2, dtype=torch.int32)
        if ids:
            out[:, :, :, 1] = -1
        else:
            out[:, :, :, 0] = torch.arange(tm, dtype=torch.int32)
        out[tile, k, r, 0] = (col.to(torch.int64) - ts[tile]).to(torch.int32)
        out[tile, k, r, 1] = w if ids else w.view(torch.int32)
        return out.contiguous()
