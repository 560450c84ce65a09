"""Per-topology graph structure for the HIP kernels, built ON THE DEVICE (``dss2_csr_build`` and friends,
csrc/dss2_topology.hip) once per distinct edge_index and cached.

Replaces, per forward call of the reference: ``MPN.is_directed`` / ``undirect_graph``
(/root/reference/networks.py:236-258: host sync + 3 cats), PyG ``gcn_norm`` (degree, pow,
masked_fill, 2 gathers per TAGConv call) and PyG's per-call scatter index handling.

Two parts, built separately:

* the CSR part (always): CSR by target / by source of the (doubled) directed edge list, incidence CSR of the stored
  edges, int32 endpoints, in-degrees, gcn_norm weights -- all the loss kernels, ``MessagePassing.propagate`` and
  ``segment_sum`` need; asynchronous, any graph size;
* the tile part (lazily, when a tile kernel asks for it): whole-graph tiles of <= 32*nrb rows and the per-tile ELL
  slices the tile kernels stage in LDS.  A connected component above 192 rows has no graph-aligned tiles:
  ``global_only`` is set, the plain GEMMs run on uniform 64-row tiles and the propagation in global memory
  (networks._tagconv_forward_global).

Host synchronisation: none when the caller passes a :class:`TopologyHint` (the device data loader does: it knows
its samples' sizes and degrees); otherwise one 24-byte copy for the cache key + directedness and one 64-byte copy of
the build statistics on a cache miss.  There is no CPU path: CPU tensors raise.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from . import _lib

_FLIP = 1 << 31
_NRB_CHOICES = (2, 4, 3, 1, 6)   # preference order on utilisation ties (32*nrb rows per tile)
_LDS_LIMIT = 160 * 1024
_ELL_MAX = 8
_TILE_ATTRS = frozenset(["global_only", "nrb", "ntiles", "tile_start", "utilisation", "max_segment", "max_nnz", "max_nnzT", "ell", "ellT", "max_tile_rows",
                         "ell_tiles", "ellT_tiles", "ell_ent_tiles", "ellT_ent_tiles"])


@dataclass(frozen=True)
class TopologyHint:
    """What a caller that assembled the batch itself knows without looking at the device (dataset.DataLoader does):
    with it the whole structure is built without a single device-to-host copy."""
    directed: bool               # MPN.is_directed of the batch: its first edge has no reverse edge (networks.py:236-238)
    nodes_per_graph: int         # every graph of the batch has this many nodes, in consecutive rows
    max_degree: int              # upper bound of the in- and out-degree on the directed (doubled) list
    max_edges_per_graph: int     # upper bound of the stored edges of one graph
    # round 6: where the graphs' ranges in the stored edge list are known too, the whole CSR part (and the folded-bias row scales) is ONE
    # launch, a wave per graph (dss2_csr_build_graphs): edges_per_graph (every graph has this many, in graph order) or edge_ptr (device int64
    # [G + 1], the batch's own prefix sum of the graphs' edge counts); neither: the general build (global scans)
    edges_per_graph: int = 0
    edge_ptr: Optional[torch.Tensor] = None


def _stream(dev) -> int:
    return _lib.stream_ptr(dev)


def probe(edge_index: torch.Tensor) -> Tuple[int, int, bool]:
    """(hash1, hash2, is_directed) of edge_index in one kernel + one 24-byte device-to-host copy."""
    ei = edge_index if edge_index.is_contiguous() else edge_index.contiguous()
    out = torch.zeros(3, dtype=torch.int64, device=ei.device)
    _lib.check(_lib.lib().dss2_topology_probe(ei.data_ptr(), ei.size(1), out.data_ptr(), _stream(ei.device)), "dss2_topology_probe")
    h1, h2, rev = out.tolist()
    return h1, h2, rev == 0


def reference_is_directed(edge_index: torch.Tensor) -> bool:
    """/root/reference/networks.py:236-238: looks only at the first edge of the batch:
    is there NO edge (v0 -> u0) among the edges leaving v0?  (One host sync; cached per topology.)"""
    if edge_index.is_cuda:
        return probe(edge_index)[2]
    u0, v0 = edge_index[0, 0], edge_index[1, 0]
    cand = edge_index[1, edge_index[0, :] == v0]
    return not bool((cand == u0).any().item())


class Topology:
    """Frozen device-side structure of one batched graph.  See include/dss2_hip.h."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, nrb: Optional[int] = None,
                 double: Optional[bool] = None, hint: Optional[TopologyHint] = None, flip: bool = True):
        """double=None: the reference's rule (MPN.is_directed on the first edge; from the hint when given); False: use
        the edge list exactly as given (standalone EdgeAggregation / TAGConv / propagate); True: always double.
        flip=False: reverse edges carry no sign-flip flag (the Multi* / MaskEmbd* variants duplicate edge_attr unchanged,
        /root/reference/networks.py:440-444, where MPN negates columns 0 and 2, :250-254)."""
        if edge_index.dim() != 2 or edge_index.size(0) != 2 or edge_index.dtype != torch.int64:
            raise ValueError("edge_index must be an int64 tensor of shape [2, E]")
        N, E = int(num_nodes), int(edge_index.size(1))
        if E == 0 or N == 0:
            raise ValueError("empty graph batch")
        if N >= 2 ** 31 - 4 or 2 * E >= 2 ** 31 - 4:
            raise ValueError("graph too large for the int32 CSR")
        if not edge_index.is_cuda:
            raise RuntimeError("DSS2 HIP path: edge_index must live on the GPU (there is no CPU fallback)")
        dev = edge_index.device
        ei = edge_index if edge_index.is_contiguous() else edge_index.contiguous()
        self.N, self.E, self.device = N, E, dev
        self.hint, self._nrb_forced = hint, nrb
        if double is not None:
            self.directed = bool(double)
        elif hint is not None:
            self.directed = bool(hint.directed)
        else:
            self.directed = probe(ei)[2]
        self.E2 = E2 = 2 * E if self.directed else E
        L = _lib.lib()
        # one int32 arena for every array (16-byte aligned slices), one for the build's scratch
        sizes = [("rowptr", N + 1), ("col", E2), ("ent", E2), ("perm", E2), ("w", E2), ("rowptrT", N + 1), ("colT", E2),
                 ("entT", E2), ("permT", E2), ("wT", E2), ("inc_rowptr", N + 1), ("inc_ent", 2 * E), ("efrom", E), ("eto", E),
                 ("deg", N), ("_lastcut", N + 1), ("_meta", 16)]
        padded = [(n + 3) // 4 * 4 for _, n in sizes]
        arena = torch.empty(sum(padded), dtype=torch.int32, device=dev)
        d = self.__dict__
        for (name, n), chunk in zip(sizes, arena.split(padded)):       # one split call: 17 views (16-byte aligned)
            d[name] = chunk[:n] if n != chunk.numel() else chunk
        for name in ("w", "wT", "deg"):
            d[name] = d[name].view(torch.float32)
        graphs = (hint is not None and (hint.edges_per_graph > 0 or hint.edge_ptr is not None) and N % hint.nodes_per_graph == 0
                  and os.environ.get("DSS2_TOPO_GRAPHS", "1") != "0"
                  and bool(L.dss2_csr_build_graphs_supported(hint.nodes_per_graph, hint.max_edges_per_graph)))
        work = None if graphs else torch.empty(int(L.dss2_csr_build_work_ints(N, E, int(self.directed))), dtype=torch.int32, device=dev)
        a = _lib.CsrBuildArgs()
        a.edge_index, a.n_edges, a.n_nodes, a.doubled = ei.data_ptr(), E, N, int(self.directed)
        a.no_flip = int(not flip)
        self.flip = bool(flip)
        base, off = arena.data_ptr(), 0
        for (name, _), pn in zip(sizes, padded):
            setattr(a, name.lstrip("_"), base + 4 * off)
            off += pn
        self._deg_pows = None
        if graphs:
            # one wave per graph: every array above AND the [N, 4] folded-bias row scales (deg_pows) in one launch
            dp = torch.empty(N, 4, dtype=torch.float32, device=dev)
            ep = hint.edge_ptr
            _lib.check(L.dss2_csr_build_graphs(C.byref(a), hint.nodes_per_graph, (ep.data_ptr() if ep is not None else None),
                                               int(hint.edges_per_graph), int(hint.max_edges_per_graph), dp.data_ptr(), _stream(dev)),
                       "dss2_csr_build_graphs")
            self._deg_pows = dp
            self._keep = (ei, ep)
        else:
            a.work = work.data_ptr()
            _lib.check(L.dss2_csr_build(C.byref(a), _stream(dev)), "dss2_csr_build")
            self._keep = (ei,)            # the build reads edge_index asynchronously
        self._stats = None
        self._tiles_built = False

    # ---- lazily built tile part -------------------------------------------------------------------------------
    def __getattr__(self, name):
        if name in _TILE_ATTRS and not self.__dict__.get("_tiles_built", True):
            self._build_tiles()
            return self.__dict__[name]
        raise AttributeError(name)

    @staticmethod
    def _read_stats(meta: torch.Tensor) -> Dict[str, int]:
        m = meta.tolist()
        if m[5] == 1:
            raise ValueError("edge_index holds node ids outside [0, num_nodes)")
        if m[5] == 2:
            raise ValueError("TopologyHint.max_degree is smaller than a row of the batch")
        return dict(max_deg=m[0], max_degT=m[1], max_segment=m[2], n_segments=m[3], min_segment=m[4], error=m[5],
                    max_nnz=m[6], max_nnzT=m[7], ntiles_cand=m[8:16])

    def stats(self) -> Dict[str, int]:
        """Exact build statistics of the PRIMARY tiling (one 64-byte device-to-host copy, cached): degrees, segments, error
        flag.  An alternate tiling (``tiles_for``) is built on its own copy of these 16 words and never shows up here."""
        if self._stats is None:
            self._stats = self._read_stats(self._meta)
        return self._stats

    def tiles_for(self, nrb: int):
        """A second set of tiles / ELL slices with a FORCED tile height of 32 * nrb rows, beside the primary one (cached).
        Used by the whole-stack kernels, which are latency-bound per tile: at small batches twice as many 32-row tiles
        put twice as many CUs to work.  Returns a namespace with the tile attributes (nrb, ntiles, tile_start, ell, ellT,
        ell_tiles, ell_ent_tiles, ellT_tiles, ellT_ent_tiles), or None when no whole-graph tiling of that height exists."""
        import types
        if not self.__dict__.get("_tiles_built", False):
            self._build_tiles()
        if self.global_only:
            return None
        if nrb == self.nrb:
            return self
        alt = self.__dict__.setdefault("_alt_tiles", {})
        if nrb not in alt:
            d = self._build_tiles(choices=(int(nrb),), store=False)
            alt[nrb] = types.SimpleNamespace(**d) if d is not None and not d["global_only"] else None
        return alt[nrb]

    def _build_tiles(self, choices=None, store=True):
        L = _lib.lib()
        dev, N = self.device, self.N
        env = os.environ.get("DSS2_NRB")
        if choices is None:
            choices = (int(self._nrb_forced),) if self._nrb_forced else ((int(env),) if env else _NRB_CHOICES)
        st = _stream(dev)
        hint = self.hint
        # an alternate tiling works on a copy of the statistics words (the walk and the ELL build write per-tiling maxima
        # into them): the primary tiling's statistics stay what they were
        # (closed-form tilings -- a hint with uniform graphs -- write no statistics: the ELL launch only raises the error flag, which may as
        #  well be the primary's; no copy, no fill: two launches less per alternate tiling of a fresh batch, round 6)
        closed_form = hint is not None and N % hint.nodes_per_graph == 0
        meta = self._meta if (store or closed_form) else self._meta.clone()
        if not store and not closed_form:
            meta[5:8].zero_()      # (error flag and per-tile maxima are THIS tiling's own: not the primary's carried over)

        def read_stats():
            if store:
                self._stats = None
                return self.stats()
            return self._read_stats(meta)
        if hint is not None and N % hint.nodes_per_graph == 0:
            # ---- no device-to-host copy: uniform graphs => closed-form tiles, degrees from the hint
            n, G = hint.nodes_per_graph, N // hint.nodes_per_graph
            best = None
            for cand in choices:
                per = (32 * cand) // n
                if per == 0:
                    continue
                nt = -(-G // per)
                util = N / float(nt * 32 * cand)
                if best is None or util > best[0] + 0.03:
                    best = (util, cand, nt, per)
            if best is None:
                return self._global_tiles(n) if store else None
            util, nrb, nt, per = best
            tile_start = torch.empty(nt + 1, dtype=torch.int32, device=dev)
            uniform_rows = per * n          # (written by the ELL build below: dss2_tiles_uniform folded in, round 6)
            max_deg = max_degT = int(hint.max_degree)
            max_segment = n
            max_tile_rows = per * n
            if os.environ.get("DSS2_CHECK", "0") == "1":      # debugging aid: read the build's error flag after all (ONE host sync)
                read_stats()
            nnz_bound = per * hint.max_edges_per_graph * (2 if self.directed else 1)
            exact_nnz = False
        else:
            # ---- general batch: greedy packing for every candidate row budget in one launch, then ONE copy of the statistics
            cap = N
            cands = [torch.empty(cap + 1, dtype=torch.int32, device=dev) for _ in choices]
            tm_host = (C.c_int32 * len(choices))(*[32 * c for c in choices])
            ptrs = (C.c_void_p * len(choices))(*[t.data_ptr() for t in cands])
            _lib.check(L.dss2_tiles_walk(self._lastcut.data_ptr(), N, tm_host, len(choices), ptrs, cap,
                                         meta[8:].data_ptr(), st), "dss2_tiles_walk")
            s = read_stats()
            max_deg, max_degT, max_segment = s["max_deg"], s["max_degT"], s["max_segment"]
            best = None
            for i, cand in enumerate(choices):
                nt = s["ntiles_cand"][i]
                if nt <= 0:
                    continue
                util = N / float(nt * 32 * cand)
                if best is None or util > best[0] + 0.03:
                    best = (util, cand, nt, i)
            if best is None:
                return self._global_tiles(max_segment) if store else None
            util, nrb, nt, i = best
            tile_start = cands[i][:nt + 1].clone()
            uniform_rows = 0
            nnz_bound, exact_nnz = 0, True
            # (known without another copy only where every graph has the same size: whole graphs per tile)
            max_tile_rows = ((32 * nrb) // max_segment) * max_segment if s["min_segment"] == max_segment else 0
        ell = max_deg if max_deg <= _ELL_MAX else 0
        ellT = max_degT if max_degT <= _ELL_MAX else 0
        tm = 32 * nrb

        def slab(width):
            return torch.empty(nt, width, tm, 2, dtype=torch.int32, device=dev) if width > 0 else None
        ell_tiles, ell_ent_tiles, ellT_tiles, ellT_ent_tiles = slab(ell), slab(ell), slab(ellT), slab(ellT)
        b = _lib.EllBuildArgs()
        b.rowptr, b.col, b.ent, b.w = self.rowptr.data_ptr(), self.col.data_ptr(), self.ent.data_ptr(), self.w.data_ptr()
        b.rowptrT, b.colT, b.entT, b.wT = self.rowptrT.data_ptr(), self.colT.data_ptr(), self.entT.data_ptr(), self.wT.data_ptr()
        b.tile_start, b.ntiles, b.tm, b.ell_width, b.ellT_width = tile_start.data_ptr(), nt, tm, ell, ellT
        b.ell_tiles, b.ell_ent_tiles = (ell_tiles.data_ptr() if ell else None), (ell_ent_tiles.data_ptr() if ell else None)
        b.ellT_tiles, b.ellT_ent_tiles = (ellT_tiles.data_ptr() if ellT else None), (ellT_ent_tiles.data_ptr() if ellT else None)
        b.meta = meta.data_ptr()
        b.uniform_rows, b.n_nodes = uniform_rows, N
        if uniform_rows and not (ell or ellT):      # (hub graphs: no ELL launch to fold the tile starts into)
            _lib.check(L.dss2_tiles_uniform(tile_start.data_ptr(), nt, uniform_rows, N, st), "dss2_tiles_uniform")
            b.uniform_rows = 0
        _lib.check(L.dss2_ell_tiles_build(C.byref(b), st), "dss2_ell_tiles_build")
        if not store and hint is not None and os.environ.get("DSS2_CHECK", "0") == "1":
            read_stats()      # (the hint path reads nothing back: with DSS2_CHECK the alternate build's error flag is read too)
        if exact_nnz and (ell == 0 or ellT == 0):      # CSR staging in the tile kernels (hub graphs): exact sizes needed
            s = read_stats()
            max_nnz, max_nnzT = s["max_nnz"], s["max_nnzT"]
        elif exact_nnz:                                 # ELL staging: the CSR size of a tile is not used by any kernel
            max_nnz, max_nnzT = max_deg * tm, max_degT * tm
        else:
            max_nnz = max_nnzT = nnz_bound
        d = dict(global_only=False, nrb=nrb, ntiles=nt, tile_start=tile_start, utilisation=util, max_segment=max_segment, max_tile_rows=max_tile_rows,
                 max_nnz=max_nnz, max_nnzT=max_nnzT, ell=ell, ellT=ellT, ell_tiles=ell_tiles,
                 ellT_tiles=ellT_tiles, ell_ent_tiles=ell_ent_tiles, ellT_ent_tiles=ellT_ent_tiles)
        if not store:
            return d
        self._stats = None            # (the ELL build added the per-tile entry counts to the statistics)
        self.__dict__.update(d)
        self._tiles_built = True

    def _global_tiles(self, max_segment: int) -> None:
        """A connected component exceeds the largest LDS-resident tile (192 rows): no graph-aligned tiles exist.  The
        plain GEMMs still run on uniform 64-row tiles; the propagation hops run in global memory on the CSR
        (dss2_csr_axpy) and the edge MLP on the row-per-wave CSR kernels (networks._tagconv_forward_global)."""
        nt = -(-self.N // 64)
        tile_start = torch.empty(nt + 1, dtype=torch.int32, device=self.device)
        _lib.check(_lib.lib().dss2_tiles_uniform(tile_start.data_ptr(), nt, 64, self.N, _stream(self.device)), "dss2_tiles_uniform")
        self.__dict__.update(global_only=True, nrb=2, ntiles=nt, tile_start=tile_start, utilisation=self.N / float(nt * 64), max_tile_rows=0,
                             max_segment=max_segment, max_nnz=0, max_nnzT=0, ell=0, ellT=0, ell_tiles=None, ellT_tiles=None,
                             ell_ent_tiles=None, ellT_ent_tiles=None)
        self._tiles_built = True

    @property
    def deg_pows(self) -> torch.Tensor:
        """[N, 4] fp32, column m = A_hat^m deg (A_hat = the gcn_norm propagation matrix): the row scales of
        a bias folded through m propagations (networks._FoldPlan).  Built once per topology, in fp64."""
        if self._deg_pows is None:
            out = torch.empty(self.N, 4, dtype=torch.float32, device=self.device)
            work = torch.empty(2 * self.N, dtype=torch.float64, device=self.device)
            _lib.check(_lib.lib().dss2_deg_pows(self.rowptr.data_ptr(), self.col.data_ptr(), self.w.data_ptr(),
                                                self.deg.data_ptr(), self.N, out.data_ptr(), work.data_ptr(),
                                                _stream(self.device)), "dss2_deg_pows")
            self._deg_pows = out
        return self._deg_pows

    def lds_check(self, nmat: int, kpad: int, ncg: int) -> None:
        if self.global_only:     # plain GEMMs only: forward K=kpad -> nmat*hout columns, data-gradient K=nmat*hout -> kpad
            f = _lib.lib().dss2_gemm_prop_lds_bytes
            need = max(f(self.nrb, 1, kpad, nmat * ncg, 0, 0), f(self.nrb, 1, nmat * ncg * 32, (kpad + 31) // 32, 0, 0))
            if need > _LDS_LIMIT:
                raise NotImplementedError(f"global-memory path: a 64-row tile x K={nmat * ncg * 32} needs {need} B of LDS (> 160 KiB)")
            return
        need = _lib.lib().dss2_gemm_prop_lds_bytes(self.nrb, nmat, kpad, ncg, max(self.max_nnz, self.max_nnzT),
                                                    min(self.ell, self.ellT))
        if need > _LDS_LIMIT:
            raise NotImplementedError(f"tile of {32 * self.nrb} rows x K={kpad} needs {need} B of LDS (> 160 KiB)")


# ------------------------------------------------------------------------------------------
# cache: identity fast path (same tensor object, unmodified) -> no sync; otherwise content hash
# ------------------------------------------------------------------------------------------
_by_hash: Dict[Tuple, Topology] = {}
_last: Dict[Tuple[int, str], Tuple] = {}   # (id(tensor), mode) -> (weakref, version, data_ptr, num_nodes, topo)
_MAX_CACHE = 64


def _mode(double: Optional[bool], flip: bool = True) -> str:
    return ("ref" if double is None else ("dbl" if double else "asis")) + ("" if flip else "-noflip")


def register_topology(edge_index: torch.Tensor, num_nodes: int, topo: Topology, double: Optional[bool] = None,
                      flip: bool = True) -> Topology:
    """Attach an already built structure to this very tensor object: the next ``get_topology(edge_index, ...)`` (the
    model's forward, the loss) returns it without hashing or synchronising.  Used by dataset.DataLoader."""
    if len(_last) >= _MAX_CACHE:
        _last.pop(next(iter(_last)))
    key = (id(edge_index), _mode(double, flip))

    def _drop(ref, key=key, table=_last):   # the tensor died: release its structure (device arrays) with it
        hit = table.get(key)
        if hit is not None and hit[0] is ref:
            table.pop(key, None)
    _last[key] = (weakref.ref(edge_index, _drop), edge_index._version, edge_index.data_ptr(), int(num_nodes), topo)
    return topo


def get_topology(edge_index: torch.Tensor, num_nodes: int, double: Optional[bool] = None, flip: bool = True) -> Topology:
    """The cached structure of this batch.  double=None: the reference's doubling rule (MPN / the loss);
    False: the edge list exactly as given (standalone EdgeAggregation / TAGConv / MessagePassing.propagate)."""
    if not edge_index.is_cuda:
        raise RuntimeError("DSS2 HIP path: edge_index must live on the GPU (there is no CPU fallback)")
    mode = _mode(double, flip)
    hit = _last.get((id(edge_index), mode))
    if hit is not None:
        ref, ver, ptr, nn, topo = hit
        if ref() is edge_index and ver == edge_index._version and ptr == edge_index.data_ptr() and nn == num_nodes:
            return topo
    h1, h2, directed = probe(edge_index)
    key = (edge_index.device.index, int(num_nodes), int(edge_index.size(1)), h1, h2, mode)
    topo = _by_hash.get(key)
    if topo is None:
        topo = Topology(edge_index, num_nodes, double=(directed if double is None else double), flip=flip)
        if len(_by_hash) >= _MAX_CACHE:
            _by_hash.pop(next(iter(_by_hash)))
        _by_hash[key] = topo
    return register_topology(edge_index, num_nodes, topo, double, flip)


def clear_cache() -> None:
    _by_hash.clear()
    _last.clear()
