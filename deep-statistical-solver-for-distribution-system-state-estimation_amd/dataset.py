"""Device-resident datasets and batch assembly.

Mirrors what the reference's driver does around the path, without leaving the GPU:

* ``data_from_pickles(folder, num_nfeat, num_efeat, num_nmeas, num_emeas, meas_v, meas_pflow)``
  <- /root/reference/data.py:96-205: same arguments, same five return values; the list of PyG ``Data``
  objects becomes a :class:`DeviceDataset` (indexable / sliceable like that list).  The pickles are read
  on the host (pandas, I/O only); the measurement model, the noise and the masked z-score run as HIP
  kernels (``dss2_measure_nodes/edges``, ``dss2_masked_zscore``).
* ``DataLoader(dataset, batch_size, shuffle)`` <- torch_geometric's loader as used at
  /root/reference/dss2_run.py:68-69,134: yields batches with ``.x .edge_index .edge_attr .y``; the
  collation (concatenate + edge_index offsets) is one ``dss2_collate`` launch over sample ids that never
  leave the device, so an epoch issues no H2D copy and no host sync.

For a case with a single topology (every reference data folder) the batch ``edge_index`` of a given
batch size is built once and re-used, which keeps the per-topology CSR cache (topology.py) on its
no-sync fast path.
"""
from __future__ import annotations

import ctypes as C
import os
import pickle
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import _lib
from . import topology as _topology
from .networks import _F32, _stream

NODE_COLS = ["vm_pu", "va_rad", "p_mw", "q_mvar", "vn_kv", "bool_slack", "bool_zero_inj"]
EDGE_COLS = ["from_bus", "to_bus", "p_from_mw", "q_from_mvar", "G", "B", "Gs", "Bs", "closed line", "phase shift",
             "imax or sn"]
NOISE_KEYS = ("p_noise", "v_noise", "pm_noise", "zero_inj_coef")


def _need_gpu(device) -> torch.device:
    device = torch.device("cuda" if device is None else device)
    if device.type != "cuda" or not torch.cuda.is_available():
        raise RuntimeError("dss2 dataset kernels need a HIP device (there is no CPU fallback)")
    return device


def masked_zscore(t: torch.Tensor, num_feat: int):
    """data.py:179-190 on the device, in place on the first ``num_feat`` columns.  Returns (t, mean, std)."""
    if t.dtype != _F32 or t.dim() != 2 or t.stride(1) != 1:
        raise ValueError("masked_zscore expects a 2-D fp32 tensor with unit column stride")
    L = _lib.lib()
    dev = t.device
    mean = torch.empty(num_feat, dtype=_F32, device=dev)
    std = torch.empty(num_feat, dtype=_F32, device=dev)
    scratch = torch.empty(int(L.dss2_masked_zscore_scratch_doubles(t.size(0))), dtype=torch.float64, device=dev)
    _lib.check(L.dss2_masked_zscore(t.data_ptr(), t.size(0), t.stride(0), num_feat, t.data_ptr(), t.stride(0),
                                    mean.data_ptr(), std.data_ptr(), scratch.data_ptr(), _stream(t)), "dss2_masked_zscore")
    return t, mean, std


class Batch:
    """What the driver reads from a PyG batch (dss2_run.py:134-141)."""

    def __init__(self, x, edge_index, edge_attr, y, num_graphs):
        self.x, self.edge_index, self.edge_attr, self.y, self.num_graphs = x, edge_index, edge_attr, y, num_graphs

    def to(self, device):
        if torch.device(device).type != "cuda":
            raise RuntimeError("a dss2 Batch lives on the HIP device")
        return self


class DeviceDataset:
    """All samples of one case resident in HBM: x [S, n, 11], edge_attr [S, e, 13], y [S, n, 2] fp32 and the
    edge lists [S, 2, e] int64 (one shared [2, e] list when every sample has the same topology)."""

    def __init__(self, x, edge_attr, y, edge_index, ids: Optional[torch.Tensor] = None):
        self.x, self.edge_attr, self.y, self.edge_index = x, edge_attr, y, edge_index
        self.S, self.n = int(x.size(0)), int(x.size(1))
        self.e = int(edge_attr.size(1))
        self.shared_topology = bool((edge_index == edge_index[0:1]).all().item())   # once per dataset
        self.ids = ids if ids is not None else torch.arange(self.S, device=x.device)
        self._ei_cache: Dict[int, torch.Tensor] = {}
        # host-side facts about the samples' edge lists, once per dataset: what lets a batch's graph structure be built
        # without any device-to-host copy (topology.TopologyHint)
        ei_h = (edge_index[0:1] if self.shared_topology else edge_index).cpu().numpy()
        # the hinted structure build never copies its error flag back (topology.TopologyHint): node ids are checked here, once,
        # on the host copy the degrees are counted from anyway (an id outside [0, n) would index HBM / LDS out of bounds later)
        if ei_h.size and (int(ei_h.min()) < 0 or int(ei_h.max()) >= self.n):
            raise ValueError(f"edge_index holds node ids outside [0, {self.n}) (min {int(ei_h.min())}, max {int(ei_h.max())})")
        first = ei_h[0]
        self.directed = not bool(((first[0] == first[1, 0]) & (first[1] == first[0, 0])).any())   # networks.py:236-238 on sample 0
        self._sample_directed = None if self.shared_topology else \
            ~((ei_h[:, 0, :] == ei_h[:, 1, 0:1]) & (ei_h[:, 1, :] == ei_h[:, 0, 0:1])).any(axis=1)
        deg_in = np.stack([np.bincount(g[1], minlength=self.n) for g in ei_h])
        deg_out = np.stack([np.bincount(g[0], minlength=self.n) for g in ei_h])
        self.max_degree_doubled = int((deg_in + deg_out).max())
        self.max_degree_asis = int(max(deg_in.max(), deg_out.max()))

    def hint(self, first_sample: Optional[int] = None) -> "_topology.TopologyHint":
        d = self.directed if (self._sample_directed is None or first_sample is None) else bool(self._sample_directed[first_sample])
        return _topology.TopologyHint(directed=d, nodes_per_graph=self.n,
                                      max_degree=self.max_degree_doubled if d else self.max_degree_asis,
                                      max_edges_per_graph=self.e, edges_per_graph=self.e)

    @property
    def device(self):
        return self.x.device

    def __len__(self):
        return int(self.ids.numel())

    def __getitem__(self, i):
        if isinstance(i, slice):     # data_list[a:b] (train / validation / test splits, dss2_run.py:60-66)
            sub = DeviceDataset.__new__(DeviceDataset)
            sub.__dict__.update(self.__dict__)
            sub.ids = self.ids[i]
            return sub
        s = int(self.ids[i])
        return Batch(self.x[s], self.edge_index[s], self.edge_attr[s], self.y[s], 1)

    def shuffled(self, generator: Optional[torch.Generator] = None) -> "DeviceDataset":
        """``random.shuffle(dataset)`` of dss2_run.py:59 as a view (permutation drawn on the device)."""
        sub = DeviceDataset.__new__(DeviceDataset)
        sub.__dict__.update(self.__dict__)
        sub.ids = self.ids[torch.randperm(self.ids.numel(), device=self.ids.device, generator=generator)]
        return sub

    # ---- collation
    def batch_edge_index(self, B: int, ids: torch.Tensor) -> torch.Tensor:
        if self.shared_topology and B in self._ei_cache:
            return self._ei_cache[B]
        out = torch.empty(2, B * self.e, dtype=torch.int64, device=self.device)
        self._launch([(self.edge_index, out, self.e, 1)], ids, B)
        if self.shared_topology:
            self._ei_cache[B] = out
        return out

    def _launch(self, items, ids: torch.Tensor, B: int) -> None:
        descs = (_lib.CollateDesc * len(items))()
        for d, (src, dst, chunk, kind) in zip(descs, items):
            d.src, d.dst, d.chunk, d.kind = src.data_ptr(), dst.data_ptr(), chunk, kind
            d.shared, d.nodes_per_sample = int(self.shared_topology), self.n
        _lib.check(_lib.lib().dss2_collate(C.addressof(descs), len(items), ids.data_ptr(), B,
                                           _lib.stream_ptr(self.device)), "dss2_collate")

    def collate(self, ids: torch.Tensor) -> Batch:
        """ids: int64 device tensor of sample numbers (positions in the underlying store)."""
        B = int(ids.numel())
        dev = self.device
        x = torch.empty(B * self.n, self.x.size(2), dtype=_F32, device=dev)
        ea = torch.empty(B * self.e, self.edge_attr.size(2), dtype=_F32, device=dev)
        y = torch.empty(B * self.n, self.y.size(2), dtype=_F32, device=dev)
        self._launch([(self.x, x, self.n * self.x.size(2), 0), (self.edge_attr, ea, self.e * self.edge_attr.size(2), 0),
                      (self.y, y, self.n * self.y.size(2), 0)], ids, B)
        ei = self.batch_edge_index(B, ids)
        if self.shared_topology:
            # the batch edge_index of this size is one cached tensor: its structure is built (and attached) once
            if _topology._last.get((id(ei), "ref")) is None:
                _topology.register_topology(ei, B * self.n, _topology.Topology(ei, B * self.n, hint=self.hint()))
        # (samples with individual edge lists: the first graph of the batch decides the doubling rule, which the host
        #  would have to read back; those batches take the cached / probed path of topology.get_topology)
        return Batch(x, ei, ea, y, B)

    # ---- collation into caller-owned (static) buffers: what a replayed training step reads (runner.EpochTrainer)
    def collate_descs(self, x: torch.Tensor, edge_attr: torch.Tensor, y: Optional[torch.Tensor] = None):
        """The descriptor table of ``collate_into`` for these destination buffers, built once (x [B*n, 11], edge_attr [B*e, 13],
        optionally y [B*n, 2]; fp32, contiguous).  Keep the returned object alive as long as it is used."""
        items = [(self.x, x, self.n * self.x.size(2)), (self.edge_attr, edge_attr, self.e * self.edge_attr.size(2))]
        if y is not None:
            items.append((self.y, y, self.n * self.y.size(2)))
        descs = (_lib.CollateDesc * len(items))()
        for d, (src, dst, chunk) in zip(descs, items):
            if dst.dtype != _F32 or not dst.is_contiguous() or dst.numel() % chunk:
                raise ValueError("collate_into: destination buffers must be contiguous fp32 of B whole samples")
            d.src, d.dst, d.chunk, d.kind = src.data_ptr(), dst.data_ptr(), chunk, 0
            d.shared, d.nodes_per_sample = int(self.shared_topology), self.n
        return descs

    def collate_into(self, descs, ids: torch.Tensor, B: int, cursor: Optional[torch.Tensor] = None, advance: bool = True) -> None:
        """Gather B samples straight into the buffers of ``descs`` -- no allocation, no second copy.  Without ``cursor``: samples
        ``ids[0:B]``.  With ``cursor`` (device int64[2] = {position, epoch length}): samples ``ids[(position + b) mod length]``, and
        (``advance``) the position moves forward by B on the device behind the gather -- the SAME two launches every step, so a
        recorded step (hipGraph / launch plan) that starts with this call walks an epoch by being replayed (dss2_collate_cursor)."""
        st = _lib.stream_ptr(self.device)
        if cursor is None:
            _lib.check(_lib.lib().dss2_collate(C.addressof(descs), len(descs), ids.data_ptr(), B, st), "dss2_collate")
        else:
            _lib.check(_lib.lib().dss2_collate_cursor(C.addressof(descs), len(descs), ids.data_ptr(), cursor.data_ptr(), B, int(advance), st),
                       "dss2_collate_cursor")

    def batch_structure(self, B: int):
        """(edge_index, Topology) of a B-sample batch of a single-topology dataset: built once per B, attached to the cached tensor."""
        if not self.shared_topology:
            raise ValueError("batch_structure: the samples have individual edge lists (use DataLoader / PrefetchLoader)")
        ei = self.batch_edge_index(B, self.ids[:1])
        if _topology._last.get((id(ei), "ref")) is None:
            _topology.register_topology(ei, B * self.n, _topology.Topology(ei, B * self.n, hint=self.hint()))
        return ei, _topology.get_topology(ei, B * self.n)

    @classmethod
    def from_batch(cls, batch: Dict[str, object], device=None) -> "DeviceDataset":
        """Per-sample store from a collated single-topology batch in the synthetic.make_batch layout."""
        dev = _need_gpu(device)
        S = int(batch["num_graphs"])
        x, ea, y, ei = batch["x"], batch["edge_attr"], batch["y"], batch["edge_index"]
        n, e = x.shape[0] // S, ea.shape[0] // S
        if n * S != x.shape[0] or e * S != ea.shape[0]:
            raise ValueError("from_batch needs a batch of one topology (uniform bus and branch counts)")
        ei = ei.view(2, S, e).permute(1, 0, 2) - (torch.arange(S) * n).view(S, 1, 1)
        return cls(x.view(S, n, -1).to(dev).contiguous(), ea.view(S, e, -1).to(dev).contiguous(),
                   y.view(S, n, -1).to(dev).contiguous(), ei.contiguous().to(dev))


class MixedDataset:
    """Samples of several cases with the SAME bus count but different closed-branch sets in one data list (BASELINE
    config C5: cigre14 + cigre14_reswitched, "variable edge_index per sample").  Global sample g = sample
    ``g - start[p]`` of part p.  The batch composition is decided on the HOST (a numpy permutation: which sample goes to
    which slot, hence every slot's edge offset), uploaded as three small index arrays per part, and collated by one
    ``dss2_collate_ragged`` launch per part; the batch's graph structure is then built on the device from a
    TopologyHint.  Nothing is read back: a shuffled mixed-topology epoch has no host synchronisation at all."""

    def __init__(self, parts: Sequence[DeviceDataset], ids: Optional[np.ndarray] = None):
        self.parts = list(parts)
        # n: the common bus count, or None when the parts differ (PyG's DataLoader, dss2_run.py:68-69, takes any data list):
        # such batches are collated the same way (per-slot node offsets) but their structure is built without a TopologyHint
        # (one 64-byte device-to-host copy of the build statistics per new batch)
        self.n = parts[0].n if len({p.n for p in parts}) == 1 else None
        self.start = np.concatenate([[0], np.cumsum([p.S for p in parts])]).astype(np.int64)
        self.ids = np.arange(self.start[-1], dtype=np.int64) if ids is None else np.asarray(ids, dtype=np.int64)

    @property
    def device(self):
        return self.parts[0].device

    def __len__(self):
        return int(self.ids.size)

    def __getitem__(self, i):
        if not isinstance(i, slice):
            raise TypeError("MixedDataset supports slices (train / test splits); iterate it through DataLoader")
        return MixedDataset(self.parts, self.ids[i])

    def shuffled(self, rng: Optional[np.random.Generator] = None) -> "MixedDataset":
        rng = rng or np.random.default_rng()
        return MixedDataset(self.parts, rng.permutation(self.ids))

    _RING_MAX = 64

    def _upload(self, tab: np.ndarray, dev) -> torch.Tensor:
        """Host -> device copy of a small index table through a RING of pinned staging buffers owned by the dataset.  (``pin_memory()``
        per batch asks torch's pinned allocator for a block whose previous copy has completed; when the host runs a step or more ahead
        of the GPU -- always, one batch ahead on a side stream -- that copy is still queued and every batch paid a fresh hipHostMalloc:
        ~1.4 ms of host time, round 6.)  The oldest slot is reused if its copy has run; otherwise the ring grows by one slot (the host
        never waits for the GPU here; the ring stops growing once it is as deep as the host's lead, _RING_MAX at most)."""
        ring = self.__dict__.setdefault("_ring", [])
        need = max(tab.size, 4 * 4096 + 8)
        slot = None
        if ring and ring[0][1].query() and ring[0][0].numel() >= tab.size:
            slot = ring.pop(0)
        elif len(ring) >= self._RING_MAX:
            slot = ring.pop(0)
            slot[1].synchronize()
            if slot[0].numel() < tab.size:
                slot = None
        if slot is None:
            slot = [torch.empty(need, dtype=torch.int64).pin_memory(), torch.cuda.Event()]
        buf, ev = slot
        stage = buf[:tab.size].view(tab.shape)
        stage.copy_(torch.from_numpy(tab))
        out = stage.to(dev, non_blocking=True)
        ev.record(torch.cuda.current_stream(dev))
        ring.append(slot)
        return out

    def collate(self, ids: np.ndarray) -> Batch:
        B, dev = int(ids.size), self.device
        part = np.searchsorted(self.start, ids, side="right") - 1           # part of every slot
        e_slot = np.asarray([p.e for p in self.parts], dtype=np.int64)[part]
        edge_off = np.concatenate([[0], np.cumsum(e_slot)]).astype(np.int64)
        E = int(edge_off[-1])
        n_slot = np.asarray([p.n for p in self.parts], dtype=np.int64)[part]
        node_off = np.concatenate([[0], np.cumsum(n_slot)]).astype(np.int64)
        Ntot = int(node_off[-1])
        p0 = self.parts[0]
        x = torch.empty(Ntot, p0.x.size(2), dtype=_F32, device=dev)
        y = torch.empty(Ntot, p0.y.size(2), dtype=_F32, device=dev)
        ea = torch.empty(E, p0.edge_attr.size(2), dtype=_F32, device=dev)
        ei = torch.empty(2, E, dtype=torch.int64, device=dev)
        L = _lib.lib()
        st = _lib.stream_ptr(dev)
        # ONE table for the whole batch -- per case {sample, node offset, edge offset} of its slots, then the batch's edge_ptr -- through
        # one pinned staging buffer and one host-to-device copy; ONE launch gathers every case (round 6: was a copy and a launch per case)
        n_parts = len(self.parts)
        slots_of = [np.nonzero(part == k)[0] for k in range(n_parts)]
        pieces, base, where = [], 0, []
        for k, sl in enumerate(slots_of):
            tab = np.stack([ids[sl] - self.start[k], node_off[sl], edge_off[sl]]).astype(np.int64).reshape(-1)
            where.append(base)
            pieces.append(tab)
            base += tab.size
        ptr_at = base
        pieces.append(edge_off)
        table = self._upload(np.concatenate(pieces).reshape(1, -1), dev).reshape(-1)
        edge_ptr = table[ptr_at:ptr_at + B + 1]
        if n_parts <= 4:
            descs = (_lib.CollateDesc * (4 * n_parts))()
            samp, noff, eoff, cnt = ((C.c_void_p * n_parts)() for _ in range(3)), None, None, (C.c_int64 * n_parts)()
            samp, noff, eoff = samp
            for k, p in enumerate(self.parts):
                n, c = p.n, int(slots_of[k].size)
                items = [(p.x, x, n * p.x.size(2), 0, 0, p.x.size(2)), (p.y, y, n * p.y.size(2), 0, 0, p.y.size(2)),
                         (p.edge_attr, ea, p.e * p.edge_attr.size(2), 0, 1, p.edge_attr.size(2)),
                         (p.edge_index, ei, p.e, 1, int(p.shared_topology), 0)]
                for i, (src, dst, chunk, kind, shared, width) in enumerate(items):
                    d = descs[4 * k + i]
                    d.src, d.dst, d.chunk, d.kind, d.shared, d.nodes_per_sample = src.data_ptr(), dst.data_ptr(), chunk, kind, shared, width
                t0 = table.data_ptr() + 8 * where[k]
                samp[k], noff[k], eoff[k], cnt[k] = t0, t0 + 8 * c, t0 + 16 * c, c
            _lib.check(L.dss2_collate_ragged_multi(C.addressof(descs), n_parts, 4, C.addressof(samp), C.addressof(noff), C.addressof(eoff),
                                                   C.addressof(cnt), E, st), "dss2_collate_ragged_multi")
        else:      # (more than four cases: a launch per case)
            for k, p in enumerate(self.parts):
                c = int(slots_of[k].size)
                if c == 0:
                    continue
                n = p.n
                descs = (_lib.CollateDesc * 4)()
                items = [(p.x, x, n * p.x.size(2), 0, 0, p.x.size(2)), (p.y, y, n * p.y.size(2), 0, 0, p.y.size(2)),
                         (p.edge_attr, ea, p.e * p.edge_attr.size(2), 0, 1, p.edge_attr.size(2)),
                         (p.edge_index, ei, p.e, 1, int(p.shared_topology), 0)]
                for d, (src, dst, chunk, kind, shared, width) in zip(descs, items):
                    d.src, d.dst, d.chunk, d.kind, d.shared, d.nodes_per_sample = src.data_ptr(), dst.data_ptr(), chunk, kind, shared, width
                t0 = table.data_ptr() + 8 * where[k]
                _lib.check(L.dss2_collate_ragged(C.addressof(descs), 4, t0, t0 + 8 * c, t0 + 16 * c, c, E, st), "dss2_collate_ragged")
        if self.n is None:       # different bus counts: no closed-form tiles, the general (statistics-reading) build
            _topology.register_topology(ei, Ntot, _topology.Topology(ei, Ntot))
            return Batch(x, ei, ea, y, B)
        n = self.n
        first = self.parts[int(part[0])]
        # (edge_ptr: the batch's own prefix sum of the graphs' edge counts, uploaded with the tables above: with it the structure is one
        #  launch, a wave per graph)
        directed = first.hint(int(ids[0] - self.start[part[0]])).directed
        hint = _topology.TopologyHint(
            directed=directed, nodes_per_graph=n,
            max_degree=max((p.max_degree_doubled if directed else p.max_degree_asis) for p in self.parts),
            max_edges_per_graph=max(p.e for p in self.parts), edge_ptr=edge_ptr)
        _topology.register_topology(ei, B * n, _topology.Topology(ei, B * n, hint=hint))
        return Batch(x, ei, ea, y, B)


class DataLoader:
    """torch_geometric.loader.DataLoader as the driver uses it: ``DataLoader(data_list, batch_size, shuffle)``;
    iterating yields collated batches, the last one smaller (drop_last=False)."""

    def __init__(self, dataset: DeviceDataset, batch_size: int = 1, shuffle: bool = False, drop_last: bool = False,
                 generator: Optional[torch.Generator] = None):
        self.dataset, self.batch_size, self.shuffle, self.drop_last, self.generator = dataset, int(batch_size), shuffle, drop_last, generator

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        ds = self.dataset
        if isinstance(ds, MixedDataset):     # batch composition on the host (numpy), data movement on the device
            ids = ds.ids
            if self.shuffle:
                self._rng = getattr(self, "_rng", None) or np.random.default_rng(
                    None if self.generator is None else int(self.generator.initial_seed()))
                ids = self._rng.permutation(ids)
            for a in range(0, ids.size, self.batch_size):
                b = min(a + self.batch_size, ids.size)
                if self.drop_last and b - a < self.batch_size:
                    break
                yield ds.collate(ids[a:b])
            return
        ids = ds.ids
        if self.shuffle:   # permutation drawn on the device: no host round trip
            ids = ids[torch.randperm(ids.numel(), device=ids.device, generator=self.generator)]
        n = int(ids.numel())
        for a in range(0, n, self.batch_size):
            b = min(a + self.batch_size, n)
            if self.drop_last and b - a < self.batch_size:
                break
            yield ds.collate(ids[a:b].contiguous())


class PrefetchLoader:
    """Iterates ``loader`` one batch AHEAD on a side stream: batch k + 1 is collated and its graph structure (CSR, tiles, ELL slices,
    the 32-row tiling of the weight gradient, the folded bias scales) is built while step k runs on the caller's stream.  For data whose
    structure changes per batch (BASELINE config C5: a new cigre14 / reswitched mix every step) the ~0.3 ms of small launch-bound
    kernels of the assembly leave the step's critical path (15.6 % of the C5 step before).  Hand-over is by stream events only -- no host
    synchronisation.  Memory safety: the side stream allocates batch k + 1 only after it has waited for everything the caller's stream
    had been given when batch k was handed out (the caller dropped batch k - 1 at that point at the latest), so a block the caching
    allocator recycles on the side stream is never still being read by a step.

    ``prepare(batch)``: extra work to run on the side stream (default: build every lazily built part of the batch's structure)."""

    def __init__(self, loader, prepare=None, priority: int = -1):
        """``priority``: of the side stream (-1 = high: the assembly's small launch-bound kernels are dispatched ahead of the step's
        waiting workgroups instead of behind them)."""
        self.loader, self.prepare, self.priority = loader, prepare, int(priority)
        self.side = None

    def __len__(self):
        return len(self.loader)

    @staticmethod
    def _build_structure(batch: "Batch") -> None:
        topo = _topology.get_topology(batch.edge_index, batch.x.size(0))
        if not topo.global_only:          # (touching nrb builds tiles + ELL slices)
            topo.tiles_for(1)
            topo.tiles_for(2)
        topo.deg_pows

    def __iter__(self):
        dev = self.loader.dataset.device
        main = torch.cuda.current_stream(dev)
        side = self.side = self.side or torch.cuda.Stream(device=dev, priority=self.priority)
        it = iter(self.loader)
        prep = self.prepare or self._build_structure

        def fetch():
            side.wait_stream(main)
            with torch.cuda.stream(side):
                try:
                    b = next(it)
                except StopIteration:
                    return None
                prep(b)
                ev = torch.cuda.Event()
                ev.record(side)
            return b, ev
        nxt = fetch()
        while nxt is not None:
            cur, ev = nxt
            main.wait_event(ev)
            nxt = fetch()
            yield cur


def data_from_tables(nodes: np.ndarray, edges: np.ndarray, labels: np.ndarray, noise: Dict[str, float], num_nfeat: int,
                     num_efeat: int, meas_v: Sequence[int], meas_pflow: Sequence[int], device=None,
                     z_nodes: Optional[np.ndarray] = None, z_edges: Optional[np.ndarray] = None,
                     generator: Optional[torch.Generator] = None):
    """The arithmetic of data_from_pickles on raw tables (uniform sample shapes):
    nodes [S, n, 7] (NODE_COLS), edges [S, e, 11] (EDGE_COLS, closed branches in stored order), labels [S, n, 2].
    ``z_*``: the standard-normal draws (default: drawn on the device in float64).
    Returns (DeviceDataset, x_mean, x_std, edge_mean, edge_std) like the reference (data.py:205)."""
    dev = _need_gpu(device)
    if num_nfeat != 8 or num_efeat != 6:
        raise ValueError("the measurement model produces 8 node and 6 edge features (dss2_run.py:72-75)")
    S, n, e = nodes.shape[0], nodes.shape[1], edges.shape[1]
    L = _lib.lib()
    st = _lib.stream_ptr(dev)
    f64 = torch.float64
    nd = torch.as_tensor(np.ascontiguousarray(nodes, dtype=np.float64)).to(dev).reshape(S * n, 7)
    ed = torch.as_tensor(np.ascontiguousarray(edges, dtype=np.float64)).to(dev).reshape(S * e, 11)
    zn = (torch.randn(S * n, 4, dtype=f64, device=dev, generator=generator) if z_nodes is None
          else torch.as_tensor(np.ascontiguousarray(z_nodes, dtype=np.float64)).to(dev).reshape(S * n, 4))
    ze = (torch.randn(S * e, 2, dtype=f64, device=dev, generator=generator) if z_edges is None
          else torch.as_tensor(np.ascontiguousarray(z_edges, dtype=np.float64)).to(dev).reshape(S * e, 2))
    mv = torch.zeros(n, dtype=torch.uint8)
    mv[torch.as_tensor(np.asarray(meas_v, dtype=np.int64))] = 1
    mp = torch.zeros(e, dtype=torch.uint8)
    mp[torch.as_tensor(np.asarray(meas_pflow, dtype=np.int64))] = 1
    mv, mp = mv.to(dev), mp.to(dev)
    x = torch.empty(S * n, 11, dtype=_F32, device=dev)
    ea = torch.empty(S * e, 13, dtype=_F32, device=dev)
    _lib.check(L.dss2_measure_nodes(nd.data_ptr(), mv.data_ptr(), n, zn.data_ptr(), float(noise["v_noise"]),
                                    float(noise["pm_noise"]), float(noise["p_noise"]), float(noise["zero_inj_coef"]),
                                    x.data_ptr(), S * n, st), "dss2_measure_nodes")
    _lib.check(L.dss2_measure_edges(ed.data_ptr(), mp.data_ptr(), e, ze.data_ptr(), float(noise["p_noise"]), ea.data_ptr(),
                                    S * e, st), "dss2_measure_edges")
    _, x_mean, x_std = masked_zscore(x, num_nfeat)
    _, e_mean, e_std = masked_zscore(ea, num_efeat)
    y = torch.as_tensor(np.ascontiguousarray(labels, dtype=np.float32)).to(dev)
    ei = torch.as_tensor(np.ascontiguousarray(edges[:, :, 0:2].astype(np.int64).transpose(0, 2, 1))).to(dev)
    ds = DeviceDataset(x.view(S, n, 11), ea.view(S, e, 13), y.contiguous(), ei.contiguous())
    return ds, x_mean, x_std, e_mean, e_std


def read_pickles(folder: str):
    """The reference's data folder layout (data.py:98-105): pickled lists of pandas DataFrames.  Host I/O only."""
    tabs = {}
    for name in ("nodes", "edges", "labels", "noise_param"):
        with open(os.path.join(folder, name), "rb") as fh:
            tabs[name] = pickle.load(fh)
    closed = [t[t["closed line"] == 1.0] for t in tabs["edges"]]
    if len({t.shape[0] for t in tabs["nodes"]}) != 1 or len({t.shape[0] for t in closed}) != 1:
        raise NotImplementedError("samples with different bus / closed-branch counts in one folder")
    nodes = np.stack([t[NODE_COLS].values.astype(np.float64) for t in tabs["nodes"]])
    edges = np.stack([t[EDGE_COLS].values.astype(np.float64) for t in closed])
    labels = np.stack([t.values.astype(np.float64) for t in tabs["labels"]])
    noise = {k: float(tabs["noise_param"][k].values[0]) for k in NOISE_KEYS}
    return nodes, edges, labels, noise


def data_from_pickles(folder, num_nfeat, num_efeat, num_nmeas, num_emeas, meas_v, meas_pflow, device=None,
                      generator: Optional[torch.Generator] = None):
    """/root/reference/data.py:96-205, same signature and return values; the data list is a DeviceDataset."""
    if num_nmeas != 4 or num_emeas != 2:
        raise ValueError("4 node and 2 edge measurement columns (dss2_run.py:74-75)")
    nodes, edges, labels, noise = read_pickles(folder)
    return data_from_tables(nodes, edges, labels, noise, num_nfeat, num_efeat, meas_v, meas_pflow, device=device,
                            generator=generator)
