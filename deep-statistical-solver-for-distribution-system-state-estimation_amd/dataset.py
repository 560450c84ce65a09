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
                                           torch.cuda.current_stream(self.device).cuda_stream), "dss2_collate")

    def collate(self, ids: torch.Tensor) -> Batch:
        """ids: int64 device tensor of sample numbers (positions in the underlying store)."""
        B = int(ids.numel())
        dev = self.device
        x = torch.empty(B * self.n, self.x.size(2), dtype=_F32, device=dev)
        ea = torch.empty(B * self.e, self.edge_attr.size(2), dtype=_F32, device=dev)
        y = torch.empty(B * self.n, self.y.size(2), dtype=_F32, device=dev)
        self._launch([(self.x, x, self.n * self.x.size(2), 0), (self.edge_attr, ea, self.e * self.edge_attr.size(2), 0),
                      (self.y, y, self.n * self.y.size(2), 0)], ids, B)
        return Batch(x, self.batch_edge_index(B, ids), ea, y, B)


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
        ids = ds.ids
        if self.shuffle:   # permutation drawn on the device: no host round trip
            ids = ids[torch.randperm(ids.numel(), device=ids.device, generator=self.generator)]
        n = int(ids.numel())
        for a in range(0, n, self.batch_size):
            b = min(a + self.batch_size, n)
            if self.drop_last and b - a < self.batch_size:
                break
            yield ds.collate(ids[a:b].contiguous())


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
    st = torch.cuda.current_stream(dev).cuda_stream
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
