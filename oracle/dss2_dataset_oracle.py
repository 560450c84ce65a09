"""CPU oracle for the dataset side of the DSS2 path: the measurement model,
the masked z-score and the batch collation that sit in front of the message-passing path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this file; only ``tests/`` may.

Restated from reading the reference (numpy float64 / torch float32 exactly where the reference uses them):

* ``measure_nodes`` / ``measure_edges``  <- /root/reference/data.py:119-141 and 144-167
  (``data_from_pickles``: measurement masks, noise model, inverse-variance features)
* ``masked_zscore``                      <- /root/reference/data.py:179-190
* ``data_from_tables``                   <- /root/reference/data.py:96-204 as a whole, on raw tables
  instead of the pandas pickles (the arithmetic is identical; the I/O is not part of the path)
* ``collate``                            <- torch_geometric ``DataLoader`` collation as the driver uses it
  (/root/reference/dss2_run.py:68-69,134): concatenate ``x / edge_attr / y`` over the samples of a
  batch, offset every sample's ``edge_index`` by the number of nodes before it.

PINNING STATUS.  The measurement model and the z-score are pinned by ``tests/golden/dataset64.npz``:
the reference's own ``data_from_pickles`` run (build container, ``tests/golden/make_goldens.py``) on
the first 64 CIGRE-14 samples with a seeded ``np.random``, together with the raw tables and the
standard-normal draws it consumed.  ``collate`` restates PyG behaviour (PyG is absent here): "parity
unpinned" for that one function; it is three concatenations and an index offset.

The one source of randomness, ``np.random.normal(loc=0, scale=|std|)`` (data.py:133,159), is made an
explicit input: ``z`` are the standard-normal draws, so that ``noise = z * |std|`` (numpy's legacy
``normal(loc, scale)`` is ``loc + scale * standard_normal`` drawn element by element in C order).
"""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import numpy as np
import torch


def measure_nodes(raw: np.ndarray, bool_slack: np.ndarray, bool_zero_inj: np.ndarray, meas_v: Sequence[int],
                  noise: Dict[str, float], z: np.ndarray) -> torch.Tensor:
    """One sample.  raw [n, 4] = (vm_pu, va_rad, p_mw, q_mvar) float64; z [n, 4] standard normal.
    Returns x [n, 8] float32 = (V, cov_V, theta, cov_theta, P, cov_P, Q, cov_Q).  data.py:119-141."""
    n = raw.shape[0]
    nodes_noises = np.array([[noise["v_noise"], noise["v_noise"], noise["pm_noise"], noise["pm_noise"]]])
    zero_inj_noises = np.array([[noise["zero_inj_coef"], noise["zero_inj_coef"]]])
    slack_noise = np.array([[noise["v_noise"], noise["zero_inj_coef"], noise["p_noise"], noise["p_noise"]]])
    mask = np.ones([n, 4]) * [0, 0, 1, 1]
    for j in meas_v:
        mask[j][0] = 1.0
    slack = np.expand_dims(bool_slack, axis=1)
    x_mean = np.multiply(raw, mask)
    x_std = x_mean * (slack_noise * slack + nodes_noises * (1 - slack))
    x = torch.tensor(x_mean + z * np.abs(x_std), dtype=torch.float32)
    x_std[:, 2:] += zero_inj_noises * np.expand_dims(bool_zero_inj, axis=1)
    x_std[:, 1:2] += slack_noise[:, 1:2] * slack
    one = torch.tensor(1, dtype=torch.float32)
    x_cov = one / torch.maximum(torch.abs(torch.tensor(x_std, dtype=torch.float32)),
                                torch.tensor(1e-6, dtype=torch.float32)) ** 2
    x_cov *= (x_cov < 1e12).type(torch.float32)
    return torch.concat([x[:, 0:1], x_cov[:, 0:1], x[:, 1:2], x_cov[:, 1:2], x[:, 2:3], x_cov[:, 2:3],
                         x[:, 3:], x_cov[:, 3:]], axis=1)


def measure_edges(pq_from: np.ndarray, gb: np.ndarray, meas_pflow: Sequence[int], noise: Dict[str, float],
                  z: np.ndarray) -> torch.Tensor:
    """One sample, closed branches only.  pq_from [e, 2] = (p_from_mw, q_from_mvar), gb [e, 2] = (G, B);
    z [e, 2].  Returns edge_attr [e, 6] float32 = (P, cov_P, Q, cov_Q, G, B).  data.py:144-167."""
    e = pq_from.shape[0]
    pflow_noises = np.array([[noise["p_noise"], noise["p_noise"]]])
    mask = np.zeros([e, 2])
    for j in meas_pflow:
        mask[j] = np.ones([1, 2])
    mean = np.multiply(pq_from, mask)
    std = mean * pflow_noises
    ea = torch.tensor(mean + z * np.abs(std), dtype=torch.float32)
    one = torch.tensor(1, dtype=torch.float32)
    cov = one / torch.maximum(torch.abs(torch.tensor(std, dtype=torch.float32)),
                              torch.tensor(1e-5, dtype=torch.float32)) ** 2
    cov *= (cov < 1e10).type(torch.float32)
    imp = torch.tensor(gb, dtype=torch.float32)
    return torch.concat([ea[:, 0:1], cov[:, 0:1], ea[:, 1:], cov[:, 1:], imp], axis=1)


def masked_zscore(t: torch.Tensor, num_feat: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """data.py:179-190: per-column mean / std over the NON-ZERO entries, z-score of those entries, zeros stay
    zero; columns >= num_feat pass through.  Returns (normalised, mean[all cols], std[all cols])."""
    mask = t != 0.0
    mean = torch.nan_to_num((t * mask).sum(dim=[0]) / mask.sum(dim=[0]))
    std = torch.nan_to_num(torch.sqrt(((t - mean) ** 2 * mask).sum(dim=[0]) / mask.sum(dim=[0])))
    out = torch.nan_to_num((t - mean) * mask / std)
    out[:, num_feat:] = t[:, num_feat:]
    return out, mean, std


def data_from_tables(nodes: np.ndarray, edges: np.ndarray, labels: np.ndarray, noise: Dict[str, float],
                     meas_v: Sequence[int], meas_pflow: Sequence[int], z_nodes: np.ndarray, z_edges: np.ndarray):
    """data.py:96-204 on raw tables (uniform sample shapes).
    nodes  [S, n, 7]  = (vm_pu, va_rad, p_mw, q_mvar, vn_kv, bool_slack, bool_zero_inj)
    edges  [S, e, 11] = (from_bus, to_bus, p_from_mw, q_from_mvar, G, B, Gs, Bs, closed, phase shift, imax or sn),
                        closed branches only, stored order
    labels [S, n, 2]  = (vm_pu, va_rad)
    Returns dict(x [S*n, 11], edge_attr [S*e, 13], y [S*n, 2], edge_index list of [2, e] int64,
    x_mean, x_std, edge_mean, edge_std) -- the tensors the reference slices its Data objects from."""
    S = nodes.shape[0]
    xs, eas, ys, eis = [], [], [], []
    for i in range(S):
        x = measure_nodes(nodes[i, :, 0:4], nodes[i, :, 5], nodes[i, :, 6], meas_v, noise, z_nodes[i])
        node_param = torch.tensor(nodes[i, :, 4:7], dtype=torch.float32)
        ea = measure_edges(edges[i, :, 2:4], edges[i, :, 4:6], meas_pflow, noise, z_edges[i])
        edge_param = torch.tensor(edges[i, :, 4:11], dtype=torch.float32)
        xs.append(torch.concat([x, node_param], axis=1))
        eas.append(torch.concat([ea, edge_param], axis=1))
        ys.append(torch.tensor(labels[i], dtype=torch.float32))
        eis.append(torch.tensor(edges[i, :, 0:2].astype(int), dtype=torch.long).t().contiguous())
    x_tensor, ea_tensor, y_tensor = torch.cat(xs, 0), torch.cat(eas, 0), torch.cat(ys, 0)
    x_set, x_mean, x_std = masked_zscore(x_tensor, 8)
    ea_set, e_mean, e_std = masked_zscore(ea_tensor, 6)
    return dict(x=x_set, edge_attr=ea_set, y=y_tensor, edge_index=eis, x_mean=x_mean[:8], x_std=x_std[:8],
                edge_mean=e_mean[:6], edge_std=e_std[:6])   # data.py:205 returns the feature columns' statistics


def collate(xs: Sequence[torch.Tensor], edge_indices: Sequence[torch.Tensor], edge_attrs: Sequence[torch.Tensor],
            ys: Sequence[torch.Tensor]):
    """PyG ``Batch.from_data_list`` as the driver's DataLoader applies it (dss2_run.py:68-69,134)."""
    off, eis = 0, []
    for x, ei in zip(xs, edge_indices):
        eis.append(ei + off)
        off += x.shape[0]
    return torch.cat(list(xs), 0), torch.cat(eis, 1), torch.cat(list(edge_attrs), 0), torch.cat(list(ys), 0)
