"""Synthetic grid batches in the reference's input layout (host side, numpy/torch CPU).

The reference ships samples only for CIGRE-14 (720 graphs); BASELINE.json's configs need
B = 4096 ... 32768 graphs on CIGRE-14, the reswitched CIGRE-14 and Oberrhein.  This module
draws bus states on the reference's own grid parameter tables (``grids.npz``, extracted from
/root/reference/data/<grid>/{bus_param,edge_param,noise_param} by tests/golden/make_goldens.py),
computes consistent branch flows / bus injections, and applies the reference's measurement
model (/root/reference/data.py:119-177: measured-quantity masks, multiplicative gaussian
noise, inverse covariances with their clamps) and its masked z-score
(/root/reference/data.py:179-190).  Result layout = what ``data_from_pickles`` +
PyG ``DataLoader`` collation hand to the hot path:

    x[N, 11]          = [V, R^-1_V, th, R^-1_th, P, R^-1_P, Q, R^-1_Q | vn_kv, slack, zero_inj]
    edge_index[2, E]   int64, stored (un-doubled) closed branches, per-graph node offsets
    edge_attr[E, 13]  = [Pf, R^-1, Qf, R^-1, G, B | G, B, Gs, Bs, closed, phase_shift, imax_or_sn]
    y[N, 2]           = [vm_pu, va_rad]

This is input generation only (fp64 numpy physics of its own); it is not the oracle and not
the product path.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))

# /root/reference/dss2_run.py:48-53
MEAS_SETS = {
    "cigre14": (np.array([0, 1, 12, 7, 11, 14]), np.array([0, 10])),
    "cigre14_reswitched": (np.array([0, 1, 12, 7, 11, 14]), np.array([0, 10])),
    "ober_sub": (np.array([35, 16, 52, 47, 6, 48, 59, 27, 37, 56]), np.array([40, 43, 11, 21, 54, 57])),
}

NOISE = dict(p_noise=0.02, v_noise=0.01, i_noise=0.01, pm_noise=0.15, sgen_noise=0.125, zero_inj_coef=0.001)


@dataclass
class Grid:
    name: str
    bus_param: np.ndarray      # [n, 3]  vn_kv, bool_slack, bool_zero_inj
    from_bus: np.ndarray       # [e]     closed branches only, stored order
    to_bus: np.ndarray         # [e]
    edge_param: np.ndarray     # [e, 7]  G, B, Gs, Bs, closed, phase shift, imax or sn
    meas_v: np.ndarray
    meas_pflow: np.ndarray

    @property
    def n(self):
        return self.bus_param.shape[0]

    @property
    def e(self):
        return self.from_bus.shape[0]


def _radial_feeder(n: int, seed: int) -> Grid:
    """Seeded random radial feeder with the Oberrhein degree mix (~{1:17%, 2:69%, 3:14%});
    stands in for the 179-bus legacy `ober2` grid whose data is not in the reference."""
    rng = np.random.default_rng(seed)
    parent = np.zeros(n, dtype=np.int64)
    deg = np.zeros(n, dtype=np.int64)
    open_ends = [0]
    for v in range(1, n):
        # mostly extend a chain end; sometimes branch off an interior degree-2 node
        if rng.random() < 0.86 or v < 4:
            u = open_ends[rng.integers(len(open_ends))]
            open_ends.remove(u)
        else:
            cand = np.nonzero(deg[:v] == 2)[0]
            u = int(cand[rng.integers(len(cand))]) if len(cand) else open_ends.pop()
        parent[v] = u
        deg[u] += 1
        deg[v] += 1
        open_ends.append(v)
        if u == 0 and deg[0] < 2 and 0 not in open_ends:
            open_ends.append(0)
    fb = parent[1:].copy()
    tb = np.arange(1, n)
    bus = np.zeros((n, 3))
    bus[:, 0] = 20.0
    bus[0, 0] = 110.0
    bus[0, 1] = 1.0
    zi = rng.choice(np.arange(1, n), size=max(1, n // 9), replace=False)
    bus[zi, 2] = 1.0
    e = n - 1
    ep = np.zeros((e, 7))
    ep[:, 0] = rng.uniform(0.5, 26.8, e)       # G
    ep[:, 1] = -rng.uniform(0.56, 20.8, e)     # B
    ep[:, 3] = rng.uniform(0.0, 2e-3, e)       # Bs (line charging)
    ep[:, 4] = 1.0
    ep[:, 6] = rng.choice([0.362, 0.421, 0.645], e)
    # branch 0 leaves the slack: make it the HV/MV transformer like ober_sub's 35->16
    ep[0, :] = [0.0139, -0.6, 0.0, 0.0, 1.0, 2.618, 25.0]
    meas_v = np.sort(rng.choice(n, size=max(2, n // 7), replace=False))
    meas_v[0] = 0
    meas_pf = np.sort(rng.choice(e, size=max(1, e // 12), replace=False))
    return Grid(f"ober{n}_synthetic", bus, fb, tb, ep, meas_v, meas_pf)


_GRIDS: Dict[str, Grid] = {}


def load_grid(name: str) -> Grid:
    """'cigre14', 'cigre14_reswitched', 'ober_sub' (real parameter tables) or 'ober179' (synthetic)."""
    if name in _GRIDS:
        return _GRIDS[name]
    if name == "ober179":
        g = _radial_feeder(179, seed=179)
    else:
        z = np.load(os.path.join(_HERE, "grids.npz"))
        bus = z[f"{name}/bus_param"].astype(np.float64)
        ep9 = z[f"{name}/edge_param"].astype(np.float64)
        closed = ep9[:, 6] == 1.0
        ep9 = ep9[closed]
        mv, mp = MEAS_SETS[name]
        g = Grid(name, bus, ep9[:, 0].astype(np.int64), ep9[:, 1].astype(np.int64), ep9[:, 2:9].copy(), mv, mp)
    _GRIDS[name] = g
    return g


def branch_flows(v, th, g: Grid, v_lv: float):
    """AC branch flows, fp64 numpy, same equations as /root/reference/data.py:370-376
    (shift = 0, scaled by V_lv^2).  v, th: [B, n]."""
    f, t = g.from_bus, g.to_bus
    G, Bb, Gs, Bs = (g.edge_param[:, k] for k in range(4))
    vi, vj, d = v[:, f], v[:, t], th[:, f] - th[:, t]
    c, s = np.cos(d), np.sin(d)
    k = v_lv ** 2
    pf = (-vi * vj * (G * c + Bb * s) + (G + Gs / 2) * vi ** 2) * k
    qf = (vi * vj * (-G * s + Bb * c) - (Bb + Bs / 2) * vi ** 2) * k
    pt = (-vi * vj * (G * c - Bb * s) + (G + Gs / 2) * vj ** 2) * k
    qt = (vi * vj * (G * s + Bb * c) - (Bb + Bs / 2) * vj ** 2) * k
    return pf, qf, pt, qt


def _raw_features(g: Grid, B: int, rng: np.random.Generator, v_lv: float, violate: float = 0.0):
    """Raw (un-normalised) x[B,n,11], edge_attr[B,e,13], y[B,n,2] for B graphs on grid g."""
    n, e = g.n, g.e
    slack = g.bus_param[:, 1]
    v = rng.uniform(0.95, 1.03, (B, n))
    th = rng.normal(-0.03, 0.01, (B, n))
    if violate > 0:  # push a fraction of graphs outside the penalty bands
        bad = rng.random((B, 1)) < violate
        v = np.where(bad, v * rng.uniform(0.8, 1.25, (B, n)), v)
        th = np.where(bad, th * rng.uniform(5.0, 40.0, (B, n)), th)
    v = np.where(slack > 0, 1.03, v)
    th = np.where(slack > 0, 0.0, th)
    pf, qf, pt, qt = branch_flows(v, th, g, v_lv)
    p = np.zeros((B, n))
    q = np.zeros((B, n))
    np.add.at(p, (slice(None), g.to_bus), -pt)
    np.add.at(p, (slice(None), g.from_bus), -pf)
    np.add.at(q, (slice(None), g.to_bus), -qt)
    np.add.at(q, (slice(None), g.from_bus), -qf)
    # ---- bus measurements (data.py:122-141)
    mask = np.ones((n, 4)) * np.array([0.0, 0.0, 1.0, 1.0])
    mask[g.meas_v, 0] = 1.0
    nodes_noise = np.array([NOISE["v_noise"], NOISE["v_noise"], NOISE["pm_noise"], NOISE["pm_noise"]])
    slack_noise = np.array([NOISE["v_noise"], NOISE["zero_inj_coef"], NOISE["p_noise"], NOISE["p_noise"]])
    zinj = g.bus_param[:, 2]
    x_mean = np.stack([v, th, p, q], -1) * mask
    coef = slack_noise * slack[:, None] + nodes_noise * (1 - slack[:, None])
    x_std = x_mean * coef
    xm = x_mean + rng.normal(0.0, 1.0, x_mean.shape) * np.abs(x_std)
    x_std[..., 2:] += NOISE["zero_inj_coef"] * zinj[:, None]
    x_std[..., 1:2] += slack_noise[1] * slack[:, None]
    cov = (1.0 / np.maximum(np.abs(x_std.astype(np.float32)), np.float32(1e-6)) ** 2).astype(np.float32)
    cov = cov * (cov < 1e12)
    x = np.zeros((B, n, 11), dtype=np.float32)
    x[..., 0:8:2] = xm
    x[..., 1:8:2] = cov
    x[..., 8:11] = g.bus_param
    # ---- branch measurements (data.py:148-167)
    emask = np.zeros((e, 2))
    emask[g.meas_pflow] = 1.0
    e_mean = np.stack([pf, qf], -1) * emask
    e_std = e_mean * NOISE["p_noise"]
    em = e_mean + rng.normal(0.0, 1.0, e_mean.shape) * np.abs(e_std)
    ecov = (1.0 / np.maximum(np.abs(e_std.astype(np.float32)), np.float32(1e-5)) ** 2).astype(np.float32)
    ecov = ecov * (ecov < 1e10)
    ea = np.zeros((B, e, 13), dtype=np.float32)
    ea[..., 0] = em[..., 0]
    ea[..., 1] = ecov[..., 0]
    ea[..., 2] = em[..., 1]
    ea[..., 3] = ecov[..., 1]
    ea[..., 4:6] = g.edge_param[:, 0:2]
    ea[..., 6:13] = g.edge_param
    y = np.stack([v, th], -1).astype(np.float32)
    return x, ea, y


def masked_zscore(t: torch.Tensor, num_feat: int):
    """/root/reference/data.py:179-190: z-score over the non-zero entries of each column; the
    trailing parameter columns are passed through.  Returns (normalised, mean[:num_feat], std[:num_feat])."""
    mask = t != 0.0
    cnt = mask.sum(dim=[0])
    mean = torch.nan_to_num((t * mask).sum(dim=[0]) / cnt)
    std = torch.nan_to_num(torch.sqrt((((t - mean) ** 2) * mask).sum(dim=[0]) / cnt))
    out = torch.nan_to_num((t - mean) * mask / std)
    out[:, num_feat:] = t[:, num_feat:]
    return out, mean[:num_feat].clone(), std[:num_feat].clone()


def make_batch(grids: Sequence[str], batch_size: int, seed: int = 0, violate: float = 0.0,
               stats: Optional[Tuple[torch.Tensor, ...]] = None) -> Dict[str, object]:
    """A collated batch of `batch_size` graphs.  `grids` with one name = single topology;
    several names = per-graph seeded uniform choice among them (BASELINE config C5 mixes
    'cigre14' and 'cigre14_reswitched').  If `stats` is None the normalisation statistics are
    computed from this batch (as data_from_pickles does from its dataset)."""
    rng = np.random.default_rng(seed)
    gl = [load_grid(nm) for nm in grids]
    v_lv = min(float(g.bus_param[:, 0].min()) for g in gl)
    choice = rng.integers(len(gl), size=batch_size) if len(gl) > 1 else np.zeros(batch_size, dtype=np.int64)
    feats = []
    for k, g in enumerate(gl):
        bk = int((choice == k).sum())
        feats.append(_raw_features(g, bk, rng, v_lv, violate) if bk else None)
    if len(gl) == 1:
        g = gl[0]
        x, ea, y = feats[0]
        off = (np.arange(batch_size, dtype=np.int64) * g.n)[:, None]
        ei = np.stack([(g.from_bus[None, :] + off).reshape(-1), (g.to_bus[None, :] + off).reshape(-1)])
        X, EA, Y = x.reshape(-1, 11), ea.reshape(-1, 13), y.reshape(-1, 2)
        graph_ptr = np.arange(batch_size + 1, dtype=np.int64) * g.n
    else:
        cursor = [0] * len(gl)
        xs, eas, ys, eis, ptr, off = [], [], [], [], [0], 0
        for b in range(batch_size):
            k = int(choice[b])
            g, i = gl[k], cursor[k]
            cursor[k] += 1
            xs.append(feats[k][0][i]); eas.append(feats[k][1][i]); ys.append(feats[k][2][i])
            eis.append(np.stack([g.from_bus + off, g.to_bus + off]))
            off += g.n
            ptr.append(off)
        X, EA, Y = np.concatenate(xs), np.concatenate(eas), np.concatenate(ys)
        ei = np.concatenate(eis, axis=1)
        graph_ptr = np.asarray(ptr, dtype=np.int64)
    xt, eat = torch.from_numpy(np.ascontiguousarray(X)), torch.from_numpy(np.ascontiguousarray(EA))
    if stats is None:
        xn, x_mean, x_std = masked_zscore(xt, 8)
        ean, e_mean, e_std = masked_zscore(eat, 6)
        stats = (x_mean, x_std, e_mean, e_std)
    else:
        x_mean, x_std, e_mean, e_std = stats
        xn = xt.clone()
        xn[:, :8] = torch.nan_to_num((xt[:, :8] - x_mean) * (xt[:, :8] != 0) / x_std)
        ean = eat.clone()
        ean[:, :6] = torch.nan_to_num((eat[:, :6] - e_mean) * (eat[:, :6] != 0) / e_std)
    return {
        "x": xn.contiguous(), "edge_index": torch.from_numpy(np.ascontiguousarray(ei)),
        "edge_attr": ean.contiguous(), "y": torch.from_numpy(np.ascontiguousarray(Y)),
        "stats": tuple(s.contiguous() for s in stats), "num_graphs": batch_size,
        "graph_ptr": torch.from_numpy(graph_ptr), "grids": list(grids),
    }


def tile_real_batch(batch: Dict[str, object], times: int) -> Dict[str, object]:
    """Repeat a collated batch `times` times (node offsets added), e.g. the 64 real CIGRE
    samples of tests/golden/cigre14_real64.npz tiled up to B=4096."""
    x, ei, ea, y = batch["x"], batch["edge_index"], batch["edge_attr"], batch["y"]
    n = x.shape[0]
    eis = torch.cat([ei + k * n for k in range(times)], dim=1)
    out = dict(batch)
    out.update(x=x.repeat(times, 1), edge_index=eis, edge_attr=ea.repeat(times, 1), y=y.repeat(times, 1),
               num_graphs=int(batch["num_graphs"]) * times)
    return out
