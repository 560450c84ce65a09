"""GPU (-m gpu): the device-side structure build (dss2_topology_probe / dss2_csr_build / dss2_tiles_* /
dss2_ell_tiles_build / dss2_deg_pows, csrc/dss2_topology.hip) against the structure oracle
(oracle/dss2_topology_oracle.py, torch index ops on the CPU): every array BIT FOR BIT -- integer work has no tolerance,
and the gcn_norm weights are two correctly rounded fp32 operations."""
import numpy as np
import pytest
import torch

import dss2_topology_oracle as topo_oracle
from conftest import load_pkg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CSR_FIELDS = ["rowptr", "col", "ent", "perm", "w", "rowptrT", "colT", "entT", "permT", "wT", "inc_rowptr", "inc_ent",
              "efrom", "eto", "deg"]
TILE_FIELDS = ["tile_start", "ell_tiles", "ellT_tiles", "ell_ent_tiles", "ellT_ent_tiles"]


@pytest.fixture(scope="module")
def pkg():
    p = load_pkg()
    p._lib.lib()
    assert torch.cuda.is_available()
    return p


def _same(a, b, name):
    if a is None or b is None:
        assert a is None and b is None, name
        return
    a = a.cpu()
    assert a.shape == b.shape and a.dtype == b.dtype, (name, a.shape, b.shape, a.dtype, b.dtype)
    if a.dtype == torch.float32:      # bitwise, including the sign of zero
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), name
    else:
        assert torch.equal(a, b), name


def _check(topo, ref, tiles=True):
    assert (topo.N, topo.E, topo.E2, topo.directed) == (ref.N, ref.E, ref.E2, ref.directed)
    for f in CSR_FIELDS:
        _same(getattr(topo, f), getattr(ref, f), f)
    if not tiles:
        return
    assert (topo.nrb, topo.ntiles, topo.ell, topo.ellT, topo.max_segment) == (ref.nrb, ref.ntiles, ref.ell, ref.ellT, ref.max_segment)
    assert abs(topo.utilisation - ref.utilisation) < 1e-12
    for f in TILE_FIELDS:
        _same(getattr(topo, f), getattr(ref, f), f)
    _same(topo.deg_pows, ref.deg_pows, "deg_pows")
    s = topo.stats()
    assert (s["max_deg"], s["max_degT"], s["max_nnz"], s["max_nnzT"], s["error"]) == (ref.max_deg, ref.max_degT, ref.max_nnz, ref.max_nnzT, 0)
    assert s["n_segments"] == len(ref.bounds) - 1 and s["min_segment"] == int(np.diff(ref.bounds).min())


def _stars(n_graphs, leaves):
    n = leaves + 1
    return torch.cat([torch.stack([torch.zeros(leaves, dtype=torch.int64), torch.arange(1, n)]) + k * n for k in range(n_graphs)], 1), n_graphs * n


@pytest.mark.parametrize("grids,B", [(["cigre14"], 9), (["cigre14", "cigre14_reswitched"], 33), (["ober_sub"], 5), (["ober179"], 3),
                                      (["cigre14"], 4096), (["cigre14", "ober_sub"], 21)])
def test_device_build_equals_the_structure_oracle(pkg, grids, B):
    """The 4 grids of BASELINE.json, the headline batch, and a batch of graphs of two different sizes (15 and 70 buses:
    non-uniform segments, greedy packing).  Reference rule for doubling (first edge), no hint: one probe + one
    statistics copy."""
    b = pkg.synthetic.make_batch(grids, B, seed=7)
    ei, N = b["edge_index"], b["x"].shape[0]
    topo = pkg.topology.Topology(ei.to(DEV), N)
    _check(topo, topo_oracle.TopologyOracle(ei, N))


def test_device_build_edge_cases(pkg):
    # hub graphs: degree above the ELL width -> CSR staging, exact per-tile entry counts
    ei, N = _stars(7, 11)
    topo = pkg.topology.Topology(ei.to(DEV), N)
    ref = topo_oracle.TopologyOracle(ei, N)
    assert ref.ell == 0 and ref.ellT == 0
    _check(topo, ref)
    assert (topo.max_nnz, topo.max_nnzT) == (ref.max_nnz, ref.max_nnzT)
    # already undirected input + an isolated node + the list used as given
    ei = torch.tensor([[0, 1, 1, 2, 4, 5], [1, 0, 2, 1, 5, 4]])
    for double in (None, False):
        topo = pkg.topology.Topology(ei.to(DEV), 7, double=double)       # nodes 3 and 6 are isolated
        ref = topo_oracle.TopologyOracle(ei, 7, double=double)
        assert topo.directed is False
        _check(topo, ref)
    # forced doubling of a list that already holds both directions (parallel edges)
    _check(pkg.topology.Topology(ei.to(DEV), 7, double=True), topo_oracle.TopologyOracle(ei, 7, double=True))
    # a component above the largest tile: the CSR part is built as always; the tile part degrades to uniform 64-row tiles
    # without graph structure (global-memory propagation path, tests/test_gpu_large_graphs.py)
    n = 400
    chain = torch.stack([torch.arange(n - 1), torch.arange(1, n)])
    topo = pkg.topology.Topology(chain.to(DEV), n)
    _check(topo, topo_oracle.TopologyOracle(chain, n), tiles=False)
    assert topo.global_only and topo.nrb == 2 and topo.ntiles == -(-n // 64) and topo.ell_tiles is None
    assert topo.tile_start.tolist() == [min(64 * i, n) for i in range(topo.ntiles + 1)]
    assert topo.stats()["max_segment"] == n
    # node ids outside [0, N) are reported, not dereferenced
    bad = pkg.topology.Topology(torch.tensor([[0, 1], [1, 9]]).to(DEV), 3)
    with pytest.raises(ValueError):
        bad.stats()


def test_probe_hash_and_directedness(pkg):
    b = pkg.synthetic.make_batch(["cigre14"], 64, seed=1)
    ei = b["edge_index"].to(DEV)
    h = pkg.topology.probe(ei)
    assert h[2] is True and h == pkg.topology.probe(ei.clone())
    und = torch.cat([ei, ei.flip(0)], 1)
    assert pkg.topology.probe(und)[2] is False
    # only the FIRST edge is inspected (networks.py:236-238): reversing any other edge keeps the batch "directed"
    ei2 = torch.cat([ei, ei[:, 5:6].flip(0)], 1)
    assert pkg.topology.probe(ei2)[2] is True
    ei3 = torch.cat([ei, ei[:, 0:1].flip(0)], 1)
    assert pkg.topology.probe(ei3)[2] is False
    swapped = ei.clone()
    swapped[:, [3, 4]] = swapped[:, [4, 3]]            # same multiset of edges, different positions -> different key
    assert pkg.topology.probe(swapped)[:2] != h[:2]
    assert pkg.topology.get_topology(ei, b["x"].shape[0]) is pkg.topology.get_topology(ei.clone(), b["x"].shape[0])


@pytest.mark.parametrize("grids,B", [(["cigre14"], 4096), (["cigre14", "cigre14_reswitched"], 512), (["ober_sub"], 37)])
def test_hinted_build_needs_no_host_sync_and_equals_the_oracle(pkg, grids, B):
    """With a TopologyHint (what dataset.DataLoader passes) nothing is copied back to the host: the whole build and a
    model step on top of it run under torch's sync-debug mode "error"."""
    b = pkg.synthetic.make_batch(grids, B, seed=3)
    ei, N = b["edge_index"], b["x"].shape[0]
    ref = topo_oracle.TopologyOracle(ei, N)
    hint = pkg.topology.TopologyHint(directed=True, nodes_per_graph=N // B, max_degree=ref.max_deg,
                                     max_edges_per_graph=max(g.e for g in (pkg.synthetic.load_grid(n) for n in grids)))
    ei_d = ei.to(DEV)
    x, ea = b["x"].to(DEV), b["edge_attr"].to(DEV)
    # (dim_hid 32: the whole-stack kernels / the fp32 fallbacks; 128: the f16x3 routes, whose headroom bits must come without a copy too)
    models = [pkg.MPN(8, 6, 2, 32, 3, 2, 0.0).to(DEV), pkg.MPN(8, 6, 2, 128, 4, 2, 0.0).to(DEV)]
    for model in models:
        model(x[:, :8], ei_d, ea[:, :6]).sum().backward()         # warm-up: plans, allocator
    pkg.topology.clear_cache()
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        topo = pkg.topology.Topology(ei_d, N, hint=hint)
        pkg.topology.register_topology(ei_d, N, topo)
        topo.nrb                                                   # builds the tile part
        for model in models:
            out = model(x[:, :8], ei_d, ea[:, :6])
            out.sum().backward()
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    assert topo._stats is None                                     # the statistics were never copied back
    for f in CSR_FIELDS + TILE_FIELDS:
        _same(getattr(topo, f), getattr(ref, f), f)
    assert (topo.nrb, topo.ntiles, topo.ell, topo.ellT) == (ref.nrb, ref.ntiles, ref.ell, ref.ellT)
    assert topo.max_nnz >= ref.max_nnz and topo.max_nnzT >= ref.max_nnzT       # upper bounds (unused with ELL slices)
    assert topo.stats()["error"] == 0


@pytest.mark.parametrize("grids,B,mode", [(["cigre14"], 4096, "uniform"), (["cigre14", "cigre14_reswitched"], 515, "edge_ptr"), (["ober_sub"], 37, "uniform"),
                                          (["ober179"], 5, "edge_ptr"), (["cigre14"], 3, "uniform-asis"), (["cigre14_reswitched"], 9, "uniform-noflip")])
def test_one_launch_build_of_equal_size_graphs_equals_the_oracle(pkg, grids, B, mode, monkeypatch):
    """Round 6: with the graphs' ranges in the stored edge list known (uniform edges_per_graph, or the batch's own edge_ptr) the whole CSR
    part AND the folded-bias row scales are ONE launch, a wave per graph (dss2_csr_build_graphs) -- every array bit for bit what the
    oracle (and the general 18-launch build) gives: row pointers, per-row order, gcn_norm weights, incidence lists, degrees, legal cuts,
    statistics, deg_pows; then tiles + ELL slices with the tile starts written by the ELL launch itself."""
    b = pkg.synthetic.make_batch(grids, B, seed=5)
    ei, N = b["edge_index"], b["x"].shape[0]
    n = N // B
    asis, noflip = mode.endswith("asis"), mode.endswith("noflip")
    if asis:      # an edge list used as given (already doubled by the caller)
        ei = torch.cat([ei, ei.flip(0)], 1)
        # (graph order must be kept: interleave per graph)
        gid = (ei[0] // n)
        ei = ei[:, torch.argsort(gid, stable=True)]
    ref = topo_oracle.TopologyOracle(ei, N, double=(False if asis else None))      # (the oracle flags reverse edges: `noflip` is compared with the general build only)
    counts = torch.bincount(ei[0] // n, minlength=B)
    assert bool((ei[1] // n == ei[0] // n).all())
    kw = dict(directed=not asis, nodes_per_graph=n, max_degree=max(ref.max_deg, ref.max_degT), max_edges_per_graph=int(counts.max()))
    if mode.startswith("uniform"):
        assert int(counts.min()) == int(counts.max())
        hint = pkg.topology.TopologyHint(edges_per_graph=int(counts[0]), **kw)
    else:
        ptr = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(counts, 0)]).to(DEV)
        hint = pkg.topology.TopologyHint(edge_ptr=ptr, **kw)
    topo = pkg.topology.Topology(ei.to(DEV), N, hint=hint, double=(False if asis else None), flip=not noflip)
    assert topo._deg_pows is not None, "the one-launch build was not taken"
    topo.nrb
    if not noflip:
        for f in CSR_FIELDS + TILE_FIELDS:
            _same(getattr(topo, f), getattr(ref, f), f)
    _same(topo.deg_pows, ref.deg_pows, "deg_pows")
    s = topo.stats()
    assert (s["max_deg"], s["max_degT"], s["error"]) == (ref.max_deg, ref.max_degT, 0)
    assert s["n_segments"] == len(ref.bounds) - 1 and s["min_segment"] == int(np.diff(ref.bounds).min()) and s["max_segment"] == ref.max_segment
    # ... and the general build of the same batch (DSS2_TOPO_GRAPHS=0) gives the same legal cuts
    monkeypatch.setenv("DSS2_TOPO_GRAPHS", "0")
    gen = pkg.topology.Topology(ei.to(DEV), N, hint=hint, double=(False if asis else None), flip=not noflip)
    assert gen._deg_pows is None
    gen.nrb
    for f in CSR_FIELDS + TILE_FIELDS + ["_lastcut", "deg_pows"]:
        _same(getattr(topo, f), (getattr(gen, f).cpu() if getattr(gen, f) is not None else None), f)
    # an edge that leaves its graph's rows is an error, not a silent wrong structure
    bad = ei.clone()
    bad[1, 0] = (bad[1, 0] + n) % N
    monkeypatch.setenv("DSS2_TOPO_GRAPHS", "1")
    t_bad = pkg.topology.Topology(bad.to(DEV), N, hint=hint, double=(False if asis else None), flip=not noflip)
    with pytest.raises(ValueError):
        t_bad.stats()
