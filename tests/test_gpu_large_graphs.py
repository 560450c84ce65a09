"""Graphs whose connected components exceed the LDS-resident tiles (> 192 buses): the modules switch to plain tile GEMMs +
propagation hops in global memory (dss2_csr_axpy) + the row-per-wave edge kernels.  Same parity bar as the tile path."""
import pytest
import torch

from conftest import load_pkg, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def pkg():
    return load_pkg()


def _feeder_batch(n_graphs, n, seed, chords=40):
    """Radial feeders of n buses (random tree) plus a few loop-closing chords, stored one direction per branch like the
    reference's data sets (the model doubles them)."""
    g = torch.Generator().manual_seed(seed)
    parts = []
    for k in range(n_graphs):
        child = torch.arange(1, n)
        parent = (torch.rand(n - 1, generator=g) * child.float()).long().clamp(max=n - 2)
        parent = torch.minimum(parent, child - 1)
        a = torch.randint(0, n, (chords,), generator=g)
        b = (a + 1 + torch.randint(0, n - 1, (chords,), generator=g)) % n
        e = torch.cat([torch.stack([parent, child]), torch.stack([a, b])], 1)
        key = torch.minimum(e[0], e[1]) * n + torch.maximum(e[0], e[1])      # no duplicate branches
        keep = torch.zeros(e.shape[1], dtype=torch.bool)
        seen = set()
        for j, kk in enumerate(key.tolist()):
            if kk not in seen:
                seen.add(kk)
                keep[j] = True
        parts.append(e[:, keep] + k * n)
    ei = torch.cat(parts, 1)
    x = torch.randn(n_graphs * n, 8, generator=g)
    ea = torch.randn(ei.shape[1], 6, generator=g)
    return x, ei, ea


@pytest.mark.parametrize("cls,args,n,n_graphs", [
    ("MPN", (8, 6, 2, 64, 3, 2, 0.0), 300, 3),             # K = 2, narrow last layer (scalar hop lanes)
    ("SkipMPN", (8, 6, 8, 32, 2, 3, 0.0), 1000, 2),        # K = 3 (four matrices), skip connection in the last hop
    ("MPN", (8, 6, 4, 48, 2, 0, 0.0), 257, 2),             # K = 0: no hops at all
    ("SkipPFN", (8, 6, 2, 32, 3, 2, 0.0, 2), 400, 2),
])
def test_large_graphs_run_the_global_memory_path(pkg, oracle, cls, args, n, n_graphs):
    torch.manual_seed(3)
    x, ei, ea = _feeder_batch(n_graphs, n, seed=11)
    dim_out = args[2]
    ref = getattr(oracle, cls)(*args).double()
    mine = getattr(pkg, cls)(*args)
    mine.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    mine = mine.to(DEV)
    topo = pkg.topology.get_topology(ei.to(DEV), x.shape[0])
    assert topo.global_only and topo.stats()["max_segment"] == n
    g = torch.randn(x.shape[0], dim_out)
    xr = x.double().requires_grad_(True)
    outr = ref(xr, ei, ea.double())
    outr.backward(g.double())
    xm = x.to(DEV).requires_grad_(True)
    outm = mine(xm, ei.to(DEV), ea.to(DEV))
    outm.backward(g.to(DEV))
    assert rel_err(outm, outr) < 1e-5
    assert rel_err(xm.grad, xr.grad) < 1e-4
    for (nm, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        assert rel_err(p.grad, q.grad) < 1e-4, nm


def test_large_graph_dropout_uses_the_same_masks_in_the_hops(pkg, oracle):
    args = (8, 6, 2, 32, 3, 2, 0.3)
    torch.manual_seed(0)
    x, ei, ea = _feeder_batch(2, 350, seed=4)
    ref = oracle.MPN(*args).double()
    mine = pkg.MPN(*args)
    mine.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    mine = mine.to(DEV)
    xm = x.to(DEV).requires_grad_(True)
    torch.manual_seed(9)
    outm = mine(xm, ei.to(DEV), ea.to(DEV))
    snap, p = mine._last_dropout
    ref.dropout_masks = [pkg.networks.dropout_mask(snap, p, l + 1, x.shape[0], 32).cpu() for l in range(2)]
    g = torch.randn(x.shape[0], 2)
    outm.backward(g.to(DEV))
    xr = x.double().requires_grad_(True)
    outr = ref(xr, ei, ea.double())
    outr.backward(g.double())
    assert rel_err(outm, outr) < 1e-5
    assert rel_err(xm.grad, xr.grad) < 1e-4
    for (nm, q), (_, r) in zip(mine.named_parameters(), ref.named_parameters()):
        assert rel_err(q.grad, r.grad) < 1e-4, nm


def test_standalone_tagconv_on_a_large_graph(pkg, oracle):
    torch.manual_seed(1)
    x, ei, _ = _feeder_batch(1, 500, seed=2)
    ei2 = torch.cat([ei, ei.flip(0)], 1)
    ref = oracle.TAGConv(8, 12, K=3).double()
    mine = pkg.TAGConv(8, 12, K=3)
    mine.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    mine = mine.to(DEV)
    xr = x.double().requires_grad_(True)
    xm = x.to(DEV).requires_grad_(True)
    g = torch.randn(x.shape[0], 12)
    outr = ref(xr, ei2)
    outr.backward(g.double())
    outm = mine(xm, ei2.to(DEV))
    outm.backward(g.to(DEV))
    assert rel_err(outm, outr) < 1e-5 and rel_err(xm.grad, xr.grad) < 1e-4
    for (nm, q), (_, r) in zip(mine.named_parameters(), ref.named_parameters()):
        assert rel_err(q.grad, r.grad) < 1e-4, nm


@pytest.mark.parametrize("cls", ["MultiMPN", "MaskEmbdMultiMPN", "MaskEmbdMPN"])
def test_multi_variants_on_a_large_graph(pkg, oracle, cls):
    from types import SimpleNamespace
    torch.manual_seed(5)
    x, ei, ea = _feeder_batch(2, 260, seed=8)
    if cls.startswith("MaskEmbd"):      # data.x = [4 node-type columns | 8 features | 8 mask columns] (networks.py:452-455)
        gen = torch.Generator().manual_seed(3)
        x = torch.cat([torch.randn(x.shape[0], 4, generator=gen), x, (torch.rand(x.shape[0], 8, generator=gen) < 0.5).float()], 1)
    args = (8, 6, 2, 32, 3, 2, 0.0)
    ref = getattr(oracle, cls)(*args).double()
    mine = getattr(pkg, cls)(*args)
    mine.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    mine = mine.to(DEV)
    xr = x.double().requires_grad_(True)
    xm = x.to(DEV).requires_grad_(True)
    outr = ref(SimpleNamespace(x=xr, edge_index=ei, edge_attr=ea.double()))
    outm = mine(SimpleNamespace(x=xm, edge_index=ei.to(DEV), edge_attr=ea.to(DEV)))
    g = torch.randn(*outr.shape)
    outr.backward(g.double())
    outm.backward(g.to(DEV))
    assert rel_err(outm, outr) < 1e-5 and rel_err(xm.grad, xr.grad) < 1e-4
    for (nm, q), (_, r) in zip(mine.named_parameters(), ref.named_parameters()):
        assert rel_err(q.grad, r.grad) < 1e-4, nm


def test_the_fused_tile_wrappers_refuse_large_graphs_loudly(pkg):
    x, ei, _ = _feeder_batch(1, 300, seed=1)
    topo = pkg.topology.get_topology(ei.to(DEV), 300)
    w = torch.zeros(4096, device=DEV)
    with pytest.raises(NotImplementedError, match="192"):
        pkg.networks._tagconv_forward(topo, x.to(DEV), w, None, 3, 8, 8)
