"""GPU (-m gpu): shapes the reference accepts and round 2 refused (missing #4): EdgeAggregation / MPN with input
widths other than (8, 6) (/root/reference/networks.py:163-174, 217-234), TAGConv with K > 3, dim_hid > 256,
MaskEmbdMPN(n_gnn_layers=1) (networks.py:408-416), mixed datasets whose parts have different bus counts (PyG's DataLoader
takes any data list, dss2_run.py:68-69) -- each against the fp64 oracle -- and the edge_attr gradient, which the kernels do not
produce and must refuse loudly instead of returning None."""
import types

import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pair(pkg, oracle, cls, args):
    torch.manual_seed(0)
    ref = getattr(oracle, cls)(*args).double()
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if p.dim() == 1:
                p.uniform_(-0.2, 0.2)
    mine = getattr(pkg, cls)(*args)
    mine.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    return ref, mine.to(DEV)


def _check(mine, ref, out, out64, tol=1e-4):
    w = torch.linspace(-1.0, 1.0, out64.numel(), dtype=torch.float64).view_as(out64)
    (out * w.float().to(DEV)).sum().backward()
    (out64 * w).sum().backward()
    assert rel_err(out, out64) < 1e-5
    for (n, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        assert p.grad is not None, n
        assert rel_err(p.grad, q.grad) < tol, n


@pytest.mark.parametrize("fn,fe,hid,dout", [(5, 4, 32, 16), (11, 3, 64, 64), (8, 6, 320, 40), (3, 0, 16, 8),
                                            (8, 9, 64, 32), (6, 16, 200, 16), (8, 17, 96, 8), (4, 32, 70, 24)])      # edge features wider than 8 (round 4: up to 32)
def test_edge_aggregation_with_other_input_widths(pkg, oracle, fn, fe, hid, dout):
    """EdgeAggregation(dim_featn, dim_feate, ...) for widths other than the reference data's (8, 6), and a hidden width above
    256 on the reference's own widths: forward, parameter gradients and the gradient of x."""
    b = pkg.synthetic.make_batch(["cigre14"], 9, seed=2)
    ref, mine = _pair(pkg, oracle, "EdgeAggregation", (fn, fe, hid, dout))
    torch.manual_seed(1)
    x = torch.randn(b["x"].shape[0], fn)
    ea = torch.randn(b["edge_index"].shape[1], fe)
    ei2, ea2 = oracle.undirect_graph(b["edge_index"], ea.double()) if fe >= 3 else (
        torch.cat([b["edge_index"], b["edge_index"].flip(0)], 1), torch.cat([ea, ea], 0).double())
    x64 = x.double().requires_grad_(True)
    xm = x.to(DEV).requires_grad_(True)
    out64 = ref(x64, ei2, ea2)
    out = mine(xm, ei2.to(DEV), ea2.float().to(DEV))
    _check(mine, ref, out, out64)
    assert rel_err(xm.grad, x64.grad) < 1e-4


@pytest.mark.parametrize("cls,args", [
    ("MPN", (5, 4, 2, 32, 3, 2, 0.0)),            # other input widths: the per-layer path
    ("SkipMPN", (5, 4, 5, 48, 2, 2, 0.0)),
    ("PFN", (5, 4, 3, 32, 2, 2, 0.0, 2)),
    ("MPN", (8, 6, 2, 320, 3, 2, 0.0)),           # dim_hid > 256
    ("MPN", (8, 6, 2, 32, 3, 4, 0.0)),            # K = 4: one plain GEMM + four propagation hops per layer
    ("MPN", (8, 6, 2, 64, 2, 5, 0.0)),            # K = 5
    ("MPN", (8, 12, 2, 64, 3, 2, 0.0)),           # twelve edge features (the reference is width-agnostic, networks.py:163-174)
    ("SkipPFN", (8, 20, 2, 32, 2, 2, 0.0, 2)),    # twenty, in a stack
])
def test_mpn_family_beyond_the_reference_data_shapes(pkg, oracle, cls, args):
    b = pkg.synthetic.make_batch(["cigre14", "cigre14_reswitched"], 10, seed=5)
    ref, mine = _pair(pkg, oracle, cls, args)
    fn, fe = args[0], args[1]
    torch.manual_seed(2)
    x = torch.randn(b["x"].shape[0], fn)
    ea = torch.randn(b["edge_index"].shape[1], fe)
    out64 = ref(x.double(), b["edge_index"], ea.double())
    out = mine(x.to(DEV), b["edge_index"].to(DEV), ea.to(DEV))
    _check(mine, ref, out, out64, tol=max(1e-4, 8.0 / x.shape[0]))


@pytest.mark.parametrize("hin,hout,K", [(32, 32, 4), (64, 24, 5), (16, 2, 6), (32, 32, 0)])
def test_tagconv_with_any_k(pkg, oracle, hin, hout, K):
    """PyG TAGConv(in, out, K) for K > 3 (the fused kernels hold K + 1 <= 4 matrices): same numbers from the plain-GEMM +
    propagation-hop path, forward and backward."""
    b = pkg.synthetic.make_batch(["cigre14"], 7, seed=3)
    ei2 = torch.cat([b["edge_index"], b["edge_index"].flip(0)], 1)
    ref, mine = _pair(pkg, oracle, "TAGConv", (hin, hout, K))
    torch.manual_seed(4)
    x = torch.randn(b["x"].shape[0], hin)
    x64 = x.double().requires_grad_(True)
    xm = x.to(DEV).requires_grad_(True)
    out64 = ref(x64, ei2)
    out = mine(xm, ei2.to(DEV))
    _check(mine, ref, out, out64)
    assert rel_err(xm.grad, x64.grad) < 1e-4


def test_maskembdmpn_with_one_layer(pkg, oracle):
    """networks.py:408-416: n_gnn_layers == 1 builds TWO dim_hid -> dim_out convs; that only runs for dim_out == dim_hid."""
    args = (8, 6, 32, 32, 1, 2, 0.0)
    ref, mine = _pair(pkg, oracle, "MaskEmbdMPN", args)
    assert list(mine.state_dict().keys()) == list(ref.state_dict().keys())
    b = pkg.synthetic.make_batch(["cigre14"], 6, seed=1)
    N = b["x"].shape[0]
    torch.manual_seed(3)
    dx = torch.cat([torch.zeros(N, 4), b["x"][:, :8], (torch.rand(N, 8) > 0.5).float()], 1)
    data64 = types.SimpleNamespace(x=dx.double(), edge_index=b["edge_index"], edge_attr=b["edge_attr"][:, :6].double())
    data = types.SimpleNamespace(x=dx.to(DEV), edge_index=b["edge_index"].to(DEV), edge_attr=b["edge_attr"][:, :6].to(DEV))
    _check(mine, ref, mine(data), ref(data64))
    with pytest.raises(ValueError):
        pkg.MaskEmbdMPN(8, 6, 2, 32, 1, 2, 0.0)


def test_edge_attr_gradient_is_refused_not_dropped(pkg):
    b = pkg.synthetic.make_batch(["cigre14"], 4, seed=1)
    x, ei = b["x"][:, :8].to(DEV), b["edge_index"].to(DEV)
    ea = b["edge_attr"][:, :6].to(DEV).requires_grad_(True)
    for m in (pkg.MPN(8, 6, 2, 32, 3, 2, 0.0), pkg.MPN(8, 6, 2, 64, 2, 2, 0.0), pkg.SkipPFN(8, 6, 2, 32, 2, 2, 0.0, 2)):
        with pytest.raises(NotImplementedError, match="edge_attr"):
            m.to(DEV)(x, ei, ea)
    with torch.no_grad():                      # no gradient asked for: runs
        pkg.MPN(8, 6, 2, 32, 3, 2, 0.0).to(DEV)(x, ei, ea)


def test_mixed_dataset_with_different_bus_counts(pkg, oracle):
    """A data list mixing 15-bus CIGRE samples and 70-bus ober_sub samples: collated like PyG's DataLoader would (node offsets
    per slot), structure built without a hint; the model and the loss run on the batch and match the oracle on it."""
    ds_mod = pkg.dataset
    parts = [ds_mod.DeviceDataset.from_batch(pkg.synthetic.make_batch([grid], S, seed=3), device=DEV)
             for grid, S in (("cigre14", 6), ("ober_sub", 4))]
    mixed = ds_mod.MixedDataset(parts)
    assert mixed.n is None and len(mixed) == 10
    loader = ds_mod.DataLoader(mixed, batch_size=10, shuffle=False)
    batch = next(iter(loader))
    assert batch.x.shape[0] == 6 * 15 + 4 * 70 and batch.edge_index.shape[1] == 6 * 14 + 4 * 69
    # PyG collation = concatenation with the node offset added to edge_index
    ei = batch.edge_index.cpu()
    off = 0
    for k, (n, e) in enumerate([(15, 14)] * 6 + [(70, 69)] * 4):
        seg = ei[:, sum(([14] * 6 + [69] * 4)[:k]):sum(([14] * 6 + [69] * 4)[:k + 1])]
        assert int(seg.min()) >= off and int(seg.max()) < off + n
        off += n
    ref, mine = _pair(pkg, oracle, "MPN", (8, 6, 2, 64, 3, 2, 0.0))
    out = mine(batch.x[:, :8], batch.edge_index, batch.edge_attr[:, :6])
    out64 = ref(batch.x[:, :8].double().cpu(), ei, batch.edge_attr[:, :6].double().cpu())
    _check(mine, ref, out, out64, tol=max(1e-4, 8.0 / batch.x.shape[0]))


@pytest.mark.parametrize("cls,args", [("MultiMPN", (8, 6, 2, 32, 2, 2, 0.0)), ("MPN", (7, 5, 2, 32, 2, 2, 0.0))])
def test_general_route_output_survives_the_in_place_mask_of_the_loss(pkg, oracle, cls, args):
    """gsp_wls_edge zeroes theta at the slack buses IN PLACE on the model output (/root/reference/data.py:413).  The per-layer
    autograd nodes of the general route used to save their un-gated last output for a gate they do not have, and autograd then
    refused the backward ("modified by an inplace operation"): model -> loss -> backward must run on every route, and match
    the fp64 oracle."""
    ref, mine = _pair(pkg, oracle, cls, args)
    b = pkg.synthetic.make_batch(["cigre14"], 8, seed=5)
    x, ei, ea = b["x"].to(DEV), b["edge_index"].to(DEV), b["edge_attr"].to(DEV)
    st = tuple(s.to(DEV) for s in b["stats"])
    fn, fe = args[0], args[1]
    if cls == "MultiMPN":
        out = mine(types.SimpleNamespace(x=x[:, :8].contiguous(), edge_index=ei, edge_attr=ea[:, :6].contiguous()))
        out64 = ref(types.SimpleNamespace(x=b["x"][:, :8].double(), edge_index=b["edge_index"], edge_attr=b["edge_attr"][:, :6].double()))
    else:
        out = mine(x[:, :fn].contiguous(), ei, ea[:, :fe].contiguous())
        out64 = ref(b["x"][:, :fn].double(), b["edge_index"], b["edge_attr"][:, :fe].double())
    kw = dict(reg_coefs=oracle.DEFAULT_REG_COEFS, num_samples=None)
    loss = pkg.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=out, x_mean=st[0], x_std=st[1], edge_mean=st[2],
                            edge_std=st[3], edge_index=ei, node_param=x[:, 8:], edge_param=ea[:, 6:], **kw)
    loss.backward()
    x64, ea64 = b["x"].double(), b["edge_attr"].double()
    st64 = tuple(s.double() for s in b["stats"])
    loss64 = oracle.gsp_wls_edge(input=x64[:, :8], edge_input=ea64[:, :6], output=out64, x_mean=st64[0], x_std=st64[1],
                                 edge_mean=st64[2], edge_std=st64[3], edge_index=b["edge_index"], node_param=x64[:, 8:],
                                 edge_param=ea64[:, 6:], **kw)
    loss64.backward()
    assert abs(loss.item() - loss64.item()) <= 1e-5 * abs(loss64.item())
    assert (out.detach()[x[:, 9] > 0, 1] == 0).all()
    for (n, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        assert p.grad is not None and rel_err(p.grad, q.grad) < 1e-4, n
