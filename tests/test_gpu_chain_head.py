"""GPU: the narrow head TAGConv fused into the chained launches (dss2_gemm_prop_chain_head, csrc/dss2_gemm_chain_sp.hip;
/root/reference/networks.py:266-275: the last TAGConv(dim_hid, dim_out) of MPN / SkipMPN) against the same model with the head as
launches of its own -- forward fusion (DSS2_CHAIN_HEAD_FWD), backward fusion (DSS2_CHAIN_HEAD), both on by default since round 5 --
and against the fp64 oracle."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(pkg, model, b, fwd, bwd, seed):
    nw = pkg.networks
    saved = pkg.flags.CHAIN_HEAD_FWD, pkg.flags.CHAIN_HEAD
    pkg.flags.CHAIN_HEAD_FWD, pkg.flags.CHAIN_HEAD = fwd, bwd
    try:
        for p in model.parameters():
            p.grad = None
        torch.manual_seed(seed)          # (the dropout snapshot is drawn from torch's generator)
        x = b["x"][:, :8].to(DEV).requires_grad_(True)
        out = model(x, b["edge_index"].to(DEV), b["edge_attr"][:, :6].to(DEV))
        w = torch.linspace(-1.0, 1.0, out.numel(), device=DEV).view_as(out)
        (out * w).sum().backward()
        return out.detach().clone(), x.grad.clone(), [p.grad.clone() for p in model.parameters()]
    finally:
        pkg.flags.CHAIN_HEAD_FWD, pkg.flags.CHAIN_HEAD = saved


@pytest.mark.parametrize("cls,args,B", [
    ("MPN", (8, 6, 2, 128, 4, 2, 0.0), 300),          # the C2 model
    ("MPN", (8, 6, 3, 128, 4, 2, 0.3), 257),          # dim_out 3, dropout mask on the head's input, odd tile count
    ("MPN", (8, 6, 4, 96, 3, 1, 0.3), 200),           # K = 1, dim_out 4, three column groups
    ("MPN", (8, 6, 2, 256, 5, 2, 0.0), 70),           # eight-wave instantiation
])
def test_fused_head_equals_separate_launches(pkg, cls, args, B):
    b = pkg.synthetic.make_batch(["cigre14"], B, seed=4)
    torch.manual_seed(0)
    model = getattr(pkg, cls)(*args).to(DEV)
    topo = pkg.topology.get_topology(b["edge_index"].to(DEV), b["x"].shape[0])
    nw = pkg.networks
    assert nw.chain_head_supported(topo, args[5] + 1, args[3], args[2], False) and nw.chain_head_supported(topo, args[5] + 1, args[3], args[2], True)
    ref = _run(pkg, model, b, False, False, 7)
    for fwd, bwd in ((True, False), (False, True), (True, True)):
        got = _run(pkg, model, b, fwd, bwd, 7)
        again = _run(pkg, model, b, fwd, bwd, 7)
        assert torch.equal(got[0], again[0]) and torch.equal(got[1], again[1])
        assert rel_err(got[0], ref[0]) < 2e-6 and rel_err(got[1], ref[1]) < 1e-5, (fwd, bwd)
        head_bias = f"convs.{args[4] - 1}.bias"
        for g, r, (n, _) in zip(got[2], ref[2], model.named_parameters()):
            if n == head_bias and bwd:
                # the head's bias gradient is the column sum of the upstream gradient -- here linspace(-1, 1), which cancels to +-0.5 out of a
                # sum of magnitudes of N / 2: two fp32 summation orders (the narrow weight-gradient launch / the fused per-tile sums of round 5)
                # agree to eps x that sum of magnitudes, not to 1e-5 of the result
                assert (g - r).abs().max().item() <= 1e-7 * b["x"].shape[0] / 2, (fwd, bwd, n)
                continue
            assert rel_err(g, r) < 1e-5, (fwd, bwd, n)


@pytest.mark.parametrize("grid,args,B", [
    ("ober_sub", (8, 6, 2, 128, 4, 2, 0.0), 41),        # the C3 model on 96-row tiles, odd tile count
    ("ober_sub", (8, 6, 3, 128, 4, 2, 0.3), 20),        # dim_out 3, dropout mask on the head's input
    ("ober_sub", (8, 6, 2, 128, 4, 2, 0.3), 23),        # dim_out 2 with dropout: the head's weight gradient reads the masked rows
    ("ober179", (8, 6, 1, 128, 3, 2, 0.3), 7),          # dim_out 1 on 192-row tiles
    ("ober179", (8, 6, 2, 128, 4, 2, 0.0), 12),         # 192-row tiles (three rows per lane in the head's hop phase)
    ("ober179", (8, 6, 4, 96, 3, 1, 0.3), 9),           # K = 1, dim_out 4, three column groups
])
def test_tall_tile_backward_head_equals_separate_launches(pkg, grid, args, B):
    """96- / 192-row tiles: only the BACKWARD head rides in the chain (gemm_chain_sp6_kernel<NRB, NMAT, 2, 2>: the data-gradient
    chain's input tile built in its staging, and with it the head's weight gradient for nout <= 2); the forward head keeps its
    launch whatever the switch says."""
    b = pkg.synthetic.make_batch([grid], B, seed=4)
    torch.manual_seed(0)
    model = pkg.MPN(*args).to(DEV)
    topo = pkg.topology.get_topology(b["edge_index"].to(DEV), b["x"].shape[0])
    nw = pkg.networks
    assert topo.nrb in (3, 6)
    assert nw.chain_head_supported(topo, args[5] + 1, args[3], args[2], True) and not nw.chain_head_supported(topo, args[5] + 1, args[3], args[2], False)
    ref = _run(pkg, model, b, False, False, 7)
    got = _run(pkg, model, b, True, True, 7)
    again = _run(pkg, model, b, True, True, 7)
    assert torch.equal(got[0], ref[0])                                   # the forward is the same launches
    assert torch.equal(got[1], again[1])
    assert rel_err(got[1], ref[1]) < 1e-5
    # ... and, for heads up to 2 wide, the head's WEIGHT gradient from the same staging (one slab per tile) instead of the narrow launch
    assert bool(pkg.ops.chain_head_wgrad_supported(topo, args[5] + 1, args[3], args[2])) == (args[2] <= 2)
    head_bias = f"convs.{args[4] - 1}.bias"
    for g, r, (n, _) in zip(got[2], ref[2], model.named_parameters()):
        if n == head_bias and args[2] <= 2:
            # cancelling column sums of linspace(-1, 1) (see test_fused_head_equals_separate_launches): +-0.5001 out of a sum of magnitudes of
            # N / 2; the two launches' fp32 summation orders each carry a few eps of THAT sum (observed: 1.8e-4 apart at N = 2148, the exact
            # value 0.500116 between them)
            assert (g - r).abs().max().item() <= 4e-7 * b["x"].shape[0] / 2, n
            exact = b["x"].shape[0] / (2.0 * b["x"].shape[0] - 1.0) if args[2] == 2 else None      # column 1 of linspace(-1, 1, 2 N) laid out [N, 2]
            if exact is not None:
                assert abs(abs(g[1].item()) - exact) <= 4e-7 * b["x"].shape[0] / 2, (g, exact)
            continue
        assert rel_err(g, r) < 1e-5, n


def test_fused_head_against_the_oracle(pkg, oracle):
    """MPN(H = 128, L = 4) with both fusions on against the fp64 oracle, gates pinned by the output tolerance only (p = 0)."""
    b = pkg.synthetic.make_batch(["cigre14"], 64, seed=9)
    torch.manual_seed(0)
    ref = oracle.MPN(8, 6, 2, 128, 4, 2, 0.0).double()
    mine = pkg.MPN(8, 6, 2, 128, 4, 2, 0.0)
    mine.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    mine = mine.to(DEV)
    out64 = ref(b["x"][:, :8].double(), b["edge_index"], b["edge_attr"][:, :6].double())
    nw = pkg.networks
    saved = pkg.flags.CHAIN_HEAD_FWD, pkg.flags.CHAIN_HEAD
    pkg.flags.CHAIN_HEAD_FWD, pkg.flags.CHAIN_HEAD = True, True
    try:
        out = mine(b["x"][:, :8].to(DEV), b["edge_index"].to(DEV), b["edge_attr"][:, :6].to(DEV))
        w = torch.linspace(-1.0, 1.0, out64.numel(), dtype=torch.float64).view_as(out64)
        (out * w.float().to(DEV)).sum().backward()
        (out64 * w).sum().backward()
    finally:
        pkg.flags.CHAIN_HEAD_FWD, pkg.flags.CHAIN_HEAD = saved
    assert rel_err(out, out64) < 1e-5
    for (n, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        assert rel_err(p.grad, q.grad) < max(1e-4, 8.0 / b["x"].shape[0]), n


def test_wide_heads_keep_their_own_launches(pkg):
    """SkipMPN's head is dim_featn = 8 wide (networks.py:336): beyond the fused head's 4 outputs, so the switch changes nothing."""
    b = pkg.synthetic.make_batch(["cigre14"], 100, seed=4)
    torch.manual_seed(0)
    model = pkg.SkipMPN(8, 6, 8, 128, 4, 2, 0.3).to(DEV)
    topo = pkg.topology.get_topology(b["edge_index"].to(DEV), b["x"].shape[0])
    assert not pkg.networks.chain_head_supported(topo, 3, 128, 8, False)
    ref, got = _run(pkg, model, b, False, False, 3), _run(pkg, model, b, True, True, 3)
    assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])
