"""GPU: the f16x3 weight gradient (csrc/dss2_wgrad16h.hip, round 5) -- TAGConv's dW_m = ((P^T)^m G)^T X, db = colsum(G) on the fp16
matrix pipe: two fp16 pieces per fp32 operand after a power-of-two scale, three MFMAs per product.  Reference mathematics:
torch_geometric TAGConv's autograd through /root/reference/networks.py:211-264 (MPN.forward); the oracle here is fp64 torch on the
same operands (the model-level parity tests in test_gpu_parity.py run this kernel inside the C2 step against oracle/).

* against fp64 on the C2 shape and on ragged / mixed batches, plain and folded (scaled bias sums) layers: 2e-6 of max |dW|, and within
  that of the bf16x6 kernel;
* operands whose magnitude changes by orders of magnitude from tile to tile and from layer to layer (the running scales move and
  the accumulators are rescaled), gradients of 1e-9, an all-zero batch: still 2e-6 of max |dW| -- the scale handling is exact;
* Inf / NaN in, Inf / NaN out (not silently finite);
* same inputs, same bits, call after call.
"""
import importlib

import pytest
import torch

from conftest import PKG_NAME

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def pkg():
    return importlib.import_module(PKG_NAME)


def _topo(pkg, grids, B, seed=0):
    b = pkg.synthetic.make_batch(grids, B, seed=seed)
    ei = b["edge_index"].to(DEV)
    N = b["x"].shape[0]
    return pkg.topology.get_topology(ei, N), N


def _reference(topo, G, X, nmat):
    N = topo.N
    rp, col, w = topo.rowptrT.cpu().long(), topo.colT.cpu().long(), topo.wT.cpu().double()
    rows = torch.repeat_interleave(torch.arange(N), rp[1:] - rp[:-1])
    P = torch.sparse_coo_tensor(torch.stack([rows, col]), w, (N, N)).to(DEV)
    Z, out = G.double(), []
    for m in range(nmat):
        if m:
            Z = torch.sparse.mm(P, Z)
        out.append(Z.t() @ X.double())
    return out, G.double().sum(0)


def _run(pkg, topo, Gs, Xs, H, nmat, rs2, f16):
    ops = pkg.ops
    nl = len(Gs)
    stride = nmat * H * H + H
    n_plain = nl - (1 if rs2 else 0)
    out = torch.full((max(n_plain, 1) * stride,), float("nan"), device=DEV)
    first = torch.full((stride + nmat * H,), float("nan"), device=DEV)
    kw = dict(first_rowscale2=topo.deg_pows, first_out=first) if rs2 else {}
    old = pkg.flags.WGRAD_F16
    try:
        pkg.flags.WGRAD_F16 = f16
        if nl == 1 and not rs2:
            ops.wgrad(topo, Gs[0], H, Xs[0], H, nmat, out)
        else:
            ops.wgrad_batched(topo, Gs, H, Xs, H, nmat, out, **kw)
    finally:
        pkg.flags.WGRAD_F16 = old
    torch.cuda.synchronize()
    res = []
    for l in range(nl):
        if rs2 and l == 0:
            res.append(first)
        else:
            j = l - (1 if rs2 else 0)
            res.append(out[j * stride:(j + 1) * stride])
    return res


def _uses_f16(pkg, topo, nmat, H):
    ts = pkg.ops._wgrad_tiles(topo, nmat, H, H, 1)
    return (pkg.ops._wgrad_mode(ts, nmat, 1) & 255) == 2


def _check(pkg, topo, Gs, Xs, H, nmat, rs2, tol=2e-6, against_bf16=True):
    got = _run(pkg, topo, Gs, Xs, H, nmat, rs2, True)
    other = _run(pkg, topo, Gs, Xs, H, nmat, rs2, False) if against_bf16 else None
    stride = nmat * H * H + H
    worst = 0.0
    for l in range(len(Gs)):
        dW, db = _reference(topo, Gs[l], Xs[l], nmat)
        for m in range(nmat):
            g = got[l][m * H * H:(m + 1) * H * H].view(H, H).double()
            sc = max(dW[m].abs().max().item(), 1e-300)
            err = (g - dW[m]).abs().max().item() / sc
            worst = max(worst, err)
            assert err <= tol, (l, m, err)
            if other is not None:
                o = other[l][m * H * H:(m + 1) * H * H].view(H, H).double()
                assert (g - o).abs().max().item() <= tol * sc
        gb = got[l][nmat * H * H:stride].double()
        assert (gb - db).abs().max().item() <= tol * max(db.abs().max().item(), 1e-300)
        if rs2 and l == 0:
            ext = got[l][stride:stride + nmat * H].view(nmat, H).double()
            ref = torch.stack([(Gs[0].double() * topo.deg_pows[:, m:m + 1].double()).sum(0) for m in range(nmat)])
            assert (ext - ref).abs().max().item() <= tol * ref.abs().max().item()
    return worst


@pytest.mark.parametrize("grids,B,nl,rs2,H,nmat", [
    (["cigre14"], 4096, 3, True, 128, 3),                          # C2: the folded layer + two plain ones in one launch
    (["cigre14"], 64, 3, True, 128, 3),
    (["cigre14", "cigre14_reswitched"], 333, 2, False, 128, 3),   # mixed topologies
    (["cigre14"], 7, 1, False, 128, 3),                            # fewer tiles than workgroups; the single-layer entry point
    (["cigre14"], 1000, 2, False, 64, 3),
    (["cigre14"], 500, 2, True, 128, 2),                           # K = 1
    (["cigre14"], 300, 1, False, 96, 3),
])
def test_f16x3_weight_gradient_matches_fp64_and_bf16x6(pkg, grids, B, nl, rs2, H, nmat):
    topo, N = _topo(pkg, grids, B, seed=5)
    if not _uses_f16(pkg, topo, nmat, H):
        pytest.skip("this shape does not take the f16x3 kernel")
    torch.manual_seed(11)
    Xs = [torch.relu(torch.randn(N, H, device=DEV)) * (3.0 ** l) for l in range(nl)]
    Gs = [torch.randn(N, H, device=DEV) * (10.0 ** (l - 1)) for l in range(nl)]
    _check(pkg, topo, Gs, Xs, H, nmat, rs2)


def test_f16x3_scales_follow_the_data(pkg):
    H, nmat = 128, 3
    topo, N = _topo(pkg, ["cigre14"], 2048, seed=6)
    if not _uses_f16(pkg, topo, nmat, H):
        pytest.skip("this shape does not take the f16x3 kernel")
    torch.manual_seed(12)
    g = torch.Generator(device="cpu").manual_seed(13)
    # magnitudes per GRAPH over twelve decades, increasing and decreasing along the batch (every workgroup's running scale moves)
    n_per = N // 2048
    dec = (torch.rand(2048, generator=g) * 12 - 6)
    dec[::2] = torch.linspace(-6, 6, 1024)
    rowscale = (10.0 ** dec).repeat_interleave(n_per).to(DEV)[:, None]
    X = torch.relu(torch.randn(N, H, device=DEV)) * rowscale
    G = torch.randn(N, H, device=DEV) * rowscale.flip(0) * 1e-3
    _check(pkg, topo, [G], [X], H, nmat, False, tol=3e-6)
    # tiny gradients (the loss's 1e-8 coefficients), large activations
    _check(pkg, topo, [torch.randn(N, H, device=DEV) * 1e-9], [torch.relu(torch.randn(N, H, device=DEV)) * 1e4], H, nmat, False)
    # zeros: an all-zero G, an X with zero tiles in front
    X0 = torch.relu(torch.randn(N, H, device=DEV)); X0[:N // 2] = 0
    got = _run(pkg, topo, [torch.zeros(N, H, device=DEV)], [X0], H, nmat, False, True)[0]
    assert torch.count_nonzero(got).item() == 0 and torch.isfinite(got).all()
    _check(pkg, topo, [torch.randn(N, H, device=DEV)], [X0], H, nmat, False)


def test_f16x3_propagates_inf_and_nan(pkg):
    H, nmat = 128, 3
    topo, N = _topo(pkg, ["cigre14"], 256, seed=7)
    if not _uses_f16(pkg, topo, nmat, H):
        pytest.skip("this shape does not take the f16x3 kernel")
    torch.manual_seed(14)
    X, G = torch.randn(N, H, device=DEV), torch.randn(N, H, device=DEV)
    for bad in (float("inf"), float("nan")):
        Xb = X.clone(); Xb[N // 3, 5] = bad
        got = _run(pkg, topo, [G], [Xb], H, nmat, False, True)[0]
        dW0 = got[:H * H].view(H, H)
        assert not torch.isfinite(dW0[:, 5]).all()
        Gb = G.clone(); Gb[N // 2, 9] = bad
        got = _run(pkg, topo, [Gb], [X], H, nmat, False, True)[0]
        assert not torch.isfinite(got[:H * H].view(H, H)[9]).all()
        assert not torch.isfinite(got[nmat * H * H + 9])


def test_f16x3_is_reproducible(pkg):
    H, nmat = 128, 3
    topo, N = _topo(pkg, ["cigre14"], 1024, seed=8)
    if not _uses_f16(pkg, topo, nmat, H):
        pytest.skip("this shape does not take the f16x3 kernel")
    torch.manual_seed(15)
    Xs = [torch.relu(torch.randn(N, H, device=DEV)) for _ in range(3)]
    Gs = [torch.randn(N, H, device=DEV) for _ in range(3)]
    a = _run(pkg, topo, Gs, Xs, H, nmat, True, True)
    b = _run(pkg, topo, Gs, Xs, H, nmat, True, True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
