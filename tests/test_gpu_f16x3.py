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
    (["ober_sub"], 64, 3, True, 128, 3),                           # 96-row tiles (csrc/dss2_wgrad16th.hip): C3's launch, 70 of 96 rows used
    (["ober_sub"], 41, 2, False, 128, 3),
    (["ober_sub"], 30, 2, True, 128, 2),                           # K = 1
    (["ober_sub"], 9, 1, False, 96, 3),                            # the single-layer entry point, three column groups
    (["ober179"], 12, 3, True, 128, 3),                            # 192-row tiles: six chunks, an exponent of X per chunk
    (["ober179"], 9, 2, False, 128, 2),
    (["ober_sub"], 700, 3, True, 256, 3),                          # H = 256: four (output, input) slices per tile-list slice
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


# ---------------------------------------------------------------------------------------------------------------------------------
# the layer chains as f16x3 (csrc/dss2_gemm_chain_sp.hip, MS = 2): the C2 model's three H -> H TAGConv layers forward and backward.
# The oracle comparison of this route is tests/test_gpu_parity.py (default flags take it: goldens, the full-size C2 step); here:
# the route is the one that runs, it agrees with the bf16x6 route far inside the north-star tolerance on adversarial magnitudes,
# the weight packing is exact to 2^-22, NaN stays where it belongs, same inputs give the same bits.
# ---------------------------------------------------------------------------------------------------------------------------------
REG = {"mu_v": 1e-1, "mu_theta": 1e-1, "lam_v": 1e-4, "lam_p": 1e-8, "lam_pf": 1e-6, "lam_reg": 1e2}


def _model_step(pkg, cls, cargs, grids, B, seed=0, xscale=None, wscale=None):
    torch.manual_seed(seed)
    b = pkg.synthetic.make_batch(grids, B, seed=seed, violate=0.3)
    x, ei, ea = b["x"].to(DEV), b["edge_index"].to(DEV), b["edge_attr"].to(DEV)
    if xscale is not None:
        x = x.clone(); x[:, :8] *= xscale(x.shape[0]).to(DEV)[:, None]
    st = tuple(s.to(DEV) for s in b["stats"])
    model = getattr(pkg, cls)(*cargs).to(DEV)
    if wscale is not None:
        with torch.no_grad():
            for i, p in enumerate(model.parameters()):
                p.mul_(wscale(i, p))
    params = list(model.parameters())

    def step():
        for p in params:
            p.grad = None
        out = model(x[:, :8], ei, ea[:, :6])
        loss = pkg.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=out, x_mean=st[0], x_std=st[1], edge_mean=st[2], edge_std=st[3],
                                edge_index=ei, reg_coefs=REG, num_samples=None, node_param=x[:, 8:], edge_param=ea[:, 6:])
        loss.backward(pkg.data.unit_grad(loss))
        torch.cuda.synchronize()
        return out.detach().clone(), loss.detach().clone(), [p.grad.detach().clone() for p in params]
    return model, step


def _rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-300)).item()


def _ab(pkg, step):
    old = pkg.flags.CHAIN_F16
    try:
        pkg.flags.CHAIN_F16 = False
        ref = step()
        pkg.flags.CHAIN_F16 = True
        got = step()
        again = step()
    finally:
        pkg.flags.CHAIN_F16 = old
    return ref, got, again


@pytest.mark.parametrize("cls,cargs,grids,B", [
    ("MPN", (8, 6, 2, 128, 4, 2, 0.0), ["cigre14"], 512),                          # the C2 model
    ("MPN", (8, 6, 2, 128, 4, 2, 0.0), ["cigre14", "cigre14_reswitched"], 100),    # mixed topologies, ragged last tile
    ("MPN", (8, 6, 2, 64, 5, 2, 0.0), ["cigre14"], 256),                           # H = 64 (two stripes: below the split-plane chain -> bf16x6 either way)
    ("MPN", (8, 6, 2, 256, 4, 1, 0.0), ["cigre14"], 128),                          # H = 256 (eight waves), K = 1
    ("SkipPFN", (8, 6, 2, 128, 3, 2, 0.2, 2), ["cigre14"], 200),                   # a stack with in-kernel dropout
])
def test_f16x3_chains_agree_with_the_bf16x6_chains(pkg, cls, cargs, grids, B):
    model, step = _model_step(pkg, cls, cargs, grids, B, seed=21)
    (o0, l0, g0), (o1, l1, g1), (o2, l2, g2) = _ab(pkg, step)
    mods = [m for m in model.modules() if m.__class__.__name__ in ("MPN", "SkipMPN") and getattr(m, "_plan", None) is not None]
    took = any(m._plan.f16 for m in mods)
    if cargs[3] in (128, 256):
        assert took, "the f16x3 route was expected for 64-row tiles with H a multiple of 32 >= 96"
    if cls != "SkipPFN":      # (dropout draws a new mask per call: only the dropout-free models compare call to call)
        assert _rel(o1, o0) < 2e-6 and abs(l1.item() - l0.item()) <= 2e-6 * abs(l0.item())
        for a, b_ in zip(g1, g0):
            assert _rel(a, b_) < 1e-5      # (the north star's own tolerance; the worst one is a two-element bias gradient that is a cancelling sum)
        assert torch.equal(o1, o2) and torch.equal(l1, l2) and all(torch.equal(a, b_) for a, b_ in zip(g1, g2))      # same inputs, same bits
    else:
        assert torch.isfinite(o1).all() and all(torch.isfinite(g).all() for g in g1)


def test_f16x3_chains_follow_the_scale_of_the_data(pkg):
    # node features over eight decades from graph to graph (the activation tiles' exponents differ tile by tile and layer by layer),
    # conv weights of one layer x 64, of another x 1/256 (the per-matrix weight exponents differ).  The WLS loss of such a batch is a
    # difference of 1e17-sized terms: its gradients move by PERCENT between any two correct evaluation orders (bf16x6 against the fp32
    # MFMA route: 4 %), so the yardstick is the third route: f16x3 must sit as close to the fp32-MFMA chains as bf16x6 does.
    n_per = 15
    def xscale(n):
        g = torch.Generator(device="cpu").manual_seed(5)
        dec = torch.rand((n + n_per - 1) // n_per, generator=g) * 8 - 4
        return (10.0 ** dec).repeat_interleave(n_per)[:n]
    def wscale(i, p):
        return 64.0 if i == 6 else (1.0 / 256.0 if i == 10 else 1.0)
    model, step = _model_step(pkg, "MPN", (8, 6, 2, 128, 4, 2, 0.0), ["cigre14"], 300, seed=22, xscale=xscale, wscale=wscale)
    FL = pkg.flags
    old = (FL.CHAIN_BF16, FL.CHAIN_F16)
    try:
        FL.CHAIN_BF16, FL.CHAIN_F16 = False, False
        o32, _, g32 = step()
        FL.CHAIN_BF16 = True
        ob, _, gb = step()
        FL.CHAIN_F16 = True
        oh, _, gh = step()
        assert model._plan.f16
    finally:
        FL.CHAIN_BF16, FL.CHAIN_F16 = old
    assert torch.isfinite(oh).all()
    assert _rel(oh, o32) < 2e-6 and _rel(ob, o32) < 2e-6
    for a, b_, c in zip(gh, gb, g32):
        assert _rel(a, c) <= 3.0 * _rel(b_, c) + 3e-6, (_rel(a, c), _rel(b_, c))


def test_f16x3_chain_keeps_a_bad_value_in_its_graph(pkg):
    # (a NaN among a tile's activations must not move the tile's scale: v_max_f32 ignores it.  What the NaN does to its OWN graph is the
    #  library's ReLU, a v_max_f32 with 0, in every route: it is quenched there -- torch.relu would carry it -- so only the other graphs
    #  of the tile are checked here)
    torch.manual_seed(23)
    b = pkg.synthetic.make_batch(["cigre14"], 64, seed=23)
    x, ei, ea = b["x"].to(DEV), b["edge_index"].to(DEV), b["edge_attr"].to(DEV)
    model = pkg.MPN(8, 6, 2, 128, 4, 2, 0.0).to(DEV)
    n_per = x.shape[0] // 64
    with torch.no_grad():
        clean = model(x[:, :8], ei, ea[:, :6]).clone()
        xb = x.clone(); xb[5 * n_per + 3, 2] = float("nan")      # one node of graph 5
        dirty = model(xb[:, :8], ei, ea[:, :6])
    assert model._plan.f16
    rows = torch.arange(x.shape[0], device=DEV)
    in_g5 = (rows >= 5 * n_per) & (rows < 6 * n_per)
    assert torch.isfinite(dirty[~in_g5]).all()
    assert _rel(dirty[~in_g5], clean[~in_g5]) < 1e-6


@pytest.mark.parametrize("nm,ho,hi,scales", [(3, 128, 128, (1.0, 1e-3, 300.0)), (2, 96, 96, (0.09, 0.09)), (3, 256, 256, (1.0, 0.0, 1e-20))])
def test_f16x2_weight_packing_is_exact_to_22_bits_of_every_column(pkg, nm, ho, hi, scales):
    pl = importlib.import_module(PKG_NAME + ".plans")
    g = torch.Generator(device="cpu").manual_seed(nm * 7 + ho)
    Ws = [(torch.randn(ho, hi, generator=g) * sc).to(DEV) for sc in scales]
    plan = pl._PackPlan([Ws], DEV, bf16_groups=(0,), f16=True)
    plan.refresh()
    torch.cuda.synchronize()
    for buf, transposed in ((plan.fwd16[0], True), (plan.bwd16[0], False)):      # forward: B[k][j] = W[j][k]; data gradient: B[k][j] = W[k][j]
        K, J = (hi, ho) if transposed else (ho, hi)
        nkk, ncg = (K + 15) // 16, (J + 31) // 32
        words = nm * ncg * nkk * 512
        raw = buf[:words].view(torch.int32).cpu()
        exps = buf[words:words + nm * ncg * 32].view(torch.int32).cpu().view(nm, ncg * 32).double()
        halves = raw.view(torch.int16).view(torch.float16).view(nm, ncg, nkk, 2, 64, 8).float().double()
        for m in range(nm):
            W = Ws[m].cpu().double()
            want_full = (W.t() if transposed else W)
            colmax = want_full.abs().max(dim=0).values
            for jcol in range(J):
                if colmax[jcol] > 0:
                    assert 2.0 ** 14 <= colmax[jcol].item() * 2.0 ** exps[m, jcol].item() < 2.0 ** 15, (m, jcol)
            val = halves[m, :, :, 0] + halves[m, :, :, 1]      # [ncg][nkk][64 lanes][8], still scaled
            B = torch.zeros(nkk * 16, ncg * 32, dtype=torch.float64)
            lane = torch.arange(64)
            for q in range(8):
                kk = (torch.arange(nkk)[:, None] * 16 + 8 * (lane[None, :] >> 5) + q)          # [nkk][64]
                jj = (torch.arange(ncg)[:, None, None] * 32 + (lane[None, None, :] & 31))        # [ncg][1][64]
                B[kk[None].expand(ncg, -1, -1), jj.expand(-1, nkk, -1)] = val[:, :, :, q]
            B = B * (2.0 ** (-exps[m]))[None, :]
            want = (W.t() if transposed else W)
            got = B[:K, :J]
            assert ((got - want).abs() <= 2.0 ** -22 * colmax[None, :J].clamp_min(1e-300) * 1.01).all(), m
            assert B[K:].abs().max().item() == 0 if K < nkk * 16 else True
