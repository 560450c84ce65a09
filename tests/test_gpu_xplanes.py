"""GPU: X plane images (round 5) -- the H -> H layers' inputs written by their producers as the bf16x3 pieces the weight-gradient
kernel's MFMA operand reads (csrc/dss2_wgrad16p.hip; include/dss2_hip.h "X plane images").

* the image a producer writes decodes to the fp32 activation (h + m + l, every piece exactly the bf16x3 split's);
* dss2_wgrad_batched_xp on such images = the weight gradients of dss2_wgrad_batched (which splits X itself) to fp32 rounding, and
  = the fp64 reference dW_m = ((A^T)^m g)^T h to 1e-6 (/root/reference/dss2_run.py:142: autograd of networks.py:266-271);
* the ranges of the (layer, tile) list: slab ids / counts for layer boundaries inside and between workgroups;
* the whole step (MPN forward + gsp_wls_edge + backward) with and without the images: same loss and outputs bit for bit (the forward
  arithmetic does not change), gradients to fp32 rounding.
"""
import importlib
import os
import sys

import pytest
import torch

from conftest import ROOT, PKG_NAME

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def pkg():
    return importlib.import_module(PKG_NAME)


def decode_xplanes(img: torch.Tensor, ntiles: int, ncols: int) -> torch.Tensor:
    """[ntiles, 64 rows, ncols] fp64: h + m + l of every element (the layout of include/dss2_hip.h)."""
    ncb = (ncols + 31) // 32
    t = img.view(torch.bfloat16).double().view(ntiles, ncb, 4, 3, 2, 32, 8)      # tile, block, k-step, piece, k half, lane slot n, i
    v = t.sum(3)                                                                # tile, block, ks, kh, n, i
    out = torch.zeros(ntiles, 64, ncb * 32, dtype=torch.float64, device=img.device)
    n = torch.arange(32, device=img.device)
    col = ((n & 7) << 2) | (n >> 3)
    for ks in range(4):
        for kh in range(2):
            for i in range(8):
                row = 2 * ks + kh + 8 * i
                for cb in range(ncb):
                    out[:, row, 32 * cb + col] = v[:, cb, ks, kh, :, i]
    return out[:, :, :ncols]


def tile_rows(topo, X):
    """[ntiles, 64, H]: the tile-local view of a node tensor (rows beyond a tile's rows: zero)."""
    ts = topo.tile_start.cpu().tolist()
    out = torch.zeros(topo.ntiles, 64, X.size(1), dtype=torch.float64, device=X.device)
    for t in range(topo.ntiles):
        out[t, :ts[t + 1] - ts[t]] = X[ts[t]:ts[t + 1]].double()
    return out


def make_case(pkg, grids, B, H, seed=0):
    b = pkg.synthetic.make_batch(grids, B, seed=seed)
    ei = b["edge_index"].to(DEV)
    N = b["x"].shape[0]
    topo = pkg.topology.get_topology(ei, N)
    return b, ei, N, topo


@pytest.mark.parametrize("grids,B", [(["cigre14"], 64), (["cigre14", "cigre14_reswitched"], 37)])
def test_chain_layers_write_the_image_of_their_output(pkg, grids, B):
    nw, ops = pkg.networks, pkg.ops
    H, nmat = 128, 3
    b, ei, N, topo = make_case(pkg, grids, B, H)
    if not ops.xplanes_supported(topo, nmat, H):
        pytest.skip("shape not covered")
    torch.manual_seed(1)
    Ws = [[torch.randn(H, H, device=DEV) * (1.2 / H ** 0.5) for _ in range(nmat)] for _ in range(3)]
    plan = nw._PackPlan(Ws, DEV, bf16_groups=(0, 1, 2))
    plan.refresh()
    h = torch.randn(N, H, device=DEV)
    bias = [torch.randn(H, device=DEV) * 0.1 for _ in range(3)]
    ys = [torch.empty(N, H, device=DEV) for _ in range(3)]
    xps = [ops.new_xplanes(topo, H, DEV) for _ in range(3)]
    for x_ in xps:
        x_.fill_(0xFF)      # NaN patterns: every position must be overwritten
    layers = [dict(Bp=plan.fwd16[i], Y=ys[i], bias=bias[i], relu=True, x_planes=xps[i]) for i in range(3)]
    ops.gemm_prop_chain(topo, h, H, nmat, layers, b_format=1)
    # ... and the same chain without images / without the fp32 copies of the inner layers
    ys2 = [torch.empty(N, H, device=DEV) for _ in range(3)]
    ops.gemm_prop_chain(topo, h, H, nmat, [dict(Bp=plan.fwd16[i], Y=ys2[i], bias=bias[i], relu=True) for i in range(3)], b_format=1)
    y3 = torch.empty(N, H, device=DEV)
    xps3 = [ops.new_xplanes(topo, H, DEV) for _ in range(2)]
    ops.gemm_prop_chain(topo, h, H, nmat, [dict(Bp=plan.fwd16[i], Y=(y3 if i == 2 else None), bias=bias[i], relu=True,
                                                 x_planes=(xps3[i] if i < 2 else None)) for i in range(3)], b_format=1)
    torch.cuda.synchronize()
    for i in range(3):
        assert torch.equal(ys[i], ys2[i]), f"layer {i}: writing the image changed the layer's output"
        dec = decode_xplanes(xps[i], topo.ntiles, H)
        assert torch.isfinite(dec).all(), "a position of the image was not written (or not finite)"
        ref = tile_rows(topo, ys[i])
        ts = topo.tile_start.cpu().tolist()
        for t in range(topo.ntiles):
            R = ts[t + 1] - ts[t]
            err = (dec[t, :R] - ref[t, :R]).abs().max().item()
            scale = ref[t, :R].abs().max().item() + 1e-30
            assert err <= 2.0 ** -23 * scale, (i, t, err, scale)      # h + m + l = v up to the third piece's rounding (2^-25 |v|)
    assert torch.equal(y3, ys[2])
    for i in range(2):
        assert torch.equal(xps3[i], xps[i]), "the image must not depend on whether the fp32 copy is written"


def test_edge_mlp_writes_the_image_of_S(pkg):
    nw, ops = pkg.networks, pkg.ops
    H = 128
    b, ei, N, topo = make_case(pkg, ["cigre14"], 96, H)
    if not ops.xplanes_supported(topo, 3, H):
        pytest.skip("shape not covered")
    torch.manual_seed(2)
    x, ea = b["x"][:, :8].contiguous().to(DEV), b["edge_attr"][:, :6].contiguous().to(DEV)
    W1, b1 = torch.randn(H, 22, device=DEV) * 0.3, torch.randn(H, device=DEV) * 0.1
    xp = ops.new_xplanes(topo, H, DEV)
    xp.fill_(0xFF)
    S, _ = nw._edge_aggr_forward(topo, x, 8, ea, 6, W1, b1, None, None, H, H, 8, 6, second_linear=False, xp=xp)
    S0, _ = nw._edge_aggr_forward(topo, x, 8, ea, 6, W1, b1, None, None, H, H, 8, 6, second_linear=False)
    torch.cuda.synchronize()
    assert torch.equal(S, S0)
    dec = decode_xplanes(xp, topo.ntiles, H)
    assert torch.isfinite(dec).all()
    ref = tile_rows(topo, S)
    ts = topo.tile_start.cpu().tolist()
    for t in range(topo.ntiles):
        R = ts[t + 1] - ts[t]
        assert (dec[t, :R] - ref[t, :R]).abs().max().item() <= 2.0 ** -23 * (ref[t, :R].abs().max().item() + 1e-30)


def _reference_wgrad(topo, G, X, nmat):
    """fp64: dW_m = ((A^T)^m G)^T X, db = colsum(G); A^T from the transposed CSR."""
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


@pytest.mark.parametrize("grids,B,nl,rs2", [(["cigre14"], 64, 3, True), (["cigre14"], 1000, 3, True), (["cigre14", "cigre14_reswitched"], 333, 2, False),
                                            (["cigre14"], 7, 1, False), (["cigre14"], 4096, 3, True)])
def test_wgrad_from_images_matches_the_fp64_reference_and_the_splitting_kernel(pkg, grids, B, nl, rs2):
    nw, ops = pkg.networks, pkg.ops
    H, nmat = 128, 3
    b, ei, N, topo = make_case(pkg, grids, B, H, seed=3)
    if not ops.xplanes_supported(topo, nmat, H):
        pytest.skip("shape not covered")
    torch.manual_seed(4)
    # images through a real producer: a chain of nl identity-free layers whose outputs are the X of the weight gradients
    Ws = [[torch.randn(H, H, device=DEV) * (1.2 / H ** 0.5) for _ in range(nmat)] for _ in range(nl)]
    plan = nw._PackPlan(Ws, DEV, bf16_groups=tuple(range(nl)))
    plan.refresh()
    h = torch.randn(N, H, device=DEV)
    Xs = [torch.empty(N, H, device=DEV) for _ in range(nl)]
    xps = [ops.new_xplanes(topo, H, DEV) for _ in range(nl)]
    if nl >= 2:
        ops.gemm_prop_chain(topo, h, H, nmat, [dict(Bp=plan.fwd16[i], Y=Xs[i], relu=True, x_planes=xps[i]) for i in range(nl)], b_format=1)
    else:      # (a chain needs two layers: run two, use the first)
        Ws2 = Ws + [[torch.randn(H, H, device=DEV) * 0.1 for _ in range(nmat)]]
        plan2 = nw._PackPlan(Ws2, DEV, bf16_groups=(0, 1))
        plan2.refresh()
        ops.gemm_prop_chain(topo, h, H, nmat, [dict(Bp=plan2.fwd16[0], Y=Xs[0], relu=True, x_planes=xps[0]),
                                               dict(Bp=plan2.fwd16[1], Y=torch.empty(N, H, device=DEV), relu=True)], b_format=1)
    Gs = [torch.randn(N, H, device=DEV) * (10.0 ** (i - 1)) for i in range(nl)]
    stride = nmat * H * H + H
    out_a, out_b = torch.full((nl * stride,), float("nan"), device=DEV), torch.full((nl * stride,), float("nan"), device=DEV)
    first_a = torch.full((stride + nmat * H,), float("nan"), device=DEV)
    first_b = first_a.clone()
    kw = dict(first_rowscale2=topo.deg_pows, first_out=None) if rs2 else {}
    n_plain = nl - (1 if rs2 else 0)
    if rs2:
        kw_a, kw_b = dict(kw, first_out=first_a), dict(kw, first_out=first_b)
    else:
        kw_a = kw_b = {}
    ops.wgrad_batched_xp(topo, Gs, H, xps, H, nmat, out_a[:max(n_plain, 1) * stride], **kw_a)
    ops.wgrad_batched(topo, Gs, H, Xs, H, nmat, out_b[:max(n_plain, 1) * stride], **kw_b)
    torch.cuda.synchronize()
    for l in range(nl):
        if rs2 and l == 0:
            ra, rb = first_a, first_b
        else:
            j = l - (1 if rs2 else 0)
            ra, rb = out_a[j * stride:(j + 1) * stride], out_b[j * stride:(j + 1) * stride]
        dW, db = _reference_wgrad(topo, Gs[l], Xs[l], nmat)
        for m in range(nmat):
            got = ra[m * H * H:(m + 1) * H * H].view(H, H).double()
            other = rb[m * H * H:(m + 1) * H * H].view(H, H).double()
            sc = dW[m].abs().max().item()
            assert (got - dW[m]).abs().max().item() <= 2e-6 * sc, (l, m, (got - dW[m]).abs().max().item() / sc)
            assert (got - other).abs().max().item() <= 2e-6 * sc
        gb = ra[nmat * H * H:nmat * H * H + H].double()
        assert (gb - db).abs().max().item() <= 2e-6 * db.abs().max().item()
        if rs2 and l == 0:
            ext = ra[stride:stride + nmat * H].view(nmat, H).double()
            ref = torch.stack([(Gs[0].double() * topo.deg_pows[:, m:m + 1].double()).sum(0) for m in range(nmat)])
            assert (ext - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()


def test_range_plan_covers_every_pair_once(pkg):
    """CPU-side arithmetic of wgrad_xp_plan against a brute-force walk of the kernel's rule (range w = items [w ipw, (w + 1) ipw),
    slab id = w + layer)."""
    ops = pkg.ops
    for ntiles, nl, ys in [(1024, 3, 2), (1, 1, 2), (5, 3, 2), (86, 3, 2), (256, 3, 2), (257, 2, 2), (1000, 8, 2), (3, 8, 2), (512, 1, 2), (85, 3, 4)]:
        n_wg, ipw, spans = ops.wgrad_xp_plan(ntiles, nl, ys)
        total = nl * ntiles
        assert n_wg * ipw >= total > (n_wg - 1) * ipw
        seen = {}
        for w in range(n_wg):
            for item in range(w * ipw, min((w + 1) * ipw, total)):
                seen.setdefault(item // ntiles, set()).add(w + item // ntiles)
        all_ids = [i for l in range(nl) for i in sorted(seen[l])]
        assert len(all_ids) == len(set(all_ids)) and max(all_ids) <= n_wg + nl - 2
        for l in range(nl):
            ids = sorted(seen[l])
            assert ids == list(range(spans[l][0], spans[l][0] + spans[l][1])), (ntiles, nl, l, ids, spans[l])


def _step(pkg, model, b, group=None):
    oracle_reg = {"mu_v": 1e-1, "mu_theta": 1e-1, "lam_v": 1e-4, "lam_p": 1e-8, "lam_pf": 1e-6, "lam_reg": 1e2}
    x, ei, ea = b["x"].to(DEV), b["edge_index"].to(DEV), b["edge_attr"].to(DEV)
    st = tuple(s.to(DEV) for s in b["stats"])
    for p in model.parameters():
        p.grad = None
    out = model(x[:, :8], ei, ea[:, :6])
    loss = pkg.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=out, x_mean=st[0], x_std=st[1], edge_mean=st[2], edge_std=st[3],
                            edge_index=ei, reg_coefs=oracle_reg, num_samples=None, node_param=x[:, 8:], edge_param=ea[:, 6:])
    loss.backward()
    torch.cuda.synchronize()
    return out.detach().clone(), loss.detach().clone(), [p.grad.detach().clone() for p in model.parameters()]


@pytest.mark.parametrize("grids,B,L", [(["cigre14"], 256, 4), (["cigre14", "cigre14_reswitched"], 100, 5), (["cigre14"], 64, 3)])
def test_whole_step_with_and_without_images(pkg, grids, B, L):
    FL = pkg.flags
    b = pkg.synthetic.make_batch(grids, B, seed=5)
    torch.manual_seed(6)
    model = pkg.MPN(8, 6, 2, 128, L, 2, 0.0).to(DEV)
    old = (FL.WGRAD_XP, FL.XP_DROP_FP32, FL.CHAIN_F16)
    try:
        FL.CHAIN_F16 = False      # (like with like: the image route is a bf16x6 route; without images the chains would run as f16x3)
        FL.WGRAD_XP, FL.XP_DROP_FP32 = True, True
        topo = pkg.topology.get_topology(b["edge_index"].to(DEV), b["x"].shape[0])
        topo.__dict__.pop("_xp_ok", None)
        if not pkg.ops.xplanes_supported(topo, 3, 128):
            pytest.skip("shape not covered")
        o1, l1, g1 = _step(pkg, model, b)
        FL.XP_DROP_FP32 = False
        o2, l2, g2 = _step(pkg, model, b)
        FL.WGRAD_XP = False
        topo.__dict__.pop("_xp_ok", None)
        o3, l3, g3 = _step(pkg, model, b)
    finally:
        FL.WGRAD_XP, FL.XP_DROP_FP32, FL.CHAIN_F16 = old
        topo.__dict__.pop("_xp_ok", None)
    assert torch.equal(o1, o2) and torch.equal(o1, o3) and torch.equal(l1, l2) and torch.equal(l1, l3)
    for a, c, d in zip(g1, g2, g3):
        assert torch.equal(a, c), "dropping the fp32 copies must not change a gradient"
        sc = d.abs().max().item() + 1e-30
        assert (a - d).abs().max().item() <= 5e-6 * sc, (a - d).abs().max().item() / sc
