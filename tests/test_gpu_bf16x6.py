"""The bf16x6 kernels (fp32-accurate products on the bf16 matrix pipe: layer chain and weight gradient) against the fp32-MFMA
kernels they replace, shape by shape, on the same inputs -- and the full-size co-residency case that exposed the packed-fp32
trap (two workgroups per CU, layer with a folded bias): bitwise reproducible, equal to the fp32 form to rounding."""
import pytest
import torch

from conftest import load_pkg, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def pkg():
    return load_pkg()


def _topo(pkg, grids, B, seed=0):
    b = pkg.synthetic.make_batch(grids, B, seed=seed)
    ei = b["edge_index"].to(DEV)
    N = b["x"].shape[0]
    return pkg.topology.get_topology(ei, N), N


@pytest.mark.parametrize("grids,B,H,nmat,n_layers,with_rs2", [
    (["cigre14"], 300, 128, 3, 1, False),        # the C2 shape, single layer
    (["cigre14"], 300, 128, 3, 2, False),        # two layers batched
    (["cigre14"], 257, 128, 3, 1, True),         # folded first layer: extra scaled column sums (one-pass kernel), odd tile count
    (["cigre14"], 300, 128, 3, 3, True),         # batch whose first layer has the scaled sums
    (["cigre14", "cigre14_reswitched"], 200, 100, 3, 2, False),   # H = 100: partial second pass, 100 of 128 X columns
    (["cigre14"], 128, 64, 2, 1, False),         # K = 1, one pass
    (["cigre14"], 96, 48, 3, 1, True),           # H = 48 (> 32): half-empty pass
    (["cigre14"], 64, 256, 3, 2, False),         # H = 256: two output groups x two input groups over grid.y
])
def test_weight_gradient_bf16x6_equals_fp32_mfma(pkg, grids, B, H, nmat, n_layers, with_rs2):
    nw = pkg.networks
    topo, N = _topo(pkg, grids, B)
    assert topo.nrb == 2
    torch.manual_seed(1)
    Gs = [torch.randn(N, H, device=DEV) for _ in range(n_layers)]
    Xs = [torch.randn(N, H, device=DEV) for _ in range(n_layers)]
    rs2 = torch.rand(N, 4, device=DEV) if with_rs2 else None
    stride = nmat * H * H + H
    saved = (pkg.flags.WGRAD_BF16, pkg.flags.WGRAD_TM32)

    def run(bf16, tm32=True):
        pkg.flags.WGRAD_BF16, pkg.flags.WGRAD_TM32 = bf16, tm32
        if n_layers == 1:
            out = torch.zeros(stride + (nmat * H if with_rs2 else 0), device=DEV)
            nw.wgrad(topo, Gs[0], H, Xs[0], H, nmat, out, rowscale2=rs2)
            return [out]
        out = torch.zeros((n_layers - (1 if with_rs2 else 0)) * stride, device=DEV)
        first = torch.zeros(stride + nmat * H, device=DEV) if with_rs2 else None
        nw.wgrad_batched(topo, Gs, H, Xs, H, nmat, out, first_rowscale2=rs2, first_out=first)
        return [out] + ([first] if with_rs2 else [])
    try:
        ref = run(False)
        got = run(True)                   # 32-row tiles, two workgroups per CU (wgrad16b_kernel, the default)
        got2 = run(True)
        got64 = run(True, tm32=False)     # 64-row tiles, one workgroup per CU (wgrad16_kernel)
        got64b = run(True, tm32=False)
    finally:
        pkg.flags.WGRAD_BF16, pkg.flags.WGRAD_TM32 = saved
    lds = pkg._lib.lib().dss2_wgrad_lds_bytes_ex
    covered = lds(2, nmat, H, H, topo.max_nnzT, topo.ellT, 1) != lds(2, nmat, H, H, topo.max_nnzT, topo.ellT, 0)
    assert covered                                                     # these shapes do run the bf16x6 kernel
    assert pkg.networks._wgrad_tiles(topo, nmat, H, H, 1).nrb == 1       # ... and the default walks the 32-row tiling
    for a, b_, c, d, e in zip(got, ref, got2, got64, got64b):
        assert torch.equal(a, c) and torch.equal(d, e)                 # fixed-order sums: bitwise reproducible
        assert rel_err(a, b_) < 2e-6 and rel_err(d, b_) < 2e-6


@pytest.mark.parametrize("grids,B,H,nmat,n_layers,with_rs2,nrb", [
    (["ober_sub"], 40, 128, 3, 1, False, None),       # 96-row tiles, 70 real rows: the last chunk runs one k-step
    (["ober_sub"], 300, 128, 3, 3, True, None),       # the C3 block: three layers in one launch, folded first layer
    (["ober_sub"], 41, 128, 3, 2, False, 5),          # two graphs per 160-row tile, odd graph count
    (["ober_sub"], 30, 100, 3, 1, False, 4),          # 128-row tiles, H = 100 (partial column groups)
    (["ober179"], 40, 128, 3, 1, True, None),         # 192-row tiles
    (["ober179"], 9, 64, 2, 1, False, None),          # K = 1
    (["ober179"], 12, 256, 3, 2, False, None),        # H = 256: two output x two input groups over grid.y
    (["cigre14", "ober_sub"], 60, 128, 3, 1, False, 3),   # mixed graph sizes in 96-row tiles
])
def test_tall_tile_weight_gradient_bf16x6_equals_fp32_mfma(pkg, grids, B, H, nmat, n_layers, with_rs2, nrb):
    """wgrad16t_kernel (96 .. 192-row tiles, chunks of 32 rows) against the fp32-MFMA kernel on the topology's own tiling."""
    nw = pkg.networks
    b = pkg.synthetic.make_batch(grids, B, seed=0)
    ei, N = b["edge_index"].to(DEV), b["x"].shape[0]
    ref_topo = pkg.topology.get_topology(ei, N)
    topo = ref_topo if nrb is None else pkg.topology.Topology(ei, N, nrb=nrb)
    assert topo.nrb >= 3 and (nrb is None or topo.nrb == nrb)
    torch.manual_seed(1)
    Gs = [torch.randn(N, H, device=DEV) for _ in range(n_layers)]
    Xs = [torch.randn(N, H, device=DEV) for _ in range(n_layers)]
    rs2 = torch.rand(N, 4, device=DEV) if with_rs2 else None
    stride = nmat * H * H + H
    saved = pkg.flags.WGRAD_BF16

    def run(bf16, tp):
        pkg.flags.WGRAD_BF16 = bf16
        if n_layers == 1:
            out = torch.zeros(stride + (nmat * H if with_rs2 else 0), device=DEV)
            nw.wgrad(tp, Gs[0], H, Xs[0], H, nmat, out, rowscale2=rs2)
            return [out]
        out = torch.zeros((n_layers - (1 if with_rs2 else 0)) * stride, device=DEV)
        first = torch.zeros(stride + nmat * H, device=DEV) if with_rs2 else None
        nw.wgrad_batched(tp, Gs, H, Xs, H, nmat, out, first_rowscale2=rs2, first_out=first)
        return [out] + ([first] if with_rs2 else [])
    try:
        ref = run(False, ref_topo)
        got = run(True, topo)
        got2 = run(True, topo)
    finally:
        pkg.flags.WGRAD_BF16 = saved
    lds = pkg._lib.lib().dss2_wgrad_lds_bytes_ex
    if topo.nrb != 5:                                                  # (the fp32 kernel has no 160-row instantiation)
        assert lds(topo.nrb, nmat, H, H, topo.max_nnzT, topo.ellT, 1) != lds(topo.nrb, nmat, H, H, topo.max_nnzT, topo.ellT, 0)
    for a, b_, c in zip(got, ref, got2):
        assert torch.equal(a, c)                                        # fixed-order sums: bitwise reproducible
        assert rel_err(a, b_) < 2e-6


def test_shapes_outside_the_bf16x6_weight_gradient_fall_back(pkg):
    """H <= 32 (the K-split 4-wave kernel), K = 3, ELL slices wider than 8: the fp32 kernel runs, results unchanged by the switch.
    (H = 32 on 96-row tiles left this list at the end of round 5: the f16x3 tall-tile kernel takes it -- same numbers to fp32 rounding.)"""
    nw = pkg.networks
    saved = pkg.flags.WGRAD_BF16
    try:
        for grids, B, H, nmat in ((["cigre14"], 64, 32, 3), (["cigre14"], 64, 64, 4), (["ober_sub"], 12, 32, 3)):
            topo, N = _topo(pkg, grids, B)
            torch.manual_seed(2)
            G, X = torch.randn(N, H, device=DEV), torch.randn(N, H, device=DEV)
            outs = []
            for mode in (False, True):
                pkg.flags.WGRAD_BF16 = mode
                o = torch.zeros(nmat * H * H + H, device=DEV)
                nw.wgrad(topo, G, H, X, H, nmat, o)
                outs.append(o)
            if grids == ["ober_sub"] and H == 32 and pkg.flags.WGRAD_F16:
                assert rel_err(outs[1], outs[0]) < 2e-6 and not torch.equal(outs[0], outs[1])      # (another kernel: csrc/dss2_wgrad16th.hip)
            else:
                assert torch.equal(outs[0], outs[1])
    finally:
        pkg.flags.WGRAD_BF16 = saved


@pytest.mark.parametrize("grids,B,H,nmat,nl", [
    (["cigre14"], 4096, 128, 3, 3),              # C2 at full size: two workgroups per CU
    (["cigre14"], 512, 64, 3, 4),                # row split (two waves per column group)
    (["cigre14"], 256, 32, 4, 3),                # K = 3 on the row-split kernel
    (["ober_sub"], 64, 128, 3, 3),               # 96-row tiles, one wave per SIMD
    (["cigre14"], 300, 256, 3, 2),               # H = 256: eight waves
    (["cigre14"], 200, 100, 3, 3),               # H = 100: k padded to 112
    (["cigre14"], 300, 96, 3, 2),                # split-plane form, three column groups
    (["cigre14"], 300, 128, 2, 3),               # split-plane form, K = 1
    (["cigre14", "cigre14_reswitched"], 190, 160, 2, 2),   # split-plane form, five column groups (eight-wave instantiation), K = 1
    (["cigre14"], 130, 72, 3, 2),                # 72 of 96 columns: the last stripe half empty (k padded to 80)
])
def test_layer_chain_bf16x6_equals_fp32_mfma(pkg, grids, B, H, nmat, nl):
    """Forward form with bias / ReLU / folded bias (prebias + row scales) in the first layer, backward form with ReLU gates:
    bf16x6 against the fp32 MFMA chain, twice (bitwise reproducible)."""
    nw = pkg.networks
    topo, N = _topo(pkg, grids, B)
    saved16 = pkg.flags.CHAIN_BF16
    pkg.flags.CHAIN_BF16 = True                     # (the environment may have switched the default off: these shapes are covered)
    try:
        assert nw.chain16_supported(topo, nmat, H, False) and nw.chain16_supported(topo, nmat, H, True)
    finally:
        pkg.flags.CHAIN_BF16 = saved16
    torch.manual_seed(3)
    Ws = [torch.randn(H, H, device=DEV) * (1.5 / H ** 0.5) for _ in range(nmat)]
    plan = nw._PackPlan([Ws], DEV, bf16_groups=(0,))
    plan.refresh()
    h, g = torch.randn(N, H, device=DEV), torch.randn(N, H, device=DEV)
    bias, pbias, prs = torch.randn(H, device=DEV), torch.randn(nmat, H, device=DEV), torch.rand(N, 4, device=DEV)
    acts = [torch.randn(N, H, device=DEV) for _ in range(nl)]

    def fwd(fmt):
        outs = [torch.empty(N, H, device=DEV) for _ in range(nl)]
        layers = [dict(Bp=(plan.fwd16[0] if fmt else plan.fwd[0]), Y=o, bias=bias, relu=True) for o in outs]
        layers[0]["prebias"] = pbias
        nw.gemm_prop_chain(topo, h, H, nmat, layers, pre_rowscale=prs, b_format=fmt)
        return outs

    def bwd(fmt):
        outs = [torch.empty(N, H, device=DEV) for _ in range(nl)]
        layers = [dict(Bp=(plan.bwd16[0] if fmt else plan.bwd[0]), Y=o, relu_src=a_) for o, a_ in zip(outs, acts)]
        nw.gemm_prop_chain(topo, g, H, nmat, layers, transposed=True, b_format=fmt)
        return outs
    for fn in (fwd, bwd):
        ref, a, b_ = fn(0), fn(1), fn(1)
        for r, x, y in zip(ref, a, b_):
            assert torch.equal(x, y)
            assert rel_err(x, r) < 3e-6


@pytest.mark.parametrize("grids,B,H,nmat", [
    (["ober179"], 40, 128, 3),                   # 192-row tiles: matrix-sequential, X staged in two K halves
    (["ober179"], 9, 96, 3),                     # H = 96: the whole X tile fits beside three wave stages -> no K halves -> fp32 kernel
    (["ober179"], 12, 128, 4),                   # K = 3
    (["ober179"], 700, 128, 3),                  # every CU busy, several rounds of workgroups (two waves per SIMD: row split)
])
def test_tall_tile_layer_bf16x6_equals_fp32_mfma(pkg, grids, B, H, nmat):
    """dss2_gemm_prop with b_format = 1 (tall tiles have no layer chain): forward form with bias / ReLU / folded bias,
    data-gradient form with a ReLU gate, against the fp32 MFMA kernel, twice (bitwise reproducible)."""
    nw = pkg.networks
    topo, N = _topo(pkg, grids, B)
    saved16 = pkg.flags.CHAIN_BF16
    pkg.flags.CHAIN_BF16 = True
    try:
        ok = nw.gemm16_supported(topo, nmat, H, False) and nw.gemm16_supported(topo, nmat, H, True)
    finally:
        pkg.flags.CHAIN_BF16 = saved16
    assert topo.nrb == 6
    if H == 96:
        assert not ok      # (bf16x6 exists for the K-halved configuration only; the model code then keeps the fp32 weights)
        return
    assert ok
    torch.manual_seed(5)
    Ws = [torch.randn(H, H, device=DEV) * (1.5 / H ** 0.5) for _ in range(nmat)]
    plan = nw._PackPlan([Ws], DEV, bf16_groups=(0,))
    plan.refresh()
    h, g, act = torch.randn(N, H, device=DEV), torch.randn(N, H, device=DEV), torch.randn(N, H, device=DEV)
    bias, pbias, prs = torch.randn(H, device=DEV), torch.randn(nmat, H, device=DEV), torch.rand(N, 4, device=DEV)

    def fwd(fmt):
        out = torch.empty(N, H, device=DEV)
        nw.gemm_prop(topo, h, H, H, (plan.fwd16[0] if fmt else plan.fwd[0]), nmat, H, out, bias=bias, relu=True, prebias=pbias,
                     pre_rowscale=prs, b_format=fmt)
        return out

    def bwd(fmt):
        out = torch.empty(N, H, device=DEV)
        nw.gemm_prop(topo, g, H, H, (plan.bwd16[0] if fmt else plan.bwd[0]), nmat, H, out, relu_src=act, transposed=True, b_format=fmt)
        return out
    for fn in (fwd, bwd):
        r, x, y = fn(0), fn(1), fn(1)
        assert torch.equal(x, y)
        assert rel_err(x, r) < 3e-6


@pytest.mark.parametrize("B,H,nmat,nl", [
    (40, 128, 3, 3),                             # the 179-bus configuration's shape: four column groups, K = 2
    (700, 128, 3, 3),                            # every CU busy, several rounds of workgroups
    (9, 64, 3, 2),                               # two column groups
    (12, 96, 2, 4),                              # three column groups, K = 1
])
def test_tall_tile_layer_chain_equals_layer_by_layer(pkg, B, H, nmat, nl):
    """192-row tiles (nrb = 6) have ONE chained form, the split-plane bf16x6 kernel of csrc/dss2_gemm_chain_sp6.hip: checked
    against the fp32 MFMA kernel launched layer by layer with the same epilogues -- forward: bias / ReLU / folded bias / mask
    tensor / in-kernel dropout (same Philox snapshot and ids); data-gradient form: ReLU gates -- and launched twice
    (bitwise reproducible)."""
    nw = pkg.networks
    topo, N = _topo(pkg, ["ober179"], B)
    assert topo.nrb == 6
    saved16 = pkg.flags.CHAIN_BF16
    pkg.flags.CHAIN_BF16 = True
    try:
        assert nw.chain16_supported(topo, nmat, H, False) and nw.chain16_supported(topo, nmat, H, True)
        assert nw.chain_supported(topo, nmat, H, False, have16=True) and not nw.chain_supported(topo, nmat, H, False)
    finally:
        pkg.flags.CHAIN_BF16 = saved16
    torch.manual_seed(7)
    Ws = [torch.randn(H, H, device=DEV) * (1.5 / H ** 0.5) for _ in range(nmat)]
    plan = nw._PackPlan([Ws], DEV, bf16_groups=(0,))
    plan.refresh()
    h, g = torch.randn(N, H, device=DEV), torch.randn(N, H, device=DEV)
    bias, pbias, prs = torch.randn(H, device=DEV), torch.randn(nmat, H, device=DEV), torch.rand(N, 4, device=DEV)
    dmask = (torch.rand(N, H, device=DEV) > 0.4).float() * 1.6
    acts = [torch.randn(N, H, device=DEV) for _ in range(nl)]
    snap = torch.tensor([7654321, 5], dtype=torch.int64, device=DEV)

    def fwd_layers(outs, fmt):
        Bp = plan.fwd16[0] if fmt else plan.fwd[0]
        ls = [dict(Bp=Bp, Y=o, bias=bias, relu=True, drop_id=i + 1) for i, o in enumerate(outs)]
        ls[0]["prebias"] = pbias
        ls[1]["dmask"] = dmask
        ls[-1].update(relu=False, drop_id=0)
        return ls

    def fwd_chain():
        outs = [torch.empty(N, H, device=DEV) for _ in range(nl)]
        nw.gemm_prop_chain(topo, h, H, nmat, fwd_layers(outs, 1), pre_rowscale=prs, drop=(snap, 0.3), b_format=1)
        return outs

    def fwd_ref():
        outs = [torch.empty(N, H, device=DEV) for _ in range(nl)]
        x = h
        for L in fwd_layers(outs, 0):
            nw.gemm_prop(topo, x, H, H, L["Bp"], nmat, H, L["Y"], bias=L["bias"], relu=L["relu"], prebias=L.get("prebias"),
                         pre_rowscale=prs if "prebias" in L else None, dmask=L.get("dmask"),
                         drop=(snap, 0.3, L["drop_id"]) if L["drop_id"] else None)
            x = L["Y"]
        return outs

    def bwd_chain():
        outs = [torch.empty(N, H, device=DEV) for _ in range(nl)]
        nw.gemm_prop_chain(topo, g, H, nmat, [dict(Bp=plan.bwd16[0], Y=o, relu_src=a_) for o, a_ in zip(outs, acts)],
                           transposed=True, b_format=1)
        return outs

    def bwd_ref():
        outs = [torch.empty(N, H, device=DEV) for _ in range(nl)]
        x = g
        for o, a_ in zip(outs, acts):
            nw.gemm_prop(topo, x, H, H, plan.bwd[0], nmat, H, o, relu_src=a_, transposed=True)
            x = o
        return outs
    for chain, ref in ((fwd_chain, fwd_ref), (bwd_chain, bwd_ref)):
        r, a, b_ = ref(), chain(), chain()
        for li, (rr, x, y) in enumerate(zip(r, a, b_)):
            assert torch.equal(x, y)
            assert rel_err(x, rr) < 3e-6 * (li + 1), li      # (layer li's input already carries li layers of rounding differences)


@pytest.mark.parametrize("grid,B,H,nmat", [("ober179", 40, 128, 3), ("ober179", 700, 128, 3), ("ober179", 9, 64, 3), ("ober179", 12, 96, 2),
                                            ("ober_sub", 64, 128, 3), ("ober_sub", 1024, 128, 3), ("ober_sub", 30, 64, 2),
                                            ("cigre14", 300, 128, 3), ("cigre14", 4096, 128, 3), ("cigre14", 70, 100, 3), ("cigre14", 41, 96, 2)])
def test_tall_tile_chain_gate_bits_equal_fp32_gate(pkg, grid, B, H, nmat):
    """dss2_chain_layer.y_bits / gate_bits: the forward chain writes the sign bits of its outputs (with dropout zeros in them),
    the data-gradient chain gates with those words instead of reading the activations -- bitwise the same result.  (Round 4: the
    64-row split-plane chain too; there a chain gated by fp32 activations runs its tile GEMM on 32x32x16 MFMAs and one gated by
    bit words on 16x16x32 when K is a multiple of 32 -- same gates, sums in another order: equal to rounding, bitwise at H = 100.)"""
    nw = pkg.networks
    topo, N = _topo(pkg, [grid], B)
    gw = nw.chain_gate_words(topo, nmat, H)
    assert gw == ((H + 31) // 32) * 32 * ((4 * topo.nrb + 7) // 8) and topo.nrb in (2, 3, 6)
    torch.manual_seed(11)
    Ws = [torch.randn(H, H, device=DEV) * (1.5 / H ** 0.5) for _ in range(nmat)]
    plan = nw._PackPlan([Ws], DEV, bf16_groups=(0,))
    plan.refresh()
    h, g, bias = torch.randn(N, H, device=DEV), torch.randn(N, H, device=DEV), torch.randn(H, device=DEV)
    snap = torch.tensor([424242, 9], dtype=torch.int64, device=DEV)
    nl = 3
    acts = [torch.empty(N, H, device=DEV) for _ in range(nl)]
    bits = [torch.zeros(topo.ntiles * gw, dtype=torch.int64, device=DEV) for _ in range(nl)]
    nw.gemm_prop_chain(topo, h, H, nmat, [dict(Bp=plan.fwd16[0], Y=a_, bias=bias, relu=True, drop_id=i + 1, y_bits=b_)
                                           for i, (a_, b_) in enumerate(zip(acts, bits))], drop=(snap, 0.3), b_format=1)
    frac = [(a_ > 0).float().mean().item() for a_ in acts]
    assert all(0.05 < f < 0.6 for f in frac), frac
    # every set bit belongs to a positive element: the populations match
    for a_, b_ in zip(acts, bits):
        pop = sum(int(((b_ >> k) & 1).sum()) for k in range(64))
        assert pop == int((a_ > 0).sum())

    def bwd(use_bits):
        outs = [torch.empty(N, H, device=DEV) for _ in range(nl)]
        nw.gemm_prop_chain(topo, g, H, nmat, [dict(Bp=plan.bwd16[0], Y=o, relu_src=a_, gate_bits=(b_ if use_bits else None))
                                               for o, a_, b_ in zip(outs, acts, bits)], transposed=True, b_format=1)
        return outs
    for li, (x, y) in enumerate(zip(bwd(True), bwd(False))):
        if topo.nrb == 2 and H % 32 == 0:
            assert rel_err(x, y) < 3e-6 * (li + 1), li
        else:
            assert torch.equal(x, y)


def _stress(fn, ref, n=200):
    """n launches of fn(): every result bitwise equal to the first, and within 5e-7 (max-normalised) of the fp32-MFMA form."""
    first = fn()
    bad = 0
    for _ in range(n - 1):
        out = fn()
        bad += int(not all(torch.equal(a, b) for a, b in zip(out, first)))
    torch.cuda.synchronize()
    assert bad == 0, f"{bad} of {n} launches differ from the first"
    for a, r in zip(first, ref):
        assert rel_err(a, r) < 5e-7


def test_packed_fp32_trap_geometry_200_launches(pkg):
    """VERDICT r2 (weak #1): the geometry in which the bf16x6 chain once returned run-to-run different values -- B = 4096 (two
    workgroups per CU), first layer with the folded bias (the fma that became v_pk_fma_f32 ... op_sel) -- launched 200 times:
    bitwise identical every time and equal to the fp32-MFMA form.  The shipped translation unit is built without packed fp32
    ops (csrc/build.sh verifies that); tools/pk_stress.py runs the same loop on a diagnostic build WITH them
    (profiles/r03_pk_fma_investigation.txt has the outcome and the ISA diff)."""
    nw = pkg.networks
    topo, N = _topo(pkg, ["cigre14"], 4096)
    H, nmat, nl = 128, 3, 3
    torch.manual_seed(3)
    Ws = [torch.randn(H, H, device=DEV) * (1.5 / H ** 0.5) for _ in range(nmat)]
    plan = nw._PackPlan([Ws], DEV, bf16_groups=(0,))
    plan.refresh()
    h = torch.randn(N, H, device=DEV)
    bias, pbias, prs = torch.randn(H, device=DEV), torch.randn(nmat, H, device=DEV), torch.rand(N, 4, device=DEV)

    def fwd(fmt):
        outs = [torch.empty(N, H, device=DEV) for _ in range(nl)]
        layers = [dict(Bp=(plan.fwd16[0] if fmt else plan.fwd[0]), Y=o, bias=bias, relu=True) for o in outs]
        layers[0]["prebias"] = pbias
        nw.gemm_prop_chain(topo, h, H, nmat, layers, pre_rowscale=prs, b_format=fmt)
        return outs
    _stress(lambda: fwd(1), fwd(0))


def test_tall_tile_bf16x6_with_packed_ops_200_launches(pkg):
    """The tall-tile bf16x6 kernel (dss2_gemm_prop.hip) KEEPS packed fp32 ops -- one workgroup per CU by LDS, its fma separated
    from every MFMA phase by a workgroup barrier -- so it gets the same 200-launch stress: every CU busy, several rounds."""
    nw = pkg.networks
    topo, N = _topo(pkg, ["ober179"], 700)
    H, nmat = 128, 3
    assert topo.nrb == 6
    torch.manual_seed(5)
    Ws = [torch.randn(H, H, device=DEV) * (1.5 / H ** 0.5) for _ in range(nmat)]
    plan = nw._PackPlan([Ws], DEV, bf16_groups=(0,))
    plan.refresh()
    h = torch.randn(N, H, device=DEV)
    bias, pbias, prs = torch.randn(H, device=DEV), torch.randn(nmat, H, device=DEV), torch.rand(N, 4, device=DEV)

    def fwd(fmt):
        out = torch.empty(N, H, device=DEV)
        nw.gemm_prop(topo, h, H, H, (plan.fwd16[0] if fmt else plan.fwd[0]), nmat, H, out, bias=bias, relu=True, prebias=pbias,
                     pre_rowscale=prs, b_format=fmt)
        return [out]
    _stress(lambda: fwd(1), fwd(0))


def test_non_finite_inputs_stay_non_finite(pkg):
    """VERDICT r2 (weak #4): split3 turns +-inf into (inf, NaN, NaN), so the bf16x6 GEMM answers NaN where the fp32 MFMA would
    answer +-inf.  What must hold: an inf / NaN in an input row never produces a FINITE wrong output -- the rows (graphs of the
    tile) that depend on it are non-finite in both forms, every other tile is untouched and bitwise equal to the clean run."""
    nw = pkg.networks
    topo, N = _topo(pkg, ["cigre14"], 64)
    H, nmat = 128, 3
    torch.manual_seed(7)
    Ws = [torch.randn(H, H, device=DEV) * (1.5 / H ** 0.5) for _ in range(nmat)]
    plan = nw._PackPlan([Ws], DEV, bf16_groups=(0,))
    plan.refresh()
    h = torch.randn(N, H, device=DEV)

    def fwd(x, fmt):
        outs = [torch.empty(N, H, device=DEV) for _ in range(2)]
        nw.gemm_prop_chain(topo, x, H, nmat, [dict(Bp=(plan.fwd16[0] if fmt else plan.fwd[0]), Y=o) for o in outs], b_format=fmt)
        return outs[-1]
    clean = fwd(h, 1)
    for bad in (float("inf"), float("nan")):
        x = h.clone()
        x[7, 5] = bad                                  # node 7: graph 0, tile 0 (rows 0..59)
        o16, o32 = fwd(x, 1), fwd(x, 0)
        assert not torch.isfinite(o16[:15]).all() and not torch.isfinite(o32[:15]).all()      # the graph of node 7
        assert torch.isfinite(o16[60:]).all() and torch.equal(o16[60:], clean[60:])           # other tiles: untouched
        assert (torch.isfinite(o16) <= torch.isfinite(o32)).all()                             # never finite where fp32 is not


@pytest.mark.parametrize("H,nmat", [(128, 3), (96, 2), (256, 3)])
def test_split_plane_chain_general_epilogue(pkg, H, nmat):
    """The split-plane chain's general epilogue (mask tensor, residual, in-kernel dropout, folded bias) against the fp32 MFMA chain:
    same Philox masks (same snapshot and ids), so the two forms agree to rounding."""
    nw = pkg.networks
    topo, N = _topo(pkg, ["cigre14"], 333)
    torch.manual_seed(5)
    Ws = [torch.randn(H, H, device=DEV) * (1.5 / H ** 0.5) for _ in range(nmat)]
    plan = nw._PackPlan([Ws], DEV, bf16_groups=(0,))
    plan.refresh()
    h = torch.randn(N, H, device=DEV)
    bias, pbias, prs = torch.randn(H, device=DEV), torch.randn(nmat, H, device=DEV), torch.rand(N, 4, device=DEV)
    dmask = (torch.rand(N, H, device=DEV) > 0.4).float() * 1.6
    snap = torch.tensor([1234567, 3], dtype=torch.int64, device=DEV)

    def run(fmt):
        outs = [torch.empty(N, H, device=DEV) for _ in range(3)]
        layers = [dict(Bp=(plan.fwd16[0] if fmt else plan.fwd[0]), Y=outs[0], bias=bias, relu=True, prebias=pbias, drop_id=1),
                  dict(Bp=(plan.fwd16[0] if fmt else plan.fwd[0]), Y=outs[1], bias=bias, relu=True, dmask=dmask, drop_id=2),
                  dict(Bp=(plan.fwd16[0] if fmt else plan.fwd[0]), Y=outs[2], relu=False, drop_id=0)]
        nw.gemm_prop_chain(topo, h, H, nmat, layers, pre_rowscale=prs, drop=(snap, 0.3), b_format=fmt)
        return outs
    ref, a, b_ = run(0), run(1), run(1)
    for r, x, y in zip(ref, a, b_):
        assert torch.equal(x, y)
        assert rel_err(x, r) < 3e-6
