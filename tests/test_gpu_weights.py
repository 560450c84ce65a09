"""GPU: the weight-space work of a step in two launches (csrc/dss2_weights.hip: dss2_prep_weights = fold + packing,
dss2_finish_weights = slab reductions + chain rule of the fold; VERDICT r4 #2b).

The reference keeps nn.Linear weights as they are and lets autograd assemble their gradients
(/root/reference/networks.py:159-209 EdgeAggregation, 211-264 MPN); the fold / packing / slab reductions are this library's own
preparation around its kernels, so the oracle here is the library's SEPARATE launches, which the model-level parity tests pin to the
CPU oracle.  The bar is bit for bit.

* the merged launches give bitwise the separate launches' outputs on random tables (dependent and independent packing descriptors
  side by side, misaligned reductions, more reduction work than one residency wave), many times over (the counter words re-arm);
* a training step with DSS2_WEIGHTS_MERGED on equals the step with it off, bit for bit, in loss and every gradient: the C2 model,
  a PFN stack (one launch for all blocks), a model whose backward is not the chained route (nothing pending: plain launches);
* the step has two launches fewer.
"""
import ctypes as C
import importlib

import numpy as np
import pytest
import torch

from conftest import PKG_NAME

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REG = {"mu_v": 1e-1, "mu_theta": 1e-1, "lam_v": 1e-4, "lam_p": 1e-8, "lam_pf": 1e-6, "lam_reg": 1e2}


@pytest.fixture(scope="module")
def pkg():
    return importlib.import_module(PKG_NAME)


def _tables(pkg, nm, ho, hid, seed):
    """A fold (Wf_m = W_m W2, bf_m = W_m b2) and a packing table: the folded matrices (dependent) and plain ones (independent)."""
    pl = importlib.import_module(PKG_NAME + ".plans")
    g = torch.Generator(device="cpu").manual_seed(seed)
    W2 = torch.randn(hid, hid, generator=g).to(DEV)
    b2 = torch.randn(hid, generator=g).to(DEV)
    ws = [torch.randn(ho, hid, generator=g).to(DEV) for _ in range(nm)]
    others = [torch.randn(hid, hid, generator=g).to(DEV) for _ in range(3)]
    fold = pl._FoldPlan(W2, b2, ws, DEV, 0, hid * hid + hid)
    fold._check()
    groups = [[pl._MatView(fold.Wf[m], ho, hid, hid, 0, dep=True) for m in range(nm)], [W2], others]
    pack = pl._PackPlan(groups, DEV, bf16_groups=(0, 2) if hid % 4 == 0 else ())
    pack._build_table()
    return fold, pack, (W2, b2, ws, others)


@pytest.mark.parametrize("nm,ho,hid", [(3, 128, 128), (3, 64, 64), (2, 96, 96), (4, 40, 72)])
def test_prep_weights_is_bitwise_fold_then_pack(pkg, nm, ho, hid):
    lib, ops = pkg._lib.lib(), pkg.ops
    fold, pack, keep = _tables(pkg, nm, ho, hid, seed=nm * 1000 + hid)
    st = pkg._lib.stream_ptr(DEV)
    bufs = pack.fwd + pack.bwd + list(pack.fwd16.values()) + list(pack.bwd16.values()) + [fold.Wf, fold.bf]
    ft, fn, fmx = fold.fwd_tab
    assert lib.dss2_small_gemm(ft.data_ptr(), fn, fmx, None, st) == 0
    assert lib.dss2_pack_weights(pack.table.data_ptr(), pack.n_desc, pack.max_elems, st) == 0
    torch.cuda.synchronize()
    want = [b.clone() for b in bufs]
    cnt = ops.weight_counters(DEV)
    for rep in range(20):
        for b in bufs:
            b.zero_()
        assert lib.dss2_prep_weights(ft.data_ptr(), fn, fmx, pack.table.data_ptr(), pack.n_desc, pack.n_dep, pack.max_elems, cnt.data_ptr(), st) == 0
        torch.cuda.synchronize()
        for b, w in zip(bufs, want):
            assert torch.equal(b, w), rep
        assert int(cnt.abs().sum()) == 0
    # without a fold: plain packing
    for b in pack.fwd + pack.bwd:
        b.zero_()
    assert pack.n_dep == nm * (4 if hid % 4 == 0 else 2)
    assert lib.dss2_prep_weights(None, 0, 0, pack.table.data_ptr(), pack.n_desc, 0, pack.max_elems, cnt.data_ptr(), st) == 0
    torch.cuda.synchronize()
    for b, w in zip(pack.fwd + pack.bwd, want):
        assert torch.equal(b, w)
    assert lib.dss2_prep_weights(ft.data_ptr(), fn, fmx, pack.table.data_ptr(), pack.n_desc, pack.n_dep, pack.max_elems, None, st) == 2
    assert lib.dss2_prep_weights(None, 0, 0, pack.table.data_ptr(), pack.n_desc, 1, pack.max_elems, cnt.data_ptr(), st) == 2


@pytest.mark.parametrize("n_slabs,extra,misalign", [(255, 3, False), (64, 1, True), (512, 6, False), (7, 0, False)])
def test_finish_weights_is_bitwise_reduce_then_chain_rule(pkg, n_slabs, extra, misalign):
    lib, ops = pkg._lib.lib(), pkg.ops
    nm, ho, hid = 3, 128, 128
    fold, pack, keep = _tables(pkg, nm, ho, hid, seed=n_slabs)
    st = pkg._lib.stream_ptr(DEV)
    g = torch.Generator(device="cpu").manual_seed(n_slabs + 1)
    glen = fold.gfold.numel()
    slab_dep = torch.randn(n_slabs, glen, generator=g).to(DEV)
    olen = nm * ho * hid + ho
    slabs = [torch.randn(n_slabs, olen + 4, generator=g).to(DEV) for _ in range(extra)]
    flat = torch.zeros(hid * hid + hid + nm * ho * hid + ho + extra * (olen + 4) + 8, dtype=torch.float32, device=DEV)
    off0 = hid * hid + hid + nm * ho * hid + ho
    pend = [(slab_dep, slab_dep.data_ptr(), n_slabs, glen, fold.gfold, glen)]
    for i, s_ in enumerate(slabs):
        o = off0 + i * (olen + 4) + (1 if (misalign and i == 0) else 0)      # (a misaligned output: the scalar form inside the launch)
        pend.append((s_, s_.data_ptr(), n_slabs, olen + 4, flat[o:o + olen], olen))
    rt, rn, rmx = fold.bwd_tab
    # separate launches
    ops.reduce_pending(list(reversed(pend)))
    assert lib.dss2_small_gemm(rt.data_ptr(), rn, rmx, flat.data_ptr(), st) == 0
    torch.cuda.synchronize()
    want_flat, want_g = flat.clone(), fold.gfold.clone()
    assert float(want_flat[:off0].abs().sum()) > 0
    cnt = ops.weight_counters(DEV)
    old = pkg.flags.WEIGHTS_MERGED
    try:
        pkg.flags.WEIGHTS_MERGED = True
        for rep in range(10):
            flat.zero_()
            fold.gfold.zero_()
            p2 = list(reversed(pend))      # (the dependency is found by its output, wherever it stands in the list)
            ops.finish_weights(p2, fold.bwd_tab, flat, {fold.gfold.data_ptr()}, DEV)
            assert p2 == []
            torch.cuda.synchronize()
            assert torch.equal(fold.gfold, want_g), rep
            assert torch.equal(flat, want_flat), rep
            assert int(cnt.abs().sum()) == 0
    finally:
        pkg.flags.WEIGHTS_MERGED = old
    descs = ops._reduce_descs(pend)
    assert lib.dss2_finish_weights(C.addressof(descs), len(pend), 1, rt.data_ptr(), rn, rmx, flat.data_ptr(), None, st) == 2
    assert lib.dss2_finish_weights(C.addressof(descs), len(pend), 0, rt.data_ptr(), rn, rmx, flat.data_ptr(), cnt.data_ptr(), st) == 2
    assert lib.dss2_finish_weights(C.addressof(descs), len(pend), len(pend) + 1, rt.data_ptr(), rn, rmx, flat.data_ptr(), cnt.data_ptr(), st) == 2


def _step_fn(pkg, cls, cargs, B, seed=0):
    torch.manual_seed(seed)
    b = pkg.synthetic.make_batch(["cigre14"], B, seed=seed, violate=0.3)
    x, ei, ea = b["x"].to(DEV), b["edge_index"].to(DEV), b["edge_attr"].to(DEV)
    st = tuple(s.to(DEV) for s in b["stats"])
    model = getattr(pkg, cls)(*cargs).to(DEV)
    params = list(model.parameters())

    def step():
        for p in params:
            p.grad = None
        out = model(x[:, :8], ei, ea[:, :6])
        loss = pkg.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=out, x_mean=st[0], x_std=st[1], edge_mean=st[2], edge_std=st[3],
                                edge_index=ei, reg_coefs=REG, num_samples=None, node_param=x[:, 8:], edge_param=ea[:, 6:])
        loss.backward(pkg.data.unit_grad(loss))
        torch.cuda.synchronize()
        return loss.detach().clone(), [p.grad.detach().clone() for p in params]
    return step


@pytest.mark.parametrize("cls,cargs,B", [
    ("MPN", (8, 6, 2, 128, 4, 2, 0.0), 256),          # the C2 model: fold + chained layers + batched weight gradients
    ("MPN", (8, 6, 2, 128, 2, 2, 0.0), 128),          # two layers: the backward records nothing (plain launches at the end)
    ("PFN", (8, 6, 2, 64, 4, 2, 0.0, 3), 128),        # a stack: every block's fold / packing / reductions / chain rule in the two launches
    ("SkipPFN", (8, 6, 2, 96, 3, 2, 0.0, 2), 100),
])
def test_training_step_with_merged_weight_launches_is_bitwise_the_separate_one(pkg, cls, cargs, B):
    step = _step_fn(pkg, cls, cargs, B)
    old = pkg.flags.WEIGHTS_MERGED
    try:
        pkg.flags.WEIGHTS_MERGED = False
        loss0, g0 = step()
        pkg.flags.WEIGHTS_MERGED = True
        loss1, g1 = step()
        loss2, g2 = step()
    finally:
        pkg.flags.WEIGHTS_MERGED = old
    assert torch.equal(loss0, loss1) and torch.equal(loss1, loss2)
    for a, b_, c in zip(g0, g1, g2):
        assert torch.equal(a, b_) and torch.equal(b_, c)
    assert int(pkg.ops.weight_counters(DEV).abs().sum()) == 0


def test_merged_weight_launches_save_two_launches_of_the_c2_step(pkg):
    step = _step_fn(pkg, "MPN", (8, 6, 2, 128, 4, 2, 0.0), 256)
    step()
    old = pkg.flags.WEIGHTS_MERGED
    counts = {}
    try:
        for on in (False, True):
            pkg.flags.WEIGHTS_MERGED = on
            plan = pkg.graphs.PlannedStep(lambda: step()[0])
            counts[on] = plan.n_launches
            del plan
    finally:
        pkg.flags.WEIGHTS_MERGED = old
    assert counts[True] == counts[False] - 2, counts


@pytest.mark.parametrize("n_slabs,length,stride", [(1024, 770, 772), (512, 2944, 2944), (300, 13, 16), (256, 8192, 8192), (255, 770, 772), (1024, 770, 770)])
def test_slab_reduction_of_many_short_slabs(pkg, n_slabs, length, stride):
    """The "tall" form of the fixed-order slab reduction (csrc/dss2_weightspace.hpp, round 5: n_slabs >= 256, len <= 8192 -- the per-tile
    slabs of the narrow head's weight gradient): 16 output floats and 64 slab lanes per workgroup.  Both entry points (they size their
    grids themselves), ragged lengths, a stride that is not a multiple of 4 (scalar form), one slab short of the threshold; against fp64."""
    ops = pkg.ops
    g = torch.Generator(device="cpu").manual_seed(n_slabs + length)
    slab = torch.randn(n_slabs, stride, generator=g).to(DEV)
    want = slab[:, :length].double().sum(0)
    scale = slab[:, :length].abs().double().sum(0).max().item()
    out1 = torch.full((length + 3,), float("nan"), device=DEV)
    ops._reduce(slab, 0, n_slabs, stride, out1[:length], length, None)                  # dss2_reduce_slabs
    out2 = torch.full((length + 3,), float("nan"), device=DEV)
    other = torch.randn(85, 49664, generator=g).to(DEV)
    o_other = torch.empty(49664, device=DEV)
    pend = []
    ops._reduce(other, 0, 85, 49664, o_other, 49664, pend)
    ops._reduce(slab, 0, n_slabs, stride, out2[:length], length, pend)
    ops.reduce_pending(pend)                                                             # dss2_reduce_slabs_multi, beside a long reduction
    torch.cuda.synchronize()
    for o in (out1, out2):
        assert torch.isnan(o[length:]).all()                                             # nothing written past the end
        assert (o[:length].double() - want).abs().max().item() <= 1e-6 * scale
    assert torch.equal(out1[:length], out2[:length])                                     # the same bits through either entry point
    assert (o_other.double() - other.double().sum(0)).abs().max().item() <= 1e-5 * other.abs().double().sum(0).max().item()
