"""GPU (-m gpu): the HIP path, called through the C ABI (ctypes -> libdss2_hip.so), against
  (a) the golden vectors produced by the reference itself, and
  (b) the CPU oracle on the same seeded inputs (fp32, plus an fp64 referee where noted).
Tolerances (max-normalised relative error, BASELINE.json north_star: 1e-5 on node states and loss):
  outputs 1e-5, loss 1e-5; gradients 1e-4 (the reference's own fp32-vs-fp64 gradient noise is 2e-5)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, CASES, LOSS_CASES, case_batch, case_grads, case_state_dict, golden, load_pkg, rel_err, t

pytestmark = pytest.mark.gpu
TOL_OUT, TOL_LOSS, TOL_GRAD = 1e-5, 1e-5, 1e-4
DEV = "cuda:0"


@pytest.fixture(scope="module")
def pkg():
    p = load_pkg()
    p._lib.lib()  # fail loudly if the extension is missing
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return p


def _loss(mod_data, x, ei, ea, st, out, reg, **kw):
    return mod_data.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=out, x_mean=st[0], x_std=st[1],
                                 edge_mean=st[2], edge_std=st[3], edge_index=ei, reg_coefs=reg, num_samples=None,
                                 node_param=x[:, 8:], edge_param=ea[:, 6:], **kw)


# ---------------------------------------------------------------------------- op level
@pytest.mark.parametrize("n_graphs,kdim,hout", [(5, 128, 128), (3, 32, 32), (7, 64, 2), (4, 8, 128), (2, 256, 64), (9, 2, 32)])
def test_gemm_linear_matches_matmul(pkg, n_graphs, kdim, hout):
    """nmat=1 path of dss2_gemm_prop: packing + MFMA fragment layout + epilogue, vs torch.matmul."""
    torch.manual_seed(kdim + hout)
    b = pkg.synthetic.make_batch(["cigre14"], n_graphs, seed=1)
    N = b["x"].shape[0]
    topo = pkg.topology.Topology(b["edge_index"].to(DEV), N)
    X = torch.randn(N, kdim, device=DEV)
    W = torch.randn(hout, kdim, device=DEV) * 0.3          # asymmetric by construction
    bias = torch.randn(hout, device=DEV)
    plan = pkg.networks._PackPlan([[W]], torch.device(DEV))
    plan.refresh()
    Y = torch.empty(N, hout, device=DEV)
    pkg.networks.gemm_prop(topo, X, kdim, kdim, plan.fwd[0], 1, hout, Y, bias=bias)
    ref = (X.double() @ W.double().t() + bias.double())
    assert rel_err(Y, ref) < 2e-6
    Y2 = torch.empty(N, kdim, device=DEV)                    # data-gradient layout: G @ W
    pkg.networks.gemm_prop(topo, Y, hout, hout, plan.bwd[0], 1, kdim, Y2)
    assert rel_err(Y2, Y.double() @ W.double()) < 2e-6


@pytest.mark.parametrize("grid,hin,hout,K", [("cigre14", 128, 128, 2), ("cigre14_reswitched", 32, 8, 3),
                                              ("ober_sub", 64, 64, 2), ("cigre14", 128, 2, 2), ("cigre14", 32, 32, 1),
                                              ("ober179", 32, 32, 2)])
def test_tagconv_fwd_bwd(pkg, oracle, grid, hin, hout, K):
    torch.manual_seed(7)
    b = pkg.synthetic.make_batch([grid], 6, seed=2)
    N = b["x"].shape[0]
    ei = b["edge_index"]
    ei2, _ = oracle.undirect_graph(ei, b["edge_attr"][:, :6])
    ref = oracle.TAGConv(hin, hout, K).double()
    ref.bias.data.uniform_(-0.5, 0.5)
    mine = pkg.TAGConv(hin, hout, K).to(DEV)
    mine.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    h = torch.randn(N, hin)
    g = torch.randn(N, hout)
    hr = h.double().requires_grad_(True)
    outr = ref(hr, ei2)
    outr.backward(g.double())
    hm = h.to(DEV).requires_grad_(True)
    outm = mine(hm, ei2.to(DEV))
    outm.backward(g.to(DEV))
    assert rel_err(outm, outr) < TOL_OUT
    assert rel_err(hm.grad, hr.grad) < TOL_OUT
    assert rel_err(mine.bias.grad, ref.bias.grad) < TOL_GRAD
    for a, bb in zip(mine.lins, ref.lins):
        assert rel_err(a.weight.grad, bb.weight.grad) < TOL_GRAD


@pytest.mark.parametrize("grid,hid,nrb", [("ober_sub", 128, None), ("ober_sub", 64, "4"), ("cigre14+ober_sub", 128, "4"), ("ober179", 128, None),
                                          ("ober179", 32, None), ("cigre14+ober179", 128, "6")])
def test_edge_aggregation_tall_tiles_without_input_gradient(pkg, oracle, grid, hid, nrb, monkeypatch):
    """The first block of a model: x needs no gradient, so the backward is the weight-gradient-only form: edge16_bwd_kernel
    without the per-row sums on 96-row tiles and -- as two PARTS of 96 rows each, end of round 5 -- on 192-row tiles (a 192-row
    instantiation had measured 132 + 68 us against 150 + 51 for the VALU tile kernels at the 179-bus shape; the parts: 81 + 45);
    128-row tiles (forced: a 70-bus graph straddles the two parts of 64 rows) run as parts since round 6 (the VALU tile kernels before).  The mixed batch has tiles of CIGRE graphs only whose rows end inside the first
    part (the second part of such a tile is skipped) beside 179-bus tiles that fill both."""
    if nrb is not None:
        monkeypatch.setenv("DSS2_NRB", nrb)
    torch.manual_seed(5)
    # (another batch for every forced height: the structure cache is keyed by content)
    b = pkg.synthetic.make_batch(grid.split("+"), 7 if nrb is None else {"4": 9, "6": 9}[nrb] + (1 if "+" in grid and nrb == "4" else 0), seed=6)
    x, ea = b["x"][:, :8], b["edge_attr"][:, :6]
    ei2, ea2 = oracle.undirect_graph(b["edge_index"], ea)
    ref = oracle.EdgeAggregation(8, 6, hid, hid).double()
    mine = pkg.EdgeAggregation(8, 6, hid, hid).to(DEV)
    mine.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    outr = ref(x.double(), ei2, ea2.double())
    g = torch.randn(outr.shape)
    outr.backward(g.double())
    ei_dev = ei2.to(DEV)
    outm = mine(x.to(DEV).contiguous(), ei_dev, ea2.to(DEV).contiguous())
    outm.backward(g.to(DEV))
    topo_ = pkg.topology.get_topology(ei_dev, x.shape[0], double=False)
    assert topo_.nrb == (int(nrb) if nrb else (3 if grid == "ober_sub" else 6))
    if "+" in grid and topo_.nrb == 6:      # some tile ends inside its first part, some tile needs both parts
        rows = (topo_.tile_start[1:] - topo_.tile_start[:-1]).cpu()
        assert int(rows.min()) <= 96 < int(rows.max()), rows.tolist()
    assert rel_err(outm, outr) < TOL_OUT
    for (n, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        assert rel_err(p.grad, q.grad) < TOL_GRAD, n


@pytest.mark.parametrize("mfma", ["bf16x6", "fp32-mfma", "valu"])     # first Linear as bf16x6 (default), on fp32 MFMAs, VALU tile kernels
# (ober_sub / ober179 at dim_hid 32 with the input gradient: the VALU tile kernel with two rows per wave, round 6 -- 70 rows per tile, and 179: an odd count,
#  the last pair's second row lies beyond the tile)
@pytest.mark.parametrize("grid,hid", [("cigre14", 128), ("ober_sub", 32), ("cigre14_reswitched", 256), ("cigre14", 64), ("ober179", 32)])
def test_edge_aggregation_fwd_bwd(pkg, oracle, grid, hid, mfma, monkeypatch):
    monkeypatch.setenv("DSS2_EDGE_MFMA", "0" if mfma == "valu" else "1")      # read per call by the library
    monkeypatch.setenv("DSS2_EDGE_BF16", "1" if mfma == "bf16x6" else "0")
    torch.manual_seed(3)
    b = pkg.synthetic.make_batch([grid], 5, seed=4)
    x, ea = b["x"][:, :8], b["edge_attr"][:, :6]
    ei2, ea2 = oracle.undirect_graph(b["edge_index"], ea)
    ref = oracle.EdgeAggregation(8, 6, hid, hid).double()
    mine = pkg.EdgeAggregation(8, 6, hid, hid).to(DEV)
    mine.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    xr = x.double().requires_grad_(True)
    outr = ref(xr, ei2, ea2.double())
    g = torch.randn(outr.shape)
    outr.backward(g.double())
    xm = x.to(DEV).contiguous().requires_grad_(True)
    outm = mine(xm, ei2.to(DEV), ea2.to(DEV).contiguous())
    outm.backward(g.to(DEV))
    assert rel_err(outm, outr) < TOL_OUT
    assert rel_err(xm.grad, xr.grad) < TOL_GRAD
    for (n, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        assert rel_err(p.grad, q.grad) < TOL_GRAD, n


def test_segment_sum(pkg):
    b = pkg.synthetic.make_batch(["cigre14", "cigre14_reswitched"], 64, seed=9)
    N = b["x"].shape[0]
    topo = pkg.topology.Topology(b["edge_index"].to(DEV), N)
    for h in (32, 128, 256):
        msg = torch.randn(topo.E2, h, device=DEV)
        out = pkg.networks.segment_sum(msg, topo.rowptr, topo.perm.to(torch.int32), N)
        tgt = torch.cat([b["edge_index"][1], b["edge_index"][0]]).to(DEV)
        ref = torch.zeros(N, h, device=DEV, dtype=torch.float64).index_add_(0, tgt, msg.double())
        assert rel_err(out, ref) < 1e-6
        out2 = pkg.networks.segment_sum(msg, topo.rowptr, topo.perm.to(torch.int32), N)
        assert torch.equal(out, out2)  # deterministic: no float atomics


# ---------------------------------------------------------------------------- golden cases
@pytest.mark.parametrize("name", list(CASES))
def test_model_matches_reference_golden(pkg, oracle, name):
    cls, args, with_loss = CASES[name]
    g = golden(f"case_{name}.npz")
    model = getattr(pkg, cls)(*args)
    res = model.load_state_dict(case_state_dict(g), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model = model.to(DEV)
    b = case_batch(g, device=DEV)
    x, ei, ea, st = b["x"], b["edge_index"], b["edge_attr"], b["stats"]
    out = model(x[:, :8], ei, ea[:, :6])
    assert rel_err(out, t(g["out"])) < TOL_OUT
    if with_loss:
        loss = _loss(pkg.data, x, ei, ea, st, out, oracle.DEFAULT_REG_COEFS)
        assert abs(loss.item() - float(g["loss"])) <= TOL_LOSS * abs(float(g["loss"]))
        assert rel_err(out, t(g["out_after_loss"])) < TOL_OUT          # theta zeroed at the slack in place
        slack = x[:, 9] > 0
        assert (out.detach()[slack, 1] == 0).all()
        loss.backward()
    else:
        out.backward(t(g["gout"], device=DEV))
    grads = case_grads(g)
    for k, p in model.named_parameters():
        assert p.grad is not None, k
        assert rel_err(p.grad, grads[k]) < TOL_GRAD, k


@pytest.mark.parametrize("name", LOSS_CASES)
def test_loss_matches_reference_golden(pkg, oracle, name):
    g = golden(f"case_{name}.npz")
    b = case_batch(g, device=DEV)
    x, ei, ea, st = b["x"], b["edge_index"], b["edge_attr"], b["stats"]
    o_leaf = t(g["output"], device=DEV).clone().requires_grad_(True)
    o = o_leaf * 1.0
    loss = _loss(pkg.data, x, ei, ea, st, o, oracle.DEFAULT_REG_COEFS)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) <= TOL_LOSS * abs(float(g["loss"]))
    assert rel_err(o, t(g["output_after"])) < 1e-7
    assert rel_err(o_leaf.grad, t(g["grad_output"])) < TOL_GRAD
    yv = torch.cat([o.detach()[:, 0:1] * st[1][:1] + st[0][:1], o.detach()[:, 1:]], 1)
    flows = torch.stack(pkg.data.get_pflow(yv, ei, x[:, 8:], ea[:, 6:]), 1)
    assert rel_err(flows, t(g["pflow"])) < TOL_OUT


# ---------------------------------------------------------------------------- full-size configs
def _referee_loss(oracle, x64, ea64, out64, st64, edge_index):
    return oracle.gsp_wls_edge(input=x64[:, :8], edge_input=ea64[:, :6], output=out64, x_mean=st64[0], x_std=st64[1],
                               edge_mean=st64[2], edge_std=st64[3], edge_index=edge_index,
                               reg_coefs=oracle.DEFAULT_REG_COEFS, num_samples=None, node_param=x64[:, 8:],
                               edge_param=ea64[:, 6:])


def _train_step_pair(pkg, oracle, grids, B, hid, L, K=2, seed=0, cls="MPN", dim_out=2):
    torch.manual_seed(seed)
    b = pkg.synthetic.make_batch(grids, B, seed=seed)
    ref = getattr(oracle, cls)(8, 6, dim_out, hid, L, K, 0.0)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if n.endswith("bias") and "convs" in n:
                p.uniform_(-0.1, 0.1)
    mine = getattr(pkg, cls)(8, 6, dim_out, hid, L, K, 0.0)
    mine.load_state_dict(ref.state_dict())
    mine = mine.to(DEV)
    out_r, loss_r = oracle.train_step(ref, b, b["stats"])
    out_r_pre = None
    x, ei, ea = b["x"].to(DEV), b["edge_index"].to(DEV), b["edge_attr"].to(DEV)
    st = tuple(s.to(DEV) for s in b["stats"])
    out_m = mine(x[:, :8], ei, ea[:, :6])
    flows_m = torch.empty(ei.shape[1], 8, device=DEV)       # get_pflow as the loss kernel itself evaluated it
    loss_m = _loss(pkg.data, x, ei, ea, st, out_m, oracle.DEFAULT_REG_COEFS, pflow_out=flows_m)
    loss_m.backward()
    return ref, mine, out_r, loss_r, out_m, loss_m, flows_m


@pytest.mark.parametrize("grids,B,hid,L", [
    (["cigre14"], 64, 32, 1),                                   # BASELINE config C1
    (["cigre14"], 4096, 128, 4),                                # C2 (the headline configuration)
    (["ober_sub"], 1024, 128, 4),                               # C3
    (["cigre14", "cigre14_reswitched"], 512, 256, 8),           # C5's model on a mixed-topology shard
    (["cigre14", "cigre14_reswitched"], 4096, 256, 8),          # ... and on a full 4096-graph shard (un-pinned tolerance 1.3e-4)
    (["ober179"], 1024, 128, 4),                                # C3 read as "~180 buses": the synthetic 179-bus feeder, full size
])
def test_baseline_configs_against_oracle(pkg, oracle, grids, B, hid, L):
    ref, mine, out_r, loss_r, out_m, loss_m, flows_m = _train_step_pair(pkg, oracle, grids, B, hid, L)
    assert rel_err(out_m, out_r) < TOL_OUT
    assert abs(loss_m.item() - loss_r.item()) <= TOL_LOSS * abs(loss_r.item())
    # gradients, first against the plain fp32 oracle with NOTHING pinned (1e-4: the oracle's own fp32-vs-fp64 gradient
    # noise is 2e-5 and a razor-edge ReLU gate that falls the other way moves a weight-gradient row by ~1/N_nodes) ...
    # (one flipped gate weighs ~1/N_nodes of a gradient row, so small batches get a tolerance of a few gates)
    # (8 gates' worth: the deep stack on the small shard -- 8 layers x 256 units on 7 680 nodes, 14 M gates -- flips a
    #  handful against the fp32 oracle; the pinned comparison below is the tight one)
    tol_unpinned = max(TOL_GRAD, 8.0 / out_m.shape[0])
    for (n, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        assert rel_err(p.grad, q.grad) < tol_unpinned, ("fp32 oracle, un-pinned", n, rel_err(p.grad, q.grad))
    # ... then tightly: fp64 referee evaluated on the HIP path's own ReLU gate pattern.  A pre-activation within
    # an ulp of 0 may take the other sign under a different fp32 summation order; such a flipped gate is
    # not an arithmetic error, but it moves one row of a weight gradient (and everything upstream of it)
    # by ~1/N_nodes.  Pinning the gates removes that ambiguity, so the tolerance can be tight (1e-5,
    # max-normalised); the flipped gates themselves must be few and must sit at |pre-activation| ~ 0.
    b = pkg.synthetic.make_batch(grids, B, seed=0)
    x, ei, ea = b["x"].to(DEV), b["edge_index"].to(DEV), b["edge_attr"].to(DEV)
    # the gates the kernels themselves applied: sign of the activations the forward pass kept for its backward (the same
    # launches as the step above -- the forward is bitwise reproducible), not of a re-evaluation by other kernels
    with torch.no_grad():
        topo = pkg.topology.get_topology(ei, x.shape[0])
        _, kept, _ = pkg.networks._mpn_forward(mine, topo, x[:, :8], ea[:, :6], mine._params())
    gates = [(a_ > 0).cpu() for a_ in kept[4:4 + L - 1]]        # kept = [x, edge_attr, S, conv-0 input, act_1 .. act_{L-1}]
    ref64 = type(ref)(8, 6, 2, hid, L, 2, 0.0).double()
    ref64.load_state_dict({k: v.double() for k, v in ref.state_dict().items()})
    x64, ea64 = b["x"].double(), b["edge_attr"].double()
    st64 = tuple(s.double() for s in b["stats"])
    ei2, ea2 = oracle.undirect_graph(b["edge_index"], ea64[:, :6])
    edge_hidden = {}

    def keep_hidden(mod, inp, outp):       # hidden activations of the edge MLP, with their gradient after backward
        outp.retain_grad()
        edge_hidden["z"], edge_hidden["h"] = inp[0], outp
    hook = ref64.edge_aggr.edge_aggr[1].register_forward_hook(keep_hidden)
    h = ref64.edge_aggr(x64[:, :8], ei2, ea2)
    hook.remove()
    n_flip, n_gate = 0, 0
    for l in range(L - 1):
        pre = ref64.convs[l](h, ei2)
        flipped = (pre > 0) != gates[l]
        n_flip, n_gate = n_flip + int(flipped.sum()), n_gate + flipped.numel()
        if flipped.any():   # only razor-edge pre-activations may differ in sign
            assert pre[flipped].abs().max() <= 1e-5 * pre.abs().max(), (l, pre[flipped].abs().max().item())
        h = pre * gates[l].double()
    out64 = ref64.convs[-1](h, ei2)
    # The loss has non-smooth points of its own, and fp32 noise decides them: get_pflow forms the branch flows as
    # differences of nearly equal terms (P = -V_i V_j (G cos + B sin) + G V_i^2), so the fp32 currents carry ~1e-5 of
    # relative noise (tools/diag_edge_mfma.py: I_from 0.6452016 in fp32 against 0.6451960 in fp64 on the same inputs).
    # With 70 000 branches one of them sits within that noise of the overload threshold of relu(loading - 1.5)
    # (data.py:455), and which side it falls on moves d loss/d output of its two buses by O(1) and every weight
    # gradient by ~3e-5 (diagnosed with tools/diag_c3.py: the round-1 "unexplained deviation" of the matrix-pipe
    # edge forward was exactly this gate, flipped by a 3e-8 change of one theta).  Like the conv gates, the loss's
    # branch decisions -- the four penalty ReLUs and max(I_from, I_to) (data.py:387-388) -- are therefore pinned to
    # the HIP path's own choices, read from the flows the loss kernel exported; all VALUES stay un-pinned.
    vhv, vlv = b["x"][:, 8].max(), b["x"][:, 8].min()
    om = out_m.detach().cpu()                                  # (theta already masked in place by the loss)
    v_m = om[:, 0:1] * b["stats"][1][:1] + b["stats"][0][:1]
    thij_m = (om[b["edge_index"][0], 1] - om[b["edge_index"][1], 1]).abs()
    fm = flows_m.cpu()
    max_pins = iter([fm[:, 6] >= fm[:, 7], fm[:, 6] * vhv >= fm[:, 7] * vlv])
    relu_pins = iter([v_m > 1.1, v_m < 0.9, thij_m > 0.5, (fm[:, 0] + fm[:, 1]) > 1.5])
    real_max, real_relu = torch.maximum, torch.relu
    torch.maximum = lambda a_, b_: torch.where(next(max_pins), a_, b_)
    torch.relu = lambda t_: t_ * next(relu_pins).to(t_.dtype)
    try:
        loss64 = _referee_loss(oracle, x64, ea64, out64, st64, b["edge_index"])
    finally:
        torch.maximum, torch.relu = real_max, real_relu
    with torch.no_grad():
        loss64_unpinned = _referee_loss(oracle, x64, ea64, out64.detach().clone(), st64, b["edge_index"])
    assert abs(loss64.item() - loss64_unpinned.item()) <= 1e-9 * abs(loss64.item())     # the pins only decide near-ties
    loss64.backward()
    assert n_flip <= max(2, 1e-5 * n_gate), (n_flip, n_gate)
    assert rel_err(out_m, out64) < TOL_OUT
    assert abs(loss_m.item() - loss64.item()) <= TOL_LOSS * abs(loss64.item())
    # ---- the plain numbers, for a reader who does not want to audit the referee above: the fp64 oracle with
    # NOTHING pinned, and the worst parameter gradient against each of the three references -- logged per configuration
    # (gpurun_out/parity_errors.jsonl; the committed copy: profiles/r06_parity_errors.jsonl) and held to the un-pinned tolerance
    ref64u = type(ref)(8, 6, 2, hid, L, 2, 0.0).double()
    ref64u.load_state_dict({k: v.double() for k, v in ref.state_dict().items()})
    out64u, loss64u = oracle.train_step(ref64u, {"x": x64, "edge_index": b["edge_index"], "edge_attr": ea64}, st64)
    worst = {"fp32_oracle_unpinned": max((rel_err(p.grad, q.grad), n) for (n, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters())),
             "fp64_oracle_unpinned": max((rel_err(p.grad, q.grad), n) for (n, p), (_, q) in zip(mine.named_parameters(), ref64u.named_parameters())),
             "fp64_referee_on_the_kernels_own_gates": max((rel_err(p.grad, q.grad), n) for (n, p), (_, q) in zip(mine.named_parameters(), ref64.named_parameters()))}
    rec = {"config": f"{'+'.join(grids)} B={B} H={hid} L={L}", "nodes": int(out_m.shape[0]), "output_vs_fp64": rel_err(out_m, out64u),
           "loss_vs_fp64": abs(loss_m.item() - loss64u.item()) / abs(loss64u.item()), "flipped_conv_gates": [n_flip, n_gate],
           "worst_gradient": {k: {"error": v[0], "parameter": v[1]} for k, v in worst.items()}, "tol_unpinned": tol_unpinned}
    print("[parity]", json.dumps(rec))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "parity_errors.jsonl"), "a") as fh:
            fh.write(json.dumps(rec) + "\n")
    except OSError:
        pass
    assert worst["fp64_oracle_unpinned"][0] < tol_unpinned, worst["fp64_oracle_unpinned"]
    for (n, p), (_, q64) in zip(mine.named_parameters(), ref64.named_parameters()):
        e = rel_err(p.grad, q64.grad)
        if e < 1e-5:
            continue
        # The edge MLP's own per-edge gates are not observable from outside (the kernels recompute them), so they
        # cannot be pinned.  Instead every deviation must be EXPLAINED by ambiguous gates: a gate is ambiguous when
        # |z| is within the fp32 rounding scale of its 22-term dot product (4e-6 * sum_k |w_k c_k|); flipping it
        # toggles that edge's contribution g * c to the row, so the deviation of row o is bounded by the sum of
        # |g| |c| over the ambiguous gates of unit o.  (With heavy-tailed inverse-variance inputs a single such
        # edge can carry more than 1e-5 of the largest gradient entry.)
        assert n.startswith("edge_aggr.edge_aggr.0."), (n, e)
        lin1 = ref64.edge_aggr.edge_aggr[0]
        c = torch.cat([x64[:, :8][ei2[1]], x64[:, :8][ei2[0]], ea2], dim=1)              # [E2, 22]: x_i | x_j | ea
        z = edge_hidden["z"].detach()
        scale = c.abs() @ lin1.weight.detach().abs().t() + lin1.bias.detach().abs()
        amb = (z.abs() <= 4e-6 * scale).double() * edge_hidden["h"].grad.abs()           # |g| on ambiguous gates
        bound = amb.t() @ c.abs() if n.endswith("weight") else amb.sum(0)
        dev = (p.grad.double().cpu() - q64.grad).abs()
        assert (dev <= bound + 1e-5 * q64.grad.abs().max()).all(), (n, e, (dev - bound).max().item())
        assert e < 1e-3, (n, e)


def test_cache_busting_batch_is_deterministic_and_linear_in_gout(pkg, oracle):
    """B = 32768 graphs on one GPU (N = 491 520 nodes: activations of 252 MB per layer, beyond the 256 MiB Infinity Cache --
    the size bench.py times the scatter-add on): the size-independent properties of the path -- bitwise run-to-run
    reproducibility and linearity of the backward in the incoming gradient -- plus block-diagonality: the batch is eight
    copies of a 4096-graph batch, so every copy's outputs must equal the first copy's bit for bit."""
    torch.manual_seed(0)
    b = pkg.synthetic.make_batch(["cigre14"], 4096, seed=5)
    m = pkg.MPN(8, 6, 2, 128, 4, 2, 0.0).to(DEV)
    reps, n1 = 8, b["x"].shape[0]
    x = b["x"].to(DEV).repeat(reps, 1)
    ea = b["edge_attr"].to(DEV).repeat(reps, 1)
    ei1 = b["edge_index"].to(DEV)
    ei = torch.cat([ei1 + k * n1 for k in range(reps)], 1)
    g = torch.randn(n1, 2, device=DEV).repeat(reps, 1)

    def run(scale):
        for p in m.parameters():
            p.grad = None
        out = m(x[:, :8], ei, ea[:, :6])
        out.backward(g * scale)
        return out.detach().clone(), [p.grad.clone() for p in m.parameters()]

    o1, g1 = run(1.0)
    o2, g2 = run(1.0)
    assert torch.equal(o1, o2) and all(torch.equal(a, c) for a, c in zip(g1, g2))
    assert all(torch.equal(o1[:n1], o1[k * n1:(k + 1) * n1]) for k in range(1, reps))
    _, g3 = run(2.0)
    for a, c in zip(g1, g3):
        assert rel_err(c, 2 * a) < 1e-6
    # against the 4096-graph batch itself: same outputs, gradients 8 x (to summation-order rounding)
    for p in m.parameters():
        p.grad = None
    o_small = m(x[:n1, :8], ei1, ea[:ei1.shape[1], :6])
    o_small.backward(g[:n1])
    assert torch.equal(o_small.detach(), o1[:n1])
    for p, a in zip(m.parameters(), g1):
        assert rel_err(a, reps * p.grad) < 5e-6


def test_full_size_is_deterministic_and_linear_in_gout(pkg, oracle):
    """Size-independent properties at C2: bitwise run-to-run reproducibility (no float atomics)
    and linearity of the backward in the incoming gradient."""
    torch.manual_seed(0)
    b = pkg.synthetic.make_batch(["cigre14"], 4096, seed=5)
    m = pkg.MPN(8, 6, 2, 128, 4, 2, 0.0).to(DEV)
    x, ei, ea = b["x"].to(DEV), b["edge_index"].to(DEV), b["edge_attr"].to(DEV)
    g = torch.randn(x.shape[0], 2, device=DEV)

    def run(scale):
        for p in m.parameters():
            p.grad = None
        out = m(x[:, :8], ei, ea[:, :6])
        out.backward(g * scale)
        return out.detach().clone(), [p.grad.clone() for p in m.parameters()]

    o1, g1 = run(1.0)
    o2, g2 = run(1.0)
    assert torch.equal(o1, o2) and all(torch.equal(a, c) for a, c in zip(g1, g2))
    _, g3 = run(2.0)
    for a, c in zip(g1, g3):
        assert rel_err(c, 2 * a) < 1e-6


# ---------------------------------------------------------------------------- interface / quirks
def test_quirks_and_errors(pkg):
    b = pkg.synthetic.make_batch(["cigre14"], 8, seed=1)
    x, ei, ea = b["x"].to(DEV), b["edge_index"].to(DEV), b["edge_attr"].to(DEV)
    m = pkg.MPN(8, 6, 2, 32, 2, 2, 0.3).to(DEV).eval()
    with torch.no_grad():   # dropout stays active in eval() (networks.py:268)
        assert (m(x[:, :8], ei, ea[:, :6]) - m(x[:, :8], ei, ea[:, :6])).abs().max() > 0
    assert m.is_directed(ei) is True
    ei2, ea2 = m.undirect_graph(ei, ea[:, :6])
    assert ei2.shape[1] == 2 * ei.shape[1] and m.is_directed(ei2) is False
    with pytest.raises(RuntimeError):
        pkg.MPN(8, 6, 2, 32, 2, 2, 0.0)(b["x"][:, :8], b["edge_index"], b["edge_attr"][:, :6])  # CPU tensors
    # (other input widths used to fail loudly; since round 3 they run the general path: tests/test_gpu_cliffs.py)
    assert pkg.EdgeAggregation(7, 6, 32, 32).to(DEV)(x[:, :7].contiguous(), ei2, ea2).shape == (x.shape[0], 32)
    with pytest.raises(NotImplementedError):      # dim_feate > 32: loud (9 .. 32 run since round 4: tests/test_gpu_cliffs.py)
        pkg.EdgeAggregation(8, 33, 32, 32).to(DEV)(x[:, :8], ei2, torch.cat([ea2] * 6, 1)[:, :33])


def test_hipgraph_replay_matches_eager(pkg, oracle):
    """A captured step (forward + loss + backward) replays to the same loss and gradients as eager."""
    torch.manual_seed(0)
    b = pkg.synthetic.make_batch(["cigre14"], 64, seed=2)
    x, ei, ea = b["x"].to(DEV), b["edge_index"].to(DEV), b["edge_attr"].to(DEV)
    st = tuple(s.to(DEV) for s in b["stats"])
    model = pkg.SkipPFN(8, 6, 2, 32, 2, 2, 0.0, 2).to(DEV)     # exercises the dx path inside the capture too

    def step():
        for p in model.parameters():
            p.grad = None
        out = model(x[:, :8], ei, ea[:, :6])
        loss = _loss(pkg.data, x, ei, ea, st, out, oracle.DEFAULT_REG_COEFS)
        loss.backward()
        return loss

    gs = pkg.graphs.GraphedStep(step)
    l_graph = gs.replay().item()
    g_graph = [p.grad.clone() for p in model.parameters()]
    with torch.cuda.stream(gs.stream):
        l_eager = step().item()
    torch.cuda.synchronize()
    assert l_graph == l_eager
    assert all(torch.equal(a, p.grad) for a, p in zip(g_graph, model.parameters()))


def test_fused_adamax_matches_torch(pkg):
    """dss2_adamax_step vs torch.optim.Adamax (the reference's optimizer, dss2_run.py:91-92) on the same
    gradients for three steps; state keys are torch's."""
    torch.manual_seed(0)
    shapes = [(128, 22), (128,), (128, 128), (2, 128), (2,), (1,)]
    ps_ref = [torch.randn(s).requires_grad_(True) for s in shapes]
    ps_gpu = [p.detach().clone().to(DEV).requires_grad_(True) for p in ps_ref]
    o_ref = torch.optim.Adamax(ps_ref, lr=3e-3)
    o_gpu = pkg.FusedAdamax(ps_gpu, lr=3e-3)
    for step in range(3):
        for a, b in zip(ps_ref, ps_gpu):
            g = torch.randn(a.shape) * (10.0 ** (step - 1))
            a.grad = g.clone()
            b.grad = g.to(DEV)
        o_ref.step()
        o_gpu.step()
    for a, b in zip(ps_ref, ps_gpu):
        assert rel_err(b, a) < 1e-6
        sa, sb = o_ref.state[a], o_gpu.state[b]
        assert set(sb.keys()) == {"step", "exp_avg", "exp_inf"} and float(sb["step"]) == float(sa["step"]) == 3.0
        assert rel_err(sb["exp_avg"], sa["exp_avg"]) < 1e-6 and rel_err(sb["exp_inf"], sa["exp_inf"]) < 1e-6


def test_runner_trains_and_tracks_the_oracle(pkg, oracle):
    """Five optimizer steps of the runner's loop (FusedAdamax, lr 3e-3) against the oracle + torch Adamax from
    the same initial weights on the same batch: the loss trajectories agree and the loss goes down."""
    torch.manual_seed(0)
    b = pkg.synthetic.make_batch(["cigre14"], 64, seed=3)
    ref = oracle.MPN(8, 6, 2, 32, 2, 2, 0.0)
    mine = pkg.MPN(8, 6, 2, 32, 2, 2, 0.0)
    mine.load_state_dict(ref.state_dict())
    mine = mine.to(DEV)
    o_ref = torch.optim.Adamax(ref.parameters(), lr=3e-3)
    o_gpu = pkg.FusedAdamax(mine.parameters(), lr=3e-3)
    dev_b = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in b.items() if k != "stats"}
    st = tuple(s.to(DEV) for s in b["stats"])
    l_ref, l_gpu = [], []
    for _ in range(5):
        _, lr_ = oracle.train_step(ref, b, b["stats"])
        o_ref.step()
        l_ref.append(lr_.item())
        l_gpu.append(pkg.runner.train_epoch(mine, o_gpu, [dev_b], st, pkg.runner.REG_COEFS))
    assert l_gpu[-1] < l_gpu[0]
    for a, c in zip(l_gpu, l_ref):
        assert abs(a - c) <= 1e-4 * abs(c), (l_gpu, l_ref)
    m = pkg.runner.evaluate(mine, [dev_b], st)
    assert all(v == v and v >= 0 for v in m.values())          # finite metrics


# ---------------------------------------------------------------------------- launch-structure variants
@pytest.mark.parametrize("grids,B,hid,L,cls,p", [
    (["cigre14"], 256, 128, 4, "MPN", 0.0),                     # C2's model: 3-layer chains, 2-layer batch
    (["cigre14", "cigre14_reswitched"], 96, 256, 5, "MPN", 0.0),  # 8-wave chain (H = 256), mixed topologies
    (["ober_sub"], 24, 64, 4, "SkipMPN", 0.0),                  # 96-row tiles, narrow H -> 8 last layer, residual
    (["cigre14"], 128, 32, 6, "MPN", 0.3),                      # dropout masks inside the chain, H = 32
    (["ober_sub"], 16, 128, 4, "MPN", 0.3),                     # 96-row tiles: two-workgroup chain, sign-bit gates WITH dropout zeros in them
    (["ober179"], 8, 128, 4, "MPN", 0.3),                       # 192-row tiles: the same, eight-wave weight gradient
])
def test_chained_and_batched_launches_equal_per_layer_launches(pkg, oracle, grids, B, hid, L, cls, p):
    """The layer chain (dss2_gemm_prop_chain), the batched weight gradient (dss2_wgrad_batched) and the folded
    second Linear are re-associations of launches, not of arithmetic inside a layer: with the same weights and the
    same dropout masks the chained forward must equal the per-layer forward BIT FOR BIT, and the gradients must
    agree to fp32 summation-order noise with the unfolded / unbatched path."""
    nw = pkg.networks
    b = pkg.synthetic.make_batch(grids, B, seed=3)
    x, ei, ea = b["x"].to(DEV), b["edge_index"].to(DEV), b["edge_attr"].to(DEV)
    st = tuple(s.to(DEV) for s in b["stats"])
    dim_out = 8 if cls == "SkipMPN" else 2
    torch.manual_seed(11)
    model = getattr(pkg, cls)(8, 6, dim_out, hid, L, 2, p).to(DEV)
    saved = (pkg.flags.CHAIN_LAYERS, pkg.flags.WGRAD_BATCH, pkg.flags.FOLD_W2, pkg.flags.CHAIN_BF16)

    def run(chain, batch, fold, bf16=False):
        pkg.flags.CHAIN_LAYERS, pkg.flags.WGRAD_BATCH, pkg.flags.FOLD_W2, pkg.flags.CHAIN_BF16 = chain, batch, fold, bf16
        for q in model.parameters():
            q.grad = None
        torch.manual_seed(5)                       # same dropout masks
        out = model(x[:, :8], ei, ea[:, :6])
        o = out.detach().clone()
        if dim_out == 2:
            loss = _loss(pkg.data, x, ei, ea, st, out, oracle.DEFAULT_REG_COEFS)
        else:
            loss = (out * torch.linspace(-1, 1, out.numel(), device=DEV).view_as(out)).sum()
        loss.backward()
        return o, [q.grad.clone() for q in model.parameters()]

    try:
        o_ref, g_ref = run(False, False, True)     # per-layer launches (fold on: same forward arithmetic)
        o_chain, g_chain = run(True, True, True)
        pkg.flags.WGRAD_JOIN_FOLDED = True                # folded conv 0 in the same batched launch as the plain layers
        o_join, g_join = run(True, True, True)
        pkg.flags.WGRAD_JOIN_FOLDED = False
        o_sep, g_sep = run(True, True, True)
        o_unfold, g_unfold = run(False, False, False)
        pkg.flags.WGRAD_JOIN_FOLDED = None
        o_16, g_16 = run(True, True, True, bf16=True)     # the chain's tile GEMM as bf16x6 (the default): fp32 rounding level
    finally:
        pkg.flags.CHAIN_LAYERS, pkg.flags.WGRAD_BATCH, pkg.flags.FOLD_W2, pkg.flags.CHAIN_BF16 = saved
        pkg.flags.WGRAD_JOIN_FOLDED = None
    assert torch.equal(o_chain, o_ref) and torch.equal(o_join, o_ref) and torch.equal(o_sep, o_ref)
    # (the batched schedules sum their slabs in the 16-lane order of reduce_slabs_multi_v4, the per-layer launches in the 4-group
    #  order of reduce_slabs: two fixed orders, fp32 rounding apart)
    for a, c, j, sp, (n, _) in zip(g_ref, g_chain, g_join, g_sep, model.named_parameters()):
        assert rel_err(c, a) < 4e-6 and rel_err(j, a) < 4e-6 and rel_err(sp, a) < 4e-6, n
    # folding the second Linear re-associates one matrix product: fp32 rounding level, not bitwise
    assert rel_err(o_unfold, o_ref) < 2e-6
    for a, c, (n, _) in zip(g_ref, g_unfold, model.named_parameters()):
        assert rel_err(c, a) < 5e-5, n
    assert rel_err(o_16, o_ref) < 5e-6
    for a, c, (n, _) in zip(g_ref, g_16, model.named_parameters()):
        assert rel_err(c, a) < 5e-5, n


@pytest.mark.parametrize("hid,L,K", [(128, 4, 2), (64, 3, 3)])
def test_tall_tiles_matrix_sequential(pkg, oracle, hid, L, K):
    """179-bus graphs need 192-row tiles (nrb = 6): NRB * NMAT accumulators no longer fit the register file, so the
    tile kernel runs one matrix at a time (T parked in the LDS stage between passes).  Against the fp64 oracle."""
    torch.manual_seed(0)
    b = pkg.synthetic.make_batch(["ober179"], 12, seed=0)
    ref = oracle.MPN(8, 6, 2, hid, L, K, 0.0).double()
    mine = pkg.MPN(8, 6, 2, hid, L, K, 0.0)
    mine.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    mine = mine.to(DEV)
    b64 = {"x": b["x"].double(), "edge_index": b["edge_index"], "edge_attr": b["edge_attr"].double()}
    out64, l64 = oracle.train_step(ref, b64, tuple(s.double() for s in b["stats"]))
    x, ei, ea = b["x"].to(DEV), b["edge_index"].to(DEV), b["edge_attr"].to(DEV)
    st = tuple(s.to(DEV) for s in b["stats"])
    assert pkg.topology.get_topology(ei, x.shape[0]).nrb == 6
    out = mine(x[:, :8], ei, ea[:, :6])
    loss = _loss(pkg.data, x, ei, ea, st, out, oracle.DEFAULT_REG_COEFS)
    loss.backward()
    assert rel_err(out, out64) < TOL_OUT
    assert abs(loss.item() - l64.item()) <= TOL_LOSS * abs(l64.item())
    for (n, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        assert rel_err(p.grad, q.grad) < 2e-5, n


def test_fused_loss_finish_is_bitwise_the_two_launch_form_under_repetition(pkg, oracle):
    """data.py:443-459's five batch sums: the LAST workgroup of the partials launch sums the workgroups' partials (DSS2_WLS_FUSED_FINISH;
    handed over through memory without a release fence, csrc/dss2_loss.hip).  Same summation order as the one-workgroup finish launch,
    so the loss and its gradient must agree bit for bit -- on every one of many repetitions with changing outputs (a stale partial of
    an earlier launch in some L2, or an arrival counted before its partial has landed, would show as a wrong sum now and then)."""
    b = pkg.synthetic.make_batch(["cigre14"], 4096, seed=2)        # 240 workgroups in the partials launch
    x, ei, ea = b["x"].to(DEV), b["edge_index"].to(DEV), b["edge_attr"].to(DEV)
    st = tuple(s.to(DEV) for s in b["stats"])
    torch.manual_seed(1)
    outs = [torch.randn(x.shape[0], 2, device=DEV) * (0.1 + 0.05 * i) for i in range(8)]

    def run(o):
        o_leaf = o.clone().requires_grad_(True)
        loss = _loss(pkg.data, x, ei, ea, st, o_leaf * 1.0, oracle.DEFAULT_REG_COEFS)
        loss.backward()
        return loss.detach().clone(), o_leaf.grad.clone()
    old = pkg.flags.WLS_FUSED_FINISH
    try:
        pkg.flags.WLS_FUSED_FINISH = False
        want = [run(o) for o in outs]
        pkg.flags.WLS_FUSED_FINISH = True
        for rep in range(150):
            i = rep % len(outs)
            got = run(outs[i])
            assert torch.equal(got[0], want[i][0]) and torch.equal(got[1], want[i][1]), (rep, got[0].item(), want[i][0].item())
    finally:
        pkg.flags.WLS_FUSED_FINISH = old
    ref = _referee_loss(oracle, b["x"].double(), b["edge_attr"].double(), outs[0].double().cpu(), tuple(s.double() for s in b["stats"]), b["edge_index"])
    assert abs(want[0][0].item() - ref.item()) <= 1e-5 * abs(ref.item())
