"""GPU (-m gpu), round-2 additions: anchors and edge cases called out by the round-1 review.
All calls go through the C ABI (ctypes -> libdss2_hip.so)."""
import numpy as np
import pytest
import torch

from conftest import (LOSS_CASES, MULTI_CASES, case_batch, case_grads, case_state_dict, golden, load_pkg, multi_case_data, rel_err, t,
                      tagconv_known_answers)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def pkg():
    p = load_pkg()
    p._lib.lib()
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return p


def test_tagconv_matches_hand_derived_known_answers(pkg):
    """PyG TAGConv semantics against paper-and-pencil literals (tests/golden/tagconv_known_answers.json), on two
    block-diagonal copies of each 3-node graph so that batching offsets are exercised too."""
    x, lins, bias, cases = tagconv_known_answers()
    for name, (ei, exp) in cases.items():
        ei2 = torch.cat([ei, ei + 3], 1).to(DEV)
        x2 = torch.cat([x, x], 0).float().to(DEV)
        for K, want in exp.items():
            conv = pkg.TAGConv(2, 2, K)
            with torch.no_grad():
                conv.bias.copy_(bias.float())
                for k in range(K + 1):
                    conv.lins[k].weight.copy_(lins[k].float())
            got = conv.to(DEV)(x2, ei2)
            want2 = torch.cat([want, want], 0)
            assert (got.double().cpu() - want2).abs().max() < 2e-6, (name, K, got, want2)


def test_get_pflow_phase_shift_false_matches_reference(pkg):
    """get_pflow(..., phase_shift=False) (shift = edge_param[:, 5], data.py:364-365) vs the reference's own output."""
    gs = golden("case_pflow_shift.npz")
    for name in LOSS_CASES:
        g = golden(f"case_{name}.npz")
        b = case_batch(g, device=DEV)
        x, ei, ea, st = b["x"], b["edge_index"], b["edge_attr"], b["stats"]
        o = t(g["output_after"], device=DEV)
        yv = torch.cat([o[:, 0:1] * st[1][:1] + st[0][:1], o[:, 1:]], 1)
        flows = torch.stack(pkg.data.get_pflow(yv, ei, x[:, 8:], ea[:, 6:], phase_shift=False), 1)
        assert rel_err(flows, t(gs[f"{name}/pflow_shift"])) < 1e-5
        flows0 = torch.stack(pkg.data.get_pflow(yv, ei, x[:, 8:], ea[:, 6:]), 1)
        assert rel_err(flows0, t(g["pflow"])) < 1e-5


@pytest.mark.parametrize("L,hid", [(10, 32), (18, 32), (10, 128)])
def test_deep_stacks_chunk_the_layer_chain(pkg, oracle, L, hid):
    """n_gnn_layers - 1 > 8 hid->hid layers: the library chains at most 8 layers per launch, deeper stacks run as
    consecutive chain launches (forward and data-gradient).  Against the fp64 oracle."""
    torch.manual_seed(1)
    b = pkg.synthetic.make_batch(["cigre14", "cigre14_reswitched"], 24, seed=4)
    ref = oracle.MPN(8, 6, 2, hid, L, 2, 0.0).double()
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if "convs" in n and "lins" in n:
                p.mul_(1.6)        # keep activations alive through 17 ReLU layers
    mine = pkg.MPN(8, 6, 2, hid, L, 2, 0.0)
    mine.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    mine = mine.to(DEV)
    launches = []
    b64 = {"x": b["x"].double(), "edge_index": b["edge_index"], "edge_attr": b["edge_attr"].double()}
    out64, l64 = oracle.train_step(ref, b64, tuple(s.double() for s in b["stats"]))
    x, ei, ea = b["x"].to(DEV), b["edge_index"].to(DEV), b["edge_attr"].to(DEV)
    st = tuple(s.to(DEV) for s in b["stats"])
    real_chain = pkg.networks.gemm_prop_chain
    def counting(topo, X, h, nmat, layers, **kw):
        if len(layers) <= pkg.flags.CHAIN_MAX:
            launches.append(len(layers))
        return real_chain(topo, X, h, nmat, layers, **kw)
    pkg.networks.gemm_prop_chain = pkg.ops.gemm_prop_chain = counting      # (ops: the chunks of a deep stack are its own recursive calls)
    try:
        out = mine(x[:, :8], ei, ea[:, :6])
        loss = pkg.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=out, x_mean=st[0], x_std=st[1], edge_mean=st[2],
                                edge_std=st[3], edge_index=ei, reg_coefs=oracle.DEFAULT_REG_COEFS, num_samples=None,
                                node_param=x[:, 8:], edge_param=ea[:, 6:])
        loss.backward()
    finally:
        pkg.networks.gemm_prop_chain = pkg.ops.gemm_prop_chain = real_chain
    if pkg.flags.CHAIN_LAYERS:      # (DSS2_CHAIN=0: one launch per layer, nothing to count)
        assert max(launches) <= 8 and sum(launches) == 2 * (L - 1), launches      # forward + data-gradient chains
    assert rel_err(out, out64) < 1e-5
    assert abs(loss.item() - l64.item()) <= 1e-5 * abs(l64.item())
    for (n, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        assert rel_err(p.grad, q.grad) < 1e-4, n


def _star_batch(n_graphs, leaves, seed):
    """Star graphs (one hub with `leaves` neighbours): doubled in-degree of the hub = leaves > the ELL width limit (8),
    so every kernel takes its general CSR path (row-per-wave edge MLP, CSR staging in the tile kernels)."""
    g = torch.Generator().manual_seed(seed)
    n = leaves + 1
    src = torch.zeros(leaves, dtype=torch.int64)
    dst = torch.arange(1, n)
    ei = torch.cat([torch.stack([src, dst]) + k * n for k in range(n_graphs)], 1)
    x = torch.randn(n_graphs * n, 8, generator=g)
    ea = torch.randn(ei.shape[1], 6, generator=g)
    return x, ei, ea


@pytest.mark.parametrize("cls,hid,L,leaves", [("MPN", 64, 3, 11), ("SkipMPN", 32, 2, 20)])
def test_hub_graphs_take_the_csr_paths(pkg, oracle, cls, hid, L, leaves):
    torch.manual_seed(2)
    x, ei, ea = _star_batch(7, leaves, seed=5)
    dim_out = 8 if cls == "SkipMPN" else 2
    ref = getattr(oracle, cls)(8, 6, dim_out, hid, L, 2, 0.0).double()
    mine = getattr(pkg, cls)(8, 6, dim_out, hid, L, 2, 0.0)
    mine.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    mine = mine.to(DEV)
    g = torch.randn(x.shape[0], dim_out)
    xr = x.double().requires_grad_(True)
    outr = ref(xr, ei, ea.double())
    outr.backward(g.double())
    xm = x.to(DEV).requires_grad_(True)
    topo = pkg.topology.get_topology(ei.to(DEV), x.shape[0])
    assert topo.ell == 0 and topo.ellT == 0           # hubs: no ELL slices, CSR everywhere
    outm = mine(xm, ei.to(DEV), ea.to(DEV))
    outm.backward(g.to(DEV))
    assert rel_err(outm, outr) < 1e-5
    assert rel_err(xm.grad, xr.grad) < 1e-4
    for (n, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        assert rel_err(p.grad, q.grad) < 1e-4, n


def test_fused_adamax_descriptor_table_is_reused(pkg):
    """The descriptor table is rebuilt only when an address changes: a steady training loop builds it once (or twice,
    while the allocator settles), and load_state_dict drops it (its state pointers are stale)."""
    torch.manual_seed(0)
    b = pkg.synthetic.make_batch(["cigre14"], 32, seed=1)
    x, ei, ea = b["x"].to(DEV), b["edge_index"].to(DEV), b["edge_attr"].to(DEV)
    model = pkg.MPN(8, 6, 2, 32, 3, 2, 0.0).to(DEV)
    opt = pkg.FusedAdamax(model.parameters(), lr=3e-3)
    for _ in range(8):
        opt.zero_grad()
        model(x[:, :8], ei, ea[:, :6]).square().mean().backward()
        opt.step()
    assert opt.table_builds == 1, opt.table_builds
    sd = opt.state_dict()
    opt.load_state_dict(sd)
    builds = opt.table_builds
    opt.zero_grad()
    model(x[:, :8], ei, ea[:, :6]).square().mean().backward()
    opt.step()
    assert opt.table_builds == builds + 1             # rebuilt against the re-loaded state tensors


def test_gradient_of_other_consumers_of_the_masked_output(pkg, oracle):
    """gsp_wls_edge masks theta in place (data.py:413); a second consumer of that (masked) output contributes a
    gradient that must not reach the pre-mask theta at slack buses -- exactly what autograd does in the reference."""
    g = golden("case_loss_violate_cigre.npz")
    b = case_batch(g, device=DEV)
    bc = case_batch(g)
    wts = torch.linspace(-1.0, 2.0, b["x"].shape[0] * 2).view(-1, 2)

    def run(mod, bb, dev, leaf):
        x, ei, ea, st = bb["x"], bb["edge_index"], bb["edge_attr"], bb["stats"]
        o = leaf * 1.0
        loss = mod.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=o, x_mean=st[0], x_std=st[1], edge_mean=st[2],
                                edge_std=st[3], edge_index=ei, reg_coefs=oracle.DEFAULT_REG_COEFS, num_samples=None,
                                node_param=x[:, 8:], edge_param=ea[:, 6:])
        (loss + (o * wts.to(dev)).sum()).backward()
        return leaf.grad

    lg = t(g["output"], device=DEV).clone().requires_grad_(True)
    lc = t(g["output"]).clone().requires_grad_(True)
    gg, gc = run(pkg.data, b, DEV, lg), run(oracle, bc, "cpu", lc)
    assert rel_err(gg, gc) < 1e-5
    slack = bc["x"][:, 9] > 0
    assert slack.any() and (gg.cpu()[slack, 1] == 0).all()


@pytest.mark.parametrize("feat", [7, 128])
def test_message_passing_propagate_with_a_user_message(pkg, oracle, feat):
    """PyG MessagePassing.propagate semantics on the HIP gather / segmented-sum kernels: *_j from the
    source end, *_i from the target end, other arguments by name, unknown keyword arguments ignored, sum per target.
    A toy subclass with its own message(), forward and backward, against index ops + index_add_ in fp64."""
    torch.manual_seed(4)
    b = pkg.synthetic.make_batch(["cigre14", "cigre14_reswitched"], 11, seed=2)
    ei, _ = oracle.undirect_graph(b["edge_index"], b["edge_attr"][:, :6])
    N, E = b["x"].shape[0], ei.shape[1]
    lin = torch.nn.Linear(5, feat).double()

    def message(x_i, x_j, edge_attr, norm, scale=0.5):
        return norm.view(-1, 1) * torch.tanh(lin(x_j - x_i)) * edge_attr[:, :3].sum(1, keepdim=True) * scale

    class Toy(pkg.MessagePassing):
        def __init__(self):
            super().__init__(aggr="add")
            self.lin = torch.nn.Linear(5, feat)

        def message(self, x_i, x_j, edge_attr, norm, scale=0.5):
            return norm.view(-1, 1) * torch.tanh(self.lin(x_j - x_i)) * edge_attr[:, :3].sum(1, keepdim=True) * scale

        def forward(self, x, edge_index, edge_attr, norm):
            return self.propagate(edge_index, x=x, edge_attr=edge_attr, norm=norm, not_a_message_argument=123)

    toy = Toy()
    toy.lin.load_state_dict({k: v.float() for k, v in lin.state_dict().items()})
    toy = toy.to(DEV)
    x, ea, nrm = torch.randn(N, 5), torch.randn(E, 4), torch.rand(E) + 0.5
    g = torch.randn(N, feat)
    xr, ear, nr = (v.double().requires_grad_(True) for v in (x, ea, nrm))
    out_r = oracle.scatter_sum(message(xr[ei[1]], xr[ei[0]], ear, nr), ei[1], N)
    out_r.backward(g.double())
    xm, eam, nm = (v.to(DEV).requires_grad_(True) for v in (x, ea, nrm))
    out_m = toy(xm, ei.to(DEV), eam, nm)
    out_m.backward(g.to(DEV))
    assert rel_err(out_m, out_r) < 1e-5
    assert rel_err(xm.grad, xr.grad) < 1e-5 and rel_err(eam.grad, ear.grad) < 1e-5 and rel_err(nm.grad, nr.grad) < 1e-5
    assert rel_err(toy.lin.weight.grad, lin.weight.grad) < 1e-5 and rel_err(toy.lin.bias.grad, lin.bias.grad) < 1e-5
    out2 = toy(xm, ei.to(DEV), eam, nm)
    assert torch.equal(out2, out_m)                       # deterministic (no float atomics)
    base = pkg.MessagePassing()                            # default message(x_j) = x_j: out[i] = sum of the sources' rows
    agg = base.propagate(ei.to(DEV), x=xm.detach())
    assert rel_err(agg, oracle.scatter_sum(x.double()[ei[0]], ei[1], N)) < 1e-6


def test_edge_aggregation_message_and_propagate_are_the_reference_expression(pkg, oracle):
    """EdgeAggregation.message() is networks.py:181 and .propagate() runs it through the generic engine; forward() (the fused
    kernels: aggregate, then the second Linear) must give the same numbers."""
    torch.manual_seed(6)
    b = pkg.synthetic.make_batch(["ober_sub"], 3, seed=8)
    x, ea = b["x"][:, :8], b["edge_attr"][:, :6]
    ei2, ea2 = oracle.undirect_graph(b["edge_index"], ea)
    ref = oracle.EdgeAggregation(8, 6, 64, 64).double()
    mine = pkg.EdgeAggregation(8, 6, 64, 64)
    mine.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    mine = mine.to(DEV)
    xd, ed, eid = x.to(DEV).contiguous(), ea2.to(DEV).contiguous(), ei2.to(DEV)
    out_ref = ref(x.double(), ei2, ea2.double())
    out_fused = mine(xd, eid, ed)
    out_prop = mine.propagate(eid, x=xd, edge_attr=ed, norm=torch.ones(ei2.shape[1], device=DEV))   # norm: ignored, as in networks.py:206
    assert rel_err(out_fused, out_ref) < 1e-5 and rel_err(out_prop, out_ref) < 1e-5
    msg = mine.message(xd[eid[1]], xd[eid[0]], ed)
    assert rel_err(msg, ref.edge_aggr(torch.cat([x.double()[ei2[1]], x.double()[ei2[0]], ea2.double()], -1))) < 1e-5


def _export_masks(pkg, block, n_rows, hid):
    snap, p = block._last_dropout
    base = getattr(block, "_drop_base", 0)          # blocks of a PFN stack share one snapshot, ids offset per block
    return [pkg.networks.dropout_mask(snap, p, base + l + 1, n_rows, hid).cpu() for l in range(block.n_gnn_layers - 1)]


@pytest.mark.parametrize("cls,args,grids,B", [
    ("MPN", (8, 6, 2, 32, 4, 2, 0.3), ["cigre14"], 64),                        # chained layers, H = 32
    ("MPN", (8, 6, 2, 128, 4, 2, 0.3), ["cigre14", "cigre14_reswitched"], 96),   # C2's model with the driver's dropout rate
    ("MPN", (8, 6, 2, 64, 2, 2, 0.5), ["ober_sub"], 6),                        # single hidden layer (no chain), 96-row tiles
    ("SkipPFN", (8, 6, 2, 32, 3, 2, 0.3, 3), ["cigre14"], 48),                 # the driver's model line, small
])
def test_in_kernel_dropout_matches_the_oracle_on_the_same_masks(pkg, oracle, cls, args, grids, B):
    """dropout_rate > 0 (the reference driver runs 0.3, dss2_run.py:80): the masks are regenerated inside the forward and
    backward epilogues (Philox, no [N, H] mask tensors).  The oracle consumes the very masks the kernels applied
    (dss2_dropout_mask); outputs, loss and every gradient must then agree like in the p = 0 tests."""
    torch.manual_seed(0)
    b = pkg.synthetic.make_batch(grids, B, seed=6)
    ref = getattr(oracle, cls)(*args).double()
    mine = getattr(pkg, cls)(*args)
    mine.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    mine = mine.to(DEV)
    x, ei, ea = b["x"].to(DEV), b["edge_index"].to(DEV), b["edge_attr"].to(DEV)
    st = tuple(s.to(DEV) for s in b["stats"])
    torch.manual_seed(123)
    out = mine(x[:, :8], ei, ea[:, :6])
    out_pre = out.detach().clone()                 # (the loss masks theta in place later)
    blocks_m = list(mine.mpns) if hasattr(mine, "mpns") else [mine]
    blocks_r = list(ref.mpns) if hasattr(ref, "mpns") else [ref]
    N, hid, p = x.shape[0], args[3], args[6]
    for bm, br in zip(blocks_m, blocks_r):
        br.dropout_masks = _export_masks(pkg, bm, N, hid)
        for m in br.dropout_masks:
            vals = set(torch.unique(m).tolist())
            assert vals <= {0.0, float(torch.tensor(1.0 / (1.0 - p), dtype=torch.float32))}
            assert abs((m == 0).float().mean().item() - p) < 0.02          # the rate is what was asked for
    masks = [m for br in blocks_r for m in br.dropout_masks]
    assert all(not torch.equal(masks[0], m) for m in masks[1:])              # every layer / block has its own mask
    loss = pkg.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=out, x_mean=st[0], x_std=st[1], edge_mean=st[2],
                            edge_std=st[3], edge_index=ei, reg_coefs=oracle.DEFAULT_REG_COEFS, num_samples=None,
                            node_param=x[:, 8:], edge_param=ea[:, 6:])
    loss.backward()
    b64 = {"x": b["x"].double(), "edge_index": b["edge_index"], "edge_attr": b["edge_attr"].double()}
    out64, l64 = oracle.train_step(ref, b64, tuple(s.double() for s in b["stats"]))
    assert rel_err(out, out64) < 1e-5
    assert abs(loss.item() - l64.item()) <= 1e-5 * abs(l64.item())
    for (n, q), (_, r) in zip(mine.named_parameters(), ref.named_parameters()):
        assert rel_err(q.grad, r.grad) < 1e-4, n
    # same torch seed -> same masks (bitwise the same output); another draw -> other masks
    torch.manual_seed(123)
    out2 = mine(x[:, :8], ei, ea[:, :6])
    out3 = mine(x[:, :8], ei, ea[:, :6])
    assert torch.equal(out2.detach(), out_pre) and not torch.equal(out3.detach(), out2.detach())


def test_in_kernel_dropout_inside_a_hipgraph_draws_new_masks_per_replay(pkg, oracle):
    torch.manual_seed(0)
    b = pkg.synthetic.make_batch(["cigre14"], 64, seed=2)
    x, ei, ea = b["x"].to(DEV), b["edge_index"].to(DEV), b["edge_attr"].to(DEV)
    model = pkg.MPN(8, 6, 2, 32, 3, 2, 0.3).to(DEV)
    outs = []

    def step():
        for q in model.parameters():
            q.grad = None
        out = model(x[:, :8], ei, ea[:, :6])
        out.square().mean().backward()
        return out

    gs = pkg.graphs.GraphedStep(step)
    for _ in range(3):
        o = gs.replay()
        snap, p = model._last_dropout
        outs.append((o.detach().clone(), snap.clone(), model.convs[0].bias.grad.clone()))
    torch.cuda.synchronize()
    assert not torch.equal(outs[0][0], outs[1][0]) and not torch.equal(outs[1][0], outs[2][0])
    offs = [int(s[1][1]) for s in outs]
    assert offs[1] == offs[0] + 1 and offs[2] == offs[1] + 1                # the captured kernel advances the device-side offset
    # forward and backward of one replay used the same masks: the replayed gradient equals an eager step on those masks
    m_last = [pkg.networks.dropout_mask(outs[2][1], 0.3, l + 1, x.shape[0], 32).cpu() for l in range(2)]
    ref = oracle.MPN(8, 6, 2, 32, 3, 2, 0.3).double()
    ref.load_state_dict({k: v.double().cpu() for k, v in model.state_dict().items()})
    ref.dropout_masks = m_last
    o64 = ref(b["x"][:, :8].double(), b["edge_index"], b["edge_attr"][:, :6].double())
    o64.square().mean().backward()
    assert rel_err(outs[2][0], o64) < 1e-5 and rel_err(outs[2][2], ref.convs[0].bias.grad) < 1e-4


def test_capturable_adamax_inside_the_step_graph(pkg, oracle):
    """FusedAdamax(capturable=True): the step count lives on the device, so forward + loss + backward + optimizer step
    replay as ONE hipGraph; three replays equal three eager steps with the host-counted optimizer."""
    torch.manual_seed(0)
    b = pkg.synthetic.make_batch(["cigre14"], 32, seed=4)
    x, ei, ea = b["x"].to(DEV), b["edge_index"].to(DEV), b["edge_attr"].to(DEV)
    st = tuple(s.to(DEV) for s in b["stats"])
    m_e = pkg.SkipPFN(8, 6, 2, 32, 3, 2, 0.0, 2).to(DEV)
    m_g = pkg.SkipPFN(8, 6, 2, 32, 3, 2, 0.0, 2).to(DEV)
    m_g.load_state_dict(m_e.state_dict())
    o_e = pkg.FusedAdamax(m_e.parameters(), lr=3e-3)
    o_g = pkg.FusedAdamax(m_g.parameters(), lr=3e-3, capturable=True)

    def make_step(model, opt):
        def step():
            for q in model.parameters():
                q.grad = None
            out = model(x[:, :8], ei, ea[:, :6])
            loss = pkg.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=out, x_mean=st[0], x_std=st[1], edge_mean=st[2],
                                    edge_std=st[3], edge_index=ei, reg_coefs=oracle.DEFAULT_REG_COEFS, num_samples=None,
                                    node_param=x[:, 8:], edge_param=ea[:, 6:])
            loss.backward()
            opt.step()
            return loss
        return step

    # warm-up WITHOUT optimizer steps (they would move the weights): plans, allocator, optimizer state
    o_g.init_state()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            m_g(x[:, :8], ei, ea[:, :6]).sum().backward()
    torch.cuda.current_stream().wait_stream(side)
    for q in m_g.parameters():
        q.grad = None
    gs = pkg.graphs.GraphedStep(make_step(m_g, o_g), warmup=0)
    lg = [gs.replay().item() for _ in range(3)]
    eager = make_step(m_e, o_e)
    with torch.cuda.stream(gs.stream):
        le = [eager().item() for _ in range(3)]
    torch.cuda.synchronize()
    # (the bias correction 1 - beta1^step is a device powf in one optimizer and a host powf in the other: ulp-level)
    assert all(abs(a - c) <= 1e-5 * abs(c) for a, c in zip(lg, le)), (lg, le)
    assert le[2] != le[0] and lg[2] != lg[0]                # the weights moved
    for a, c in zip(m_g.parameters(), m_e.parameters()):
        assert rel_err(a, c) < 1e-5
    assert float(o_g.state[next(iter(m_g.parameters()))]["step"]) == 3.0


@pytest.mark.parametrize("name", list(MULTI_CASES))
def test_multi_variants_match_reference_golden(pkg, name):
    """MaskEmbdMPN / MultiMPN / MaskEmbdMultiMPN / MaskEmbdMultiMPN_NoMP on the HIP kernels against the reference's own
    outputs and gradients: same state_dict keys, forward(data), no sign flips on the reverse edges, EdgeAggregation on
    the hidden activation (node features of width dim_hid)."""
    cls, args = MULTI_CASES[name]
    g = golden(f"case_{name}.npz")
    model = getattr(pkg, cls)(*args)
    res = model.load_state_dict(case_state_dict(g), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model = model.to(DEV)
    out = model(multi_case_data(g, device=DEV))
    assert rel_err(out, t(g["out"])) < 1e-5
    out.backward(t(g["gout"], device=DEV))
    grads = case_grads(g)
    for k, p in model.named_parameters():
        assert p.grad is not None, k
        assert rel_err(p.grad, grads[k]) < 1e-4, k


def test_multimpn_with_dropout_and_general_dims(pkg, oracle):
    """MultiMPN with dropout 0.3 (masks handed to the oracle) and an EdgeAggregation whose dims are not the data's 8 / 6."""
    import types
    torch.manual_seed(3)
    b = pkg.synthetic.make_batch(["cigre14_reswitched"], 20, seed=5)
    x, ei, ea = b["x"][:, :8], b["edge_index"], b["edge_attr"][:, :6]
    ref = oracle.MultiMPN(8, 6, 2, 64, 3, 2, 0.3).double()
    mine = pkg.MultiMPN(8, 6, 2, 64, 3, 2, 0.3)
    mine.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    mine = mine.to(DEV)
    out = mine(types.SimpleNamespace(x=x.to(DEV), edge_index=ei.to(DEV), edge_attr=ea.to(DEV)))
    N = x.shape[0]
    ref.dropout_masks = []
    for layer in list(mine.layers)[:-1]:
        snap, p = layer._last_dropout
        ref.dropout_masks.append(pkg.networks.dropout_mask(snap, p, 1, N, 64).cpu())
    g = torch.randn(N, 2)
    out.backward(g.to(DEV))
    out64 = ref(types.SimpleNamespace(x=x.double(), edge_index=ei, edge_attr=ea.double()))
    out64.backward(g.double())
    assert rel_err(out, out64) < 1e-5
    for (n, q), (_, r) in zip(mine.named_parameters(), ref.named_parameters()):
        assert rel_err(q.grad, r.grad) < 1e-4, n
    # standalone EdgeAggregation with arbitrary feature widths (5 node features, 3 edge features), edge list as given
    ei2, _ = oracle.undirect_graph_same_attr(ei, ea)
    ea2 = torch.randn(ei2.shape[1], 3)
    x5 = torch.randn(N, 5)
    r2 = oracle.EdgeAggregation(5, 3, 48, 10).double()
    m2 = pkg.EdgeAggregationGeneral(5, 3, 48, 10)
    m2.load_state_dict({k: v.float() for k, v in r2.state_dict().items()})
    m2 = m2.to(DEV)
    xr = x5.double().requires_grad_(True)
    o_r = r2(xr, ei2, ea2.double())
    gg = torch.randn(N, 10)
    o_r.backward(gg.double())
    xm = x5.to(DEV).requires_grad_(True)
    o_m = m2(xm, ei2.to(DEV), ea2.to(DEV))
    o_m.backward(gg.to(DEV))
    assert rel_err(o_m, o_r) < 1e-5 and rel_err(xm.grad, xr.grad) < 1e-4
    for (n, q), (_, r) in zip(m2.named_parameters(), r2.named_parameters()):
        assert rel_err(q.grad, r.grad) < 1e-4, n
