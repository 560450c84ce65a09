"""GPU (-m gpu): the whole-stack kernels (csrc/dss2_stack.hip, stack.py) -- the reference driver's own model line
SkipPFN(dim_hid 32, 8 layers, K 2, dropout 0.3, L 5) (/root/reference/dss2_run.py:72-88, networks.py:340-388) as ONE
forward and ONE backward launch.  Checked against the fp64 oracle (on the very dropout masks the kernels applied), against
the per-block kernels (DSS2_STACK_KERNEL=0 path, same library), for bitwise reproducibility, on ragged mixed-topology
tiles, with an input that requires a gradient, and inside a hipGraph with the optimizer.  The golden cases skipmpn / pfn /
skippfn / mpn_undirected_input of tests/test_gpu_parity.py run through this path too (it is the default route)."""
import importlib

import pytest
import torch

from conftest import PKG_NAME, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _stack_mod():
    return importlib.import_module(PKG_NAME + ".stack")


def _blocks(m):
    return list(m.mpns) if hasattr(m, "mpns") else [m]


def _train(pkg, oracle, model, b, with_input_grad=False):
    x, ei, ea = b["x"].to(DEV), b["edge_index"].to(DEV), b["edge_attr"].to(DEV)
    st = tuple(s.to(DEV) for s in b["stats"])
    for q in model.parameters():
        q.grad = None
    xin = x[:, :8].clone().requires_grad_(True) if with_input_grad else x[:, :8]
    out = model(xin, ei, ea[:, :6])
    _train.gates = _kernel_gates(out) if type(out.grad_fn).__name__.startswith("_FusedStackFn") else None
    if out.size(1) == 2:
        loss = pkg.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=out, x_mean=st[0], x_std=st[1], edge_mean=st[2],
                                edge_std=st[3], edge_index=ei, reg_coefs=oracle.DEFAULT_REG_COEFS, num_samples=None,
                                node_param=x[:, 8:], edge_param=ea[:, 6:])
    else:       # a SkipMPN block on its own has 8 outputs: no WLS loss on those
        loss = (out * torch.linspace(-1.0, 1.0, out.numel(), device=out.device).view_as(out)).sum()
    loss.backward()
    grads = [q.grad.detach().clone() for q in model.parameters()]
    # (the loss masks theta at the slack buses in place, data.py:413: `out` is returned as the oracle's train_step returns it)
    return out.detach().clone(), loss.detach().clone(), grads, (xin.grad.detach().clone() if with_input_grad else None)


def _kernel_gates(out):
    """The conv ReLU gates the kernels applied, block by block and layer by layer: the sign of the activations the forward
    launch saved for its backward (stack._FusedStackFn saves acts [blocks][n_hh + 1][N][32]; slot l + 1 = output of conv l)."""
    acts = out.grad_fn.saved_tensors[2]
    return [(acts[b, l] > 0).cpu() for b in range(acts.shape[0]) for l in range(1, acts.shape[1])]


def _oracle_run(pkg, oracle, cls, args, model, b, with_input_grad=False, gates=None):
    """fp64 oracle on the masks the kernels applied in the model's LAST forward.  gates: the kernels' own conv ReLU decisions
    (_kernel_gates).  A pre-activation within fp32 rounding of 0 may take the other sign in another summation order; such a
    flipped gate is not an arithmetic error but moves every gradient below it by ~1/N_nodes, so the
    referee evaluates the kernels' decisions -- which must differ from its own only at razor-edge pre-activations."""
    ref = getattr(oracle, cls)(*args).double()
    ref.load_state_dict({k: v.double().cpu() for k, v in model.state_dict().items()})
    N, hid, p = b["x"].shape[0], args[3], args[6]
    if p > 0:
        for bm, br in zip(_blocks(model), _blocks(ref)):
            snap, pp = bm._last_dropout
            base = getattr(bm, "_drop_base", 0)
            br.dropout_masks = [pkg.networks.dropout_mask(snap, pp, base + l + 1, N, hid).cpu() for l in range(bm.n_gnn_layers - 1)]
    b64 = {"x": b["x"].double(), "edge_index": b["edge_index"], "edge_attr": b["edge_attr"].double()}
    if with_input_grad:
        xin = b64["x"][:, :8].clone().requires_grad_(True)
        b64 = dict(b64, x=torch.cat([xin, b64["x"][:, 8:]], 1))
    real_relu, glist, pos, flips = torch.relu, list(gates or []), [0], [0, 0]

    def pinned_relu(t_):
        # the model's conv ReLUs, in order; the edge MLP's nn.ReLU ([E2, 32]) and the loss's penalty ReLUs are not pinned
        if pos[0] >= len(glist) or tuple(t_.shape) != tuple(glist[pos[0]].shape):
            return real_relu(t_)
        g = glist[pos[0]]
        pos[0] += 1
        flipped = (t_ > 0) != g
        flips[0] += int(flipped.sum())
        flips[1] += flipped.numel()
        if flipped.any():
            assert t_[flipped].abs().max() <= 1e-5 * t_.abs().max(), "a kernel gate differs from the referee's away from 0"
        return t_ * g.to(t_.dtype)
    if gates is not None:
        torch.relu = pinned_relu
    try:
        if args[2] == 2:
            out64, l64 = oracle.train_step(ref, b64, tuple(s.double() for s in b["stats"]))
        else:
            out64 = ref(b64["x"][:, :8], b64["edge_index"], b64["edge_attr"][:, :6])
            l64 = (out64 * torch.linspace(-1.0, 1.0, out64.numel(), dtype=torch.float64).view_as(out64)).sum()
            l64.backward()
    finally:
        torch.relu = real_relu
    if gates is not None:
        assert pos[0] == len(glist), (pos[0], len(glist))
        assert flips[0] <= max(2, 1e-5 * flips[1]), flips
    return ref, out64, l64, (xin.grad if with_input_grad else None)


def _tol(name, tol):
    """The edge MLP's per-edge ReLU gates cannot be pinned from outside (the kernels recompute them); an edge at a razor-edge
    pre-activation that falls the other way under another summation order toggles its whole contribution g * [x_i | x_j | ea]
    to one row of dW1, and the inverse-variance inputs are heavy-tailed: such a row may move by ~1e-2 of the largest entry
    (tests/test_gpu_parity.py::test_baseline_configs_against_oracle bounds exactly this term edge by edge)."""
    return max(tol, 2e-2) if "edge_aggr.edge_aggr.0." in name else tol


@pytest.mark.parametrize("cls,args,grids,B", [
    ("SkipPFN", (8, 6, 2, 32, 8, 2, 0.3, 5), ["cigre14"], 64),                       # the driver's line at the driver's batch size
    ("SkipPFN", (8, 6, 2, 32, 8, 2, 0.0, 5), ["cigre14"], 64),
    ("SkipPFN", (8, 6, 2, 32, 8, 2, 0.3, 5), ["cigre14", "cigre14_reswitched"], 301),   # ragged tiles, a one-cycle topology, a short last tile
    ("PFN", (8, 6, 2, 32, 4, 2, 0.3, 3), ["cigre14"], 37),
    ("MPN", (8, 6, 2, 32, 5, 2, 0.3), ["cigre14"], 50),                               # one block
    ("SkipMPN", (8, 6, 8, 32, 3, 2, 0.0), ["cigre14_reswitched"], 19),                # one block with the residual, 8 outputs
    ("MPN", (8, 6, 2, 32, 2, 2, 0.5), ["cigre14"], 8),                                # a single H -> H layer
])
def test_whole_stack_kernels_match_the_oracle(pkg, oracle, cls, args, grids, B):
    st = _stack_mod()
    torch.manual_seed(0)
    b = pkg.synthetic.make_batch(grids, B, seed=11)
    model = getattr(pkg, cls)(*args).to(DEV)
    with torch.no_grad():                      # biases away from zero (TAGConv's default init) so their gradients' paths carry signal
        for q in model.parameters():
            if q.dim() == 1:
                q.uniform_(-0.2, 0.2)
    torch.manual_seed(5)
    out, loss, grads, _ = _train(pkg, oracle, model, b)
    assert model.__dict__.get("_fused_plan") is not None and _train.gates is not None, "the whole-stack path was not taken"
    ref, out64, l64, _ = _oracle_run(pkg, oracle, cls, args, model, b, gates=_train.gates)
    assert rel_err(out, out64) < 1e-5
    assert abs(loss.item() - l64.item()) <= 1e-5 * abs(l64.item())
    for (n, _), g, r in zip(model.named_parameters(), grads, ref.parameters()):
        assert rel_err(g, r.grad) < _tol(n, 1e-4), n           # conv gates pinned: tight
    # the two GPU routes against each other, nothing pinned: a razor-edge gate that differs between them toggles ONE node's
    # contribution to a weight-gradient row, which weighs ~1 / sqrt(N_nodes) of that row's randomly-signed sum.  (The tight
    # checks are the pinned referee above and the B = 4096 route-vs-route test below.)
    tol = max(1e-4, 1.0 / b["x"].shape[0] ** 0.5)
    # bitwise reproducible: same torch seed -> same masks -> same bits, forward and backward
    torch.manual_seed(5)
    out2, loss2, grads2, _ = _train(pkg, oracle, model, b)
    assert torch.equal(out, out2) and torch.equal(loss, loss2)
    for (n, _), g, g2 in zip(model.named_parameters(), grads, grads2):
        assert torch.equal(g, g2), n
    # the per-block kernels (the route for every other shape) on the same masks: same numbers to fp32 rounding
    st.STACK_KERNEL = False
    try:
        model.__dict__.pop("_fused_route", None)
        for m in _blocks(model):
            m.__dict__.pop("_fused_route", None)
        torch.manual_seed(5)
        out3, loss3, grads3, _ = _train(pkg, oracle, model, b)
    finally:
        st.STACK_KERNEL = True
        model.__dict__.pop("_fused_route", None)
        for m in _blocks(model):
            m.__dict__.pop("_fused_route", None)
    assert rel_err(out3, out) < 1e-5 and abs(loss3.item() - loss.item()) <= 1e-5 * abs(loss.item())
    for (n, _), g, g3 in zip(model.named_parameters(), grads, grads3):
        assert rel_err(g, g3) < _tol(n, tol), n


def test_whole_stack_input_gradient(pkg, oracle):
    """x.requires_grad: the backward launch also returns dx of block 0 (the reference's autograd would)."""
    cls, args = "SkipPFN", (8, 6, 2, 32, 3, 2, 0.3, 3)
    torch.manual_seed(1)
    b = pkg.synthetic.make_batch(["cigre14"], 21, seed=4)
    model = getattr(pkg, cls)(*args).to(DEV)
    torch.manual_seed(9)
    out, loss, grads, dx = _train(pkg, oracle, model, b, with_input_grad=True)
    ref, out64, l64, dx64 = _oracle_run(pkg, oracle, cls, args, model, b, with_input_grad=True, gates=_train.gates)
    tol = 1e-4
    assert rel_err(out, out64) < 1e-5 and rel_err(dx, dx64) < tol
    for (n, _), g, r in zip(model.named_parameters(), grads, ref.parameters()):
        assert rel_err(g, r.grad) < _tol(n, tol), n


def test_whole_stack_full_batch_against_the_per_block_kernels(pkg, oracle):
    """B = 4096 (1024 tiles, four per persistent backward workgroup): the two routes agree, and the fused one is bitwise
    reproducible at that size."""
    st = _stack_mod()
    torch.manual_seed(0)
    b = pkg.synthetic.make_batch(["cigre14"], 4096, seed=3)
    model = pkg.SkipPFN(8, 6, 2, 32, 8, 2, 0.3, 5).to(DEV)
    torch.manual_seed(2)
    out, loss, grads, _ = _train(pkg, oracle, model, b)
    torch.manual_seed(2)
    out2, loss2, grads2, _ = _train(pkg, oracle, model, b)
    assert torch.equal(out, out2) and all(torch.equal(g, g2) for g, g2 in zip(grads, grads2))
    st.STACK_KERNEL = False
    try:
        model.__dict__.pop("_fused_route", None)
        torch.manual_seed(2)
        out3, loss3, grads3, _ = _train(pkg, oracle, model, b)
    finally:
        st.STACK_KERNEL = True
        model.__dict__.pop("_fused_route", None)
    assert rel_err(out3, out) < 1e-5 and abs(loss3.item() - loss.item()) <= 1e-5 * abs(loss.item())
    for (n, _), g, g3 in zip(model.named_parameters(), grads, grads3):
        assert rel_err(g, g3) < 2e-4, n


@pytest.mark.parametrize("n_graphs", [64, 800])      # 800 tiles: the single-wave split-plane chain (f16x3) of one column group, round 6
def test_driver_line_on_ober_sub_matches_the_oracle_on_the_per_block_kernels(pkg, oracle, n_graphs):
    """The reference driver's OTHER branch (/root/reference/dss2_run.py:51-53: Oberrhein, 70 buses) with the driver's model
    (dss2_run.py:72-88: SkipPFN dim_hid 32, 8 layers, K 2, dropout 0.3, L 5).  70-bus graphs need 96-row tiles, which the
    whole-stack kernels do not cover: the stack must run block by block -- one autograd node, in-kernel
    dropout, the layer chains of 96-row tiles -- and agree with the fp64 oracle on the very masks the kernels applied."""
    args = (8, 6, 2, 32, 8, 2, 0.3, 5)
    torch.manual_seed(0)
    b = pkg.synthetic.make_batch(["ober_sub"], n_graphs, seed=3)
    single_wave = n_graphs >= int(pkg._lib.lib().dss2_chain_sp6_single_group_min_tiles()) > 0
    model = pkg.SkipPFN(*args).to(DEV)
    with torch.no_grad():
        for q in model.parameters():
            if q.dim() == 1:
                q.uniform_(-0.2, 0.2)
    torch.manual_seed(5)
    out, loss, grads, _ = _train(pkg, oracle, model, b)
    assert _train.gates is None and model.__dict__.get("_fused_plan") is None, "96-row tiles took the whole-stack kernels?"
    assert bool(model.mpns[0]._plan.f16) == single_wave      # (from 768 tiles on the blocks' chains run as f16x3 on single-wave workgroups)
    ref, out64, l64, _ = _oracle_run(pkg, oracle, "SkipPFN", args, model, b)
    assert rel_err(out, out64) < 1e-5
    assert abs(loss.item() - l64.item()) <= 1e-5 * abs(l64.item())
    # conv gates NOT pinned here (40 gated layers deep): a razor-edge gate that falls the other way moves a gradient by ~1 / N_nodes
    # (the number of razor-edge gates grows with the batch as fast as one gate's weight shrinks: 72 M gates at 800 graphs, 2.1e-4 measured)
    tol = max(2e-4 if n_graphs <= 64 else 4e-4, 8.0 / b["x"].shape[0])
    for (n, _), g, r in zip(model.named_parameters(), grads, ref.parameters()):
        assert rel_err(g, r.grad) < _tol(n, tol), (n, rel_err(g, r.grad))
    torch.manual_seed(5)      # bitwise reproducible on this route too
    out2, loss2, grads2, _ = _train(pkg, oracle, model, b)
    assert torch.equal(out, out2) and torch.equal(loss, loss2) and all(torch.equal(g, g2) for g, g2 in zip(grads, grads2))


def test_whole_stack_route_covers_only_what_it_was_built_for(pkg):
    st = _stack_mod()
    b = pkg.synthetic.make_batch(["cigre14"], 8, seed=0)
    bo = pkg.synthetic.make_batch(["ober_sub"], 4, seed=0)
    topo = pkg.topology.get_topology(b["edge_index"].to(DEV), b["x"].shape[0])
    topo_o = pkg.topology.get_topology(bo["edge_index"].to(DEV), bo["x"].shape[0])
    yes = [pkg.SkipPFN(8, 6, 2, 32, 8, 2, 0.3, 5), pkg.PFN(8, 6, 2, 32, 2, 2, 0.0, 2), pkg.MPN(8, 6, 2, 32, 4, 2, 0.0)]
    no = [pkg.MPN(8, 6, 2, 64, 4, 2, 0.0), pkg.MPN(8, 6, 2, 32, 1, 2, 0.0), pkg.MPN(8, 6, 2, 32, 3, 3, 0.0),
          pkg.MPN(8, 6, 2, 32, 10, 2, 0.0)]
    for m in yes:
        assert st.supported(_blocks(m), topo) is not None, type(m).__name__
        assert st.supported(_blocks(m), topo_o) is None           # 70-bus graphs: 96-row tiles, per-block kernels
    for m in no:
        assert st.supported(_blocks(m), topo) is None


def test_whole_stack_training_step_inside_a_hipgraph(pkg, oracle):
    """forward + loss + backward + FusedAdamax(capturable) of the driver's model as ONE replayed hipGraph: every replay draws
    new masks and trains; the parameters after k replays equal k eager steps on the same masks (same device-side offsets)."""
    torch.manual_seed(0)
    b = pkg.synthetic.make_batch(["cigre14"], 64, seed=1)
    x, ei, ea = b["x"].to(DEV), b["edge_index"].to(DEV), b["edge_attr"].to(DEV)
    st = tuple(s.to(DEV) for s in b["stats"])
    model = pkg.SkipPFN(8, 6, 2, 32, 8, 2, 0.3, 5).to(DEV)
    opt = pkg.FusedAdamax(model.parameters(), lr=3e-3, capturable=True)
    losses = []

    def train_step():
        for q in model.parameters():
            q.grad = None
        out = model(x[:, :8], ei, ea[:, :6])
        loss = pkg.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=out, x_mean=st[0], x_std=st[1], edge_mean=st[2],
                                edge_std=st[3], edge_index=ei, reg_coefs=oracle.DEFAULT_REG_COEFS, num_samples=None,
                                node_param=x[:, 8:], edge_param=ea[:, 6:])
        loss.backward()
        opt.step()
        return loss
    g = pkg.graphs.GraphedStep(train_step)
    for _ in range(30):
        losses.append(float(g.replay().item()))
    snap, _ = model._last_dropout
    assert int(snap[1].item()) >= 29                                   # the captured pack kernel advances the dropout offset
    assert all(l == l and l < 1e30 for l in losses)                   # finite
    assert min(losses[-5:]) < losses[0]                                # it trains
    assert len(set(losses)) > 25                                       # new masks every replay


def test_graphed_trainer_equals_the_eager_training_loop(pkg, oracle):
    """runner.GraphedTrainer (the training step captured once per batch shape, replayed on static buffers) against
    runner.train_epoch (eager) on the same batches from the same initial weights, dropout 0: same losses, same weights."""
    torch.manual_seed(0)
    batches = [pkg.synthetic.make_batch(["cigre14"], 64, seed=20 + k) for k in range(3)]
    batches.append(pkg.synthetic.make_batch(["cigre14"], 24, seed=99, stats=batches[0]["stats"]))      # a short last batch: second shape
    st = tuple(s.to(DEV) for s in batches[0]["stats"])
    dev_b = [{k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in b.items() if k != "stats"} for b in batches]
    m_e = pkg.SkipPFN(8, 6, 2, 32, 4, 2, 0.0, 3).to(DEV)
    m_g = pkg.SkipPFN(8, 6, 2, 32, 4, 2, 0.0, 3).to(DEV)
    m_g.load_state_dict(m_e.state_dict())
    o_e = pkg.FusedAdamax(m_e.parameters(), lr=3e-3)
    o_g = pkg.FusedAdamax(m_g.parameters(), lr=3e-3, capturable=True)
    trainer = pkg.runner.GraphedTrainer(m_g, o_g, st, pkg.runner.REG_COEFS)
    for epoch in range(3):
        l_e = pkg.runner.train_epoch(m_e, o_e, dev_b, st, pkg.runner.REG_COEFS)
        l_g = pkg.runner.train_epoch_graphed(trainer, dev_b)
        # (same kernels; the only difference is Adamax's bias correction 1 - beta1^t, evaluated by the host's powf in the eager
        #  optimizer and by the device's in the capturable one: an ulp that twelve training steps amplify to ~1e-5)
        assert abs(l_e - l_g) <= 1e-4 * abs(l_e), (epoch, l_e, l_g)
    assert len(trainer.graphs) == 2
    for (n, a), (_, c) in zip(m_e.named_parameters(), m_g.named_parameters()):
        assert rel_err(c, a) < 1e-4, n
