"""GPU: the data-parallel step with GLOBAL sums that differ from the local ones, emulated on ONE GPU.

The reference loss is built from means over the whole batch and SQUARES of such means (/root/reference/data.py:450-455), so a
rank's loss value and every gradient scale depend on the other ranks' sums and counts.  At world size 1 the all-reduce is the
identity, and a kernel that used its local N or E in one gradient scale would pass every world-size-1 test.  Here a
mixed-topology batch with all three penalties active is cut into unequal whole-graph shards (2 and 8) and each shard runs the
PRODUCT's data-parallel branch -- ``gsp_wls_edge(..., group=...)``: ``dss2_wls_loss_partials`` -> ``parallel.allreduce_loss_sums``
-> ``dss2_wls_loss_value`` -> the gradient kernel scaled by the global counts -> the model backward -> the flat gradient bucket
handed to ``parallel.allreduce_flat_grads`` -- with the two collectives replaced by stand-ins that do on one device what RCCL
does over eight: pass 1 collects every shard's eight sums, pass 2 hands each shard the total and adds up the buckets.

Bars: loss and every gradient equal the single-batch HIP step to 1e-6 and the fp64 oracle on the whole batch to 1e-5 (loss) /
2e-5 (gradients, max-normalised; the bar test_tall_tiles_matrix_sequential uses against the fp64 oracle; the edge MLP's first
Linear 2e-3: its per-edge gates are un-pinnable).  The first bar needs every shard on the same kernels as the whole batch:
found on the way, a 29-graph shard tiled at 96 rows and left the whole-stack kernels (stack.tiles_of now asks for its own
64-row tiling), and the general route kept its un-gated last output alive across the loss's in-place mask (multi.py)."""
import pytest
import torch

from conftest import load_pkg, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL_SHARD, TOL_ORACLE = 1e-6, 2e-5


@pytest.fixture(autouse=True)
def same_tile_height_for_every_shard(monkeypatch):
    """The 1e-6 bar compares a sharded step with the single-batch step, so every shard must run the kernels the whole batch runs.
    A 5-graph shard tiles best at 32 or 96 rows and then takes another form of the layer chain; until round 4 all forms summed in
    the same order (bitwise equal results), since the 64-row chain's tile GEMM moved to 16x16x32 MFMAs they differ by rounding --
    nothing a data-parallel run cares about, but 1e-6 is below it.  So the tile height is pinned for this file (read per structure
    build, topology.py)."""
    monkeypatch.setenv("DSS2_NRB", "2")


@pytest.fixture(scope="module")
def pkg():
    p = load_pkg()
    p._lib.lib()
    assert torch.cuda.is_available()
    return p


MODELS = {
    # BASELINE C2's model: split-plane layer chain, bf16x6 weight gradient, folded conv 0, narrow head
    "MPN_C2_model": ("MPN", (8, 6, 2, 128, 4, 2, 0.0)),
    # the reference driver's shape of model (dss2_run.py:72-82): five blocks, ONE bucket for the stack, whole-stack kernels
    "SkipPFN_5_blocks": ("SkipPFN", (8, 6, 2, 32, 3, 2, 0.0, 5)),
    # input widths other than (8, 6): the general route (per-layer autograd nodes; ADVICE r3: its bucket hooks)
    "MPN_general_dims": ("MPN", (7, 5, 2, 32, 3, 2, 0.0)),
}


def _build(pkg, oracle, name):
    cls, args = MODELS[name]
    torch.manual_seed(0)
    ref = getattr(oracle, cls)(*args).double()
    with torch.no_grad():       # outputs outside the penalty bands: J_v, J_theta and J_loading all active
        blocks = list(ref.mpns) if hasattr(ref, "mpns") else [ref]
        for lin in blocks[-1].convs[-1].lins:
            lin.weight *= 3.0
        for n, p in ref.named_parameters():
            if n.endswith("bias") and "convs" in n:
                p.uniform_(-0.1, 0.1)
    mine = getattr(pkg, cls)(*args)
    mine.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    return ref, mine.to(DEV)


def _batch(pkg, n_graphs, fn, fe):
    b = pkg.synthetic.make_batch(["cigre14", "cigre14_reswitched"], n_graphs, seed=3, violate=0.5)
    st = [s.clone() for s in b["stats"]]
    st[1][0] *= 8.0             # sigma_V: v = out * sigma + mu leaves [0.9, 1.1] at many buses
    b["stats"] = tuple(st)
    b["fn"], b["fe"] = fn, fe
    return b


def _hip_step(pkg, oracle, model, b, group):
    """One forward + loss + backward of the product on batch `b` (host tensors).  Feature widths below (8, 6) drop trailing
    feature columns for the model only; the loss always sees the reference's 8 / 6 columns."""
    x, ei, ea = b["x"].to(DEV), b["edge_index"].to(DEV), b["edge_attr"].to(DEV)
    st = tuple(s.to(DEV) for s in b["stats"])
    for p in model.parameters():
        p.grad = None
    xin = x[:, :b["fn"]] if b["fn"] == 8 else x[:, :b["fn"]].contiguous()
    ein = ea[:, :b["fe"]] if b["fe"] == 6 else ea[:, :b["fe"]].contiguous()
    out = model(xin, ei, ein)
    loss = pkg.data.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=out, x_mean=st[0], x_std=st[1], edge_mean=st[2],
                                 edge_std=st[3], edge_index=ei, reg_coefs=oracle.DEFAULT_REG_COEFS, num_samples=None,
                                 node_param=x[:, 8:], edge_param=ea[:, 6:], group=group)
    loss.backward()
    torch.cuda.synchronize()
    return loss.detach().double().cpu(), [p.grad.detach().clone() for p in model.parameters()]


def _cuts(n_graphs, n_shards):
    """Unequal whole-graph cuts: shard k gets a share proportional to k + 1."""
    tot = n_shards * (n_shards + 1) // 2
    cuts, acc = [0], 0
    for k in range(n_shards):
        acc += k + 1
        cuts.append(max(cuts[-1] + 1, round(n_graphs * acc / tot)))
    cuts[-1] = n_graphs
    return cuts


@pytest.mark.parametrize("n_shards", [2, 8])
@pytest.mark.parametrize("name", list(MODELS))
def test_sharded_hip_step_equals_single_batch_step(pkg, oracle, monkeypatch, name, n_shards):
    par = pkg.parallel
    ref, model = _build(pkg, oracle, name)
    fn, fe = MODELS[name][1][0], MODELS[name][1][1]
    full = _batch(pkg, 44, fn, fe)
    cuts = _cuts(44, n_shards)
    shards = [par.cut_batch(full, a, c) for a, c in zip(cuts, cuts[1:])]
    assert len({s["x"].shape[0] for s in shards}) > 1 and sum(s["num_graphs"] for s in shards) == 44

    mode = {"pass": 0, "seen": [], "total": None, "buckets": []}

    def fake_loss_allreduce(sums, group=None):
        assert group == "emulated-world"
        if mode["pass"] == 1:
            mode["seen"].append(sums.detach().clone())
        else:
            sums.copy_(mode["total"])
        return sums

    def fake_bucket_allreduce(flat, group=None, pending=None):
        mode["buckets"].append(flat.detach().clone())
        return flat

    monkeypatch.setattr(par, "allreduce_loss_sums", fake_loss_allreduce)
    monkeypatch.setattr(par, "allreduce_flat_grads", fake_bucket_allreduce)
    assert par.attach_grad_allreduce(model, "emulated-world") >= 1

    # ---- the whole batch as one step of the product (no group: the loss finishes on its own sums)
    loss_one, grads_one = _hip_step(pkg, oracle, model, full, None)
    buckets_one = mode["buckets"]
    mode["buckets"] = []
    assert buckets_one, "the model's backward never handed a gradient bucket to the data-parallel hook"
    assert sum(b.numel() for b in buckets_one) == sum(p.numel() for p in model.parameters())

    # ---- pass 1: every shard's local sums and counts
    mode["pass"] = 1
    for s in shards:
        _hip_step(pkg, oracle, model, s, "emulated-world")
    assert len(mode["seen"]) == n_shards
    total = torch.stack(mode["seen"]).sum(0)
    assert int(total[5].item()) == full["x"].shape[0] and int(total[6].item()) == full["edge_index"].shape[1]
    assert bool((total[2:5] > 0).all()), total                  # the three squared-mean penalties are really exercised
    local_means = torch.stack(mode["seen"])[:, 2:5] / torch.stack(mode["seen"])[:, [5, 6, 6]]
    glob_means = total[2:5] / total[[5, 6, 6]]
    assert ((local_means - glob_means).abs() > 1e-3 * glob_means.abs()).any()       # global != local: the case under test

    # ---- pass 2: every shard steps with the GLOBAL sums; buckets are summed (what the SUM all-reduce returns)
    mode.update({"pass": 2, "total": total, "buckets": []})
    losses, grads = [], None
    for s in shards:
        l, g = _hip_step(pkg, oracle, model, s, "emulated-world")
        losses.append(l)
        grads = g if grads is None else [a + c for a, c in zip(grads, g)]
    per_shard = len(mode["buckets"]) // n_shards
    assert per_shard == len(buckets_one)
    flat_sum = [sum(mode["buckets"][k * per_shard + j] for k in range(n_shards)) for j in range(per_shard)]

    # every rank reports the loss of the GLOBAL batch
    for l in losses:
        assert abs(l.item() - loss_one.item()) <= 1e-6 * abs(loss_one.item()), (l.item(), loss_one.item())
    names = [n for n, _ in model.named_parameters()]
    errs = {n: rel_err(a, c) for n, a, c in zip(names, grads, grads_one)}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
    assert all(e < TOL_SHARD for e in errs.values()), ("sharded vs single-batch HIP step", worst)
    for a, c in zip(flat_sum, buckets_one):
        assert rel_err(a, c) < TOL_SHARD

    # ---- the fp64 oracle on the whole batch (the reference's single-process step)
    b64 = {"x": full["x"].double(), "edge_index": full["edge_index"], "edge_attr": full["edge_attr"].double()}
    if (fn, fe) != (8, 6):
        _, loss64 = _oracle_step_general(oracle, ref, b64, tuple(s.double() for s in full["stats"]), fn, fe)
    else:
        _, loss64 = oracle.train_step(ref, b64, tuple(s.double() for s in full["stats"]))
    assert abs(losses[0].item() - loss64.item()) <= 1e-5 * abs(loss64.item())
    errs64 = {n: rel_err(a, q.grad) for n, a, q in zip(names, grads, ref.parameters())}
    worst64 = sorted(errs64.items(), key=lambda kv: -kv[1])[:6]
    # (the gates of the edge MLP are per edge and recomputed inside the kernels: they cannot be pinned from outside, and a
    #  razor-edge one that falls the other way than in fp64 moves its block's first-Linear gradients by ~1/N_edges --
    #  tests/test_gpu_stack.py gives those rows 2e-2 at this batch size; measured here: 3.4e-4 on one bias of the SkipPFN)
    def tol(n):
        return 2e-3 if ".edge_aggr.edge_aggr.0." in n or n.startswith("edge_aggr.edge_aggr.0.") else TOL_ORACLE
    assert all(e < tol(n) for n, e in errs64.items()), ("sharded HIP step vs fp64 oracle", worst64)


def _oracle_step_general(oracle, model, b, st, fn, fe):
    for p in model.parameters():
        p.grad = None
    x, ei, ea = b["x"], b["edge_index"], b["edge_attr"]
    out = model(x[:, :fn], ei, ea[:, :fe])
    loss = oracle.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=out, x_mean=st[0], x_std=st[1], edge_mean=st[2],
                               edge_std=st[3], edge_index=ei, reg_coefs=oracle.DEFAULT_REG_COEFS, num_samples=None,
                               node_param=x[:, 8:], edge_param=ea[:, 6:])
    loss.backward()
    return out, loss
