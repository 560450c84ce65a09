"""GPU: host-free training epochs (the reference's loop is /root/reference/dss2_run.py:131-147 with the loader of
:68-69).

* ``DeviceDataset.collate_into``: the gather straight into caller-owned buffers equals ``collate``; with a device-side cursor the same
  launch, repeated, walks the permutation, wraps at the epoch's end, and takes the smaller last batch;
* ``runner.EpochTrainer`` (an epoch = N replays of one recorded step that starts with the collation and ends with the optimizer and the
  loss accumulation) equals the eager epoch (``DataLoader`` + ``train_epoch`` + FusedAdamax) bit for bit in every parameter -- as a
  launch plan and as a hipGraph, with a smaller last batch, over two epochs; constructing the trainer leaves model and optimizer untouched;
* ``PrefetchLoader`` hands out the very batches of the loader it wraps (mixed topologies: a new structure per batch, built one batch
  ahead on a side stream) and a training epoch through it equals the epoch without it;
* the entry points that are not launches of a step refuse to run while a plan records (include/dss2_hip.h, "launch plans")."""
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


def _dataset(pkg, S, grid="cigre14", seed=0):
    full = pkg.synthetic.make_batch([grid], S, seed=seed, violate=0.2)
    ds = pkg.dataset.DeviceDataset.from_batch(full, device=DEV)
    return ds, tuple(s.to(DEV) for s in full["stats"])


def test_collate_into_static_buffers_and_the_device_cursor(pkg):
    ds, _ = _dataset(pkg, 50)
    B = 16
    x = torch.empty(B * ds.n, 11, device=DEV)
    ea = torch.empty(B * ds.e, 13, device=DEV)
    y = torch.empty(B * ds.n, 2, device=DEV)
    descs = ds.collate_descs(x, ea, y)
    ids = torch.randperm(50, device=DEV)
    ds.collate_into(descs, ids[7:], B)
    ref = ds.collate(ids[7:7 + B].contiguous())
    assert torch.equal(x, ref.x) and torch.equal(ea, ref.edge_attr) and torch.equal(y, ref.y)
    # the cursor: {position, epoch length}; four launches of the SAME call walk 50 samples in batches of 16 and wrap
    cursor = torch.tensor([0, 50], dtype=torch.int64, device=DEV)
    seen = []
    for k in range(4):
        ds.collate_into(descs, ids, B, cursor=cursor, advance=True)
        want = ids[(torch.arange(B, device=DEV) + 16 * k) % 50]
        ref = ds.collate(want.contiguous())
        assert torch.equal(x, ref.x) and torch.equal(ea, ref.edge_attr), k
        seen.append(int(cursor[0]))
    assert seen == [16, 32, 48, 14]
    # advance=False leaves the position where it is
    ds.collate_into(descs, ids, B, cursor=cursor, advance=False)
    assert int(cursor[0]) == 14


def _eager_epochs(pkg, model, opt, ds, stats, B, epochs):
    losses = []
    for _ in range(epochs):
        loader = pkg.dataset.DataLoader(ds, batch_size=B, shuffle=False)
        losses.append(pkg.runner.train_epoch(model, opt, loader, stats, REG))
    return losses


@pytest.mark.parametrize("mode", ["plan", "graph"])
@pytest.mark.parametrize("cls,cargs,S,B", [
    ("MPN", (8, 6, 2, 128, 4, 2, 0.0), 200, 64),          # the C2 model; 3 full batches + one of 8
    ("SkipPFN", (8, 6, 2, 32, 3, 2, 0.0, 2), 96, 32),     # a stack on the whole-stack kernels; no remainder
])
def test_an_epoch_of_replays_equals_the_eager_epoch(pkg, mode, cls, cargs, S, B):
    ds, stats = _dataset(pkg, S, seed=4)
    torch.manual_seed(1)
    m1 = getattr(pkg, cls)(*cargs).to(DEV)
    m2 = getattr(pkg, cls)(*cargs).to(DEV)
    m2.load_state_dict(m1.state_dict())
    o1 = pkg.optim.FusedAdamax(m1.parameters(), lr=3e-3, capturable=True)
    o2 = pkg.optim.FusedAdamax(m2.parameters(), lr=3e-3, capturable=True)
    before = [p.detach().clone() for p in m2.parameters()]
    tr = pkg.runner.EpochTrainer(m2, o2, stats, REG, ds, B, shuffle=False, mode=mode)
    torch.cuda.synchronize()
    # recording the steps trained on real batches; the trainer restored what it consumed
    assert all(torch.equal(a, b) for a, b in zip(before, m2.parameters()))
    assert float(o2.param_groups[0]["_step"]) == 0.0 and tr.cursor.tolist() == [0, S]
    want = _eager_epochs(pkg, m1, o1, ds, stats, B, 2)
    got = []
    for _ in range(2):
        tr.train_epoch()
        got.append(tr.mean_loss())
    torch.cuda.synchronize()
    for a, b in zip(m1.parameters(), m2.parameters()):
        assert torch.equal(a, b), (a - b).abs().max().item()
    assert float(o2.param_groups[0]["_step"]) == 2 * -(-S // B)
    assert np.allclose(got, want, rtol=1e-6, atol=0), (got, want)       # (the eager mean is accumulated in fp32, the trainer's in fp64)


def test_a_shuffled_epoch_visits_every_sample_once(pkg):
    ds, stats = _dataset(pkg, 70, seed=2)
    m = pkg.MPN(8, 6, 2, 32, 2, 2, 0.0).to(DEV)
    o = pkg.optim.FusedAdamax(m.parameters(), lr=1e-3, capturable=True)
    g = torch.Generator(device=DEV)
    g.manual_seed(5)
    tr = pkg.runner.EpochTrainer(m, o, stats, REG, ds, 32, shuffle=True, generator=g)
    tr.train_epoch()
    first = tr.ids.clone()
    tr.train_epoch()
    torch.cuda.synchronize()
    assert sorted(first.tolist()) == list(range(70)) and sorted(tr.ids.tolist()) == list(range(70)) and not torch.equal(first, tr.ids)
    assert tr.acc.tolist()[1] == 3.0 and int(tr.cursor[0]) == 0          # 2 full + 1 smaller step; the position wrapped to the start


def test_prefetch_loader_hands_out_the_same_batches_one_ahead(pkg):
    S = 96
    full = pkg.synthetic.make_batch(["cigre14", "cigre14_reswitched"], 64, seed=1)
    parts = [pkg.dataset.DeviceDataset.from_batch(pkg.synthetic.make_batch([g], S, seed=2 + k, stats=full["stats"]), device=DEV)
             for k, g in enumerate(["cigre14", "cigre14_reswitched"])]
    stats = tuple(s.to(DEV) for s in full["stats"])

    def epoch(prefetch):
        torch.manual_seed(3)
        m = pkg.MPN(8, 6, 2, 64, 3, 2, 0.0).to(DEV)
        o = pkg.optim.FusedAdamax(m.parameters(), lr=3e-3)
        gen = torch.Generator()
        gen.manual_seed(9)
        loader = pkg.dataset.DataLoader(pkg.dataset.MixedDataset(parts), batch_size=48, shuffle=True, generator=gen)
        if prefetch:
            loader = pkg.dataset.PrefetchLoader(loader)
        seen = []

        class Spy:
            def __len__(self):
                return len(loader)

            def __iter__(self):
                for b in loader:
                    seen.append((b.x.clone(), b.edge_index.clone(), b.edge_attr.clone()))
                    yield b
        loss = pkg.runner.train_epoch(m, o, Spy(), stats, REG)
        torch.cuda.synchronize()
        return loss, seen, [p.detach().clone() for p in m.parameters()]
    l0, s0, p0 = epoch(False)
    l1, s1, p1 = epoch(True)
    assert len(s0) == len(s1) == 4
    for a, b in zip(s0, s1):
        assert all(torch.equal(u, v) for u, v in zip(a, b))
    assert l0 == l1 and all(torch.equal(a, b) for a, b in zip(p0, p1))


def test_structure_builds_refuse_to_run_while_a_plan_records(pkg):
    L = pkg._lib.lib()
    b = pkg.synthetic.make_batch(["cigre14"], 4, seed=0)
    ei = b["edge_index"].to(DEV)
    h = C.c_void_p()
    assert L.dss2_plan_begin(C.byref(h)) == 0
    try:
        with pytest.raises(RuntimeError, match="not available while a launch plan records"):
            pkg.topology.Topology(ei, b["x"].shape[0], double=True)
    finally:
        assert L.dss2_plan_end(h) == 0
    assert L.dss2_plan_size(h) == 0
    L.dss2_plan_destroy(h)
    pkg.topology.Topology(ei, b["x"].shape[0], double=True)      # ... and are available again afterwards


def test_planned_step_rejects_a_detached_loss_and_verifies_a_replay(pkg):
    b = pkg.synthetic.make_batch(["cigre14"], 32, seed=0)
    x, ei, ea = b["x"].to(DEV), b["edge_index"].to(DEV), b["edge_attr"].to(DEV)
    st = tuple(s.to(DEV) for s in b["stats"])
    m = pkg.MPN(8, 6, 2, 64, 3, 2, 0.0).to(DEV)
    params = list(m.parameters())
    extra = torch.zeros(4, device=DEV)

    def step(detach=False, torch_kernel=False):
        for p in params:
            p.grad = None
        out = m(x[:, :8], ei, ea[:, :6])
        loss = pkg.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=out, x_mean=st[0], x_std=st[1], edge_mean=st[2], edge_std=st[3],
                                edge_index=ei, reg_coefs=REG, num_samples=None, node_param=x[:, 8:], edge_param=ea[:, 6:])
        loss.backward(pkg.data.unit_grad(loss))
        if torch_kernel:
            extra.add_(1.0)            # a launch the plan does not carry
        return loss.detach() if detach else loss
    with pytest.raises(ValueError, match="ATTACHED loss"):
        pkg.graphs.PlannedStep(lambda: step(detach=True))
    pkg.graphs.PlannedStep(step, verify=lambda: [p.grad for p in params])           # a pure step verifies
    with pytest.raises(RuntimeError, match="does not reproduce the recorded step"):
        pkg.graphs.PlannedStep(lambda: step(torch_kernel=True), verify=lambda: [extra] + [p.grad for p in params])
