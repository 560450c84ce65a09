"""CPU, world_size 2, gloo: the data-parallel host logic of parallel.py.

The HIP kernels cannot run here, so each rank's per-shard arithmetic is supplied by the oracle
(tests may use it as a stand-in for the kernels); what is under test is the recipe the GPU path
uses: contiguous whole-graph shards, all-reduce of the five loss sums + counts between the two loss
phases, SUM all-reduce of the flat gradient bucket - which must reproduce the single-process loss
and gradients of the global batch exactly, including the squared-mean penalties."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_pkg


def _worker(rank, world, port, ret, n_graphs=10):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dss2_oracle as oracle
    pkg = load_pkg()
    torch.set_num_threads(1)
    env = pkg.parallel.init_from_env("gloo")
    assert env["world"] == world
    torch.manual_seed(0)
    # ragged: 10 graphs of mixed sizes on 2 ranks, half of them outside the penalty bands
    full = pkg.synthetic.make_batch(["cigre14", "cigre14_reswitched"], n_graphs, seed=3, violate=0.5)
    model = oracle.MPN(8, 6, 2, 16, 2, 2, 0.0).double()
    with torch.no_grad():   # push the outputs outside the penalty bands so J_v, J_theta, J_loading are all active
        for lin in model.convs[-1].lins:
            lin.weight *= 40.0
    pkg.parallel.broadcast_parameters(model, 0)
    st = [s.double().clone() for s in full["stats"]]
    st[1][0] *= 20.0        # sigma_V: v = out * sigma + mu leaves [0.9, 1.1]
    st = tuple(st)
    reg = oracle.DEFAULT_REG_COEFS

    def local_sums(b):
        x, ei, ea = b["x"].double(), b["edge_index"], b["edge_attr"].double()
        out = model(x[:, :8], ei, ea[:, :6])
        out = torch.cat([out[:, :1], out[:, 1:] * (1.0 - x[:, 9:10])], 1)
        sums = oracle.wls_partial_sums(x[:, :8], ea[:, :6], out, st[0], st[1], st[2], st[3], ei, x[:, 8:], ea[:, 6:], reg)
        return sums, x.shape[0], ei.shape[1]

    # ---- sharded: partial sums -> all-reduce (values only) -> loss -> backward -> SUM all-reduce of the bucket
    shard = pkg.parallel.shard_batch(full, rank, world)
    sums, n, e = local_sums(shard)
    glob = torch.cat([sums.detach(), torch.tensor([float(n), float(e)], dtype=torch.float64)])
    pkg.parallel.allreduce_loss_sums(glob)
    sums_g = sums + (glob[:5] - sums.detach())           # global value, local gradient path
    loss = oracle.loss_from_sums(sums_g, glob[5], glob[6], reg["lam_reg"])
    loss.backward()
    flat = torch.cat([p.grad.flatten() for p in model.parameters()])
    pkg.parallel.allreduce_flat_grads(flat)
    # ---- single process reference on the whole batch
    for p in model.parameters():
        p.grad = None
    b64 = {"x": full["x"].double(), "edge_index": full["edge_index"], "edge_attr": full["edge_attr"].double()}
    _, loss_ref = oracle.train_step(model, b64, st, reg)
    flat_ref = torch.cat([p.grad.flatten() for p in model.parameters()])
    ok_loss = abs(loss.item() - loss_ref.item()) <= 1e-10 * abs(loss_ref.item())
    ok_grad = (flat - flat_ref).abs().max().item() <= 1e-9 * flat_ref.abs().max().item()
    ok_shard = int(glob[5].item()) == full["x"].shape[0] and int(glob[6].item()) == full["edge_index"].shape[1]
    ok_pen = bool((glob[2:5] > 0).all())                 # the penalty terms are really exercised
    ret[rank] = (ok_loss, ok_grad, ok_shard, ok_pen, shard["num_graphs"])
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_loss_and_grads_equal_single_process():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29600 + os.getpid() % 300
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        ok_loss, ok_grad, ok_shard, ok_pen, ng = ret[r]
        assert ok_loss and ok_grad and ok_shard and ok_pen, (r, ret[r])
    assert sum(ret[r][4] for r in range(world)) == 10


def test_sharded_loss_and_grads_equal_single_process_at_world_8_with_unequal_shards():
    """VERDICT r5 next #7c: the shard recipe at the world size the 8-GPU node runs, 21 mixed-topology graphs cut 3,3,3,3,3,2,2,2 -- every
    rank's local means differ from the global ones, all three squared-mean penalties active."""
    world, n_graphs = 8, 21
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, 29300 + os.getpid() % 200, ret, n_graphs), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        ok_loss, ok_grad, ok_shard, ok_pen, ng = ret[r]
        assert ok_loss and ok_grad and ok_shard and ok_pen, (r, ret[r])
    assert [ret[r][4] for r in range(world)] == [3, 3, 3, 3, 3, 2, 2, 2]


def test_shard_bounds_cover_and_are_disjoint():
    pkg = load_pkg()
    for ng, world in [(10, 2), (4096, 8), (7, 8), (1, 1)]:
        spans = [pkg.parallel.shard_bounds(ng, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == ng
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def test_attach_grad_allreduce_hooks_every_block():
    pkg = load_pkg()
    m = pkg.SkipPFN(8, 6, 2, 16, 2, 2, 0.0, 3)
    assert pkg.parallel.attach_grad_allreduce(m) == 3
    assert pkg.parallel.attach_grad_allreduce(pkg.MPN(8, 6, 2, 16, 2, 2, 0.0)) == 1


class _BucketBlock(torch.nn.Module):
    """A block with the attributes attach_grad_allreduce looks for, whose backward follows networks._MPNFn's protocol on
    CPU tensors: gradients are produced in ONE flat bucket, the bucket hook is called, VIEWS of the bucket are returned."""

    def __init__(self):
        super().__init__()
        self.convs, self.edge_aggr, self._plan = torch.nn.ModuleList(), torch.nn.Identity(), None
        self.w = torch.nn.Parameter(torch.arange(6.0).view(2, 3))
        self.b = torch.nn.Parameter(torch.ones(4))

    def forward(self, scale):
        return _BucketFn.apply(scale, self, self.w, self.b)


class _BucketFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scale, mod, w, b):
        ctx.mod, ctx.scale = mod, float(scale)
        return (w.sum() + b.sum()) * scale

    @staticmethod
    def backward(ctx, g):
        flat = torch.full((10,), ctx.scale, dtype=torch.float32) * g
        hook = getattr(ctx.mod, "_grad_bucket_hook", None)
        if hook is not None:
            hook(flat)
        return None, None, flat[:6].view(2, 3), flat[6:]


def _async_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    pkg = load_pkg()
    torch.set_num_threads(1)
    pkg.parallel.init_from_env("gloo")
    m = _BucketBlock()
    assert pkg.parallel.attach_grad_allreduce(m, async_op=True) == 1
    tot = float(sum(r + 1 for r in range(world)))            # rank r contributes (r + 1) per element
    # step 1: .grad is None -> asynchronous collective, joined by wait_grad_allreduce
    m(torch.tensor(rank + 1.0)).backward()
    joined1 = pkg.parallel.wait_grad_allreduce(m)
    ok1 = bool((m.w.grad == tot).all() and (m.b.grad == tot).all())
    # step 2 WITHOUT clearing the gradients (accumulation): autograd adds the views into the existing .grad right away, so
    # the hook must have reduced the bucket before it returns (blocking fallback) -- nothing is left to join
    m(torch.tensor(rank + 1.0)).backward()
    joined2 = pkg.parallel.wait_grad_allreduce(m)
    ok2 = bool((m.w.grad == 2 * tot).all() and (m.b.grad == 2 * tot).all())
    ret[rank] = (ok1, joined1, ok2, joined2)
    dist.barrier()
    dist.destroy_process_group()


def test_async_bucket_hook_with_gradients_already_in_place():
    """async_op=True is only sound while .grad is None; with gradients in place the hook must fall back
    to a blocking collective, otherwise `p.grad += view` races the all-reduce and the reduced values never arrive."""
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_async_worker, args=(world, 29950 + os.getpid() % 40, ret), nprocs=world, join=True)
    for r in range(world):
        ok1, j1, ok2, j2 = ret[r]
        assert ok1 and j1 == 1, ret[r]
        assert ok2 and j2 == 0, ret[r]


class _ChunkOpt:
    """Optimizer stub with FusedAdamax's step_group interface: plain SGD, records the order of group steps."""

    def __init__(self, param_groups, lr):
        self.param_groups, self.lr, self.order = param_groups, lr, []

    def step_group(self, gi):
        self.order.append(gi)
        with torch.no_grad():
            for p in self.param_groups[gi]["params"]:
                p -= self.lr * p.grad

    def step(self):
        for gi in range(len(self.param_groups)):
            self.step_group(gi)


def _overlap_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    pkg = load_pkg()
    torch.set_num_threads(1)
    pkg.parallel.init_from_env("gloo")
    torch.manual_seed(0)
    model = pkg.SkipPFN(8, 6, 2, 16, 3, 2, 0.0, 2)            # CPU: only its structure and parameters are used
    table = pkg.parallel.bucket_param_offsets(model)
    total = max(o + n for _, o, n in table)
    assert total == sum(p.numel() for p in model.parameters())
    assert sorted(id(p) for p, _, _ in table) == sorted(id(p) for p in model.parameters())
    groups = pkg.parallel.overlap_param_groups(model, 2)
    assert len(groups) == 2 and sum(len(g["params"]) for g in groups) == len(list(model.parameters()))
    assert pkg.parallel.attach_grad_allreduce(model, async_op=True, n_chunks=2) == 2
    opt = _ChunkOpt(groups, 0.5)
    before = [p.detach().clone() for p, _, _ in table]
    # what the backward of the stack does: one flat bucket, the hook of the first block, views handed to autograd
    flat = torch.arange(total, dtype=torch.float32) * (rank + 1)
    model.mpns[0]._grad_bucket_hook(flat)
    n_pending = len(model._dss2_pending_allreduce)
    for prm, off, n in table:
        prm.grad = flat[off:off + n].view_as(prm)
    pkg.parallel.step_overlapped(model, opt)
    expect = torch.arange(total, dtype=torch.float32) * sum(r + 1 for r in range(world))
    ok = all(torch.equal(prm.detach(), b0 - 0.5 * expect[off:off + n].view_as(prm)) for (prm, off, n), b0 in zip(table, before))
    ret[rank] = (ok, n_pending, list(opt.order), len(model._dss2_pending_allreduce))
    dist.barrier()
    dist.destroy_process_group()


def test_chunked_bucket_and_overlapped_optimizer_step():
    """SURVEY 8f rank 2 / VERDICT r2 next #3: the gradient bucket travels as two collectives and the optimizer steps the first
    half's parameters while the second half is still in flight; the result equals the plain all-reduce + step."""
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_overlap_worker, args=(world, 29880 + os.getpid() % 60, ret), nprocs=world, join=True)
    for r in range(world):
        ok, n_pending, order, left = ret[r]
        assert ok and n_pending == 2 and order == [0, 1] and left == 0, ret[r]


class _TwoBlocksAndAHead(torch.nn.Module):
    """Two bucket blocks run as SEPARATE autograd nodes plus a plain Linear whose gradients come in no bucket (what the
    general route's per-layer nodes, MaskEmbd*'s embedding MLP and the MultiMPN family look like to parallel.py)."""

    def __init__(self):
        super().__init__()
        self.a, self.b = _BucketBlock(), _BucketBlock()
        self.head = torch.nn.Linear(1, 3)

    def forward(self, scale):
        return self.head((self.a(scale) + 2.0 * self.b(scale)).view(1, 1)).sum()


def _param_hook_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    pkg = load_pkg()
    torch.set_num_threads(1)
    pkg.parallel.init_from_env("gloo")
    torch.manual_seed(0)
    m = _TwoBlocksAndAHead()
    single = _TwoBlocksAndAHead()
    single.load_state_dict(m.state_dict())
    assert pkg.parallel.attach_grad_allreduce(m, async_op=True) == 2
    assert len(m._dss2_param_hook_handles) == 2               # head.weight, head.bias: no bucket -> the coalescer's tensor hooks
    m(torch.tensor(rank + 1.0)).backward()
    co = m._dss2_loose_grads
    one_flat = (co.collectives, co.fallback_collectives) == (1, 0)       # both loose gradients in ONE collective
    adopted = all(p.grad.untyped_storage().data_ptr() == m.head.weight.grad.untyped_storage().data_ptr() for p in m.head.parameters())
    # both blocks' buckets travelled asynchronously: block a's hook ran after block b's .grad was set (separate nodes) and
    # must look at its OWN parameters only
    joined = pkg.parallel.wait_grad_allreduce(m)
    # single-process reference: the sum over ranks of the per-rank losses
    sum(single(torch.tensor(r + 1.0)) for r in range(world)).backward()
    ok = all(torch.allclose(p.grad, q.grad, rtol=1e-6, atol=1e-6) for p, q in zip(m.parameters(), single.parameters()))
    # a second attach (blocking) replaces the tensor hooks instead of stacking a second collective on them
    pkg.parallel.attach_grad_allreduce(m)
    n_handles = len(m._dss2_param_hook_handles)
    for p in m.parameters():
        p.grad = None
    m(torch.tensor(rank + 1.0)).backward()
    ok2 = all(torch.allclose(p.grad, q.grad, rtol=1e-6, atol=1e-6) for p, q in zip(m.parameters(), single.parameters()))
    # a third backward WITHOUT clearing the gradients (accumulation): the loose gradients fall back to one blocking collective each
    # on this backward's contribution, and the sums are exact
    co = m._dss2_loose_grads
    m(torch.tensor(rank + 1.0)).backward()
    ok3 = all(torch.allclose(p.grad, 2.0 * q.grad, rtol=1e-6, atol=1e-6) for p, q in zip(m.parameters(), single.parameters()))
    ok3 = ok3 and (co.collectives, co.fallback_collectives) == (1, 2)
    ret[rank] = (ok and one_flat and adopted, joined, ok2 and ok3, n_handles)
    dist.barrier()
    dist.destroy_process_group()


def test_parameters_outside_the_buckets_are_reduced_and_async_checks_only_the_own_block():
    """ADVICE r3 (medium + low): gradients that do not come in a flat bucket must still be all-reduced (tensor hooks), and a
    block's asynchronous collective must not be downgraded because the blocks above it already have their .grad."""
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_param_hook_worker, args=(world, 29800 + os.getpid() % 60, ret), nprocs=world, join=True)
    for r in range(world):
        ok, joined, ok2, n_handles = ret[r]
        assert ok and joined == 2 and ok2 and n_handles == 2, ret[r]
