"""Data-parallel host logic: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" for the CPU tests).  The reference is single-process; what is added here is
exactly what sharding its batch needs:

* graphs are independent, so a batch is cut into contiguous per-rank shards of whole graphs
  (``shard_batch``) with no halo and no data-path collective in the message passing;
* the loss is ``mean_nodes + mean_edges + lam_reg * (mean)^2 x 3`` over the GLOBAL batch
  (/root/reference/data.py:450-455), so the five batch sums and the node / edge counts are
  all-reduced between the two loss phases (``gsp_wls_edge(..., group=pg)``; 7 doubles);
* every MPN block produces its parameter gradients in one flat fp32 bucket; the bucket is
  all-reduced (SUM: each rank's loss is already normalised by the global counts) once per block
  per step (``attach_grad_allreduce``) - a single latency-bound collective of 0.67 MB at C2/C4;
  ``async_op=True`` lets the collectives of a PFN stack overlap the backward of the blocks below.

These helpers are device agnostic (plain torch tensors + a process group), which is what lets the
world_size-2 gloo tests exercise them on CPU.
"""
from __future__ import annotations

import datetime
import os
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from . import graphs as _graphs


def init_from_env(backend: Optional[str] = None) -> Dict[str, int]:
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run) and initialise."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ      # torch.distributed.run sets both
    if (world > 1 or launched) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        # every collective wait is bounded: past the timeout the process-group watchdog aborts the communicator and
        # the process exits non-zero instead of hanging the node (DSS2_COLLECTIVE_TIMEOUT_S, default 120 s)
        timeout = datetime.timedelta(seconds=float(os.environ.get("DSS2_COLLECTIVE_TIMEOUT_S", "120")))
        if backend == "nccl":
            os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=timeout,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=timeout)
    return {"rank": rank, "world": world, "local": local}


def shard_bounds(num_graphs: int, rank: int, world: int):
    """Contiguous, near-equal graph ranges [g0, g1) per rank."""
    base, rem = divmod(num_graphs, world)
    g0 = rank * base + min(rank, rem)
    return g0, g0 + base + (1 if rank < rem else 0)


def cut_batch(batch: Dict[str, object], g0: int, g1: int) -> Dict[str, object]:
    """Graphs [g0, g1) of a collated batch (synthetic.make_batch layout, with 'graph_ptr') as a batch of their own; node
    indices are re-based to the cut.  Edges are assumed grouped by graph in graph order (what PyG collation and
    synthetic.make_batch produce)."""
    gp = batch["graph_ptr"]
    n0, n1 = int(gp[g0]), int(gp[g1])
    ei = batch["edge_index"]
    sel = (ei[0] >= n0) & (ei[0] < n1)
    idx = sel.nonzero().flatten()
    e0, e1 = (int(idx[0]), int(idx[-1]) + 1) if idx.numel() else (0, 0)
    if idx.numel() != e1 - e0:
        raise ValueError("edges of a rank's graphs are not contiguous in edge_index")
    out = dict(batch)
    out.update(x=batch["x"][n0:n1], y=batch["y"][n0:n1], edge_attr=batch["edge_attr"][e0:e1],
               edge_index=(ei[:, e0:e1] - n0), num_graphs=g1 - g0, graph_ptr=gp[g0:g1 + 1] - n0)
    return out


def shard_batch(batch: Dict[str, object], rank: int, world: int) -> Dict[str, object]:
    """This rank's contiguous share of whole graphs (``shard_bounds``) of a collated batch."""
    g0, g1 = shard_bounds(int(batch["num_graphs"]), rank, world)
    return cut_batch(batch, g0, g1)


def allreduce_loss_sums(sums: torch.Tensor, group=None) -> torch.Tensor:
    """sums[0..4] = batch sums, sums[5] = node count, sums[6] = edge count -> global values.  56 bytes: pure latency.  The
    collective is posted asynchronously (it runs on the backend's own stream as soon as the partials kernel has finished) and
    joined by a STREAM dependency, so the host goes on enqueueing; no kernel of the step is independent of these sums (the loss
    value and every gradient need the global penalty means), so there is nothing to overlap it with on the device."""
    if dist.is_initialized():
        def coll():
            work = dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group, async_op=True)
            work.wait()          # current stream waits for the collective's stream; the host does not block (NCCL / RCCL)
        _graphs.plan_collective(coll)      # (under a recording launch plan: a segment boundary, graphs.PlannedStep)
    return sums


def allreduce_flat_grads(flat: torch.Tensor, group=None, pending: Optional[List] = None) -> torch.Tensor:
    """SUM all-reduce of one block's flat gradient bucket.  With ``pending`` (a list) the collective is issued
    asynchronously (it runs on the backend's own stream while the main stream goes on with the backward of the block
    below) and its work handle is appended; ``wait_grad_allreduce`` joins all of them before the optimizer."""
    if dist.is_initialized():
        if pending is not None:
            _graphs.plan_collective(lambda: pending.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)))
        else:
            _graphs.plan_collective(lambda: dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group))
    return flat


def bucket_param_offsets(model: torch.nn.Module):
    """[(parameter, element offset, numel)] of every parameter inside the model's flat gradient bucket, for models whose backward
    produces ONE bucket: an MPN / SkipMPN block, or a PFN / SkipPFN stack (one autograd node).  Layout per block =
    MPN._flat_offsets(): [W1 | b1] [W2 | b2] then per conv [W_0 .. W_K | bias]; a stack's blocks follow each other."""
    blocks = list(model.mpns) if hasattr(model, "mpns") else [model]
    out, base = [], 0
    for m in blocks:
        offs = m._flat_offsets()
        hid, nmat = m.dim_hid, m.K + 1
        lin1, lin2 = m.edge_aggr.edge_aggr[0], m.edge_aggr.edge_aggr[2]
        out += [(lin1.weight, base + offs[0], lin1.weight.numel()), (lin1.bias, base + offs[0] + lin1.weight.numel(), hid),
                (lin2.weight, base + offs[1], hid * hid), (lin2.bias, base + offs[1] + hid * hid, hid)]
        for l, c in enumerate(m.convs):
            hout = c.out_channels
            for k, lin in enumerate(c.lins):
                out.append((lin.weight, base + offs[2 + l] + k * hout * hid, hout * hid))
            out.append((c.bias, base + offs[2 + l] + nmat * hout * hid, hout))
        base += offs[-1]
    return out


def overlap_param_groups(model: torch.nn.Module, n_chunks: int = 2):
    """Parameter groups for an optimizer that is stepped chunk by chunk while the later chunks of the gradient bucket are still
    being all-reduced (``attach_grad_allreduce(..., async_op=True, n_chunks=n)`` + ``step_overlapped``).  The bucket is cut
    at element n * total / n_chunks (rounded to 64 elements: collectives do not care about parameter boundaries); group i holds
    the parameters that END inside chunk i, so a parameter straddling a cut is stepped with the later chunk."""
    table = bucket_param_offsets(model)
    total = max(off + n for _, off, n in table)
    cuts = [min(total, (total * (i + 1) // n_chunks + 63) // 64 * 64) for i in range(n_chunks)]
    cuts[-1] = total
    groups = [[] for _ in range(n_chunks)]
    for prm, off, n in table:
        gi = next(i for i, c in enumerate(cuts) if off + n <= c)
        groups[gi].append(prm)
    model._dss2_bucket_cuts = cuts
    return [{"params": g} for g in groups if g]


def step_overlapped(model: torch.nn.Module, optimizer) -> None:
    """Optimizer step overlapped with the tail of the gradient all-reduce: chunk i's collective is joined, then the parameter
    group that lives in chunk i is stepped (one fused launch) while chunk i + 1 is still in flight on the collective stream.
    Needs ``attach_grad_allreduce(async_op=True, n_chunks=n)`` and an optimizer built on ``overlap_param_groups(model, n)``
    with a ``step_group(i)`` method (optim.FusedAdamax).  Without pending collectives it is a plain ``optimizer.step()``."""
    pending = getattr(model, "_dss2_pending_allreduce", None)
    if not pending or not hasattr(optimizer, "step_group") or len(pending) != len(optimizer.param_groups):
        wait_grad_allreduce(model)
        optimizer.step()
        return
    for gi, work in enumerate(pending):
        work.wait()
        optimizer.step_group(gi)
    pending.clear()


class LooseGradCoalescer:
    """ONE collective per backward for all the gradients that do not come in a flat bucket (blocks on the general route, the
    mask-embedding MLP of MaskEmbd*, the MultiMPN family: dozens of small parameters; one blocking all-reduce each was dozens of
    latency-bound collectives per step at N > 1 -- ADVICE r4, VERDICT r5 missing #5).

    Every such parameter gets a tensor hook.  The first hook that fires in a backward pass allocates one flat buffer and queues an
    end-of-backward callback on the autograd engine; each hook copies this backward's gradient into the parameter's slice and hands
    autograd a VIEW of that slice (AccumulateGrad adopts it as ``.grad`` while ``.grad`` is None); the callback all-reduces the whole
    buffer in place -- the ``.grad`` tensors are views of it -- and repairs any ``.grad`` autograd chose to copy instead of adopt.
    Parameters that take no part in a backward keep a zero slice (every rank sends the same layout).

    Sound only while every participating ``.grad`` is None (``zero_grad(set_to_none=True)``, torch's default): with a gradient already
    in place autograd ADDS the view right away and a later in-place reduction of the buffer would never reach it, so a backward that
    finds a ``.grad`` in place reduces parameter by parameter with blocking collectives on its own contribution (gradient accumulation
    stays exact), as before."""

    def __init__(self, params, group=None):
        self.params, self.group = list(params), group
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += p.numel()                            # (contiguous: the buffer holds exactly the parameters' elements)
        self.total = off
        self.flat = None            # the buffer of the backward in flight (None between backward passes)
        self.mode = None
        self.fired = []
        self.collectives = 0        # flat collectives issued so far (tests)
        self.fallback_collectives = 0

    def hooks(self):
        return [p.register_hook(self._make_hook(i)) for i, p in enumerate(self.params)]

    def _begin(self, like: torch.Tensor) -> None:
        self.fired = []
        if any(p.grad is not None for p in self.params):
            self.mode = "per-parameter"
        else:
            self.mode = "flat"
            self.flat = torch.zeros(self.total, dtype=like.dtype, device=like.device)
        torch.autograd.Variable._execution_engine.queue_callback(self._finish)

    def _make_hook(self, i: int):
        def hook(grad):
            if self.mode is None:
                self._begin(grad)
            p, off = self.params[i], self.offsets[i]
            if self.mode == "flat" and grad.dtype == self.flat.dtype and grad.device == self.flat.device:
                view = self.flat[off:off + p.numel()].view(grad.shape)
                view.copy_(grad)
                self.fired.append(i)
                return view
            red = grad.clone(memory_format=torch.contiguous_format)     # (a copy: the input may be shared with other consumers of the edge)
            allreduce_flat_grads(red, self.group, None)
            self.fallback_collectives += 1
            return red
        return hook

    def _finish(self) -> None:
        mode, flat, fired = self.mode, self.flat, self.fired
        self.mode, self.flat, self.fired = None, None, []
        if mode != "flat" or not fired:
            return
        allreduce_flat_grads(flat, self.group, None)
        self.collectives += 1
        for i in fired:
            p, off = self.params[i], self.offsets[i]
            g = p.grad
            if g is None:
                continue
            view = flat[off:off + p.numel()].view(g.shape)
            if g.data_ptr() != view.data_ptr():       # autograd copied instead of adopting the view: hand it the reduced values
                g.copy_(view)


def attach_grad_allreduce(model: torch.nn.Module, group=None, async_op: bool = False, n_chunks: int = 1) -> int:
    """Install the flat-bucket all-reduce on every MPN block of `model` (MPN / SkipMPN themselves,
    or the blocks inside PFN / SkipPFN).  Returns the number of blocks hooked.  A PFN / SkipPFN stack runs as one
    autograd node with ONE gradient bucket (networks._PFNFn): it issues a single collective per step, through the hook
    of its first block.

    ``async_op=True``: each block's collective is launched as soon as that block's backward has produced its bucket
    and overlaps the backward of the blocks below it (a PFN / SkipPFN stack then has its L collectives in flight
    instead of L serialised ones); call ``wait_grad_allreduce(model)`` after ``loss.backward()`` and before reading
    the gradients (the wait is a stream dependency, not a host block).  The asynchronous mode requires the ``.grad`` of
    every parameter to be None when backward runs (see ``hook`` below); a step that finds gradients in place reduces its
    buckets with blocking collectives instead.  ``n_chunks > 1`` (with ``async_op`` and ``overlap_param_groups``): the bucket
    travels as n_chunks collectives so that ``step_overlapped`` can step the first chunks' parameters while the last ones are
    still in flight ("optimizer overlapped with the all-reduce").

    Parameters whose gradients do not come in a bucket -- blocks on the general route (input widths other than (8, 6),
    dim_hid > 256: per-layer autograd nodes), the mask-embedding MLP of MaskEmbd*, the MultiMPN family -- get a tensor hook
    each and travel together: ONE flat buffer and ONE collective per backward (``LooseGradCoalescer``; a backward that finds a
    ``.grad`` in place reduces them one by one on its own contribution instead)."""
    n = 0
    pending = [] if async_op else None
    model._dss2_pending_allreduce = pending
    for h in getattr(model, "_dss2_param_hook_handles", ()):       # a second attach replaces the first one's hooks
        h.remove()
    model._dss2_param_hook_handles = []
    params = [p for p in model.parameters()]

    def make_hook(own):
        own_numel = sum(p.numel() for p in own)

        def hook(flat, g=group, q=pending):
            # The asynchronous mode hands autograd VIEWS of a bucket whose collective is still in flight.  That is only
            # sound while AccumulateGrad adopts those views as the .grad tensors, i.e. while every .grad the bucket feeds is
            # None (``zero_grad(set_to_none=True)``, torch's default).  With a .grad already in place (gradient accumulation,
            # ``set_to_none=False``) autograd would run ``p.grad += view`` on the compute stream beside the collective and
            # the reduced values would never reach p.grad: such a step falls back to the blocking collective.  Only the
            # parameters THIS bucket feeds are looked at: the block itself, or (a PFN / SkipPFN stack as one autograd node:
            # one bucket through its first block's hook) the whole model -- blocks that run as separate nodes have had the
            # .grad of the blocks above them set by the time their own bucket is ready.
            if q is not None and any(p.grad is not None for p in (own if flat.numel() == own_numel else params)):
                q = None
            cuts = getattr(model, "_dss2_bucket_cuts", None)
            if q is not None and n_chunks > 1 and cuts is not None and len(cuts) == n_chunks and cuts[-1] == flat.numel():
                a = 0
                for c in cuts:            # one collective per chunk, issued in order: chunk i completes before chunk i + 1
                    allreduce_flat_grads(flat[a:c], g, q)
                    a = c
                return flat
            return allreduce_flat_grads(flat, g, q)
        return hook

    bucketed = set()
    for m in model.modules():
        if hasattr(m, "convs") and hasattr(m, "edge_aggr") and hasattr(m, "_plan"):
            n += 1
            if m.edge_aggr.fused_dims() if hasattr(m.edge_aggr, "fused_dims") else True:
                own = list(m._params()) if hasattr(m, "_params") else list(m.parameters())
                m._grad_bucket_hook = make_hook(own)
                bucketed.update(id(p) for p in own)
            else:
                m._grad_bucket_hook = None          # the general route (other input widths, dim_hid > 256): per-layer nodes
    # everything the buckets do not carry -- the general route's blocks, MaskEmbd*'s embedding MLP, the MultiMPN family --
    # travels in one flat buffer and one collective per backward
    loose = [p for p in params if id(p) not in bucketed and p.requires_grad]
    model._dss2_loose_grads = LooseGradCoalescer(loose, group) if loose else None
    if loose:
        model._dss2_param_hook_handles = model._dss2_loose_grads.hooks()
    return n


def wait_grad_allreduce(model: torch.nn.Module) -> int:
    """Join the asynchronous gradient collectives issued during the last backward (no-op for the blocking mode).
    Returns how many were joined."""
    pending = getattr(model, "_dss2_pending_allreduce", None)
    if not pending:
        return 0

    def join():
        n_ = len(pending)
        for w in pending:
            w.wait()          # current stream waits for the collective; bounded by the process group's timeout
        pending.clear()
        return n_
    return _graphs.plan_collective(join)      # (a recording launch plan joins again at this point of every replay)


def broadcast_parameters(model: torch.nn.Module, src: int = 0, group=None) -> None:
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        for p in model.parameters():
            dist.broadcast(p.data, src=src, group=group)
