"""Whole-stack path of MPN / SkipMPN / PFN / SkipPFN at the reference driver's own model line (dim_hid 32, K = 2:
/root/reference/dss2_run.py:72-88) on the kernels of csrc/dss2_stack.hip: ONE pack launch, ONE forward launch for all
blocks of the stack, ONE backward launch (data- and weight-gradients fused) and ONE reduction launch (with the chain rule
of the folded second Linear) per training step, instead of ~13 launches per block (networks._PFNFn).

``networks.PFN.forward`` / ``networks.MPN.forward`` come here when ``supported()`` says the shape is covered; everything
else keeps the per-block kernels.  ``DSS2_STACK_KERNEL=0`` switches this path off (tests run both).  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os as _os
from typing import List, Optional, Sequence

import numpy as np
import torch


from . import _lib


def _static_replay() -> bool:
    """True while the step is being captured into a hipGraph or recorded into a launch plan (graphs.PlannedStep): by-value seeds /
    cached batch constants must not be baked in -- the device-side state is used instead."""
    from .graphs import plan_recording
    return plan_recording() or torch.cuda.is_current_stream_capturing()


_F32 = torch.float32
STACK_KERNEL = _os.environ.get("DSS2_STACK_KERNEL", "1") == "1"
DROP_STRIDE = 64          # dropout mask id of block b, conv l: b * DROP_STRIDE + l + 1 (= networks._StackPlan.DROP_STRIDE)
_CU_COUNT = {}


def _cu_count(dev) -> int:
    n = _CU_COUNT.get(dev)
    if n is None:
        n = _CU_COUNT[dev] = int(torch.cuda.get_device_properties(dev).multi_processor_count)
    return n


def dims_of(blocks) -> Optional[_lib.StackDims]:
    """dss2_stack_dims of a list of MPN / SkipMPN blocks, or None when they do not form a uniform stack."""
    b0, bl = blocks[0], blocks[-1]
    for m in blocks:
        if (m.dim_hid, m.K, m.dim_featn, m.dim_feate, m.n_gnn_layers, float(m.dropout_rate)) != \
                (b0.dim_hid, b0.K, b0.dim_featn, b0.dim_feate, b0.n_gnn_layers, float(b0.dropout_rate)):
            return None
    for m in blocks[:-1]:
        if m.dim_out != b0.dim_out or m.skip != b0.skip:
            return None
    d = _lib.StackDims()
    d.n_blocks, d.n_hh = len(blocks), b0.n_gnn_layers - 1
    d.dout_inner = b0.dim_out if len(blocks) > 1 else bl.dim_out
    d.dout_last = bl.dim_out
    d.skip_inner = int(bool(b0.skip)) if len(blocks) > 1 else 0
    d.skip_last = int(bool(bl.skip))
    return d


STACK_NRB = _os.environ.get("DSS2_STACK_NRB", "auto")      # tile height of the whole-stack kernels: auto | 1 | 2


def tiles_of(topo):
    """The tile set the whole-stack kernels run on.  They are latency-bound per tile (one workgroup walks a tile through ~45
    dependent phases), so as long as 64-row tiles leave CUs idle, 32-row tiles -- twice as many workgroups, half the work
    per phase -- are faster (B = 64: 16 -> 32 workgroups); big batches keep the 64-row tiles (less overhead per graph)."""
    if topo.global_only:
        return None
    if STACK_NRB in ("1", "2"):
        return topo.tiles_for(int(STACK_NRB))
    base = topo
    if topo.nrb > 2:
        # small graphs the primary tiling happened to pack into taller tiles (ragged or tiny batches: 29 CIGRE graphs tile as
        # well at 96 rows as at 64): the stack kernels run them on a 64-row tiling of their own, so that the route -- and
        # with it every razor-edge ReLU gate -- does not depend on the batch size (tests/test_gpu_shard_emulation.py)
        base = topo.tiles_for(2)
        if base is None or base.ell_tiles is None or base.ellT_tiles is None:
            return None
    if base.nrb == 2 and base.ntiles <= _cu_count(topo.device):
        alt = topo.tiles_for(1)
        if alt is not None and alt.ell_tiles is not None and alt.ellT_tiles is not None:
            return alt
    return base


def supported(blocks, topo) -> Optional[_lib.StackDims]:
    """The stack's dims when the whole-stack kernels cover it on this topology, else None."""
    if not STACK_KERNEL or topo.global_only:
        return None
    d = dims_of(blocks)
    ts = tiles_of(topo) if d is not None else None
    if ts is None or ts.ell_tiles is None or ts.ellT_tiles is None or ts.ell_ent_tiles is None or ts.ellT_ent_tiles is None:
        return None
    b0 = blocks[0]
    if len(blocks) >= DROP_STRIDE or b0.n_gnn_layers + 1 >= DROP_STRIDE:
        return None
    ok = _lib.lib().dss2_stack_supported(C.byref(d), b0.dim_hid, b0.K + 1, b0.dim_featn, b0.dim_feate, ts.nrb, ts.ell, ts.ellT)
    return d if ok else None


class _Plan:
    """Per (owner module, device): the device table of parameter pointers, the packed-weight scratch, the dropout state."""

    def __init__(self, dims: _lib.StackDims, device):
        self.dims, self.device = dims, device
        L = _lib.lib()
        self.wpack = torch.empty(int(L.dss2_stack_wpack_words(C.byref(dims))), dtype=torch.int32, device=device)
        self.total = int(L.dss2_stack_flat_floats(C.byref(dims)))
        self.stride = (self.total + 3) // 4 * 4
        self.fold_floats = int(L.dss2_stack_fold_scratch_floats(C.byref(dims)))
        self.ptrs = None
        self.table = None
        self.version = 0
        self.table_builds = 0
        self.rng_state = None

    def key(self):
        return _dims_key(self.dims)

    def params_table(self, ps: Sequence[torch.Tensor]) -> torch.Tensor:
        ptrs = tuple(p.data_ptr() for p in ps)
        if ptrs != self.ptrs:
            for p in ps:
                if not p.is_contiguous():
                    raise RuntimeError("weight matrices must be contiguous")
            self.table = torch.from_numpy(np.array(ptrs, dtype=np.uint64).view(np.int64).copy()).to(self.device)
            self.ptrs = ptrs
            self.table_builds += 1
        return self.table

    def pack(self, ps, snap: Optional[torch.Tensor], host_seed: int, capturing: bool, tick: Optional[torch.Tensor] = None,
             tiles: Optional["_lib.StackArgs"] = None) -> int:
        tab = self.params_table(ps)
        st = self.rng_state
        _lib.check(_lib.lib().dss2_stack_pack(C.byref(self.dims), tab.data_ptr(), self.wpack.data_ptr(),
                                              (st.data_ptr() if st is not None else None),
                                              (snap.data_ptr() if snap is not None else None), host_seed, int(not capturing),
                                              (tick.data_ptr() if tick is not None else None),
                                              (C.byref(tiles) if tiles is not None else None), _lib.stream_ptr(self.device)),
                   "dss2_stack_pack")
        self.version += 1
        return self.version


def _dims_key(d) -> tuple:
    return (d.n_blocks, d.n_hh, d.dout_inner, d.dout_last, d.skip_inner, d.skip_last)


def _plan_of(owner, dims, device) -> _Plan:
    plan = owner.__dict__.get("_fused_plan")
    if plan is None or plan.device != device or plan.key() != _dims_key(dims):
        plan = owner.__dict__["_fused_plan"] = _Plan(dims, device)
    return plan


def _fill_common(a: "_lib.StackArgs", plan: _Plan, topo, x, ldx, ea, ldea, acts, xs, snap, p_drop, eacache) -> None:
    from .networks import _dropout_params
    d = plan.dims
    a.dims = d
    a.x, a.ldx, a.ea, a.ldea = x.data_ptr(), ldx, ea.data_ptr(), ldea
    a.wpack = plan.wpack.data_ptr()
    ts = tiles_of(topo)
    a.tile_start, a.ntiles, a.tm = ts.tile_start.data_ptr(), ts.ntiles, 32 * ts.nrb
    a.ell_w, a.ell_e, a.ell_width = ts.ell_tiles.data_ptr(), ts.ell_ent_tiles.data_ptr(), ts.ell
    a.ellT_w, a.ellT_e, a.ellT_width = ts.ellT_tiles.data_ptr(), ts.ellT_ent_tiles.data_ptr(), ts.ellT
    a.deg_pows = topo.deg_pows.data_ptr()
    a.eacache = eacache.data_ptr()
    a.acts = acts.data_ptr()
    a.xs = xs.data_ptr() if xs is not None else None
    a.n_nodes = topo.N
    a.drop_stride = DROP_STRIDE
    if snap is not None:
        a.drop_state = snap.data_ptr()
        a.drop_thr, a.drop_scale = _dropout_params(p_drop)


class _FusedStackFn(torch.autograd.Function):
    """forward(x, edge_attr) of a whole MPN / SkipMPN / PFN / SkipPFN as ONE autograd node on the whole-stack kernels."""

    @staticmethod
    def forward(ctx, x, ea, topo, owner, blocks, dims, *ps):
        from .networks import _rows
        x, ldx = _rows(x)
        ea, ldea = _rows(ea)
        dev = x.device
        plan = _plan_of(owner, dims, dev)
        N, n_hh, NB = topo.N, dims.n_hh, dims.n_blocks
        p_drop = float(blocks[0].dropout_rate)
        snap = None
        host_seed, capturing = 0, False
        if p_drop > 0.0:
            # one draw from torch's CPU generator per forward call, as nn.Dropout consumes torch's generator in the reference
            # (networks.dropout_snapshot); inside a hipGraph capture the device-side state is advanced by the pack kernel
            host_seed = int(torch.empty((), dtype=torch.int64).random_().item())
            if plan.rng_state is None:
                plan.rng_state = torch.tensor([host_seed ^ 0x5DEECE66D, 0], dtype=torch.int64).to(dev)
            snap = torch.empty(2, dtype=torch.int64, device=dev)
            capturing = _static_replay()
        acts = torch.empty(NB, n_hh + 1, N, 32, dtype=_F32, device=dev)
        xs = torch.empty(NB, N, 8, dtype=_F32, device=dev) if NB > 1 else None
        out = torch.empty(N, dims.dout_last, dtype=_F32, device=dev)
        ts = tiles_of(topo)
        eacache = torch.empty(ts.ntiles, ts.ell + ts.ellT, 64, 8, dtype=_F32, device=dev)
        a = _lib.StackArgs()
        _fill_common(a, plan, topo, x, ldx, ea, ldea, acts, xs, snap, p_drop, eacache)
        a.out, a.ldo = out.data_ptr(), out.stride(0)
        # ONE launch: fold + fragment packing of every block's weights, the dropout state hand-over, and this batch's per-tile
        # edge-feature cache
        ver = plan.pack(ps, snap, host_seed, capturing, None, a)
        _lib.check(_lib.lib().dss2_stack_forward(C.byref(a), _lib.stream_ptr(dev)), "dss2_stack_forward")
        for bi, m in enumerate(blocks):      # what the tests read to hand the very same masks to the oracle
            m._last_dropout, m._drop_base = (snap, p_drop), bi * DROP_STRIDE
        owner._last_dropout = (snap, p_drop)
        saved = [x, ea, acts, eacache] + ([xs] if xs is not None else []) + ([snap] if snap is not None else [])
        ctx.save_for_backward(*saved, *ps)
        ctx.meta = (topo, owner, blocks, dims, ldx, ldea, len(saved), xs is not None, snap is not None, p_drop, ver)
        return out

    @staticmethod
    def backward(ctx, gout):
        topo, owner, blocks, dims, ldx, ldea, n_saved, has_xs, has_snap, p_drop, ver = ctx.meta
        st = ctx.saved_tensors
        saved, ps = st[:n_saved], st[n_saved:]
        x, ea, acts, eacache = saved[0:4]
        xs = saved[4] if has_xs else None
        snap = saved[4 + int(has_xs)] if has_snap else None
        dev = gout.device
        plan = owner.__dict__["_fused_plan"]
        if plan.version != ver:           # another forward re-packed in between (weights are unchanged: autograd checks that)
            plan.pack(ps, None, 0, False)
        g = gout.contiguous()
        n_wg = max(1, min(tiles_of(topo).ntiles, _cu_count(dev)))
        slab = torch.empty(n_wg, plan.stride, dtype=_F32, device=dev)
        flat = torch.empty(plan.total, dtype=_F32, device=dev)
        need_dx = bool(ctx.needs_input_grad[0])
        dxbuf = torch.empty(topo.N, 8, dtype=_F32, device=dev) if dims.n_blocks > 1 else None
        dx = torch.empty(topo.N, 8, dtype=_F32, device=dev) if need_dx else None
        a = _lib.StackArgs()
        _fill_common(a, plan, topo, x, ldx, ea, ldea, acts, xs, snap, p_drop, eacache)
        a.gout, a.ldg = g.data_ptr(), g.stride(0)
        a.dxbuf = dxbuf.data_ptr() if dxbuf is not None else None
        a.dx_out = dx.data_ptr() if dx is not None else None
        a.slab, a.slab_stride, a.n_wg = slab.data_ptr(), plan.stride, n_wg
        L = _lib.lib()
        s_ = _lib.stream_ptr(dev)
        _lib.check(L.dss2_stack_backward(C.byref(a), s_), "dss2_stack_backward")
        fold_scratch = torch.empty(plan.fold_floats, dtype=_F32, device=dev)
        _lib.check(L.dss2_stack_reduce(C.byref(dims), slab.data_ptr(), n_wg, plan.stride, plan.params_table(ps).data_ptr(),
                                       flat.data_ptr(), fold_scratch.data_ptr(), s_), "dss2_stack_reduce")
        hook = getattr(blocks[0], "_grad_bucket_hook", None)
        if hook is not None:      # data-parallel: ONE all-reduce for the whole stack's bucket (parallel.py)
            hook(flat)
        lay = owner.__dict__.get("_fused_grad_layout")
        if lay is None:
            lay = owner.__dict__["_fused_grad_layout"] = _grad_layout(blocks)
        parts = flat.split(lay[0])
        grads = [parts[i] if lay[1][i] is None else parts[i].view(lay[1][i]) for i in lay[2]]
        return (dx, None, None, None, None, None, *grads)


def _grad_layout(blocks):
    """(piece sizes of the flat buffer, piece shapes, piece index per parameter in _params() order) for all blocks."""
    sizes, shapes, order = [], [], []
    for m in blocks:
        hid, fn, fe, nmat, L = m.dim_hid, m.dim_featn, m.dim_feate, m.K + 1, m.n_gnn_layers
        nc = 2 * fn + fe
        base = len(sizes)
        sizes += [hid * nc, hid, hid * hid, hid]
        shapes += [(hid, nc), None, (hid, hid), None]
        order += [base, base + 1, base + 2, base + 3]
        for l in range(L):
            hout = m.dim_out if l == L - 1 else hid
            b0 = len(sizes)
            sizes += [hout * hid] * nmat + [hout]                      # stored [W_0 .. W_K | bias]
            shapes += [(hout, hid)] * nmat + [None]
            order += [b0 + nmat] + list(range(b0, b0 + nmat))          # returned bias first (MPN._params())
    return sizes, shapes, order


def route(owner, blocks, topo) -> Optional[_lib.StackDims]:
    """``supported`` with the answer cached per (owner, topology object)."""
    hit = owner.__dict__.get("_fused_route")
    if hit is not None and hit[0] is topo and hit[2] == STACK_KERNEL:
        return hit[1]
    d = supported(blocks, topo)
    owner.__dict__["_fused_route"] = (topo, d, STACK_KERNEL)
    return d


def run(owner, blocks, dims, topo, x, edge_attr):
    ps: List[torch.Tensor] = [t for m in blocks for t in m._params()]
    return _FusedStackFn.apply(x, edge_attr, topo, owner, blocks, dims, *ps)
