#!/usr/bin/env python3
"""GPU: launch times of the X-plane route at C2 (or --batch / --grid), kernel by kernel, against the route that splits X in the
weight-gradient kernel: forward chain with / without images, edge MLP forward with / without, dss2_wgrad_batched vs
dss2_wgrad_batched_xp.  HIP events around 50 back-to-back launches each (cold-cache effects of the step are not in here; the
step-level A/B is bench.py with DSS2_WGRAD_XP=0 / 1).  DSS2_LIB / DSS2_WGRAD_XP_MODE select a build / the range pairing."""
import argparse, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--grid", default="cigre14")
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--only-wgrad", action="store_true")
ap.add_argument("--no-rs2", action="store_true", help="plain layers only (no folded layer with its scaled bias sums)")
args = ap.parse_args()
DEV = "cuda:0"
nw, ops = pkg.networks, pkg.ops
H, nmat, nl = 128, 3, 3
b = pkg.synthetic.make_batch([args.grid], args.batch, seed=0)
ei = b["edge_index"].to(DEV); N = b["x"].shape[0]
topo = pkg.topology.get_topology(ei, N)
torch.manual_seed(0)
Ws = [[torch.randn(H, H, device=DEV) * (1.2 / H ** 0.5) for _ in range(nmat)] for _ in range(nl)]
plan = nw._PackPlan(Ws, DEV, bf16_groups=tuple(range(nl))); plan.refresh()
h = torch.randn(N, H, device=DEV)
Ys = [torch.empty(N, H, device=DEV) for _ in range(nl)]
xps = [ops.new_xplanes(topo, H, DEV) for _ in range(nl)]
gw = ops.chain_gate_words(topo, nmat, H)
bits = [torch.empty(topo.ntiles * gw, dtype=torch.int64, device=DEV) for _ in range(nl)]


def timed(fn, reps=args.reps):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def chain(planes, fp32=True):
    layers = [dict(Bp=plan.fwd16[i], Y=(Ys[i] if (fp32 or i == nl - 1 or not planes) else None), relu=True, y_bits=bits[i],
                   x_planes=(xps[i] if (planes and i < nl - 1) else None)) for i in range(nl)]
    ops.gemm_prop_chain(topo, h, H, nmat, layers, b_format=1)


print(f"B={args.batch} {args.grid}: N={N}, {topo.ntiles} tiles of {32 * topo.nrb} rows, ell {topo.ell}/{topo.ellT}; lib {os.environ.get('DSS2_LIB', '(shipped)')}, "
      f"XP_MODE {os.environ.get('DSS2_WGRAD_XP_MODE', '(default)')}")
if not args.only_wgrad:
    print(f"forward chain, 3 layers: no images {timed(lambda: chain(False)):.1f} us | images of layers 0,1 + fp32 {timed(lambda: chain(True)):.1f} us | "
          f"images, no fp32 copies of layers 0,1 {timed(lambda: chain(True, False)):.1f} us")
    x, ea = b["x"][:, :8].contiguous().to(DEV), b["edge_attr"][:, :6].contiguous().to(DEV)
    W1, b1 = torch.randn(H, 22, device=DEV) * 0.3, torch.randn(H, device=DEV) * 0.1
    f0 = lambda: nw._edge_aggr_forward(topo, x, 8, ea, 6, W1, b1, None, None, H, H, 8, 6, second_linear=False)
    f1 = lambda: nw._edge_aggr_forward(topo, x, 8, ea, 6, W1, b1, None, None, H, H, 8, 6, second_linear=False, xp=xps[0])
    print(f"edge MLP forward: no image {timed(f0):.1f} us | with the image of S {timed(f1):.1f} us")
chain(True)
for i in range(nl - 1):      # the last layer's image too (a chain never writes its last layer's): shift by one through a second pass
    pass
ops.gemm_prop_chain(topo, h, H, nmat, [dict(Bp=plan.fwd16[i], Y=Ys[i], relu=True, x_planes=xps[i]) for i in range(nl)], b_format=1)
Gs = [torch.randn(N, H, device=DEV) for _ in range(nl)]
stride = nmat * H * H + H
out = torch.empty(nl * stride, device=DEV); first = torch.empty(stride + nmat * H, device=DEV)


KW = {} if args.no_rs2 else dict(first_rowscale2=topo.deg_pows, first_out=first)
NOUT = nl if args.no_rs2 else nl - 1


def w_old():
    pend = []
    ops.wgrad_batched(topo, Gs, H, Ys, H, nmat, out[:NOUT * stride], pending=pend, **KW)
    return pend


def w_new():
    pend = []
    ops.wgrad_batched_xp(topo, Gs, H, xps, H, nmat, out[:NOUT * stride], pending=pend, **KW)
    return pend


t_old, t_new = timed(w_old), timed(w_new)
p_old, p_new = w_old(), w_new()
r_old, r_new = timed(lambda: ops.reduce_pending(list(p_old))), timed(lambda: ops.reduce_pending(list(p_new)))
print(f"weight gradients of {nl} layers: dss2_wgrad_batched {t_old:.1f} us (+ slab reduction {r_old:.1f}) | dss2_wgrad_batched_xp {t_new:.1f} us (+ {r_new:.1f})")
