#!/usr/bin/env python3
"""GPU micro-benchmark of the individual HIP kernels at BASELINE config C2 shapes (HIP events, in one
process, interleaved rounds).  Usage: python tools/kbench.py [--B 4096] [--H 128] [--grid cigre14]"""
import argparse, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
nw = pkg.networks

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=4096); ap.add_argument("--H", type=int, default=128)
ap.add_argument("--grid", default="cigre14"); ap.add_argument("--K", type=int, default=2)
ap.add_argument("--reps", type=int, default=30); ap.add_argument("--rounds", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0")
b = pkg.synthetic.make_batch([a.grid], a.B, seed=0)
x, ei, ea = b["x"].to(dev), b["edge_index"].to(dev), b["edge_attr"].to(dev)
N, H, nmat = x.shape[0], a.H, a.K + 1
topo = pkg.topology.get_topology(ei, N)
print(f"N={N} E2={topo.E2} nrb={topo.nrb} ntiles={topo.ntiles} util={topo.utilisation:.3f} max_nnz={topo.max_nnz}")
torch.manual_seed(0)
Ws = [torch.randn(H, H, device=dev) * 0.1 for _ in range(nmat)]
W2 = torch.randn(H, H, device=dev) * 0.1
Wl = [torch.randn(2, H, device=dev) * 0.1 for _ in range(nmat)]
W1 = torch.randn(H, 22, device=dev) * 0.1; b1 = torch.randn(H, device=dev) * 0.1
plan = nw._PackPlan([Ws, [W2], Wl], dev); plan.refresh()
h = torch.randn(N, H, device=dev); g = torch.randn(N, H, device=dev); g2 = torch.randn(N, 2, device=dev)
bias = torch.randn(H, device=dev); bias2 = torch.randn(2, device=dev); out = torch.empty(N, H, device=dev); out2 = torch.empty(N, 2, device=dev)
S = torch.empty(N, H, device=dev)
flat = torch.empty(nmat * H * H + H, device=dev); flat1 = torch.empty(H * H + H, device=dev); flatl = torch.empty(nmat * 2 * H + 2, device=dev)
xin, ein = x[:, :8], ea[:, :6]
L = pkg._lib.lib(); st = torch.cuda.current_stream().cuda_stream
slab1 = torch.empty(512 * (H * 23), device=dev); g1 = torch.empty(H * 23, device=dev)
slab2 = torch.empty(topo.ntiles * (H * 23), device=dev)

fl_gemm = 2.0 * N * H * nmat * H + 2.0 * a.K * topo.E2 * H
cases = {
  "tag_fwd  H->H (gemm_prop nmat=3)": (lambda: nw.gemm_prop(topo, h, H, H, plan.fwd[0], nmat, H, out, bias=bias, relu=True), fl_gemm),
  "tag_dgrad H->H (gemm_prop nmat=3, A^T, relu mask)": (lambda: nw.gemm_prop(topo, g, H, H, plan.bwd[0], nmat, H, out, relu_src=h, transposed=True), fl_gemm),
  "tag_wgrad H->H (wgrad+reduce)": (lambda: nw.wgrad(topo, g, H, h, H, nmat, flat), fl_gemm),
  "linear fwd H->H (gemm_prop nmat=1)": (lambda: nw.gemm_prop(topo, h, H, H, plan.fwd[1], 1, H, out, bias=bias, rowscale=topo.deg), 2.0 * N * H * H),
  "linear wgrad (nmat=1)": (lambda: nw.wgrad(topo, g, H, h, H, 1, flat1, rowscale=topo.deg), 2.0 * N * H * H),
  "tag_fwd  H->2": (lambda: nw.gemm_prop(topo, h, H, H, plan.fwd[2], nmat, 2, out2, bias=bias2, narrow_h=2), 2.0 * N * H * nmat * 2),
  "tag_dgrad 2->H": (lambda: nw.gemm_prop(topo, g2, 2, nmat * 2, plan.bwd[2], 1, H, out, relu_src=h, transposed=True, prop_in=nmat - 1), 2.0 * N * H * nmat * 2),
  "tag_wgrad H->2": (lambda: nw.wgrad(topo, g2, 2, h, H, nmat, flatl), 2.0 * N * H * nmat * 2),
  "edge_hidden_fwd": (lambda: pkg._lib.check(L.dss2_edge_hidden_fwd(xin.data_ptr(), 11, ein.data_ptr(), 13, W1.data_ptr(), b1.data_ptr(), topo.rowptr.data_ptr(), topo.col.data_ptr(), topo.ent.data_ptr(), S.data_ptr(), N, H, 8, 6, st), "f"), 2.0 * topo.E2 * 14 * H + 2.0 * N * 8 * H),
  "edge_hidden_bwd": (lambda: pkg._lib.check(L.dss2_edge_hidden_bwd(xin.data_ptr(), 11, ein.data_ptr(), 13, W1.data_ptr(), b1.data_ptr(), g.data_ptr(), topo.rowptr.data_ptr(), topo.col.data_ptr(), topo.ent.data_ptr(), slab1.data_ptr(), 512, None, H, N, H, 8, 6, 0, st), "b"), 4.0 * topo.E2 * 22 * H),
  "edge_tile_fwd": (lambda: pkg._lib.check(L.dss2_edge_tile_fwd(xin.data_ptr(), 11, ein.data_ptr(), 13, W1.data_ptr(), b1.data_ptr(), topo.tile_start.data_ptr(), topo.ell_ent_tiles.data_ptr(), topo.ell, topo.nrb, topo.ntiles, S.data_ptr(), H, 8, 6, st), "f"), 2.0 * topo.E2 * 14 * H + 2.0 * N * 8 * H),
  "edge_tile_bwd": (lambda: pkg._lib.check(L.dss2_edge_tile_bwd(xin.data_ptr(), 11, ein.data_ptr(), 13, W1.data_ptr(), b1.data_ptr(), g.data_ptr(), topo.tile_start.data_ptr(), topo.ell_ent_tiles.data_ptr(), topo.ell, topo.nrb, topo.ntiles, slab2.data_ptr(), 512, None, H, H, 8, 6, 0, st), "b"), 4.0 * topo.E2 * 22 * H),
  "pack_weights": (lambda: plan.refresh(), 1.0),
}
res = {k: [] for k in cases}
for k, (fn, _) in cases.items():
    fn()
torch.cuda.synchronize()
for r in range(a.rounds):
    for k, (fn, _) in cases.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) / a.reps * 1e3)
for k, (fn, fl) in cases.items():
    v = sorted(res[k]); med = v[len(v) // 2]
    print(f"{k:52s} median {med:8.1f} us  min {v[0]:8.1f} us  {fl / med / 1e6:7.1f} TFLOP/s ({100 * fl / med / 1e6 / 157.3:5.1f}% of fp32 MFMA peak)")
