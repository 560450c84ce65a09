#!/usr/bin/env python3
"""GPU diagnostic (needs the -DDSS2_STAMPS build: DSS2_LIB=tools/diag_lib/libdss2_hip_stamps.so): per-wave phase durations of
wgrad_kernel<2,3,4> on the second tile of every workgroup (s_memtime ticks = 100 MHz constant clock)."""
import ctypes as C, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
nw = pkg.networks
dev = torch.device("cuda:0"); nmat = 3
H = int(sys.argv[3]) if len(sys.argv) > 3 else 128      # argv: grid, graphs, hidden width
GRID = sys.argv[1] if len(sys.argv) > 1 else "cigre14"; NB_ = int(sys.argv[2]) if len(sys.argv) > 2 else 4096      # argv: grid, graphs
b = pkg.synthetic.make_batch([GRID], NB_, seed=0)
ei = b["edge_index"].to(dev); N = b["x"].shape[0]
topo = pkg.topology.get_topology(ei, N)
h = torch.randn(N, H, device=dev); g = torch.randn(N, H, device=dev); flat = torch.empty(nmat * H * H + H, device=dev)
for _ in range(20):
    nw.wgrad(topo, g, H, h, H, nmat, flat)
torch.cuda.synchronize()
lib = C.CDLL(pkg._lib.LIB_PATH)
n = 256 * 8 * 16
buf = (C.c_ulonglong * n)()
assert lib.dss2_debug_read_wstamps(buf, n) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8, 16).astype(np.int64)
h0, h1 = t[:, :4, :], t[:, 4:, :]          # half 0: MFMA then propagation; half 1: bias + propagation then MFMA
def show(name, v):
    print(f"{name:46s} mean {v.mean():8.0f}  median {np.median(v):8.0f}  p90 {np.percentile(v, 90):8.0f} ticks")
show("staging (slab writes + ELL copy + barrier)", t[:, :, 1] - t[:, :, 0])
show("issue next-tile loads", t[:, :, 2] - t[:, :, 1])
if os.environ.get("WSTAMP_PF", "1") == "1":     # propagate-first schedule (NMAT == 3, tiles <= 64 rows)
    show("bias sums", t[:, :, 3] - t[:, :, 2])
    show("propagation 1 + barrier", t[:, :, 4] - t[:, :, 3])
    show("propagation 2 + barrier", t[:, :, 5] - t[:, :, 4])
    show("MFMA over the three slabs", t[:, :, 6] - t[:, :, 5])
    show("closing barrier", t[:, :, 14] - t[:, :, 6])
    show("tile total", t[:, :, 14] - t[:, :, 0])
    sys.exit(0)
if topo.nrb >= 4 and (topo.nrb != 6 or os.environ.get("DSS2_WGRAD_W8", "1") == "0"):      # tall tiles: the 4-wave kernel (NB = 1; 192-row tiles: DSS2_WGRAD_W8=0): every wave runs MFMA -> bias sums -> propagation per phase
    for ph, s0 in (("phase 0", 3), ("phase 1", 7), ("phase 2", 11)):
        start = t[:, :4, 2] if s0 == 3 else t[:, :4, s0 - 1]
        show(f"{ph}: MFMA", h0[:, :, s0] - start)
        show(f"{ph}: bias sums", h0[:, :, s0 + 1] - h0[:, :, s0])
        show(f"{ph}: propagation", h0[:, :, s0 + 2] - h0[:, :, s0 + 1])
        end = h0[:, :, 6] if s0 == 3 else (h0[:, :, 10] if s0 == 7 else h0[:, :, 14])
        show(f"{ph}: closing barrier", end - h0[:, :, s0 + 2])
    show("tile total", h0[:, :, 14] - h0[:, :, 0])
    sys.exit(0)
for ph, s0 in (("phase 0", 3), ("phase 1", 7), ("phase 2", 11)):
    start = t[:, :, 2] if s0 == 3 else t[:, :, s0 - 1]
    show(f"{ph} half0: MFMA", h0[:, :, s0] - start[:, :4])
    show(f"{ph} half0: propagation", h0[:, :, s0 + 2] - h0[:, :, s0 + 1])
    show(f"{ph} half1: bias sums" if s0 == 3 else f"{ph} half1: -", h1[:, :, s0] - start[:, 4:])
    show(f"{ph} half1: propagation", h1[:, :, s0 + 1] - h1[:, :, s0])
    show(f"{ph} half1: MFMA", h1[:, :, s0 + 2] - h1[:, :, s0 + 1])
    end = t[:, :, 6] if s0 == 3 else (t[:, :, 10] if s0 == 7 else t[:, :, 14])
    show(f"{ph} total incl. closing barrier", end - start)
show("tile total", t[:, :, 14] - t[:, :, 0])
