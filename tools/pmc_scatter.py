#!/usr/bin/env python3
"""Launch ONLY the standalone scatter-add (dss2::segment_sum_kernel) on the cache-busting working set (B = 32768 CIGRE-14
graphs: messages 470 MB + sums 252 MB > 256 MiB Infinity Cache), for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
dev = torch.device("cuda:0"); H = 128
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
b = pkg.synthetic.make_batch(["cigre14"], 4096, seed=0)
ei0, n0 = b["edge_index"].to(dev), b["x"].shape[0]
reps = B // 4096
ei = torch.cat([ei0 + k * n0 for k in range(reps)], 1); N = n0 * reps
topo = pkg.topology.get_topology(ei, N)
msg = torch.randn(topo.E2, H, device=dev)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    pkg.networks.segment_sum(msg, topo.rowptr, topo.perm, N)
torch.cuda.synchronize()
