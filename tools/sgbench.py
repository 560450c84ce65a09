#!/usr/bin/env python3
"""GPU: the weight-space small-GEMM launches of the fold (forward table, chain-rule table, and subsets of it) timed alone."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
nw = pkg.networks
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = pkg.MPN(8, 6, 2, 128, 4, 2, 0.0).to(dev)
offs = m._flat_offsets()
fold = nw._FoldPlan(m.edge_aggr.edge_aggr[2].weight, m.edge_aggr.edge_aggr[2].bias, [l.weight for l in m.convs[0].lins], dev, int(offs[1]), int(offs[2]))
fwd, bwd = fold.records()
flat = torch.zeros(offs[-1], device=dev)
fold.gfold.normal_()


def timeit(tab, base, n=200):
    for _ in range(20):
        nw._small_gemm(tab, base, dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        nw._small_gemm(tab, base, dev)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("forward table (6 records):      %.2f us per launch (back to back, incl. launch gap)" % timeit(nw._sg_table(fwd, dev), 0))
print("chain-rule table (6 records):   %.2f us" % timeit(nw._sg_table(bwd, dev), flat.data_ptr()))
for name, sel in [("dW_m x3 (tB, rank-1)", bwd[0:3]), ("bias copy", bwd[3:4]), ("dW2 (tA, 3 batches)", bwd[4:5]), ("db2", bwd[5:6]), ("dW_m x1", bwd[0:1])]:
    print("  %-24s %.2f us" % (name, timeit(nw._sg_table(sel, dev), flat.data_ptr())))
