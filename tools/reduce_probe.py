#!/usr/bin/env python3
"""GPU: what the step-end slab reduction of a C2 step spends its time on.  Captures the reduction list of a real backward
(ops.reduce_pending), then times the launch with subsets of its descriptors (20 launches inside a hipGraph: GPU time), and the whole
step as a launch plan.  Record with three variants that were not kept: profiles/experiments/r05_slab_reduction_probe.txt.
Usage: tools/reduce_probe.py [graphs]"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
REG = {"mu_v": 1e-1, "mu_theta": 1e-1, "lam_v": 1e-4, "lam_p": 1e-8, "lam_pf": 1e-6, "lam_reg": 1e2}
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
b = pkg.synthetic.make_batch(["cigre14"], B, seed=1)
x, ei, ea = b["x"].to(dev), b["edge_index"].to(dev), b["edge_attr"].to(dev)
st = tuple(s.to(dev) for s in b["stats"])
model = pkg.MPN(8, 6, 2, 128, 4, 2, 0.0).to(dev)


def step():
    for p in model.parameters():
        p.grad = None
    out = model(x[:, :8], ei, ea[:, :6])
    loss = pkg.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=out, x_mean=st[0], x_std=st[1], edge_mean=st[2],
                            edge_std=st[3], edge_index=ei, reg_coefs=REG, num_samples=None, node_param=x[:, 8:], edge_param=ea[:, 6:])
    loss.backward(pkg.data.unit_grad(loss))
    return loss


def timed(run, n=200):
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


ops = pkg.ops
real = ops.reduce_pending
for _once in (0,):
    captured = []

    def spy(pending):
        captured.clear()
        captured.extend(pending)
        return real(pending)
    ops.reduce_pending = spy
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ops.reduce_pending = real
    recs = list(captured)
    print(f"--- {len(recs)} reductions in the step-end launch")
    for r in recs:
        print(f"    n_slabs {r[2]:5d}  stride {r[3]:7d}  len {r[5]:7d}")
    # the launches around the reduction keep their slabs warm in the caches the way the step does not: only relative numbers
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def t_subset(sel):      # 20 launches inside a hipGraph: GPU time, not the host's launch rate
        chunk = [recs[i] for i in sel]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                real(list(chunk))
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20):
                real(list(chunk))
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(10):
            g.replay()
        ev1.record()
        torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) * 1000.0 / 200.0
    print(f"    all {t_subset(range(len(recs))):6.1f} us" + "".join(f" | only #{i} {t_subset([i]):6.1f}" for i in range(len(recs))))
    print(f"    #1 + #2 {t_subset([1, 2]):6.1f} us | #0 + #3 {t_subset([0, 3]):6.1f} us | all but #0 {t_subset([1, 2, 3]):6.1f} us | all but #3 {t_subset([0, 1, 2]):6.1f} us")
    pl = pkg.graphs.PlannedStep(step)
    print(f"    whole step as a launch plan: {timed(pl.replay):7.1f} us")
    del pl
