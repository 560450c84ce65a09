#!/usr/bin/env python3
"""GPU: the reference driver's model line on its OTHER case (dss2_run.py:51-53: Oberrhein, 70 buses -> 96-row tiles): step time as a replayed
launch plan at B = 64 and B = 1024, and the kernels of one step (rocprofv3 --kernel-trace --stats around this script)."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
REG = {"mu_v": 1e-1, "mu_theta": 1e-1, "lam_v": 1e-4, "lam_p": 1e-8, "lam_pf": 1e-6, "lam_reg": 1e2}
dev = torch.device("cuda:0")
for B in [int(a) for a in sys.argv[1:]] or [64, 1024]:
    torch.manual_seed(0)
    b = pkg.synthetic.make_batch(["ober_sub"], B, seed=1)
    x, ei, ea = b["x"].to(dev), b["edge_index"].to(dev), b["edge_attr"].to(dev)
    st = tuple(s.to(dev) for s in b["stats"])
    model = pkg.SkipPFN(8, 6, 2, 32, 8, 2, 0.3, 5).to(dev)
    params = list(model.parameters())
    def step():
        for q in params: q.grad = None
        out = model(x[:, :8], ei, ea[:, :6])
        loss = pkg.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=out, x_mean=st[0], x_std=st[1], edge_mean=st[2],
                                edge_std=st[3], edge_index=ei, reg_coefs=REG, num_samples=None, node_param=x[:, 8:], edge_param=ea[:, 6:])
        loss.backward(pkg.data.unit_grad(loss)); return loss
    s = torch.cuda.Stream(); torch.cuda.set_stream(s)
    for _ in range(20): step()
    pl = pkg.graphs.PlannedStep(step, stream=s)
    for _ in range(20): pl.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 300
    for _ in range(n): pl.replay()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / n
    print(f"SkipPFN driver line on ober_sub B={B}: {t * 1e3:.3f} ms per step as a launch plan ({pl.n_launches} launches), loss {float(pl.loss):.6g}", flush=True)
    del pl
