#!/usr/bin/env python3
"""GPU box: where the host time of a step goes (cProfile of 300 CPU-bound steps at B=512)."""
import cProfile, importlib, io, os, pstats, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
dev = torch.device("cuda:0")
REG = {"mu_v": 1e-1, "mu_theta": 1e-1, "lam_v": 1e-4, "lam_p": 1e-8, "lam_pf": 1e-6, "lam_reg": 1e2}
batch = pkg.synthetic.make_batch(["cigre14"], 512, seed=1000)
x, ei, ea = batch["x"].to(dev), batch["edge_index"].to(dev), batch["edge_attr"].to(dev)
stats = tuple(s.to(dev) for s in batch["stats"])
model = pkg.MPN(8, 6, 2, 128, 4, 2, 0.0).to(dev)
xin, ein, npar, epar = x[:, :8], ea[:, :6], x[:, 8:], ea[:, 6:]
def step():
    for p in model.parameters(): p.grad = None
    out = model(xin, ei, ein)
    loss = pkg.gsp_wls_edge(input=xin, edge_input=ein, output=out, x_mean=stats[0], x_std=stats[1], edge_mean=stats[2],
                            edge_std=stats[3], edge_index=ei, reg_coefs=REG, num_samples=None, node_param=npar, edge_param=epar)
    loss.backward()
for _ in range(200): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
