#!/usr/bin/env python3
"""GPU: cProfile of the host side of the eager C2 step (300 steps): where the ~0.5 ms of Python / ctypes / torch time per step goes."""
import cProfile, importlib, os, pstats, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
dev = torch.device("cuda:0")
REG = {"mu_v": 1e-1, "mu_theta": 1e-1, "lam_v": 1e-4, "lam_p": 1e-8, "lam_pf": 1e-6, "lam_reg": 1e2}
b = pkg.synthetic.make_batch(["cigre14"], 4096, seed=1)
x, ei, ea = b["x"].to(dev), b["edge_index"].to(dev), b["edge_attr"].to(dev)
st = tuple(s.to(dev) for s in b["stats"])
model = pkg.MPN(8, 6, 2, 128, 4, 2, 0.0).to(dev)
xin, ein, npar, epar = x[:, :8], ea[:, :6], x[:, 8:], ea[:, 6:]
params = list(model.parameters())
def step():
    for p in params: p.grad = None
    out = model(xin, ei, ein)
    loss = pkg.gsp_wls_edge(input=xin, edge_input=ein, output=out, x_mean=st[0], x_std=st[1], edge_mean=st[2], edge_std=st[3],
                            edge_index=ei, reg_coefs=REG, num_samples=None, node_param=npar, edge_param=epar)
    loss.backward()
if os.environ.get('SINGLE_THREAD_AUTOGRAD', '1') == '1':
    torch.autograd.set_multithreading_enabled(False)      # backward on the calling thread: visible to cProfile
params = list(model.parameters())
for _ in range(100): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300): step()
pr.disable()
torch.cuda.synchronize()
ps = pstats.Stats(pr); ps.sort_stats("tottime")
ps.print_stats(28)
