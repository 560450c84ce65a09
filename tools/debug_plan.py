#!/usr/bin/env python3
"""GPU diagnostic: record one training step of a model as a launch plan and replay it with a synchronisation after every launch
(DSS2_PLAN_SYNC=2), to find a launch whose recorded arguments do not survive.  argv: class, B"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("DSS2_PLAN_SYNC", "2")
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
DEV = "cuda:0"
REG = {"mu_v": 1e-1, "mu_theta": 1e-1, "lam_v": 1e-4, "lam_p": 1e-8, "lam_pf": 1e-6, "lam_reg": 1e2}
cls = sys.argv[1] if len(sys.argv) > 1 else "SkipPFN"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cargs = (8, 6, 2, 32, 3, 2, 0.0, 3) if cls.endswith("PFN") else (8, 6, 2, 128, 4, 2, 0.0)
torch.manual_seed(0)
b = pkg.synthetic.make_batch(["cigre14"], B, seed=0, violate=0.3)
x, ei, ea = b["x"].to(DEV), b["edge_index"].to(DEV), b["edge_attr"].to(DEV)
st = tuple(s.to(DEV) for s in b["stats"])
model = getattr(pkg, cls)(*cargs).to(DEV)
params = list(model.parameters())


def step():
    for p in params:
        p.grad = None
    out = model(x[:, :8], ei, ea[:, :6])
    loss = pkg.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=out, x_mean=st[0], x_std=st[1], edge_mean=st[2], edge_std=st[3],
                            edge_index=ei, reg_coefs=REG, num_samples=None, node_param=x[:, 8:], edge_param=ea[:, 6:])
    loss.backward(pkg.data.unit_grad(loss))
    return loss


le = step().item()
plan = pkg.graphs.PlannedStep(step)
print("recorded", plan.n_launches, "launches; eager loss", le, flush=True)
lp = plan.replay()
torch.cuda.synchronize()
print("replayed loss", lp.item())
