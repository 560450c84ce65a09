#!/usr/bin/env python3
"""GPU: the C2 step (or another cfgbench-style configuration) under module-flag settings, eager and replayed.
    python tools/exp_c2.py [B=4096] [grid=cigre14] [H=128] [L=4] [FLAG=value ...]      (FLAG: an attribute of <pkg>.flags)
Several settings in one call: separate them with '--'."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
REG = {"mu_v": 1e-1, "mu_theta": 1e-1, "lam_v": 1e-4, "lam_p": 1e-8, "lam_pf": 1e-6, "lam_reg": 1e2}
dev = torch.device("cuda:0")


def run(setting):
    kv = dict(a.split("=", 1) for a in setting)
    B, grid, H, L = int(kv.pop("B", 4096)), kv.pop("grid", "cigre14"), int(kv.pop("H", 128)), int(kv.pop("L", 4))
    saved = {}
    for k, v in kv.items():
        saved[k] = getattr(pkg.flags, k)
        setattr(pkg.flags, k, {"True": True, "False": False, "None": None}.get(v, v if not v.lstrip("-").isdigit() else int(v)))
    torch.manual_seed(0)
    b = pkg.synthetic.make_batch(grid.split("+"), B, seed=1)
    x, ei, ea = b["x"].to(dev), b["edge_index"].to(dev), b["edge_attr"].to(dev)
    st = tuple(s.to(dev) for s in b["stats"])
    model = pkg.MPN(8, 6, 2, H, L, 2, 0.0).to(dev)
    params = list(model.parameters())

    def step():
        for p in params:
            p.grad = None
        out = model(x[:, :8], ei, ea[:, :6])
        loss = pkg.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=out, x_mean=st[0], x_std=st[1], edge_mean=st[2],
                                edge_std=st[3], edge_index=ei, reg_coefs=REG, num_samples=None, node_param=x[:, 8:], edge_param=ea[:, 6:])
        loss.backward(pkg.data.unit_grad(loss))
        return loss

    def timed(f, secs=1.5):
        for _ in range(20):
            f()
        torch.cuda.synchronize()
        best = 1e9
        t_end = time.perf_counter() + secs
        while time.perf_counter() < t_end:
            t0 = time.perf_counter()
            for _ in range(50):
                f()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 50)
        return best * 1e3
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        e = timed(step)
        g = pkg.graphs.GraphedStep(step, stream=s)
        r = timed(g.replay)
        l = g.replay().item()
        gsum = sum(float(p.grad.double().abs().sum()) for p in params)
    print(f"{' '.join(setting) or '(defaults)':60s} eager {e:.4f} ms  replay {r:.4f} ms  loss {l:.9g}  sum|grad| {gsum:.9g}", flush=True)
    for k, v in saved.items():
        setattr(pkg.flags, k, v)


args, cur = [], []
for a in sys.argv[1:]:
    if a == "--":
        args.append(cur); cur = []
    else:
        cur.append(a)
args.append(cur)
for setting in args:
    run(setting)
