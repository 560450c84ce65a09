#!/usr/bin/env python3
"""GPU: the reference driver's own model line -- SkipPFN(dim_hid=32, gnn_layers=8, K=2, dropout_rate=0.3, L=5) with Adamax
(dss2_run.py:72-92) -- at the driver's batch size (64) and at B=4096: forward + gsp_wls_edge + backward + optimizer step,
eager and as a replayed hipGraph (in-kernel dropout: every replay draws new masks).  Also p=0 for comparison."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
REG = {"mu_v": 1e-1, "mu_theta": 1e-1, "lam_v": 1e-4, "lam_p": 1e-8, "lam_pf": 1e-6, "lam_reg": 1e2}
dev = torch.device("cuda:0")
for B in (64, 4096):
    for p in (0.3, 0.0):
        torch.manual_seed(0)
        b = pkg.synthetic.make_batch(["cigre14"], B, seed=1)
        x, ei, ea = b["x"].to(dev), b["edge_index"].to(dev), b["edge_attr"].to(dev)
        st = tuple(s.to(dev) for s in b["stats"])
        model = pkg.SkipPFN(8, 6, 2, 32, 8, 2, p, 5).to(dev)
        opt = pkg.FusedAdamax(model.parameters(), lr=3e-3)

        def step(with_opt=True):
            for q in model.parameters(): q.grad = None
            out = model(x[:, :8], ei, ea[:, :6])
            loss = pkg.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=out, x_mean=st[0], x_std=st[1], edge_mean=st[2],
                                    edge_std=st[3], edge_index=ei, reg_coefs=REG, num_samples=None, node_param=x[:, 8:], edge_param=ea[:, 6:])
            loss.backward(pkg.data.unit_grad(loss))
            return loss

        opt_c = pkg.FusedAdamax(model.parameters(), lr=3e-3, capturable=True)

        def train_step():
            loss = step(); opt_c.step(); return loss
        gst = pkg.graphs.GraphedStep(train_step)       # forward + loss + backward + Adamax as ONE graph
        for _ in range(10): gst.replay()
        nn_ = 50 if B == 4096 else 200
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(nn_): gst.replay()
        torch.cuda.synchronize(); tgt = (time.perf_counter() - t0) / nn_
        gs = pkg.graphs.GraphedStep(step)              # forward + loss + backward only
        for _ in range(10): gs.replay()
        n = 50 if B == 4096 else 200
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): gs.replay()
        torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / n
        with torch.cuda.stream(gs.stream):
            for _ in range(10): step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): step()
            torch.cuda.synchronize(); te = (time.perf_counter() - t0) / n
            for _ in range(3): step(); opt.step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): step(); opt.step()
            torch.cuda.synchronize(); to = (time.perf_counter() - t0) / n
        print(f"SkipPFN(H=32, 8 layers, 5 blocks, dropout {p}) B={B:5d}: eager {te*1e3:7.3f} ms/step ({B/te/1e6:6.3f} M graphs/s) | "
              f"eager + FusedAdamax {to*1e3:7.3f} ms | hipGraph replay {tg*1e3:7.3f} ms/step ({B/tg/1e6:6.3f} M graphs/s) | "
              f"hipGraph incl. Adamax {tgt*1e3:7.3f} ms/step", flush=True)
