#!/usr/bin/env python3
"""GPU: cProfile of the host side of a fresh mixed-topology batch (ragged collation + structure build) and of one eager C5 step."""
import cProfile, importlib, os, pstats, sys, time, io
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
REG = {"mu_v": 1e-1, "mu_theta": 1e-1, "lam_v": 1e-4, "lam_p": 1e-8, "lam_pf": 1e-6, "lam_reg": 1e2}
dev = torch.device("cuda:0")
B, S = 4096, 8192
full = pkg.synthetic.make_batch(["cigre14", "cigre14_reswitched"], 256, seed=1)
parts = [pkg.dataset.DeviceDataset.from_batch(pkg.synthetic.make_batch([g], S, seed=2 + k, stats=full["stats"]), device=dev)
         for k, g in enumerate(["cigre14", "cigre14_reswitched"])]
ds = pkg.dataset.MixedDataset(parts)
st = tuple(s_.to(dev) for s_ in full["stats"])
model = pkg.MPN(8, 6, 2, 256, 8, 2, 0.0).to(dev)
params = list(model.parameters())
gen = torch.Generator(); gen.manual_seed(0)
plain = lambda: pkg.dataset.DataLoader(ds, batch_size=B, shuffle=True, generator=gen)
def step(bt):
    for p in params: p.grad = None
    out = model(bt.x[:, :8], bt.edge_index, bt.edge_attr[:, :6])
    loss = pkg.gsp_wls_edge(input=bt.x[:, :8], edge_input=bt.edge_attr[:, :6], output=out, x_mean=st[0], x_std=st[1], edge_mean=st[2],
                            edge_std=st[3], edge_index=bt.edge_index, reg_coefs=REG, num_samples=None, node_param=bt.x[:, 8:], edge_param=bt.edge_attr[:, 6:])
    loss.backward(pkg.data.unit_grad(loss)); return loss
def assembly(n):
    k = 0
    out = None
    while k < n:
        for bt in plain():
            pkg.dataset.PrefetchLoader._build_structure(bt); out = bt
            k += 1
            if k >= n: break
    return out
bt = assembly(4); 
for _ in range(5): step(bt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40): step(bt)
th = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"eager C5 step on a resident batch: host enqueue {th / 40 * 1e3:.3f} ms per step, wall {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms")
pr = cProfile.Profile(); pr.enable(); assembly(40); pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
