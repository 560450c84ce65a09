#!/usr/bin/env python3
"""GPU (under rocprofv3 --kernel-trace --stats): N eager C5 steps, argv[1] = resident | fresh | prefetch.  Prints host enqueue time per step."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
REG = {"mu_v": 1e-1, "mu_theta": 1e-1, "lam_v": 1e-4, "lam_p": 1e-8, "lam_pf": 1e-6, "lam_reg": 1e2}
dev = torch.device("cuda:0")
B, S, N = 4096, 8192, 48
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
mode = sys.argv[1]
bt0 = next(iter(plain()))
for _ in range(4): step(bt0)
def batches():
    if mode == "resident":
        while True: yield bt0
    while True:
        for b in (plain() if mode == "fresh" else pkg.dataset.PrefetchLoader(plain())): yield b
it = batches()
for _ in range(4): step(next(it))
torch.cuda.synchronize(); t0 = time.perf_counter(); th_step = 0.0
for _ in range(N):
    b = next(it); t1 = time.perf_counter(); step(b); th_step += time.perf_counter() - t1
th = time.perf_counter() - t0; torch.cuda.synchronize(); tw = time.perf_counter() - t0
print(f"{mode}: host {th / N * 1e3:.3f} ms per step to enqueue (the step alone {th_step / N * 1e3:.3f}), wall {tw / N * 1e3:.3f} ms per step", flush=True)
