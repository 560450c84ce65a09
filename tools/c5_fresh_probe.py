#!/usr/bin/env python3
"""GPU: where a fresh mixed-topology batch per step (BASELINE C5) spends its time: the assembly alone (host wall, GPU time), the step with
the assembly in line, one batch ahead on a side stream of normal / of high priority.  `asm` as argv[1]: only the assembly loop (for rocprofv3)."""
import importlib, os, sys, time
import numpy as np
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
H, L = (int(os.environ.get("C5_H", "256")), int(os.environ.get("C5_L", "8")))
model = pkg.MPN(8, 6, 2, H, L, 2, 0.0).to(dev)
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
    while k < n:
        for bt in plain():
            pkg.dataset.PrefetchLoader._build_structure(bt)
            k += 1
            if k >= n: break


assembly(4); torch.cuda.synchronize()
if len(sys.argv) > 1 and sys.argv[1] == "asm":
    assembly(40); torch.cuda.synchronize(); sys.exit(0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record(); assembly(40); e1.record(); t_host = time.perf_counter() - t0; torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print(f"assembly alone (ragged collate + structure incl. the alternate tilings): host {t_host / 40 * 1e3:.3f} ms per batch to enqueue, {t_all / 40 * 1e3:.3f} ms wall, {e0.elapsed_time(e1) / 40:.3f} ms GPU")


def run(loader_factory, seconds=1.5):
    n, t0 = 0, None
    while True:
        for bt in loader_factory():
            step(bt); n += 1
            if t0 is None and n == 4:
                torch.cuda.synchronize(); t0, n0 = time.perf_counter(), n
        torch.cuda.synchronize()
        if time.perf_counter() - t0 >= seconds:
            return (time.perf_counter() - t0) / (n - n0) * 1e3


bt0 = next(iter(plain()))
t_res = run(lambda: [bt0] * 8)
print(f"resident batch {t_res:.3f} ms/step")
for name, f in [("in line", plain), ("prefetched, normal priority", lambda: pkg.dataset.PrefetchLoader(plain(), priority=0)),
                ("prefetched, high priority", lambda: pkg.dataset.PrefetchLoader(plain(), priority=-1))]:
    t = run(f)
    print(f"fresh batch per step, {name}: {t:.3f} ms/step (+{100 * (t / t_res - 1):.1f} %)")
