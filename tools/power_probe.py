#!/usr/bin/env python3
"""GPU: socket power and shader clock (rocm-smi, sampled once a second from a side thread) while the C2 step replays for ~12 s."""
import importlib, json, os, subprocess, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
REG = {"mu_v": 1e-1, "mu_theta": 1e-1, "lam_v": 1e-4, "lam_p": 1e-8, "lam_pf": 1e-6, "lam_reg": 1e2}
dev = torch.device("cuda:0")
b = pkg.synthetic.make_batch(["cigre14"], 4096, seed=1)
x, ei, ea = b["x"].to(dev), b["edge_index"].to(dev), b["edge_attr"].to(dev)
st = tuple(s.to(dev) for s in b["stats"])
model = pkg.MPN(8, 6, 2, 128, 4, 2, 0.0).to(dev)
params = list(model.parameters())
def step():
    for p in params: p.grad = None
    out = model(x[:, :8], ei, ea[:, :6])
    loss = pkg.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=out, x_mean=st[0], x_std=st[1], edge_mean=st[2], edge_std=st[3],
                            edge_index=ei, reg_coefs=REG, num_samples=None, node_param=x[:, 8:], edge_param=ea[:, 6:])
    loss.backward(pkg.data.unit_grad(loss)); return loss
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
for _ in range(100): step()
pl = pkg.graphs.PlannedStep(step, stream=s)
samples, stop = [], [False]
def sampler():
    while not stop[0]:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
            j = json.loads(out[out.index("{"):])
            c = next(iter(j.values()))
            samples.append({k: v for k, v in c.items() if "Power" in k or "sclk" in k or "mclk" in k})
        except Exception as e:
            samples.append({"error": str(e)[:100]})
        time.sleep(1.0)
t = threading.Thread(target=sampler); t.start()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 12:
    for _ in range(200): pl.replay()
    torch.cuda.synchronize(); n += 200
dt = time.perf_counter() - t0
stop[0] = True; t.join()
print(f"C2 step replayed {n} times: {dt / n * 1e3:.4f} ms per step")
for smp in samples: print(smp)
