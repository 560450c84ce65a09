#!/usr/bin/env python3
"""GPU debug: the adversarial-scale model step under three chain routes (fp32 MFMA, bf16x6, f16x3): pairwise gradient differences."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
import test_gpu_f16x3 as T
n_per = 15
DEC = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
def xscale(n):
    g = torch.Generator(device="cpu").manual_seed(5)
    dec = torch.rand((n + n_per - 1) // n_per, generator=g) * DEC - DEC / 2
    return (10.0 ** dec).repeat_interleave(n_per)[:n]
def wscale(i, p):
    return 64.0 if i == 6 else (1.0 / 256.0 if i == 10 else 1.0)
model, step = T._model_step(pkg, "MPN", (8, 6, 2, 128, 4, 2, 0.0), ["cigre14"], 300, seed=22, xscale=xscale, wscale=wscale)
FL = pkg.flags
res = {}
for name, (b16, f16) in {"fp32": (False, False), "bf16x6": (True, False), "f16x3": (True, True)}.items():
    FL.CHAIN_BF16, FL.CHAIN_F16 = b16, f16
    res[name] = step()
    print(name, "plan.f16", model._plan.f16, "loss", res[name][1].item())
names = [n for n, _ in model.named_parameters()]
for a, b in (("bf16x6", "fp32"), ("f16x3", "fp32"), ("f16x3", "bf16x6")):
    print(a, "vs", b, "out", T._rel(res[a][0], res[b][0]))
    for n, x, y in zip(names, res[a][2], res[b][2]):
        print(f"   {n:28s} {T._rel(x, y):.3e}   max|g| {y.abs().max().item():.3e}")
