#!/usr/bin/env python3
"""GPU: per-parameter error of the whole-stack kernels against the fp64 oracle (diagnostics)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dss2_oracle as oracle
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
DEV = "cuda:0"


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def run(cls, args, grids, B, p_note=""):
    torch.manual_seed(0)
    b = pkg.synthetic.make_batch(grids, B, seed=11)
    model = getattr(pkg, cls)(*args).to(DEV)
    with torch.no_grad():
        for q in model.parameters():
            if q.dim() == 1:
                q.uniform_(-0.2, 0.2)
    x, ei, ea = b["x"].to(DEV), b["edge_index"].to(DEV), b["edge_attr"].to(DEV)
    st = tuple(s.to(DEV) for s in b["stats"])
    torch.manual_seed(5)
    out = model(x[:, :8], ei, ea[:, :6])
    wts = torch.linspace(-1.0, 1.0, out.numel(), device=DEV).view_as(out)
    (out * wts).sum().backward()
    ref = getattr(oracle, cls)(*args).double()
    ref.load_state_dict({k: v.double().cpu() for k, v in model.state_dict().items()})
    blocks_m = list(model.mpns) if hasattr(model, "mpns") else [model]
    blocks_r = list(ref.mpns) if hasattr(ref, "mpns") else [ref]
    if args[6] > 0:
        for bm, br in zip(blocks_m, blocks_r):
            snap, pp = bm._last_dropout
            base = getattr(bm, "_drop_base", 0)
            br.dropout_masks = [pkg.networks.dropout_mask(snap, pp, base + l + 1, x.shape[0], 32).cpu() for l in range(bm.n_gnn_layers - 1)]
    o64 = ref(b["x"][:, :8].double(), b["edge_index"], b["edge_attr"][:, :6].double())
    (o64 * wts.double().cpu()).sum().backward()
    print(f"{cls}{args} B={B} fused={'_fused_plan' in model.__dict__}: out {rel(out, o64):.2e}")
    for (n, q), (_, r) in zip(model.named_parameters(), ref.named_parameters()):
        e = rel(q.grad, r.grad)
        print(f"   {n:40s} {e:.2e} {'  <<<<' if e > 1e-4 else ''}")


import sys
if len(sys.argv) > 1 and sys.argv[1] == "driver":
    run("SkipPFN", (8, 6, 2, 32, 8, 2, 0.3, 5), ["cigre14"], 64)
else:
    run("MPN", (8, 6, 2, 32, 2, 2, 0.0), ["cigre14"], 8)
    run("MPN", (8, 6, 2, 32, 3, 2, 0.0), ["cigre14"], 8)
    run("SkipPFN", (8, 6, 2, 32, 3, 2, 0.0, 2), ["cigre14"], 8)
