#!/usr/bin/env python3
"""Per-launch timing of one training step at C2: every gemm_prop / wgrad / fold launch, by position in
the step (HIP events around each call, averaged over steps).  python tools/launch_probe.py [steps]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
nw = pkg.networks
dev = torch.device("cuda", 0)
B, HID, L, K = int(os.environ.get("B", 4096)), int(os.environ.get("H", 128)), int(os.environ.get("L", 4)), 2
REG = {"mu_v": 1e-1, "mu_theta": 1e-1, "lam_v": 1e-4, "lam_p": 1e-8, "lam_pf": 1e-6, "lam_reg": 1e2}
batch = pkg.synthetic.make_batch(["cigre14"], B, seed=1000)
x, ei, ea = batch["x"].to(dev), batch["edge_index"].to(dev), batch["edge_attr"].to(dev)
stats = tuple(s.to(dev) for s in batch["stats"])
torch.manual_seed(0)
model = pkg.MPN(8, 6, 2, HID, L, K, 0.0).to(dev)
xin, ein, npar, epar = x[:, :8], ea[:, :6], x[:, 8:], ea[:, 6:]


def step():
    for p in model.parameters():
        p.grad = None
    out = model(xin, ei, ein)
    loss = pkg.gsp_wls_edge(input=xin, edge_input=ein, output=out, x_mean=stats[0], x_std=stats[1], edge_mean=stats[2],
                            edge_std=stats[3], edge_index=ei, reg_coefs=REG, num_samples=None, node_param=npar, edge_param=epar)
    loss.backward()


rec = []


def wrap(name, fn, label):
    def w(*a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **kw)
        e1.record()
        rec.append((label(*a, **kw), e0, e1))
        return r
    return w


for _ in range(5):
    step()
og, ow = nw.gemm_prop, nw.wgrad
nw.gemm_prop = wrap("gemm", og, lambda topo, X, ldx, kreal, Bp, nmat, hout, Y, **kw:
                    f"gemm_prop k={kreal} nmat={nmat} hout={hout}" + (" prebias" if kw.get("prebias") is not None else "")
                    + (" T" if kw.get("transposed") else "") + (f" narrow" if kw.get("narrow_h") else "")
                    + (f" prop_in" if kw.get("prop_in") else ""))
nw.wgrad = wrap("wgrad", ow, lambda topo, G, hout, X, hin, nmat, out, **kw:
                f"wgrad hout={hout} hin={hin} nmat={nmat}" + (" rowscale2" if kw.get("rowscale2") is not None else "")
                + (" rowscale" if kw.get("rowscale") is not None else ""))
if hasattr(nw, "_FoldPlan"):
    nw._FoldPlan.refresh_forward = wrap("foldf", nw._FoldPlan.refresh_forward, lambda *a, **k: "fold forward (small_gemm)")
    nw._FoldPlan.backward = wrap("foldb", nw._FoldPlan.backward, lambda *a, **k: "fold backward (small_gemm)")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
per = None
for s in range(steps):
    rec.clear()
    step()
    torch.cuda.synchronize()
    t = [(l, a.elapsed_time(b) * 1e3) for l, a, b in rec]
    if per is None:
        per = [[l, 0.0] for l, _ in t]
    for i, (l, us) in enumerate(t):
        per[i][1] += us
tot = 0.0
for l, us in per:
    print(f"{us / steps:8.1f} us  {l}")
    tot += us / steps
print(f"{tot:8.1f} us  total of the listed launches")
