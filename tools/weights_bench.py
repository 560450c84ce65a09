#!/usr/bin/env python3
"""GPU: the weight-space launches of a C2 training step (or --batch / --layers), one by one: fold, packing, the merged step-start
launch (dss2_prep_weights); slab reductions, chain rule of the fold, the merged step-end launch (dss2_finish_weights).  The arguments
are the ones a real step passes (captured from one step of the model); HIP events around back-to-back launches."""
import argparse, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--layers", type=int, default=4)
ap.add_argument("--hid", type=int, default=128)
ap.add_argument("--reps", type=int, default=200)
args = ap.parse_args()
DEV = "cuda:0"
ops, nw = pkg.ops, pkg.networks
REG = {"mu_v": 1e-1, "mu_theta": 1e-1, "lam_v": 1e-4, "lam_p": 1e-8, "lam_pf": 1e-6, "lam_reg": 1e2}
b = pkg.synthetic.make_batch(["cigre14"], args.batch, seed=0, violate=0.3)
x, ei, ea = b["x"].to(DEV), b["edge_index"].to(DEV), b["edge_attr"].to(DEV)
st = tuple(s.to(DEV) for s in b["stats"])
torch.manual_seed(0)
model = pkg.MPN(8, 6, 2, args.hid, args.layers, 2, 0.0).to(DEV)


def step():
    for p in model.parameters():
        p.grad = None
    out = model(x[:, :8], ei, ea[:, :6])
    loss = pkg.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=out, x_mean=st[0], x_std=st[1], edge_mean=st[2], edge_std=st[3],
                            edge_index=ei, reg_coefs=REG, num_samples=None, node_param=x[:, 8:], edge_param=ea[:, 6:])
    loss.backward(pkg.data.unit_grad(loss))


step()
cap = {}
orig_prep, orig_fin = ops.prep_weights, ops.finish_weights


def cap_prep(fold_tab, pack_tab, device):
    cap["prep"] = (fold_tab, pack_tab, device)
    return orig_prep(fold_tab, pack_tab, device)


def cap_fin(pending, rule_tab, base, dep_outs, device):
    cap["fin"] = (list(pending), rule_tab, base, set(dep_outs), device)
    return orig_fin(pending, rule_tab, base, dep_outs, device)


nw.finish_weights = cap_fin
pkg.plans.prep_weights = cap_prep
step()
nw.finish_weights, pkg.plans.prep_weights = orig_fin, orig_prep
torch.cuda.synchronize()


def timed(fn, reps=args.reps):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


L = pkg._lib.lib()
sp = pkg._lib.stream_ptr(DEV)
fold_tab, pack_tab, _ = cap["prep"]
ft, fcnt, fmx = fold_tab
t, cnt, mx, n_dep = pack_tab
cw = ops.weight_counters(DEV)
print(f"B={args.batch} L={args.layers} H={args.hid}: fold {fcnt} products x {fmx} tiles; packing {cnt} descriptors ({n_dep} of folded matrices) x {(mx + 255) // 256} workgroups")
t_fold = timed(lambda: L.dss2_small_gemm(ft.data_ptr(), fcnt, fmx, None, sp))
t_pack = timed(lambda: L.dss2_pack_weights(t.data_ptr(), cnt, mx, sp))
t_both = timed(lambda: (L.dss2_small_gemm(ft.data_ptr(), fcnt, fmx, None, sp), L.dss2_pack_weights(t.data_ptr(), cnt, mx, sp)))
t_prep = timed(lambda: L.dss2_prep_weights(ft.data_ptr(), fcnt, fmx, t.data_ptr(), cnt, n_dep, mx, cw.data_ptr(), sp))
t_prep0 = timed(lambda: L.dss2_prep_weights(None, 0, 0, t.data_ptr(), cnt, 0, mx, cw.data_ptr(), sp))
print(f"step start: fold {t_fold:.1f} us | packing {t_pack:.1f} us | the two back to back {t_both:.1f} us | merged {t_prep:.1f} us | merged kernel, packing only {t_prep0:.1f} us")

pending, rule_tab, base, dep_outs, _ = cap["fin"]
rt, rcnt, rmx = rule_tab
print(f"step end: {len(pending)} reductions " + ", ".join(f"{p[2]}x{p[5]}" for p in pending) + f"; chain rule {rcnt} products x {rmx} tiles")
old = pkg.flags.WEIGHTS_MERGED
pkg.flags.WEIGHTS_MERGED = False
t_red = timed(lambda: ops.reduce_pending(list(pending)))
t_rule = timed(lambda: L.dss2_small_gemm(rt.data_ptr(), rcnt, rmx, base.data_ptr(), sp))
t_sep = timed(lambda: ops.finish_weights(list(pending), rule_tab, base, dep_outs, DEV))
pkg.flags.WEIGHTS_MERGED = True
t_fin = timed(lambda: ops.finish_weights(list(pending), rule_tab, base, dep_outs, DEV))
dep = [p for p in pending if p[4].data_ptr() in dep_outs]
t_dep = timed(lambda: ops.reduce_pending(list(dep)))
pkg.flags.WEIGHTS_MERGED = old
print(f"step end: reductions {t_red:.1f} us (the fold's alone {t_dep:.1f}) | chain rule {t_rule:.1f} us | the two back to back {t_sep:.1f} us | merged {t_fin:.1f} us")
