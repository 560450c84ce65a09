#!/usr/bin/env python3
"""GPU: what a NaN / an Inf does on the HIP path against the CPU oracle (torch semantics = the reference's), on the C2 model:
which output rows are non-finite, what the loss is.  (The findings are pinned by tests/test_gpu_nonfinite.py.)"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
import dss2_oracle as oracle
dev = "cuda:0"
REG = oracle.DEFAULT_REG_COEFS


def run(tag, mutate_x=None, mutate_w=None, hid=128, layers=4, B=8):
    torch.manual_seed(0)
    b = pkg.synthetic.make_batch(["cigre14"], B, seed=5)
    ref = oracle.MPN(8, 6, 2, hid, layers, 2, 0.0)
    if mutate_w:
        with torch.no_grad():
            mutate_w(ref)
    mine = pkg.MPN(8, 6, 2, hid, layers, 2, 0.0)
    mine.load_state_dict(ref.state_dict())
    mine = mine.to(dev)
    x = b["x"].clone()
    if mutate_x:
        mutate_x(x)
    bb = {"x": x, "edge_index": b["edge_index"], "edge_attr": b["edge_attr"]}
    out_r, loss_r = oracle.train_step(ref, bb, b["stats"], REG)
    xd, ei, ea = x.to(dev), b["edge_index"].to(dev), b["edge_attr"].to(dev)
    st = tuple(s.to(dev) for s in b["stats"])
    out = mine(xd[:, :8], ei, ea[:, :6])
    loss = pkg.gsp_wls_edge(input=xd[:, :8], edge_input=ea[:, :6], output=out, x_mean=st[0], x_std=st[1], edge_mean=st[2], edge_std=st[3],
                            edge_index=ei, reg_coefs=REG, num_samples=None, node_param=xd[:, 8:], edge_param=ea[:, 6:])
    loss.backward()
    torch.cuda.synchronize()
    bad_r = (~torch.isfinite(out_r).all(1)).nonzero().flatten().tolist()
    bad_m = (~torch.isfinite(out.cpu()).all(1)).nonzero().flatten().tolist()
    gr = {n: bool(torch.isfinite(p.grad).all()) for n, p in ref.named_parameters()}
    gm = {n: bool(torch.isfinite(p.grad).all()) for n, p in mine.named_parameters()}
    print(f"{tag}: oracle non-finite rows {bad_r[:20]}{'...' if len(bad_r) > 20 else ''} ({len(bad_r)}), loss {loss_r.item()} | HIP rows {bad_m[:20]}{'...' if len(bad_m) > 20 else ''} ({len(bad_m)}), loss {loss.item()}")
    print(f"    finite grads oracle: {sum(gr.values())}/{len(gr)}  HIP: {sum(gm.values())}/{len(gm)}   differing: {[n for n in gr if gr[n] != gm[n]]}")


nan, inf = float("nan"), float("inf")
run("clean")
run("NaN in x[20, 0] (graph 1)", mutate_x=lambda x: x.__setitem__((20, 0), nan))
run("Inf in x[20, 0] (graph 1)", mutate_x=lambda x: x.__setitem__((20, 0), inf))
run("NaN in convs[1].lins[0].weight[3, 5]", mutate_w=lambda m: m.convs[1].lins[0].weight.__setitem__((3, 5), nan))
run("Inf in convs[1].lins[0].weight[3, 5]", mutate_w=lambda m: m.convs[1].lins[0].weight.__setitem__((3, 5), inf))
run("NaN in convs[3] (head) lins[1].weight[0, 5]", mutate_w=lambda m: m.convs[3].lins[1].weight.__setitem__((0, 5), nan))
run("NaN in edge MLP W1[3, 2]", mutate_w=lambda m: m.edge_aggr.edge_aggr[0].weight.__setitem__((3, 2), nan))
run("H=32 L=2 (whole-stack kernels): NaN in x[20, 0]", mutate_x=lambda x: x.__setitem__((20, 0), nan), hid=32, layers=2)
run("H=32 L=2 (whole-stack kernels): NaN in convs[0].lins[0].weight[3, 5]", mutate_w=lambda m: m.convs[0].lins[0].weight.__setitem__((3, 5), nan), hid=32, layers=2)
