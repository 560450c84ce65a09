"""Diagnosis of the round-1 anomaly: with the edge-MLP forward on the matrix pipe (DSS2_EDGE_MFMA_FWD=1) the 70-bus
full-size gradient test found a first-layer weight-gradient deviation of ~3e-5 that the pinned conv gates did not
explain.  Hypothesis: a non-smooth point of the LOSS (penalty ReLUs, |theta_ij|) flipped by the 1e-7 output change.
This script runs the same step with both forwards and localises every difference:
  outputs, loss, d loss/d output (node by node), conv ReLU gates per layer, parameter gradients.
Run on the GPU box:  python tools/diag_edge_mfma.py"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
import dss2_oracle as oracle  # noqa: E402

DEV = "cuda:0"


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def main():
    torch.manual_seed(0)
    grids, B, hid, L = ["ober_sub"], 1024, 128, 4
    b = pkg.synthetic.make_batch(grids, B, seed=0)
    ref = oracle.MPN(8, 6, 2, hid, L, 2, 0.0)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if n.endswith("bias") and "convs" in n:
                p.uniform_(-0.1, 0.1)
    mine = pkg.MPN(8, 6, 2, hid, L, 2, 0.0)
    mine.load_state_dict(ref.state_dict())
    mine = mine.to(DEV)
    x, ei, ea = b["x"].to(DEV), b["edge_index"].to(DEV), b["edge_attr"].to(DEV)
    st = tuple(s.to(DEV) for s in b["stats"])

    def loss_of(out):
        return pkg.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=out, x_mean=st[0], x_std=st[1], edge_mean=st[2],
                                edge_std=st[3], edge_index=ei, reg_coefs=oracle.DEFAULT_REG_COEFS, num_samples=None,
                                node_param=x[:, 8:], edge_param=ea[:, 6:])

    def run(fwd_mfma):
        os.environ["DSS2_EDGE_MFMA_FWD"] = "1" if fwd_mfma else "0"
        for p in mine.parameters():
            p.grad = None
        out = mine(x[:, :8], ei, ea[:, :6])
        o0 = out.detach().clone()
        leaf = o0.clone().requires_grad_(True)
        l2 = loss_of(leaf * 1.0)
        l2.backward()
        loss = loss_of(out)
        loss.backward()
        gates = []
        sd = mine.state_dict()
        with torch.no_grad():
            for l in range(L - 1):
                tr = pkg.MPN(8, 6, hid, hid, l + 1, 2, 0.0)
                tr.load_state_dict({k: v for k, v in sd.items() if k in tr.state_dict()})
                gates.append(tr.to(DEV)(x[:, :8], ei, ea[:, :6]) > 0)
        torch.cuda.synchronize()
        return dict(out=o0, loss=loss.item(), dout=leaf.grad.clone(), gates=gates,
                    grads={n: p.grad.clone() for n, p in mine.named_parameters()})

    a, m = run(False), run(True)
    # ---- HIP's d loss / d output against the fp64 oracle's, both evaluated at the HIP (MFMA forward) output
    o64 = m["out"].double().cpu().requires_grad_(True)
    x64, ea64 = b["x"].double(), b["edge_attr"].double()
    st64 = tuple(s.double() for s in b["stats"])
    l64 = oracle.gsp_wls_edge(input=x64[:, :8], edge_input=ea64[:, :6], output=o64 * 1.0, x_mean=st64[0], x_std=st64[1],
                              edge_mean=st64[2], edge_std=st64[3], edge_index=b["edge_index"], reg_coefs=oracle.DEFAULT_REG_COEFS,
                              num_samples=None, node_param=x64[:, 8:], edge_param=ea64[:, 6:])
    l64.backward()
    g64 = o64.grad
    dd = (m["dout"].double().cpu() - g64).abs()
    sc = g64.abs().max()
    bad = (dd > 1e-5 * sc).any(1).nonzero().flatten()
    print(f"HIP vs fp64 d loss/d out at the same output: rel {dd.max().item() / sc.item():.3e}; nodes off by > 1e-5 max: {bad.numel()} {bad[:12].tolist()}")
    ei = b["edge_index"]
    yv = torch.cat([m["out"][:, 0:1] * st[1][:1] + st[0][:1], m["out"][:, 1:] * (1 - x[:, 9:10])], 1)
    fl = torch.stack(pkg.data.get_pflow(yv, b["edge_index"].to(DEV), x[:, 8:], ea[:, 6:]), 1).cpu()
    fl64 = torch.stack(oracle.get_pflow(yv.double().cpu(), ei, x64[:, 8:], ea64[:, 6:]), 1)
    for i in bad[:6].tolist():
        for e in ((ei[0] == i) | (ei[1] == i)).nonzero().flatten().tolist()[:4]:
            print(f"      edge {e} ({ei[0, e].item()}->{ei[1, e].item()}): HIP loading_line {fl[e, 0]:.6g} loading_trafo {fl[e, 1]:.6g} "
                  f"I_from {fl[e, 6]:.9g} I_to {fl[e, 7]:.9g} | fp64 I_from {fl64[e, 6]:.12g} I_to {fl64[e, 7]:.12g} "
                  f"theta_ij {(yv[ei[0, e], 1] - yv[ei[1, e], 1]).item():.6g}")
    for i in bad[:6].tolist():
        inc = ((ei[0] == i) | (ei[1] == i)).nonzero().flatten().tolist()
        print(f"   node {i}: HIP {m['dout'][i].tolist()} fp64 {g64[i].tolist()}  incident stored edges {inc} "
              f"params {[b['edge_attr'][e, 6:].tolist() for e in inc[:3]]}")

    print(f"out   rel diff {rel(m['out'], a['out']):.3e}   loss {a['loss']:.9g} vs {m['loss']:.9g}")
    d = (m["dout"] - a["dout"]).abs()
    scale = a["dout"].abs().max()
    big = (d > 1e-4 * scale).any(1).nonzero().flatten()
    print(f"dL/dout rel diff {rel(m['dout'], a['dout']):.3e}; nodes with |diff| > 1e-4 max: {big.numel()} {big[:10].tolist()}")
    for i in big[:5].tolist():
        print(f"   node {i}: out VALU {a['out'][i].tolist()} MFMA {m['out'][i].tolist()}  dout VALU {a['dout'][i].tolist()} MFMA {m['dout'][i].tolist()}")
    # where do the loss's non-smooth points sit?  v in physical units against 0.9 / 1.1
    v = a["out"][:, 0] * st[1][0] + st[0][0]
    for thr in (0.9, 1.1):
        k = (v - thr).abs().argmin()
        print(f"   closest v to {thr}: {v[k].item():.9f} (|d| = {(v[k] - thr).abs().item():.3e}); after MFMA fwd: "
              f"{(m['out'][k, 0] * st[1][0] + st[0][0]).item():.9f}")
    for l, (ga, gm) in enumerate(zip(a["gates"], m["gates"])):
        print(f"conv {l}: gates that differ between the two forwards: {int((ga != gm).sum())} of {ga.numel()}")
    for n in a["grads"]:
        print(f"grad {n:40s} rel diff {rel(m['grads'][n], a['grads'][n]):.3e}")


if __name__ == "__main__":
    main()
