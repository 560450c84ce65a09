"""Where does the C3 (70-bus, B=1024) weight-gradient deviation of the full-size test come from?  Replays the test's fp64
referee (conv gates and the loading-max branch pinned to the HIP path's choices) and prints, stage by stage, how far the
HIP path is from it: d loss/d output, then every parameter gradient, with and without each pin."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
import dss2_oracle as oracle
DEV = "cuda:0"


def rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def main():
    os.environ["DSS2_EDGE_MFMA_FWD"] = sys.argv[1] if len(sys.argv) > 1 else "1"
    grids, B, hid, L = ["ober_sub"], 1024, 128, 4
    torch.manual_seed(0)
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
    out_m = mine(x[:, :8], ei, ea[:, :6])
    leaf = out_m.detach().clone().requires_grad_(True)      # d loss / d output of the HIP loss at the HIP output
    pkg.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=leaf * 1.0, x_mean=st[0], x_std=st[1], edge_mean=st[2],
                     edge_std=st[3], edge_index=ei, reg_coefs=oracle.DEFAULT_REG_COEFS, num_samples=None,
                     node_param=x[:, 8:], edge_param=ea[:, 6:]).backward()
    gout_m = leaf.grad
    flows = torch.empty(ei.shape[1], 8, device=DEV)
    loss_m = pkg.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=out_m, x_mean=st[0], x_std=st[1], edge_mean=st[2],
                              edge_std=st[3], edge_index=ei, reg_coefs=oracle.DEFAULT_REG_COEFS, num_samples=None,
                              node_param=x[:, 8:], edge_param=ea[:, 6:], pflow_out=flows)
    loss_m.backward()
    gates = []
    sd = mine.state_dict()
    with torch.no_grad():
        for l in range(L - 1):
            tr = pkg.MPN(8, 6, hid, hid, l + 1, 2, 0.0)
            tr.load_state_dict({k: v for k, v in sd.items() if k in tr.state_dict()})
            gates.append((tr.to(DEV)(x[:, :8], ei, ea[:, :6]) > 0).cpu())
    x64, ea64 = b["x"].double(), b["edge_attr"].double()
    st64 = tuple(s.double() for s in b["stats"])
    ei2, ea2 = oracle.undirect_graph(b["edge_index"], ea64[:, :6])
    vhv, vlv = b["x"][:, 8].max(), b["x"][:, 8].min()
    i_f, i_t = flows[:, 6].cpu(), flows[:, 7].cpu()

    def referee(pin_gates, pin_max, at_hip_output=False):
        r = oracle.MPN(8, 6, 2, hid, L, 2, 0.0).double()
        r.load_state_dict({k: v.double() for k, v in ref.state_dict().items()})
        h = r.edge_aggr(x64[:, :8], ei2, ea2)
        for l in range(L - 1):
            pre = r.convs[l](h, ei2)
            h = pre * gates[l].double() if pin_gates else torch.relu(pre)
        o = r.convs[-1](h, ei2)
        o.retain_grad()
        real = torch.maximum
        if pin_max:
            pins = iter([i_f >= i_t, i_f * vhv >= i_t * vlv])
            torch.maximum = lambda a_, b_: torch.where(next(pins), a_, b_)
        try:
            ls = oracle.gsp_wls_edge(input=x64[:, :8], edge_input=ea64[:, :6], output=o, x_mean=st64[0], x_std=st64[1],
                                     edge_mean=st64[2], edge_std=st64[3], edge_index=b["edge_index"],
                                     reg_coefs=oracle.DEFAULT_REG_COEFS, num_samples=None, node_param=x64[:, 8:], edge_param=ea64[:, 6:])
        finally:
            torch.maximum = real
        ls.backward()
        return r, o

    for pg, pm in [(True, True), (True, False), (False, True)]:
        r, o = referee(pg, pm)
        d = (gout_m.double().cpu() - o.grad).abs()
        sc = o.grad.abs().max()
        bad = (d > 1e-5 * sc).any(1).nonzero().flatten()
        print(f"pin gates={pg} max={pm}: out rel {rel(out_m, o):.2e}  dL/dout rel {d.max().item() / sc.item():.3e} nodes>1e-5: {bad.numel()} {bad[:8].tolist()}")
        for i in bad[:3].tolist():
            print(f"     node {i}: HIP {gout_m[i].tolist()} ref {o.grad[i].tolist()}")
        errs = {n: rel(p.grad, q.grad) for (n, p), (_, q) in zip(mine.named_parameters(), r.named_parameters())}
        print("     grads:", {k.replace('edge_aggr.edge_aggr', 'ea').replace('convs', 'c').replace('.lins', ''): f"{v:.1e}" for k, v in errs.items()})


if __name__ == "__main__":
    main()
