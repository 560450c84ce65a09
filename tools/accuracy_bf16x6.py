#!/usr/bin/env python3
"""GPU: accuracy of the three arithmetic routes of the H -> H layers -- f16x3 (default where the kernels have the form: DSS2_CHAIN_F16=1
DSS2_WGRAD_F16=1), bf16x6 (both =0) and the fp32-MFMA path (DSS2_CHAIN_BF16=0 DSS2_WGRAD_BF16=0) --, each measured against
the fp64 CPU oracle on the same weights and batch: max-normalised error of the output, relative error of the loss, worst
max-normalised error over the parameter gradients.  Runs both settings in child processes."""
import importlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))


def child():
    import torch
    pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
    import dss2_oracle as oracle
    dev = "cuda:0"
    res = {}
    for name, cls, args, grids, B in [("C2 model MPN H=128 L=4", "MPN", (8, 6, 2, 128, 4, 2, 0.0), ["cigre14"], 256),
                                      ("MPN H=64 L=6", "MPN", (8, 6, 2, 64, 6, 2, 0.0), ["cigre14", "cigre14_reswitched"], 128),
                                      ("SkipPFN H=32 8 layers x 5 blocks", "SkipPFN", (8, 6, 2, 32, 8, 2, 0.0, 5), ["cigre14"], 64)]:
        torch.manual_seed(0)
        b = pkg.synthetic.make_batch(grids, B, seed=3)
        ref = getattr(oracle, cls)(*args).double()
        mine = getattr(pkg, cls)(*args)
        mine.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
        mine = mine.to(dev)
        x, ei, ea = b["x"].to(dev), b["edge_index"].to(dev), b["edge_attr"].to(dev)
        st = tuple(s.to(dev) for s in b["stats"])
        out = mine(x[:, :8], ei, ea[:, :6])
        loss = pkg.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=out, x_mean=st[0], x_std=st[1], edge_mean=st[2],
                                edge_std=st[3], edge_index=ei, reg_coefs=oracle.DEFAULT_REG_COEFS, num_samples=None,
                                node_param=x[:, 8:], edge_param=ea[:, 6:])
        loss.backward()
        b64 = {"x": b["x"].double(), "edge_index": b["edge_index"], "edge_attr": b["edge_attr"].double()}
        out64, l64 = oracle.train_step(ref, b64, tuple(s.double() for s in b["stats"]))
        rel = lambda a, r: float((a.detach().double().cpu() - r.detach().double()).abs().max() / r.detach().double().abs().max())
        res[name] = dict(out=rel(out, out64), loss=abs(loss.item() - l64.item()) / abs(l64.item()),
                         grad=max(rel(p.grad, q.grad) for p, q in zip(mine.parameters(), ref.parameters())))
    print(json.dumps(res))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        rows = {}
        MODES = {"f16x3": dict(DSS2_CHAIN_BF16="1", DSS2_WGRAD_BF16="1", DSS2_CHAIN_F16="1", DSS2_WGRAD_F16="1"),
                 "bf16x6": dict(DSS2_CHAIN_BF16="1", DSS2_WGRAD_BF16="1", DSS2_CHAIN_F16="0", DSS2_WGRAD_F16="0"),
                 "fp32": dict(DSS2_CHAIN_BF16="0", DSS2_WGRAD_BF16="0", DSS2_CHAIN_F16="0", DSS2_WGRAD_F16="0")}
        for mode, envs in MODES.items():
            env = dict(os.environ, **envs)
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if not line:
                print(p.stdout[-2000:], p.stderr[-3000:]); sys.exit(1)
            rows[mode] = json.loads(line[-1])
        print(f"{'configuration':36s} {'path':8s} {'output':>10s} {'loss':>10s} {'worst grad':>10s}   (errors against the fp64 oracle)")
        for name in rows["fp32"]:
            for mode in MODES:
                label = mode
                r = rows[mode][name]
                print(f"{name:36s} {label:8s} {r['out']:10.2e} {r['loss']:10.2e} {r['grad']:10.2e}")
