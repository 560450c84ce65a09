#!/usr/bin/env python3
"""GPU: timeline of ONE replayed hipGraph step of the reference driver's model line (SkipPFN H=32, 8 layers, 5 blocks, p=0.3,
B from argv, default 64).  Two modes:
  run   (under rocprofv3 --kernel-trace --output-format csv): builds the step graph and replays it 30 times;
  post <kernel_trace.csv>: prints, for the LAST replay, every kernel with its duration and the idle gap before it, and the
  totals (sum of kernel time, sum of gaps) -- what a launch-count reduction can win at this batch size."""
import csv, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)


def run(B):
    import torch
    pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
    REG = {"mu_v": 1e-1, "mu_theta": 1e-1, "lam_v": 1e-4, "lam_p": 1e-8, "lam_pf": 1e-6, "lam_reg": 1e2}
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    b = pkg.synthetic.make_batch(["cigre14"], B, seed=1)
    x, ei, ea = b["x"].to(dev), b["edge_index"].to(dev), b["edge_attr"].to(dev)
    st = tuple(s.to(dev) for s in b["stats"])
    model = pkg.SkipPFN(8, 6, 2, 32, 8, 2, 0.3, 5).to(dev)
    opt = pkg.FusedAdamax(model.parameters(), lr=3e-3, capturable=True)

    def train_step():
        for q in model.parameters(): q.grad = None
        out = model(x[:, :8], ei, ea[:, :6])
        loss = pkg.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=out, x_mean=st[0], x_std=st[1], edge_mean=st[2],
                                edge_std=st[3], edge_index=ei, reg_coefs=REG, num_samples=None, node_param=x[:, 8:], edge_param=ea[:, 6:])
        loss.backward(pkg.data.unit_grad(loss))
        opt.step()
        return loss
    g = pkg.graphs.GraphedStep(train_step)
    for _ in range(30): g.replay()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(200): g.replay()
    torch.cuda.synchronize()
    print(f"B={B}: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per replayed step", flush=True)


def post(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the replays are identical kernel sequences: find the period from the last occurrence of the first kernel of a step
    names = [r["Kernel_Name"] for r in rows]
    key = "rng_next"
    idx = [i for i, n in enumerate(names) if key in n]
    per = None
    for cand in range(1, 400):        # kernels per step = distance between steps' first rng_next
        if len(idx) > 2 * cand and all(names[idx[-1 - cand] + j] == names[idx[-1 - 2 * cand] + j] for j in range(5)):
            pass
    # simpler: steps are separated by the largest gaps; take the last 1/40 of the trace between two Adamax kernels
    ad = [i for i, n in enumerate(names) if "adamax_kernel" in n or "adamax_flat_kernel" in n]
    a0, a1 = ad[-2], ad[-1]
    # several adamax launches per step may exist: walk back to the previous step's last adamax
    j = len(ad) - 1
    while j > 0 and ad[j] - ad[j - 1] < 5: j -= 1
    a1 = ad[-1]; a0 = ad[j - 1]
    step = rows[a0 + 1:a1 + 1]
    t_prev = int(rows[a0]["End_Timestamp"])
    tot_k = tot_g = 0
    agg = {}
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = s - t_prev
        nm = r["Kernel_Name"].split("(")[0][-60:]
        print(f"{nm:60s} dur {1e-3 * (e - s):8.1f} us  gap {1e-3 * gap:7.1f} us")
        tot_k += e - s; tot_g += max(gap, 0); t_prev = max(t_prev, e)
        a = agg.setdefault(nm, [0, 0, 0]); a[0] += 1; a[1] += e - s; a[2] += max(gap, 0)
    print(f"\n{len(step)} kernels per step; kernel time {tot_k * 1e-3:.1f} us, gaps {tot_g * 1e-3:.1f} us, span "
          f"{(int(step[-1]['End_Timestamp']) - int(rows[a0]['End_Timestamp'])) * 1e-3:.1f} us")
    for nm, a in sorted(agg.items(), key=lambda kv: -kv[1][1] - kv[1][2]):
        print(f"  {nm:60s} x{a[0]:3d}  kernels {a[1] * 1e-3:7.1f} us  gaps before {a[2] * 1e-3:7.1f} us")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "post":
        post(sys.argv[2])
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 64)
