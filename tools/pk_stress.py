#!/usr/bin/env python3
"""GPU diagnostic for the packed-fp32 anomaly of round 2 (csrc/dss2_gemm_chain16.hip header): the failing geometry -- bf16x6
layer chain, B = 4096 (two workgroups per CU), folded bias in the first layer -- launched N times with the library named by
DSS2_LIB (build one WITH packed ops: DSS2_NOPK_SRCS="" DSS2_OUT=<pkg>/libdss2_pk.so DSS2_OBJ=/tmp/obj_pk bash csrc/build.sh).
Reports how many launches differ from the first one, where (layer, lanes of the epilogue's 64-lane row pieces, tile rows),
and the distance to the fp32-MFMA form."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
nw = pkg.networks
DEV = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
b = pkg.synthetic.make_batch(["cigre14"], B, seed=0)
ei = b["edge_index"].to(DEV); N = b["x"].shape[0]
topo = pkg.topology.get_topology(ei, N)
H, nmat, nl = 128, 3, 3
torch.manual_seed(3)
Ws = [torch.randn(H, H, device=DEV) * (1.5 / H ** 0.5) for _ in range(nmat)]
plan = nw._PackPlan([Ws], DEV, bf16_groups=(0,)); plan.refresh()
h = torch.randn(N, H, device=DEV)
bias, pbias, prs = torch.randn(H, device=DEV), torch.randn(nmat, H, device=DEV), torch.rand(N, 4, device=DEV)


def fwd(fmt, with_pre=True):
    outs = [torch.empty(N, H, device=DEV) for _ in range(nl)]
    layers = [dict(Bp=(plan.fwd16[0] if fmt else plan.fwd[0]), Y=o, bias=bias, relu=True) for o in outs]
    if with_pre:
        layers[0]["prebias"] = pbias
    nw.gemm_prop_chain(topo, h, H, nmat, layers, pre_rowscale=prs, b_format=fmt)
    return outs


ref, first = fwd(0), fwd(1)
torch.cuda.synchronize()
print(f"library {os.environ.get('DSS2_LIB', '(shipped)')}: B={B}, {topo.ntiles} tiles; first bf16x6 launch vs fp32 MFMA form: "
      + ", ".join(f"layer {i} {((a - r).abs().max() / r.abs().max()).item():.2e}" for i, (a, r) in enumerate(zip(first, ref))))
bad, lanes, rows_in_tile, layers_hit = 0, torch.zeros(64, dtype=torch.long), torch.zeros(64, dtype=torch.long), [0] * nl
for it in range(n - 1):
    out = fwd(1)
    differs = False
    for li, (a, f) in enumerate(zip(out, first)):
        d = (a != f)
        if d.any():
            differs = True
            layers_hit[li] += 1
            idx = d.nonzero()
            r_, c_ = idx[:, 0].cpu(), idx[:, 1].cpu()
            lane = ((r_ % 60) % 8) * 8 + (c_ % 32) // 4            # lane of the epilogue piece: r8 * 8 + (col in group) / 4
            lanes += torch.bincount(lane, minlength=64)
            rows_in_tile += torch.bincount(r_ % 60, minlength=64)
    bad += int(differs)
print(f"{bad} of {n - 1} repeat launches differ from the first; layers hit: {layers_hit}")
if bad:
    print("differing elements by epilogue lane group [0-15 | 16-31 | 32-47 | 48-63]:", [int(lanes[g * 16:(g + 1) * 16].sum()) for g in range(4)])
    print("differing elements by row inside the 60-row tile:", rows_in_tile[:60].tolist())
    ok = fwd(1, with_pre=False); ok2 = fwd(1, with_pre=False)
    print("same chain WITHOUT the folded-bias fma: two launches bitwise equal:", all(torch.equal(a, c) for a, c in zip(ok, ok2)))
