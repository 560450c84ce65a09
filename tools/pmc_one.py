#!/usr/bin/env python3
"""Launch ONLY one kernel shape repeatedly (for rocprofv3 --pmc): argv[1] in {fwd, dgrad, wgrad, chain, wgrad2, wgrad3, stack64, stack4096}."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
nw = pkg.networks
dev = torch.device("cuda:0"); H, nmat = 128, 3
b = pkg.synthetic.make_batch([os.environ.get("PMC_GRID", "cigre14")], int(os.environ.get("PMC_B", "4096")), seed=0)      # (PMC_GRID=ober_sub PMC_B=1024: the C3 shape)
ei = b["edge_index"].to(dev); N = b["x"].shape[0]
topo = pkg.topology.get_topology(ei, N)
Ws = [torch.randn(H, H, device=dev) * 0.1 for _ in range(nmat)]
B16 = nw.chain16_supported(topo, nmat, H, False)          # the product path: tile GEMM of the chain as bf16x6
F16 = bool(B16 and os.environ.get("PMC_F16", "1") == "1" and pkg.ops.chain_f16_supported(topo, nmat, H))      # ... as f16x3 where the chain has the form (PMC_F16=0: bf16x6)
plan = nw._PackPlan([Ws], dev, bf16_groups=((0,) if B16 else ()), f16=F16); plan.refresh()
h = torch.randn(N, H, device=dev); g = torch.randn(N, H, device=dev); out = torch.empty(N, H, device=dev)
bias = torch.randn(H, device=dev); flat = torch.empty(nmat * H * H + H, device=dev)
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
if which.startswith("stack"):      # the whole-stack kernels on the driver's model line: stack64 / stack4096 = graphs in the batch
    Bs = int(which[5:] or 4096)
    bs = pkg.synthetic.make_batch(["cigre14"], Bs, seed=1)
    xs, eis, eas = bs["x"].to(dev), bs["edge_index"].to(dev), bs["edge_attr"].to(dev)
    model = pkg.SkipPFN(8, 6, 2, 32, 8, 2, 0.3, 5).to(dev)
    for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
        for q in model.parameters(): q.grad = None
        model(xs[:, :8], eis, eas[:, :6]).square().sum().backward()
    torch.cuda.synchronize()
    sys.exit(0)
NREP = int(sys.argv[2]) if len(sys.argv) > 2 else 10
T0, T1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(NREP):
    if it == NREP // 5:
        T0.record()
    if which == "fwd":
        nw.gemm_prop(topo, h, H, H, plan.fwd[0], nmat, H, out, bias=bias, relu=True)
    elif which == "dgrad":
        nw.gemm_prop(topo, g, H, H, plan.bwd[0], nmat, H, out, relu_src=h, transposed=True)
    elif which == "chain":       # 3 chained H -> H layers (the C2 forward chain)
        outs = [torch.empty(N, H, device=dev) for _ in range(3)]
        nw.gemm_prop_chain(topo, h, H, nmat, [dict(Bp=(plan.fwd16[0] if B16 else plan.fwd[0]), Y=o, bias=bias, relu=True) for o in outs],
                           b_format=(2 if F16 else int(B16)))
    elif which == "wgrad2":      # two layers batched in one launch (the C2 backward of round 3)
        flat2 = torch.empty(2 * (nmat * H * H + H), device=dev)
        nw.wgrad_batched(topo, [g, g], H, [h, h], H, nmat, flat2)
    elif which == "wgrad3":      # the C2 backward of round 4: the folded conv 0 (extra scaled bias sums) + two plain layers in ONE launch
        flat2 = torch.empty(2 * (nmat * H * H + H), device=dev)
        first = torch.empty(nmat * H * H + H + nmat * H, device=dev)
        nw.wgrad_batched(topo, [g, g, g], H, [h, h, h], H, nmat, flat2, first_rowscale2=topo.deg_pows, first_out=first)
    else:
        nw.wgrad(topo, g, H, h, H, nmat, flat)
T1.record()
torch.cuda.synchronize()
if os.environ.get("PMC_TIME"):      # (launch + its slab reduction, back to back)
    print(f"{which} {os.environ.get('DSS2_LIB', 'product library')}: {1e3 * T0.elapsed_time(T1) / (NREP - NREP // 5):.1f} us per call")
