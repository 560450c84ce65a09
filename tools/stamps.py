#!/usr/bin/env python3
"""GPU diagnostics: in-kernel s_memtime phase stamps of the library's kernels, one subcommand per kernel family (round 5: the seven
stamp scripts folded into one).  Each needs a diagnostic build of the library with the family's stamp macro and DSS2_LIB pointing at it:

    DSS2_OUT=/tmp/libdss2_st.so DSS2_OBJ=/tmp/obj_st bash <pkg>/csrc/build.sh -D<MACRO>;  DSS2_LIB=/tmp/libdss2_st.so python tools/stamps.py <which> [args]

    which     macro               kernel
    gemm      DSS2_STAMPS         gemm_prop_kernel (H -> H forward, C2)
    teams     DSS2_STAMPS         gemm_prop_kernel, work / barrier-wait per step
    chain     DSS2_CHAIN_STAMPS   the bf16x6 layer chains (args: graphs, hidden width, layers)
    stack     DSS2_STACK_STAMPS   the whole-stack kernels (args: graphs, dropout p)
    wgrad     DSS2_STAMPS         wgrad_kernel<2,3,4> (args: grid, graphs, hidden width)
    wgradh    DSS2_STAMPS         wgrad16h_kernel (f16x3, 32-row tiles; args: grid, graphs)
"""

def _refuse_a_stale_diagnostic_library():
    """The stamp profiles must describe HEAD: tools/build_diag_libs.sh leaves the hash of every source it compiled beside the diagnostic
    libraries (tools/diag_lib/SOURCES.sha256); a library (DSS2_LIB) under tools/diag_lib/ whose sources have changed since is refused."""
    import glob, hashlib, os, sys
    lib = os.environ.get("DSS2_LIB")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not lib or os.path.dirname(os.path.abspath(lib)) != os.path.join(root, "tools", "diag_lib"):
        return
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "deep-*", "csrc", "*.h*")) + glob.glob(os.path.join(root, "include", "*.h"))):
        h.update(open(f, "rb").read())
    stamp = os.path.join(root, "tools", "diag_lib", "SOURCES.sha256")
    have = open(stamp).read().strip() if os.path.exists(stamp) else "(none)"
    if have != h.hexdigest():
        sys.exit(f"tools/stamps.py: {lib} was not built from the sources as they are now: run tools/build_diag_libs.sh")


_refuse_a_stale_diagnostic_library()

import ctypes as C, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")


def cmd_gemm(argv):
    """GPU diagnostic (needs the -DDSS2_STAMPS build, DSS2_LIB=.../libdss2_hip_stamps.so): per-wave phase
durations of gemm_prop (H->H forward, C2) from in-kernel s_memtime stamps."""
    sys_argv = [""] + list(argv)
    nw = pkg.networks
    dev = torch.device("cuda:0"); H, nmat = 128, 3
    b = pkg.synthetic.make_batch(["cigre14"], 4096, seed=0)
    ei = b["edge_index"].to(dev); N = b["x"].shape[0]
    topo = pkg.topology.get_topology(ei, N)
    Ws = [torch.randn(H, H, device=dev) * 0.1 for _ in range(nmat)]
    plan = nw._PackPlan([Ws], dev); plan.refresh()
    h = torch.randn(N, H, device=dev); out = torch.empty(N, H, device=dev); bias = torch.randn(H, device=dev)
    for _ in range(5):
        nw.gemm_prop(topo, h, H, H, plan.fwd[0], nmat, H, out, bias=bias, relu=True)
    torch.cuda.synchronize()
    lib = C.CDLL(pkg._lib.LIB_PATH)
    n = topo.ntiles * 4 * 8
    buf = (C.c_ulonglong * n)()
    assert lib.dss2_debug_read_stamps(buf, n) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.int64)   # s_memtime: 100 MHz constant clock? (ticks)
    names = ["staging+barrier (1->2)", "MFMA loop (2->3)", "Horner (3->4)", "stores (4->5)", "tile total (1->5)"]
    d = [t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3], t[:, 5] - t[:, 4], t[:, 5] - t[:, 1]]
    t0 = t[:, 1].min()
    print(f"tiles={topo.ntiles} waves={t.shape[0]} kernel span (first tile start -> last store) {t[:, 5].max() - t0} ticks")
    for nm, v in zip(names, d):
        print(f"{nm:24s} mean {v.mean():9.0f}  median {np.median(v):9.0f}  p90 {np.percentile(v, 90):9.0f} ticks  ({100 * v.mean() / d[4].mean():5.1f}% of tile)")
    g = t.reshape(topo.ntiles, 4, 8)
    nwg = min(topo.ntiles, 512)
    first, second = g[:nwg], g[nwg:2 * nwg]
    print("tile-1 start (median over WGs, rel. kernel start): round-1 WGs", int(np.median(first[:256, 0, 1]) - t0), " round-2 WGs", int(np.median(first[256:, 0, 1]) - t0))
    if len(second):
        print("gap tile-1 end -> tile-2 start (barrier):", int(np.median(second[:, 0, 1] - first[:len(second), :, 5].max(axis=1))),
              "ticks; tile-2 staging:", int(np.median(second[:, 0, 2] - second[:, 0, 1])), " tile-1 staging:", int(np.median(first[:, 0, 2] - first[:, 0, 1])))
        print("tile-1 total:", int(np.median(first[:, :, 5].max(axis=1) - first[:, 0, 1])), " tile-2 total:", int(np.median(second[:, :, 5].max(axis=1) - second[:, 0, 1])))


def cmd_teams(argv):
    """GPU diagnostic (-DDSS2_STAMPS build): per-step work / barrier-wait durations of the two-team kernel."""
    sys_argv = [""] + list(argv)
    nw = pkg.networks
    dev = torch.device("cuda:0"); H, nmat = 128, 3
    b = pkg.synthetic.make_batch(["cigre14"], 4096, seed=0)
    ei = b["edge_index"].to(dev); N = b["x"].shape[0]
    topo = pkg.topology.get_topology(ei, N)
    Ws = [torch.randn(H, H, device=dev) * 0.1 for _ in range(nmat)]
    plan = nw._PackPlan([Ws], dev); plan.refresh()
    h = torch.randn(N, H, device=dev); out = torch.empty(N, H, device=dev); bias = torch.randn(H, device=dev)
    for _ in range(5):
        nw.gemm_prop(topo, h, H, H, plan.fwd[0], nmat, H, out, bias=bias, relu=True)
    torch.cuda.synchronize()
    lib = C.CDLL(pkg._lib.LIB_PATH)
    n = 256 * 8 * 16
    buf = (C.c_ulonglong * n)()
    assert lib.dss2_debug_read_stamps(buf, n) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8, 16).astype(np.int64)
    for s in range(5):
        w0, w1, w2 = t[:, :, 3 * s], t[:, :, 3 * s + 1], t[:, :, 3 * s + 2]
        for team in (0, 1):
            sl = slice(0, 4) if team == 0 else slice(4, 8)
            role = "MFMA" if (s & 1) == team else ("epilogue+stage" if s >= 1 else "idle")
            work = (w1 - w0)[:, sl]; wait = (w2 - w1)[:, sl]
            print(f"step {s} team {team} {role:15s} work mean {work.mean():8.0f} p90 {np.percentile(work, 90):8.0f}   barrier wait mean {wait.mean():8.0f}")
    tot = t[:, :, 14] - t[:, :, 0]
    print("whole loop per wave: mean", int(tot.mean()), "max", int(tot.max()))


def cmd_chain(argv):
    """GPU diagnostic (needs the -DDSS2_CHAIN_STAMPS build: DSS2_OUT=tools/diag_lib/libdss2_cstamps.so DSS2_OBJ=/tmp/obj_cst bash csrc/build.sh -DDSS2_CHAIN_STAMPS; run with DSS2_LIB=tools/diag_lib/libdss2_cstamps.so): s_memtime phase stamps of the bf16x6
layer chain (forward; argv: graphs, hidden width (128), layers (3): the C2 shape by default) -- per layer: GEMM phase, barrier wait, Horner, epilogue, barrier wait, as the
median over workgroups and waves.  argv[1] = graphs in the batch (1024: one workgroup per CU; 4096: two per CU, two rounds)."""
    sys_argv = [""] + list(argv)
    nw = pkg.networks
    dev = torch.device("cuda:0"); nmat = 3
    B = int(sys_argv[1]) if len(sys_argv) > 1 else 4096
    H = int(sys_argv[2]) if len(sys_argv) > 2 else 128
    nl = int(sys_argv[3]) if len(sys_argv) > 3 else 3
    GRID = sys_argv[4] if len(sys_argv) > 4 else "cigre14"
    b = pkg.synthetic.make_batch([GRID], B, seed=0)
    ei = b["edge_index"].to(dev); N = b["x"].shape[0]
    topo = pkg.topology.get_topology(ei, N)
    Ws = [torch.randn(H, H, device=dev) * 0.1 for _ in range(nmat)]
    f16 = os.environ.get("STAMPS_F16", "1") == "1" and pkg.ops.chain_f16_supported(topo, nmat, H)      # (STAMPS_F16=0: the bf16x6 form)
    plan = nw._PackPlan([Ws], dev, bf16_groups=(0,), f16=f16); plan.refresh()
    h = torch.randn(N, H, device=dev); bias = torch.randn(H, device=dev)
    outs = [torch.empty(N, H, device=dev) for _ in range(nl)]
    print("tile GEMM as", "f16x3" if f16 else "bf16x6")
    run = lambda: nw.gemm_prop_chain(topo, h, H, nmat, [dict(Bp=plan.fwd16[0], Y=o, bias=bias, relu=True) for o in outs], b_format=(2 if f16 else 1))
    for _ in range(200): run()
    torch.cuda.synchronize()
    run(); torch.cuda.synchronize()
    nwg = min(topo.ntiles, 2048)
    buf = (C.c_ulonglong * (nwg * 8 * 64))()
    lib = pkg._lib.lib()
    sp = H >= 96 and os.environ.get("DSS2_CHAIN_SP", "1") != "0"      # the split-plane kernel keeps its stamps in its own translation unit
    sp3 = topo.nrb == 3 and H >= 64 and os.environ.get("DSS2_CHAIN_SP", "1") != "0"
    reader = lib.dss2_debug_read_cstamps_sp6 if sp3 else (lib.dss2_debug_read_cstamps_sp if sp else lib.dss2_debug_read_cstamps)      # (96-row tiles: gemm_chain_sp6_kernel<3, .>)
    reader.argtypes = [C.c_void_p, C.c_int]
    assert reader(buf, nwg * 8 * 64) == 0
    ncg = (H + 31) // 32
    nwav = min(8, ncg * (2 if (ncg <= 2 and topo.nrb != 3) else 1))      # waves per workgroup (row split for narrow layers)
    st = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 8, 64)[:, :nwav, :].astype(np.int64)
    us = lambda d: float(np.median(d))      # s_memtime ticks = shader cycles
    print(f"B={B}: {topo.ntiles} tiles; shader cycles (median over workgroups x waves)")
    print(f"  first barrier wait: {us(st[:, :, 1] - st[:, :, 0]):.0f}")
    tot = 0
    for li in range(nl):
        s = lambda i: st[:, :, 2 + li * 6 + i]
        prev = st[:, :, 1] if li == 0 else st[:, :, 2 + (li - 1) * 6 + 4]
        gemm, bar1, horner, epi, bar2 = us(s(0) - prev), us(s(1) - s(0)), us(s(2) - s(1)), us(s(3) - s(2)), us(s(4) - s(3))
        epi_a = us(s(5) - s(2))
        print(f"  layer {li}: GEMM {gemm:7.0f}  barrier {bar1:5.0f}  Horner {horner:6.0f}  epilogue {epi:6.0f} (T -> stage {epi_a:5.0f}, rows -> HBM / X tile {epi - epi_a:5.0f})  barrier {bar2:5.0f}   sum {gemm + bar1 + horner + epi + bar2:7.0f} cycles")
    if sp and not sp3:
        dt, drt = st[:, :, 2 + (nl - 1) * 6 + 4] - st[:, :, 1], st[:, :, 63] - st[:, :, 62]
        print(f"  in-kernel clock (d s_memtime / d s_memrealtime x 100 MHz), median: {np.median(dt / np.maximum(drt, 1)) * 0.1:.2f} GHz")
    if sp and not sp3:
        # wall-clock picture (100 MHz s_memrealtime): when workgroups start their first layer and when they end
        t0 = st[:, :, 62].min(); a = (st[:, :, 62].min(axis=1) - t0) / 100.0; e = (st[:, :, 63].max(axis=1) - t0) / 100.0
        q = lambda v: " ".join(f"{x:6.1f}" for x in np.percentile(v, [0, 10, 50, 90, 100]))
        n1 = min(nwg, 512)
        print(f"  wall clock, us after the first workgroup's start (min p10 p50 p90 max):")
        print(f"    workgroups 0..{n1 - 1}: start {q(a[:n1])} | end {q(e[:n1])} | duration {q(e[:n1] - a[:n1])}")
        if nwg > n1: print(f"    workgroups {n1}..{nwg - 1}: start {q(a[n1:])} | end {q(e[n1:])} | duration {q(e[n1:] - a[n1:])}")
    wg = st[:, :, 2 + (nl - 1) * 6 + 3].max(axis=1) - st[:, :, 0].min(axis=1)
    print(f"  per workgroup, first stamp -> last epilogue: median {np.median(wg):.0f} cycles, max {wg.max():.0f}")


def cmd_stack(argv):
    """GPU diagnostic: s_memtime phase stamps of the whole-stack kernels (csrc/dss2_stack.hip).  Needs the stamps build:
    DSS2_OUT=tools/diag_lib/libdss2_sstamps.so DSS2_OBJ=/tmp/obj_sst bash <pkg>/csrc/build.sh -DDSS2_STACK_STAMPS
    DSS2_LIB=tools/diag_lib/libdss2_sstamps.so python tools/stamps.py stack [graphs]
Prints, per phase, the median over workgroups of (max over waves) in shader cycles: forward block 0, backward last block."""
    sys_argv = [""] + list(argv)
    dev = torch.device("cuda:0")
    B = int(sys_argv[1]) if len(sys_argv) > 1 else 64
    p = float(sys_argv[2]) if len(sys_argv) > 2 else 0.3
    n_hh = 7
    b = pkg.synthetic.make_batch(["cigre14"], B, seed=1)
    x, ei, ea = b["x"].to(dev), b["edge_index"].to(dev), b["edge_attr"].to(dev)
    model = pkg.SkipPFN(8, 6, 2, 32, n_hh + 1, 2, p, 5).to(dev)


    def step():
        for q in model.parameters(): q.grad = None
        model(x[:, :8], ei, ea[:, :6]).square().sum().backward()


    for _ in range(50): step()
    torch.cuda.synchronize()
    step(); torch.cuda.synchronize()
    lib = pkg._lib.lib()
    lib.dss2_debug_read_sstamps.argtypes = [C.c_void_p, C.c_int]
    stack_mod = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd.stack")
    tset = stack_mod.tiles_of(pkg.topology.get_topology(ei, x.shape[0]))
    nwg = min(64, tset.ntiles)
    print(f"tiles: {tset.ntiles} of {32 * tset.nrb} rows")


    def read(which):
        buf = (C.c_ulonglong * (64 * 8 * 128))()
        assert lib.dss2_debug_read_sstamps(buf, which) == 0
        return np.frombuffer(buf, dtype=np.uint64).reshape(64, 8, 128)[:nwg].astype(np.int64)


    def span(st, a, b_):      # median over workgroups of (latest wave at b_) - (latest wave at a)
        return float(np.median(st[:, :, b_].max(axis=1) - st[:, :, a].max(axis=1)))


    f = read(0)
    print(f"B={B} p={p}: forward, block 0 (cycles, median over {nwg} workgroups)")
    print(f"  staging {span(f, 0, 1):6.0f}   edge MLP {span(f, 1, 2):6.0f} (+ barrier {span(f, 2, 3):4.0f})")
    for l in range(n_hh):
        s0 = 3 + 6 * l if l else 3
        base = 4 + 6 * l
        prev = 3 if l == 0 else base - 1
        print(f"  conv {l}: MFMA+mask {span(f, prev, base):6.0f}  bar {span(f, base, base + 1):4.0f}  hop1 {span(f, base + 1, base + 2):5.0f}  bar {span(f, base + 2, base + 3):4.0f}  "
              f"hop2+epilogue {span(f, base + 3, base + 4):5.0f}  bar {span(f, base + 4, base + 5):4.0f}   layer {span(f, prev, base + 5):6.0f}")
    h0 = 4 + 6 * n_hh
    print(f"  head {span(f, h0 - 1, h0):6.0f}    block 0 total {span(f, 1, h0):7.0f}   whole kernel (5 blocks) {span(f, 0, h0 + 1):8.0f}")
    g = read(1)
    print(f"backward, last block, first tile")
    print(f"  staging {span(g, 0, 1):6.0f} + barrier {span(g, 1, 2):5.0f}")
    for i in range(n_hh + 1):
        u = n_hh - i
        base = 3 + 5 * i
        prev = 2 if i == 0 else base - 1
        print(f"  unit {u} ({'head' if i == 0 else 'conv'}): hops {span(g, prev, base):6.0f}  MFMA phase {span(g, base, base + 1):6.0f}  bar {span(g, base + 1, base + 2):4.0f}  "
              f"gate {span(g, base + 2, base + 3):5.0f}  bar {span(g, base + 3, base + 4):4.0f}   unit {span(g, prev, base + 4):6.0f}")
        if i:      # per-role MFMA phase time
            q0 = 3 * u
            d = g[:, :, base + 1] - g[:, :, base]
            roles = {"wgrad": [(q0 + k) % 8 for k in range(3)], "dgrad": [(q0 + 3 + k) % 8 for k in range(4)], "bias sums": [(q0 + 7) % 8]}
            print("           " + "  ".join(f"{r}: {np.median(d[:, w].max(axis=1)):.0f}" for r, w in roles.items()))
    e0 = 3 + 5 * (n_hh + 1)
    print(f"  edge passes {span(g, e0 - 1, e0):6.0f}   dx + barrier {span(g, e0, e0 + 1):6.0f}   tile total {span(g, 0, e0 + 1):8.0f}")


def cmd_wgrad(argv):
    """GPU diagnostic (needs the -DDSS2_STAMPS build: DSS2_LIB=tools/diag_lib/libdss2_hip_stamps.so): per-wave phase durations of
wgrad_kernel<2,3,4> on the second tile of every workgroup (s_memtime ticks = 100 MHz constant clock)."""
    sys_argv = [""] + list(argv)
    nw = pkg.networks
    dev = torch.device("cuda:0"); nmat = 3
    H = int(sys_argv[3]) if len(sys_argv) > 3 else 128      # argv: grid, graphs, hidden width
    GRID = sys_argv[1] if len(sys_argv) > 1 else "cigre14"; NB_ = int(sys_argv[2]) if len(sys_argv) > 2 else 4096      # argv: grid, graphs
    b = pkg.synthetic.make_batch([GRID], NB_, seed=0)
    ei = b["edge_index"].to(dev); N = b["x"].shape[0]
    topo = pkg.topology.get_topology(ei, N)
    h = torch.randn(N, H, device=dev); g = torch.randn(N, H, device=dev); flat = torch.empty(nmat * H * H + H, device=dev)
    for _ in range(20):
        nw.wgrad(topo, g, H, h, H, nmat, flat)
    torch.cuda.synchronize()
    lib = C.CDLL(pkg._lib.LIB_PATH)
    n = 256 * 8 * 16
    buf = (C.c_ulonglong * n)()
    assert lib.dss2_debug_read_wstamps(buf, n) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8, 16).astype(np.int64)
    h0, h1 = t[:, :4, :], t[:, 4:, :]          # half 0: MFMA then propagation; half 1: bias + propagation then MFMA
    def show(name, v):
        print(f"{name:46s} mean {v.mean():8.0f}  median {np.median(v):8.0f}  p90 {np.percentile(v, 90):8.0f} ticks")
    show("staging (slab writes + ELL copy + barrier)", t[:, :, 1] - t[:, :, 0])
    show("issue next-tile loads", t[:, :, 2] - t[:, :, 1])
    if os.environ.get("WSTAMP_PF", "1") == "1":     # propagate-first schedule (NMAT == 3, tiles <= 64 rows)
        show("bias sums", t[:, :, 3] - t[:, :, 2])
        show("propagation 1 + barrier", t[:, :, 4] - t[:, :, 3])
        show("propagation 2 + barrier", t[:, :, 5] - t[:, :, 4])
        show("MFMA over the three slabs", t[:, :, 6] - t[:, :, 5])
        show("closing barrier", t[:, :, 14] - t[:, :, 6])
        show("tile total", t[:, :, 14] - t[:, :, 0])
        return
    if topo.nrb >= 4 and (topo.nrb != 6 or os.environ.get("DSS2_WGRAD_W8", "1") == "0"):      # tall tiles: the 4-wave kernel (NB = 1; 192-row tiles: DSS2_WGRAD_W8=0): every wave runs MFMA -> bias sums -> propagation per phase
        for ph, s0 in (("phase 0", 3), ("phase 1", 7), ("phase 2", 11)):
            start = t[:, :4, 2] if s0 == 3 else t[:, :4, s0 - 1]
            show(f"{ph}: MFMA", h0[:, :, s0] - start)
            show(f"{ph}: bias sums", h0[:, :, s0 + 1] - h0[:, :, s0])
            show(f"{ph}: propagation", h0[:, :, s0 + 2] - h0[:, :, s0 + 1])
            end = h0[:, :, 6] if s0 == 3 else (h0[:, :, 10] if s0 == 7 else h0[:, :, 14])
            show(f"{ph}: closing barrier", end - h0[:, :, s0 + 2])
        show("tile total", h0[:, :, 14] - h0[:, :, 0])
        return
    for ph, s0 in (("phase 0", 3), ("phase 1", 7), ("phase 2", 11)):
        start = t[:, :, 2] if s0 == 3 else t[:, :, s0 - 1]
        show(f"{ph} half0: MFMA", h0[:, :, s0] - start[:, :4])
        show(f"{ph} half0: propagation", h0[:, :, s0 + 2] - h0[:, :, s0 + 1])
        show(f"{ph} half1: bias sums" if s0 == 3 else f"{ph} half1: -", h1[:, :, s0] - start[:, 4:])
        show(f"{ph} half1: propagation", h1[:, :, s0 + 1] - h1[:, :, s0])
        show(f"{ph} half1: MFMA", h1[:, :, s0 + 2] - h1[:, :, s0 + 1])
        end = t[:, :, 6] if s0 == 3 else (t[:, :, 10] if s0 == 7 else t[:, :, 14])
        show(f"{ph} total incl. closing barrier", end - start)
    show("tile total", t[:, :, 14] - t[:, :, 0])


def cmd_wgradh(argv):
    """GPU diagnostic (needs a -DDSS2_STAMPS build of csrc/dss2_wgrad16h.hip: DSS2_LIB=<that library>): per-wave phase durations of
wgrad16h_kernel on the third tile of every workgroup's walk (layer 1 of three; s_memtime ticks = 10 ns), C2 by default."""
    sys_argv = [""] + list(argv)
    ops = pkg.ops
    DEV = "cuda:0"; H, nmat, nl = 128, 3, 3
    GRID = sys_argv[1] if len(sys_argv) > 1 else "cigre14"; B = int(sys_argv[2]) if len(sys_argv) > 2 else 4096
    b = pkg.synthetic.make_batch([GRID], B, seed=0)
    ei = b["edge_index"].to(DEV); N = b["x"].shape[0]
    topo = pkg.topology.get_topology(ei, N)
    torch.manual_seed(0)
    Xs = [torch.relu(torch.randn(N, H, device=DEV)) for _ in range(nl)]
    Gs = [torch.randn(N, H, device=DEV) for _ in range(nl)]
    stride = nmat * H * H + H
    out = torch.empty(nl * stride, device=DEV); first = torch.empty(stride + nmat * H, device=DEV)
    big = torch.empty(300 << 20, dtype=torch.uint8, device=DEV)
    for _ in range(5):
        big.fill_(1)      # (cold caches, as inside the step)
        ops.wgrad_batched(topo, Gs, H, Xs, H, nmat, out[:(nl - 1) * stride], first_rowscale2=topo.deg_pows, first_out=first, pending=[])
    torch.cuda.synchronize()
    lib = C.CDLL(pkg._lib.LIB_PATH)
    n = 512 * 4 * 16
    buf = (C.c_ulonglong * n)()
    assert lib.dss2_debug_read_hstamps(buf, n) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(512, 4, 16).astype(np.int64)
    t = t[(t[:, :, 11] > 0).all(axis=1)]
    def show(name, v):
        print(f"{name:58s} mean {v.mean():8.0f}  median {np.median(v):8.0f}  p90 {np.percentile(v, 90):8.0f} ticks")
    print(f"{t.shape[0]} workgroups stamped")
    show("scales (LDS maxima, running exponents)", t[:, :, 1] - t[:, :, 0])
    show("staging: splits of X and G -> planes, fp32 G, ELL", t[:, :, 2] - t[:, :, 1])
    show("barrier", t[:, :, 3] - t[:, :, 2])
    show("next tile's loads issued", t[:, :, 4] - t[:, :, 3])
    show("hop 1 (gathers, fma, split -> planes)", t[:, :, 5] - t[:, :, 4])
    show("barrier", t[:, :, 6] - t[:, :, 5])
    show("hop 2", t[:, :, 7] - t[:, :, 6])
    show("barrier", t[:, :, 8] - t[:, :, 7])
    show("MFMA phase (36 MFMAs = 1152 cycles of matrix pipe)", t[:, :, 9] - t[:, :, 8])
    show("next tile's maxima (waits for its rows)", t[:, :, 10] - t[:, :, 9])
    show("closing barrier", t[:, :, 11] - t[:, :, 10])
    show("tile total", t[:, :, 11] - t[:, :, 0])


COMMANDS = {"wgradh": cmd_wgradh, "gemm": cmd_gemm, "teams": cmd_teams, "chain": cmd_chain, "stack": cmd_stack, "wgrad": cmd_wgrad}

if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] not in COMMANDS:
        print(__doc__)
        sys.exit(2)
    COMMANDS[sys.argv[1]](sys.argv[2:])
