#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its headline configuration (C2).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one resident batch: MPN forward -> gsp_wls_edge -> backward
(/root/reference/dss2_run.py:138-142; optimizer excluded, gradient all-reduce included for N > 1).
Workload at every N (weak scaling): CIGRE-14, B = 4096 graphs per GPU, MPN(8, 6, 2, H=128, L=4, K=2),
dropout 0, fp32, synthetic states on the reference's CIGRE-14 parameter tables, random-init weights.
Rank 0 prints ONE JSON line (contract in the task brief) with two extra objects:
  roofline     : the dominant kernel (fused MFMA GEMM + propagation over chains of hid->hid TAGConv
                 layers), timed in situ with HIP events in an instrumented pass after the timed region
  cpu_baseline : the CPU oracle (a port of the reference's PyTorch-eager path, dead dense Laplacian
                 stripped) timed on this box's host cores on the same batch (N = 1 only)
"""
import argparse
import contextlib
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "deep-statistical-solver-for-distribution-system-state-estimation_amd"

REG = {"mu_v": 1e-1, "mu_theta": 1e-1, "lam_v": 1e-4, "lam_p": 1e-8, "lam_pf": 1e-6, "lam_reg": 1e2}
B_PER_GPU, HID, LAYERS, KHOPS = 4096, 128, 4, 2
FP32_MFMA_PEAK_TF = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: Peak FP32 (matrix)
BF16_MFMA_PEAK_TF = 2500.0     # same table: Peak BF16 MFMA, dense
HBM_PEAK_GBS = 8000.0


# The arithmetic routes of the H -> H layers' products: all accumulate in fp32 and are fp32-accurate against fp64
# (cpu_baseline.accuracy_vs_fp64_oracle); they differ in how a product of two fp32 operands reaches the matrix pipe.
ROUTES = {
    "f16x3": dict(CHAIN_BF16=True, WGRAD_BF16=True, CHAIN_F16=True, WGRAD_F16=True),
    "bf16x6": dict(CHAIN_BF16=True, WGRAD_BF16=True, CHAIN_F16=False, WGRAD_F16=False),
    "fp32-mfma": dict(CHAIN_BF16=False, WGRAD_BF16=False, CHAIN_F16=False, WGRAD_F16=False),
}
DTYPE_BY_ROUTE = {
    "f16x3": "f32 (H->H products as f16x3: two fp16 pieces per operand, 3 fp16 MFMAs per fp32 product, f32 accumulate; edge MLP bf16x6; everything else f32/f64)",
    "bf16x6": "f32 (H->H products as bf16x6: three bf16 pieces per operand, 6 bf16 MFMAs per fp32 product, f32 accumulate)",
    "fp32-mfma": "f32 (H->H products on fp32 MFMAs; edge MLP bf16x6)",
    "mixed": "f32 (mixed f16x3 / bf16x6 / fp32-MFMA routes: see config.arithmetic_route_flags)",
}


def route_name(pkg) -> str:
    f = pkg.flags
    cur = {k: bool(getattr(f, k)) for k in ("CHAIN_BF16", "WGRAD_BF16", "CHAIN_F16", "WGRAD_F16")}
    if not cur["CHAIN_BF16"] and not cur["WGRAD_BF16"]:
        return "fp32-mfma"
    for name, setting in ROUTES.items():
        if cur == setting:
            return name
    return "mixed"


def spawn_ranks(n: int) -> int:
    """One child per GPU via `python -m torch.distributed.run` on a free loopback port; stdout / stderr pass through."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env).returncode


HANG_RC = int(os.environ.get("DSS2_BENCH_HANG_RC", "3"))      # exit status of a run one of whose legs hung


def capture_watchdog(result: dict, rank: int, world: int, timeout: float, leg: str = "hipGraph replay"):
    """Dead-man timer of a replay leg (hipGraph capture / launch plan; every rank runs one).  If the leg does not answer within
    `timeout` seconds -- e.g. a collective that never completes inside a capture -- the process cannot be recovered (a hung HIP call
    holds the GPU), so it ends here, VISIBLY: rank 0 prints the run's FINAL JSON line = everything measured so far (value /
    ms_per_step: the best leg that completed) + `"hang": {"leg", "timeout_s", "exit_status"}` and `legs[leg] = "hung"`, and every rank
    leaves with exit status HANG_RC (3; DSS2_BENCH_HANG_RC) -- the measurement survives on stdout, the launcher reports the failure.
    Never a re-exec of this process (it has initialised the GPU).  Returns the started threading.Timer (cancel() it when the leg
    finishes)."""
    import threading

    def bail():
        msg = f"{leg} leg: no answer within {timeout:.0f} s; the measurements of the legs that completed stand"
        print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)
        if rank == 0:
            result.pop("partial", None)
            result["hang"] = {"leg": leg, "timeout_s": timeout, "exit_status": HANG_RC, "note": msg}
            result.setdefault("legs", {})[leg] = "hung"
            print(json.dumps(result), flush=True)
        sys.stdout.flush()
        os._exit(HANG_RC)
    t = threading.Timer(timeout, bail)
    t.daemon = True
    t.start()
    return t


def dry_run(args) -> int:
    """The launch path without a GPU: gloo process group from the launcher's environment, one all-reduce, one JSON line.
    --simulate-hung-capture: after the eager line, every rank starts a replay leg's watchdog and then blocks forever
    (what a capture with a never-completing collective looks like): the run must end with the hang recorded in its LAST
    JSON line (with the eager measurement in it) and exit status HANG_RC."""
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        print(f"--gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 2
    total = rank + 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        t = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(t)
        total = int(t.item())
        dist.barrier()
    if total != world * (world + 1) // 2:
        return 3
    res = {"dry_run": True, "n_gpus": world, "ranks_seen": total, "steps": args.steps, "warmup": args.warmup}
    if args.simulate_hung_capture:
        res["config"] = {"mode": "eager"}
        res["legs"] = {"eager": "ok"}
        if rank == 0 and world > 1:
            early = dict(res)
            early["partial"] = "eager measurement, printed before the replay legs; a later line supersedes it"
            print(json.dumps(early), flush=True)
        capture_watchdog(res, rank, world, args.graph_timeout, leg="hipGraph replay")
        import threading
        threading.Event().wait()          # never returns: the watchdog ends the process
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(res), flush=True)
    return 0


def flops_per_graph(n, e, hid, layers, khops, fn=8, fe=6, fo=2, blocks=1, fo_inner=None):
    """forward FLOPs of one graph = e2 (2 (2 Fn + Fe) H + 2 H^2) + sum over layers [n (K + 1) 2 H H_out + K e2 2 H];
    forward + backward = 3 x forward.  ``blocks`` > 1: a PFN / SkipPFN stack (inner blocks end at ``fo_inner`` columns)."""
    e2 = 2 * e
    total = 0.0
    for b in range(blocks):
        out = fo if b == blocks - 1 else (fo_inner if fo_inner is not None else fo)
        fin = fn if b == 0 else (fo_inner if fo_inner is not None else fo)
        f = e2 * (2.0 * (2 * fin + fe) * hid + 2.0 * hid * hid)
        for l in range(layers):
            hout = out if l == layers - 1 else hid
            f += n * (khops + 1) * 2.0 * hid * hout + khops * e2 * 2.0 * hid
        total += f
    return 3.0 * total


# The other BASELINE.json configurations and the reference driver's own model line, timed on ONE GPU beside the headline
# ("make the driver see more than C2").  (tag, grids, graphs, class, ctor args, (n, e) per graph, blocks)
OTHER_CONFIGS = [
    ("C1 cigre14 B=64 H=32 L=1", ["cigre14"], 64, "MPN", (8, 6, 2, 32, 1, 2, 0.0), (15, 14), 1),
    ("C3 ober_sub B=1024 H=128 L=4", ["ober_sub"], 1024, "MPN", (8, 6, 2, 128, 4, 2, 0.0), (70, 69), 1),
    ("C3' ober179 (synthetic 179-bus feeder) B=1024 H=128 L=4", ["ober179"], 1024, "MPN", (8, 6, 2, 128, 4, 2, 0.0), (179, 178), 1),
    ("C5 model, 4096-graph mixed-topology shard, H=256 L=8", ["cigre14", "cigre14_reswitched"], 4096, "MPN", (8, 6, 2, 256, 8, 2, 0.0), (15, 14.5), 1),
    ("C2 shape, B=32768 on one GPU (cache-busting)", ["cigre14"], 32768, "MPN", (8, 6, 2, 128, 4, 2, 0.0), (15, 14), 1),
    ("driver line: SkipPFN H=32 8 layers x 5 blocks, dropout 0.3, B=4096", ["cigre14"], 4096, "SkipPFN", (8, 6, 2, 32, 8, 2, 0.3, 5), (15, 14), 5),
    # the reference driver's other branch (dss2_run.py:51-53: Oberrhein; 70 buses => 96-row tiles).  The whole-stack kernels cover <= 64-row
    # tiles only (their backward takes 158 of the CU's 160 KB of LDS at 64 rows; 96 rows would need ~226 KB: DESIGN section 9), so these run the per-block
    # kernels -- timed here so that the gap is a number
    ("driver line on ober_sub: SkipPFN H=32 8 layers x 5 blocks, dropout 0.3, B=64", ["ober_sub"], 64, "SkipPFN", (8, 6, 2, 32, 8, 2, 0.3, 5), (70, 69), 5),
    ("driver line on ober_sub: SkipPFN H=32 8 layers x 5 blocks, dropout 0.3, B=1024", ["ober_sub"], 1024, "SkipPFN", (8, 6, 2, 32, 8, 2, 0.3, 5), (70, 69), 5),
]


def time_other_config(pkg, dev, stream, tag, grids, B, cls, cargs, ne, blocks, seconds):
    """ms/step of forward + gsp_wls_edge + backward on a resident batch, eager and as a replayed hipGraph, each over
    >= `seconds` of timed work; fractions from SURVEY 8(d)'s FLOP count."""
    torch.manual_seed(0)
    b = pkg.synthetic.make_batch(grids, B, seed=1)
    x, ei, ea = b["x"].to(dev), b["edge_index"].to(dev), b["edge_attr"].to(dev)
    st = tuple(s.to(dev) for s in b["stats"])
    model = getattr(pkg, cls)(*cargs).to(dev)
    params = list(model.parameters())
    xin, ein, npar, epar = x[:, :8], ea[:, :6], x[:, 8:], ea[:, 6:]

    def step():
        for p in params:
            p.grad = None
        out = model(xin, ei, ein)
        loss = pkg.gsp_wls_edge(input=xin, edge_input=ein, output=out, x_mean=st[0], x_std=st[1], edge_mean=st[2], edge_std=st[3],
                                edge_index=ei, reg_coefs=REG, num_samples=None, node_param=npar, edge_param=epar)
        loss.backward(pkg.data.unit_grad(loss))
        return loss

    def timed(run):
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        n, t0 = 0, time.perf_counter()
        while True:
            for _ in range(10):
                run()
            torch.cuda.synchronize()
            n += 10
            el = time.perf_counter() - t0
            if el >= seconds or n >= 20000:
                return el / n * 1e3

    ms_eager = timed(step)
    rec = {"graphs": B, "ms_per_step_eager": ms_eager}
    g = None          # (a failed capture must not lose the eager measurement: `del g` below)
    try:
        g = pkg.graphs.GraphedStep(step, stream=stream, capture_error_mode="thread_local")
        rec["ms_per_step_replay"] = timed(g.replay)
    except Exception as exc:
        rec["ms_per_step_replay"] = None
        rec["replay_error"] = f"{type(exc).__name__}: {exc}"[:200]
    # ... and as a launch plan (graphs.PlannedStep: the library's own record of the step's launches, re-issued from ONE C call): what an
    # eager step costs without the Python around every launch -- for steps that cannot be captured into a hipGraph
    if B <= 4096:
        try:
            pl = pkg.graphs.PlannedStep(step, stream=stream, verify=(None if cargs[6] > 0 else (lambda: [p.grad for p in params])))
            rec["ms_per_step_plan"], rec["plan_launches"] = timed(pl.replay), pl.n_launches
            del pl
        except Exception as exc:
            rec["ms_per_step_plan"], rec["plan_error"] = None, f"{type(exc).__name__}: {exc}"[:200]
    best = min(v for v in (rec["ms_per_step_eager"], rec["ms_per_step_replay"], rec.get("ms_per_step_plan")) if v is not None)
    hid, layers, khops = cargs[3], cargs[4], cargs[5]
    ne = (x.shape[0] / float(B), ei.shape[1] / float(B))      # nodes / stored edges per graph of THIS batch (mixed topologies: the mean)
    fpg = flops_per_graph(ne[0], ne[1], hid, layers, khops, blocks=blocks, fo_inner=(8 if blocks > 1 else None))
    tf = fpg * B / (best * 1e-3) / 1e12
    topo = pkg.topology.get_topology(ei, x.shape[0])
    rec.update(nodes_per_graph=ne[0], edges_per_graph=ne[1], ms_per_step=best, graphs_per_s=B / (best * 1e-3), flops_per_graph_fwd_bwd=fpg, tflops_fp32_equiv=tf,
               frac_of_fp32_mfma_peak=tf / FP32_MFMA_PEAK_TF, frac_of_bf16_pipe_div6=tf / (BF16_MFMA_PEAK_TF / 6.0),
               tile_rows=32 * int(topo.nrb), tile_utilisation=float(topo.utilisation), loss=float(step().item()))
    del g, model, x, ei, ea
    torch.cuda.empty_cache()
    return rec


# End-to-end training epochs (the reference's loop is dss2_run.py:131-147 with the loader of :68-69): device-resident dataset ->
# shuffle on the device -> collation straight into the recorded step's static buffers -> forward + gsp_wls_edge + backward -> fused Adamax,
# an epoch = N replays of one recorded step (runner.EpochTrainer).  (tag, grid, graphs per batch, class, ctor args, batches per epoch)
TRAIN_EPOCHS = [
    ("train epoch C2: cigre14 B=4096 H=128 L=4, 64 batches/epoch", "cigre14", 4096, "MPN", (8, 6, 2, 128, 4, 2, 0.0), 64),
    ("train epoch SkipPFN driver line: cigre14 B=64 H=32 8 layers x 5 blocks dropout 0.3, 64 batches/epoch", "cigre14", 64, "SkipPFN", (8, 6, 2, 32, 8, 2, 0.3, 5), 64),
]


def time_train_epoch(pkg, dev, stream, tag, grid, B, cls, cargs, n_batches, epochs=3):
    """ms per training step END TO END over `epochs` shuffled epochs of `n_batches` batches, beside the same training step (with the optimizer)
    replayed on ONE resident batch, and the host's wall time per batch (how long the loop needs to hand a batch to the GPU)."""
    torch.manual_seed(0)
    base = pkg.synthetic.make_batch([grid], min(B, 4096), seed=7)
    ds0 = pkg.dataset.DeviceDataset.from_batch(base, device=dev)
    reps = -(-B * n_batches // ds0.S)
    ds = pkg.dataset.DeviceDataset(ds0.x.repeat(reps, 1, 1)[:B * n_batches].contiguous(), ds0.edge_attr.repeat(reps, 1, 1)[:B * n_batches].contiguous(),
                                   ds0.y.repeat(reps, 1, 1)[:B * n_batches].contiguous(), ds0.edge_index.repeat(reps, 1, 1)[:B * n_batches].contiguous())
    st = tuple(t_.to(dev) for t_ in base["stats"])
    rec = {"graphs_per_batch": B, "batches_per_epoch": n_batches, "epochs_timed": epochs, "samples_resident": ds.S}
    for mode in ("plan", "graph"):
        model = getattr(pkg, cls)(*cargs).to(dev)
        opt = pkg.optim.FusedAdamax(model.parameters(), lr=3e-3, capturable=True)
        try:
            tr = pkg.runner.EpochTrainer(model, opt, st, REG, ds, B, shuffle=True, mode=mode)
            tr.train_epoch()                                 # warm-up epoch
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(epochs):
                tr.train_epoch()
            t_host = time.perf_counter() - t0                # the host has handed over every batch of every epoch
            loss = tr.mean_loss()                            # (the epochs' only synchronisation)
            t_all = time.perf_counter() - t0
            rec[f"ms_per_step_end_to_end_{mode}"] = t_all / (epochs * n_batches) * 1e3
            rec[f"loader_and_launch_host_us_per_batch_{mode}"] = t_host / (epochs * n_batches) * 1e6
            rec[f"launches_per_step_{mode}"] = getattr(tr.steps[B][0], "n_launches", None)
            rec["last_epoch_mean_loss"] = loss
            # the same recorded step WITHOUT the loader in front of it: one resident batch, replayed
            sx = tr.steps[B][1]
            del tr
            m2 = getattr(pkg, cls)(*cargs).to(dev)
            o2 = pkg.optim.FusedAdamax(m2.parameters(), lr=3e-3, capturable=True)
            ei, _ = ds.batch_structure(B)
            bt = ds.collate(ds.ids[:B].contiguous())
            p2 = list(m2.parameters())

            def resident():
                for p in p2:
                    p.grad = None
                out = m2(bt.x[:, :8], ei, bt.edge_attr[:, :6])
                l_ = pkg.gsp_wls_edge(input=bt.x[:, :8], edge_input=bt.edge_attr[:, :6], output=out, x_mean=st[0], x_std=st[1], edge_mean=st[2],
                                      edge_std=st[3], edge_index=ei, reg_coefs=REG, num_samples=None, node_param=bt.x[:, 8:], edge_param=bt.edge_attr[:, 6:])
                l_.backward(pkg.data.unit_grad(l_))
                o2.step()
                return l_
            g = pkg.graphs.PlannedStep(resident, stream=stream) if mode == "plan" else pkg.graphs.GraphedStep(resident, stream=stream)
            for _ in range(10):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(epochs * n_batches):
                g.replay()
            torch.cuda.synchronize()
            rec[f"ms_per_step_resident_batch_{mode}"] = (time.perf_counter() - t0) / (epochs * n_batches) * 1e3
            del g, m2, o2, sx
        except Exception as exc:
            rec[f"error_{mode}"] = f"{type(exc).__name__}: {exc}"[:300]
        torch.cuda.empty_cache()
    e2e = [rec[k] for k in ("ms_per_step_end_to_end_plan", "ms_per_step_end_to_end_graph") if k in rec]
    res = [rec[k] for k in ("ms_per_step_resident_batch_plan", "ms_per_step_resident_batch_graph") if k in rec]
    if e2e and res:
        rec.update(ms_per_step=min(e2e), ms_per_step_resident_batch=min(res), end_to_end_over_resident=min(e2e) / min(res),
                   graphs_per_s=B / (min(e2e) * 1e-3))
    rec["what"] = ("shuffled epochs from a device-resident dataset: permutation drawn on the device per epoch; per step ONE C call (plan) / one "
                   "hipGraph launch (graph) whose first launch gathers the batch into the step's static buffers (dss2_collate_cursor) and whose last "
                   "ones are the fused Adamax and the loss accumulation; one host synchronisation per timed run")
    return rec


def time_c5_fresh_batches(pkg, dev, seconds=1.5, B=4096, S=8192):
    """BASELINE config C5 as it words it ("variable edge_index per sample"): a NEW Bernoulli(0.5) mix of cigre14 / cigre14_reswitched graphs
    every step -- ragged collation + the batch's graph structure (CSR, tiles, ELL) built on the device per step.  ms/step on a resident batch,
    with a fresh batch per step assembled in line (dataset.DataLoader), and assembled one batch ahead on a side stream (dataset.PrefetchLoader)."""
    import numpy as np
    full = pkg.synthetic.make_batch(["cigre14", "cigre14_reswitched"], 256, seed=1)
    parts = [pkg.dataset.DeviceDataset.from_batch(pkg.synthetic.make_batch([g], S, seed=2 + k, stats=full["stats"]), device=dev)
             for k, g in enumerate(["cigre14", "cigre14_reswitched"])]
    ds = pkg.dataset.MixedDataset(parts)
    st = tuple(s_.to(dev) for s_ in full["stats"])
    model = pkg.MPN(8, 6, 2, 256, 8, 2, 0.0).to(dev)
    params = list(model.parameters())

    def step(bt):
        for p in params:
            p.grad = None
        out = model(bt.x[:, :8], bt.edge_index, bt.edge_attr[:, :6])
        loss = pkg.gsp_wls_edge(input=bt.x[:, :8], edge_input=bt.edge_attr[:, :6], output=out, x_mean=st[0], x_std=st[1], edge_mean=st[2],
                                edge_std=st[3], edge_index=bt.edge_index, reg_coefs=REG, num_samples=None, node_param=bt.x[:, 8:],
                                edge_param=bt.edge_attr[:, 6:])
        loss.backward(pkg.data.unit_grad(loss))
        return loss

    def run(loader_factory):
        n, t0 = 0, None
        while True:
            for bt in loader_factory():
                step(bt)
                n += 1
                if t0 is None and n == 4:          # (warm-up batches)
                    torch.cuda.synchronize()
                    t0, n0 = time.perf_counter(), n
            torch.cuda.synchronize()
            if time.perf_counter() - t0 >= seconds:
                return (time.perf_counter() - t0) / (n - n0) * 1e3
    gen = torch.Generator()
    gen.manual_seed(0)
    plain = lambda: pkg.dataset.DataLoader(ds, batch_size=B, shuffle=True, generator=gen)
    bt0 = next(iter(plain()))
    t_res = run(lambda: [bt0] * 8)
    t_inline = run(plain)
    t_pref = run(lambda: pkg.dataset.PrefetchLoader(plain()))
    return {"graphs": B, "ms_per_step_resident_batch": t_res, "ms_per_step_fresh_batch_inline": t_inline, "ms_per_step_fresh_batch_prefetched": t_pref,
            "fresh_batch_overhead_inline": t_inline / t_res - 1.0, "fresh_batch_overhead_prefetched": t_pref / t_res - 1.0,
            "ms_per_step": min(t_inline, t_pref), "graphs_per_s": B / (min(t_inline, t_pref) * 1e-3),
            "what": "MPN H=256 L=8, eager steps; every step a new mix of the two topologies: ragged collation + device CSR / tile / ELL build, "
                    "in line on the step's stream vs one batch ahead on a side stream (PrefetchLoader).  The step's big kernels are persistent and take "
                    "every CU's LDS and registers, so the side stream's ~25 small dependent launches only find room at kernel boundaries: the "
                    "prefetch cannot hide the assembly here (DESIGN section 9); ms_per_step = the better of the two"}


def route_accuracy(pkg, oracle, dev):
    """Errors of the f16x3 / bf16x6 / fp32-MFMA routes against the fp64 CPU oracle (same weights, same 256-graph CIGRE batch, the C2 model):
    {route: {"output", "loss", "worst_gradient"}} -- the oracle is the checker here, nothing of it is timed."""
    torch.manual_seed(0)
    b = pkg.synthetic.make_batch(["cigre14"], 256, seed=3)
    ref = oracle.MPN(8, 6, 2, HID, LAYERS, KHOPS, 0.0).double()
    b64 = {"x": b["x"].double(), "edge_index": b["edge_index"], "edge_attr": b["edge_attr"].double()}
    threads = torch.get_num_threads()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    out64, l64 = oracle.train_step(ref, b64, tuple(t.double() for t in b["stats"]), REG)
    torch.set_num_threads(threads)
    x, ei, ea = b["x"].to(dev), b["edge_index"].to(dev), b["edge_attr"].to(dev)
    st = tuple(t.to(dev) for t in b["stats"])
    rel = lambda a, r: float((a.detach().double().cpu() - r.detach().double()).abs().max() / r.detach().double().abs().max())
    saved = {k: getattr(pkg.flags, k) for k in ("CHAIN_BF16", "WGRAD_BF16", "CHAIN_F16", "WGRAD_F16")}
    res = {}
    try:
        for rname, setting in ROUTES.items():
            for k, v in setting.items():
                setattr(pkg.flags, k, v)
            mine = pkg.MPN(8, 6, 2, HID, LAYERS, KHOPS, 0.0)
            mine.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
            mine = mine.to(dev)
            out = mine(x[:, :8], ei, ea[:, :6])
            loss = pkg.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=out, x_mean=st[0], x_std=st[1], edge_mean=st[2],
                                    edge_std=st[3], edge_index=ei, reg_coefs=REG, num_samples=None, node_param=x[:, 8:], edge_param=ea[:, 6:])
            loss.backward()
            torch.cuda.synchronize()
            res[rname] = {"output": rel(out, out64), "loss": abs(loss.item() - l64.item()) / abs(l64.item()),
                          "worst_gradient": max(rel(p.grad, q.grad) for p, q in zip(mine.parameters(), ref.parameters()))}
    finally:
        for k, v in saved.items():
            setattr(pkg.flags, k, v)
    res["note"] = "max-normalised errors against the fp64 CPU oracle, C2 model on a 256-graph CIGRE-14 batch, un-pinned ReLU gates; the parity bar is 1e-5"
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=B_PER_GPU, help="graphs per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="time the eager step only (default: eager and hipGraph replay, best reported)")
    ap.add_argument("--graph-timeout", type=float, default=90.0, help="seconds the hipGraph leg may take before the eager result is printed")
    ap.add_argument("--cpu-seconds", type=float, default=3.0, help="CPU time budget per thread count of the cpu_baseline thread sweep")
    ap.add_argument("--cpu-protocol-steps", type=int, default=50, help="timed steps of the cpu_baseline's protocol leg at the best thread count (10 warm-ups before them)")
    ap.add_argument("--no-routes", action="store_true", help="skip timing the step on the bf16x6 and fp32-MFMA routes beside the default one")
    ap.add_argument("--route-seconds", type=float, default=1.5, help="timed work per extra arithmetic route")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the other_configs block (C1, C3, C3', C5 shard, B=32768, driver line)")
    ap.add_argument("--no-train-epochs", action="store_true", help="skip the end-to-end training epochs in other_configs")
    ap.add_argument("--other-seconds", type=float, default=1.0, help="timed work per mode of every other_configs entry")
    ap.add_argument("--min-window-seconds", type=float, default=5.0,
                    help="repeat the K-step timed window until this much timed work has accumulated; the median window is "
                         "reported (5 s per mode: the eager and the replay leg together keep the GPU busy >= 10 s in a row, so "
                         "that an outside utilisation sampler sees the run)")
    ap.add_argument("--ramp-seconds", type=float, default=2.0,
                    help="untimed load before the W warm-up steps: a fresh MI355X needs ~1 s of sustained work to reach "
                         "its steady clocks (measured: the first ~100 steps of a process run 40 %% slower)")
    ap.add_argument("--simulate-hung-capture", action="store_true", help="with --dry-run: block after the eager line as a hung capture would (tests/test_bench_launch_cpu.py)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch / rendezvous path only (gloo on CPU, no GPU call): every rank joins the process group, one "
                         "all-reduce checks it, rank 0 prints a JSON line marked dry_run (tests/test_bench_launch_cpu.py)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python3 bench.py --gpus N` without a launcher: start one fresh child process per GPU through torch's own
        # launcher BEFORE this process has touched the GPU (nothing above initialises HIP), relay what the ranks print
        # and leave with their exit code.  A process that has initialised the GPU is never re-executed.
        raise SystemExit(spawn_ranks(args.gpus))
    if args.dry_run:
        raise SystemExit(dry_run(args))

    pkg = importlib.import_module(PKG)
    pkg._lib.lib()  # fail loudly without the HIP extension
    import torch.distributed as dist
    env = pkg.parallel.init_from_env("nccl")
    rank, world, local = env["rank"], env["world"], env["local"]
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N > 1 as `python -m torch.distributed.run "
                         f"--nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...`")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    distributed = dist.is_initialized()          # true under torch.distributed.run, also at world size 1
    group = dist.group.WORLD if distributed else None

    # ---- resident inputs and model
    torch.manual_seed(0)
    batch = pkg.synthetic.make_batch(["cigre14"], args.batch, seed=1000 + rank)
    x, ei, ea = batch["x"].to(dev), batch["edge_index"].to(dev), batch["edge_attr"].to(dev)
    stats = tuple(s.to(dev) for s in batch["stats"])
    model = pkg.MPN(8, 6, 2, HID, LAYERS, KHOPS, 0.0).to(dev)
    ranks_seen = 1
    if distributed:
        # every rank adds one over RCCL: the JSON line then SHOWS that `world` ranks took part in the collectives of this run
        seen = torch.ones(1, dtype=torch.int32, device=dev)
        dist.all_reduce(seen, op=dist.ReduceOp.SUM, group=group)
        ranks_seen = int(seen.item())
        if ranks_seen != world:
            raise SystemExit(f"RCCL all-reduce of ones over the process group returned {ranks_seen}, expected {world}")
        pkg.parallel.broadcast_parameters(model, 0, group)
        pkg.parallel.attach_grad_allreduce(model, group)
    xin, ein, npar, epar = x[:, :8], ea[:, :6], x[:, 8:], ea[:, 6:]
    # every step of this process runs on ONE side stream: the step can then be captured into a hipGraph later on the stream
    # its autograd state (AccumulateGrad nodes) already belongs to
    work_stream = torch.cuda.Stream()
    work_stream.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(work_stream)
    params = list(model.parameters())        # (walking the module tree every step costs the host ~70 us of a 0.58 ms step)

    def step():
        for p in params:
            p.grad = None
        out = model(xin, ei, ein)
        loss = pkg.gsp_wls_edge(input=xin, edge_input=ein, output=out, x_mean=stats[0], x_std=stats[1],
                                edge_mean=stats[2], edge_std=stats[3], edge_index=ei, reg_coefs=REG,
                                num_samples=None, node_param=npar, edge_param=epar, group=group)
        loss.backward(pkg.data.unit_grad(loss))      # (= loss.backward(); the root gradient is a cached scalar 1 instead of a fill kernel per step)
        return loss

    def sync():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def ramp(run):          # clock ramp (see --ramp-seconds), untimed
        n, t_ramp = 0, time.perf_counter() + args.ramp_seconds
        while args.ramp_seconds > 0:
            for _ in range(20):
                run()
            torch.cuda.synchronize()
            n += 20
            go = time.perf_counter() < t_ramp
            if distributed:      # the steps contain collectives: every rank must run the same number of them
                flag = torch.tensor([int(go)], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                go = bool(flag.item())
            if not go:
                break
        return n

    def timed_windows(run):
        """EXACTLY K steps between barrier + synchronize on both sides.  K steps of this workload can be as short as 12 ms,
        a fragile sample, so the K-step window is repeated until >= --min-window-seconds of timed work has accumulated and
        the MEDIAN window is reported (every window is bracketed the same way; its duration is the max over ranks)."""
        out, last = [], None
        while True:
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                last = run()
            sync()
            dtw = time.perf_counter() - t0
            if distributed:
                tt = torch.tensor([dtw], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dtw = float(tt.item())
            out.append(dtw)
            go = sum(out) < args.min_window_seconds and len(out) < 2000
            if distributed:
                flag = torch.tensor([int(go)], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                go = bool(flag.item())
            if not go:
                break
        return out, last

    ramp_steps = ramp(step)
    for _ in range(args.warmup):
        step()
    windows, loss = timed_windows(step)
    mode = "eager"
    dt = sorted(windows)[len(windows) // 2]
    ms = dt / args.steps * 1e3
    value = args.batch * world / (dt / args.steps)

    result = {
        "metric": "grid-samples/sec fwd+bwd (WLS loss), CIGRE-14 batch=4096, 1/2/4/8 GPU", "value": value, "unit": "graphs/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": DTYPE_BY_ROUTE[route_name(pkg)], "data": "synthetic",
        "config": {"workload": f"CIGRE-14 (15 buses, 14 closed branches) B={args.batch} graphs/GPU, "
                               f"MPN L={LAYERS} H={HID} K={KHOPS} dropout=0: forward + gsp_wls_edge + backward"
                               + (" + RCCL loss-sum and gradient all-reduce" if distributed else ""),
                   "graphs_per_gpu": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}",
                   "loss": float(loss.item()), "clock_ramp_steps_before_warmup": ramp_steps,
                   "timed_windows": len(windows), "window_ms_min_median_max": [min(windows) * 1e3, dt * 1e3, max(windows) * 1e3],
                   "mode": mode, "ms_per_step_by_mode": {"eager": ms}, "ranks_seen": ranks_seen,
                   "ranks_seen_source": ("SUM all-reduce of a ones tensor over RCCL at start-up" if distributed else "single process, no process group"),
                   "arithmetic_route": route_name(pkg)},
        "legs": {"eager": "ok"},
    }

    # ---- instrumented pass: EVERY rank runs it (the steps contain collectives); rank 0 reports
    # ---- roofline of the dominant kernel, timed in situ (instrumented pass, not the timed region).
    # The H -> H TAGConv layers (forward) and their data-gradients (backward) run as layer chains
    # (gemm_chain_kernel: one launch per chain, activation tile resident in LDS); FLOPs per launch =
    # layers in the chain x the per-layer GEMM + propagation FLOPs.
    N, E2 = x.shape[0], 2 * ei.shape[1]
    events = []
    nw_mod = pkg.networks
    orig_chain, orig_single = nw_mod.gemm_prop_chain, nw_mod.gemm_prop

    by_dir = {}                 # average launch time (us) of the forward chain / of the data-gradient chain
    head_launches = [0, 0]      # chained launches that carry the narrow head (forward: its TAGConv; backward: its data gradient), all launches

    def timed_chain(topo, X, hid, nmat, layers, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig_chain(topo, X, hid, nmat, layers, **kw)
        e1.record()
        events.append((e0, e1, len(layers), bool(kw.get("transposed", False))))
        head_launches[0] += int(kw.get("head") is not None)
        head_launches[1] += 1

    def timed_single(topo, X, ldx, kreal, Bp, nmat, hout, Y, **kw):   # DSS2_CHAIN=0: one launch per layer
        dominant = (kreal == HID and hout == HID and nmat == KHOPS + 1)
        if dominant:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        orig_single(topo, X, ldx, kreal, Bp, nmat, hout, Y, **kw)
        if dominant:
            e1.record()
            events.append((e0, e1, 1, bool(kw.get("transposed", False))))

    wg_events = []
    wg_topo = [None]
    orig_wgb = nw_mod.wgrad_batched

    def timed_wgrad_batched(topo, Gs, hout, Xs, hin, nmat, out_flat, **kw):   # the block's H -> H weight gradients (one launch)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig_wgb(topo, Gs, hout, Xs, hin, nmat, out_flat, **kw)
        e1.record()
        wg_events.append((e0, e1, len(Gs), hout, hin, nmat))
        wg_topo[0] = topo

    def timed_pass(n_steps):
        events.clear()
        wg_events.clear()
        nw_mod.gemm_prop_chain, nw_mod.gemm_prop, nw_mod.wgrad_batched = timed_chain, timed_single, timed_wgrad_batched
        for _ in range(n_steps):
            # the GPU waits ~0.8 ms (a spin kernel) while the host enqueues the whole instrumented step behind it: the step's launches then
            # run back to back and the event pairs bracket KERNEL time.  Without it the eager step is host-bound by now (0.39 ms of GPU work
            # under ~0.4 ms of Python), the GPU idles between an event record and the launch that follows it, and the "launch duration"
            # read 101.7 us where rocprofv3's kernel average says 82-86.
            torch.cuda._sleep(1_600_000)
            step()
        torch.cuda.synchronize()
        nw_mod.gemm_prop_chain, nw_mod.gemm_prop, nw_mod.wgrad_batched = orig_chain, orig_single, orig_wgb
        durs = sorted(a.elapsed_time(b) for a, b, _, _ in events)
        layers = sum(n for _, _, n, _ in events)
        by_dir.clear()
        for tr in (False, True):
            d = [a.elapsed_time(b) for a, b, _, t in events if t == tr]
            if d:
                by_dir["data-gradient chain" if tr else "forward chain"] = sum(d) / len(d) * 1e3
        return sum(durs) / len(durs), durs[len(durs) // 2], len(durs), layers / len(durs)

    # the clock the chip HOLDS under the dominant kernel: the split-plane chain leaves {s_memtime, s_memrealtime} at the
    # start and the end of four of its workgroups (dss2_debug_chain_clock_probe; off outside this pass): shader cycles per 100 MHz tick
    probe = torch.zeros(16, dtype=torch.int64, device=dev)
    pkg._lib.lib().dss2_debug_chain_clock_probe(probe.data_ptr())
    try:
        avg_ms, med_ms, n_l, layers_per_launch = timed_pass(min(args.steps, 20))
    finally:
        pkg._lib.lib().dss2_debug_chain_clock_probe(None)
    pr = probe.cpu().view(4, 4).tolist()
    clocks = [(q[2] - q[0]) / float(q[3] - q[1]) * 0.1 for q in pr if q[3] > q[1] and q[2] > q[0]]
    held_clock_ghz = (sum(clocks) / len(clocks)) if clocks else None


    if rank == 0:
        flops_layer = 2.0 * N * HID * (KHOPS + 1) * HID + 2.0 * KHOPS * E2 * HID
        bytes_layer = 4.0 * N * HID + 4.0 * (KHOPS + 1) * HID * HID          # write the output once + the weights
        chained = layers_per_launch > 1
        bf16x6 = bool(chained and pkg.flags.CHAIN_BF16 and pkg.networks.chain16_supported(
            pkg.topology.get_topology(ei, N), KHOPS + 1, HID, False))
        kname = ("gemm_chain_kernel<2,3,4,1,true>" if bf16x6 else "gemm_chain_kernel<2,3>") if chained else "gemm_prop_kernel<2,3,false>"
        topo_ = pkg.topology.get_topology(ei, N)
        if bf16x6 and os.environ.get("DSS2_CHAIN_SP", "1") != "0" and pkg._lib.lib().dss2_gemm_prop_chain_head_supported(
                topo_.nrb, KHOPS + 1, HID, HID, topo_.ell, 2):
            # 64-row tiles, one wave per column group: the split-plane form (csrc/dss2_gemm_chain_sp.hip); the narrow head rides in the
            # same launches (forward: <3,4,1>, data gradients: <3,4,2>) -- its time is inside the launch durations below and its FLOPs
            # (0.095 GFLOP per launch) are counted in `achieved`
            kname = "gemm_chain_sp_kernel<3,4>"
        traffic, traffic_source = None, None
        try:   # HBM bytes per launch from the committed PMC runs (FETCH_SIZE x2 on gfx950 + WRITE_SIZE): a pointer to
               # the rocprofv3 --pmc evidence under profiles/, NOT a counter read during this run
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
                rec = json.load(fh)[kname]
            traffic, traffic_source = rec["hbm_bytes_per_launch"], rec.get("source", "profiles/pmc_traffic.json")
        except Exception:
            pass
        # the narrow head TAGConv (H -> 2) rides inside the chained launches (forward: after the last layer, DSS2_CHAIN_HEAD_FWD; backward:
        # its data gradient in the staging, DSS2_CHAIN_HEAD): its work is part of what the launch does, so its algorithmic FLOPs
        # (2 N H (K+1) nout + 2 K E2 nout = 0.095 GFLOP, 0.5 % of the launch) are part of `achieved`
        nout = 2
        head_frac = head_launches[0] / max(head_launches[1], 1)
        head_flops = (2.0 * N * HID * (KHOPS + 1) * nout + 2.0 * KHOPS * E2 * nout) * head_frac
        flops = flops_layer * layers_per_launch + head_flops
        # in situ, exactly as the timed region runs (single stream): the average
        # agrees with `rocprofv3 --kernel-trace --stats` of this same command (profiles/)
        result["roofline"] = {
            "kernel": f"dss2::{kname} (TAGConv H->H layers, forward and data-gradient"
                      + (f", {layers_per_launch:.0f} layers chained per launch)" if chained else ")"),
            "bound": "mfma", "achieved": flops / (avg_ms * 1e-3) / 1e12, "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
            "frac": flops / (avg_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF, "peak_pipe": "fp32 MFMA", "traffic": traffic,
            "traffic_source": traffic_source,
            "launches_timed": n_l, "layers_per_launch": layers_per_launch, "avg_launch_us": avg_ms * 1e3,
            "median_launch_us": med_ms * 1e3, "algorithmic_flops_per_launch": flops,
            "head_flops_per_launch_included": head_flops, "launches_with_fused_head": head_frac,
            "avg_launch_us_by_direction": dict(by_dir),
            "algorithmic_bytes_per_launch": 4.0 * N * HID + bytes_layer * layers_per_launch,
            "mode": "single stream",
        }
        if wg_events:       # the second-largest kernel of the step, same reading: algorithmic FLOPs / in-situ launch time
            nl, ho, hi, nm = wg_events[0][2:]
            wflops = nl * 2.0 * N * nm * ho * hi
            wus = sum(a.elapsed_time(b) for a, b, *_ in wg_events) / len(wg_events) * 1e3
            wbf16 = bool(pkg.flags.WGRAD_BF16)
            # which kernel took the launch: f16x3 (two fp16 pieces per operand, THREE matrix instructions per fp32 product; 32-row tiles) or
            # bf16x6 (three bf16 pieces, SIX).  The bound is the pipe's dense peak (2500 TF for fp16 and bf16 alike) over that count, so a
            # kernel that needs half the instructions has twice the bound: `frac` says how busy the pipe is, ms/step what the work costs.
            wts = pkg.ops._wgrad_tiles(wg_topo[0], nm, ho, hi, int(wbf16))
            wf16 = bool(wbf16 and (pkg.ops._wgrad_mode(wts, nm, 1) & 255) == 2)
            wprod = 3.0 if wf16 else 6.0
            wpeak = BF16_MFMA_PEAK_TF / wprod if wbf16 else FP32_MFMA_PEAK_TF
            wname = ("(dss2::wgrad16h_kernel, f16x3)" if wf16 else "(dss2::wgrad16b_kernel / wgrad16_kernel, bf16x6)") if wbf16 else "(dss2::wgrad_kernel, fp32 MFMA)"
            result["roofline_wgrad"] = {
                "kernel": f"the block's {nl} H->H weight gradients dW_k = ((A^T)^k g)^T h in one launch " + wname,
                "bound": "mfma", "achieved": wflops / (wus * 1e-6) / 1e12, "peak": wpeak, "unit": "TFLOP/s",
                "frac": wflops / (wus * 1e-6) / 1e12 / wpeak, "avg_launch_us": wus, "launches_timed": len(wg_events),
                "algorithmic_flops_per_launch": wflops, "traffic": None,
                # truly algorithmic: every layer's G and X read ONCE (+ the weight gradients written once); what the kernel's design moves
                # on top of that (X by both output-half workgroups, slabs written and re-read) is waste and shows in `traffic`
                "algorithmic_bytes_per_launch": nl * (4.0 * N * ho + 4.0 * N * hi) + nl * nm * 4.0 * ho * hi,
                "peak_pipe": (f"fp16 / bf16 MFMA dense peak (2500 TF) / {int(wprod)} instructions per fp32 product" if wbf16 else "fp32 MFMA"),
            }
            try:   # (a pointer to the committed rocprofv3 --pmc passes, as for the chain above; the f16x3 kernel of the C2 shape only)
                if wf16 and nl == 3:
                    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
                        wrec = json.load(fh)["wgrad16h_kernel<3,true> (3 layers in one launch)"]
                    result["roofline_wgrad"].update(traffic=wrec["hbm_bytes_per_launch"], traffic_source=wrec.get("source"),
                                                    bytes_this_kernel_design_moves=wrec.get("design_bytes_per_launch"))
            except Exception:
                pass
        if bf16x6:
            # The tile GEMM runs on the 16-bit matrix pipe: as f16x3 (round 5: operands as TWO fp16 pieces after an exact power-of-two
            # scale, THREE v_mfma_f32_16x16x32_f16 per fp32 product group; csrc/dss2_gemm_chain_sp.hip MS = 2) where the chain has the
            # form, else as bf16x6 (three bf16 pieces, SIX instructions).  fp32 accumulation, errors of fp32 arithmetic's own size
            # (tests/test_gpu_f16x3.py, tools/accuracy_bf16x6.py).  The bound of the pipe the kernel executes on is its dense peak
            # (2500 TF for fp16 and bf16 alike) / executed flops per algorithmic flop: THAT is `peak` and `frac` -- a kernel that needs
            # half the instructions has twice the bound; what the work costs is ms_per_step.  The ratio to the fp32 MFMA peak (which
            # this kernel exceeds) is kept as a secondary field only.
            f16x3 = bool(pkg.flags.CHAIN_F16 and pkg.ops.chain_f16_supported(topo_, KHOPS + 1, HID) and kname.startswith("gemm_chain_sp"))
            nprod = 3.0 if f16x3 else 6.0
            r = result["roofline"]
            r["frac_of_fp32_mfma_peak"] = r["achieved"] / FP32_MFMA_PEAK_TF
            r["peak"] = BF16_MFMA_PEAK_TF / nprod
            r["frac"] = r["achieved"] / r["peak"]
            r["peak_pipe"] = (f"fp16 MFMA dense peak (2500 TF) / 3 instructions per fp32 product group (f16x3)" if f16x3 else
                              "bf16 MFMA dense peak (2500 TF) / 6 instructions per fp32 product group (bf16x6)")
            r["mfma_instructions_per_fp32_product"] = int(nprod)
            r["frac_vs_bf16x6_bound"] = r["achieved"] / (BF16_MFMA_PEAK_TF / 6.0)      # (round 4's yardstick, for comparison across rounds)
            r["frac_bf16_pipe"] = r["frac"]
            if held_clock_ghz:
                # the matrix pipe's bound scales with the clock: 2.5 PF is the figure at 2.4 GHz.  Under this kernel the chip holds less
                # (DVFS: a cycle saved inside the kernel comes back partly as a lower clock), so two fractions are reported:
                # `frac` against the headline peak, frac_at_held_clock against the pipe's bound at the clock the kernel actually ran at
                r["held_clock_ghz"] = held_clock_ghz
                r["frac_at_held_clock"] = r["frac"] * 2.4 / held_clock_ghz
                r["held_clock_source"] = "s_memtime / s_memrealtime (100 MHz) over the life of workgroups 0, 256, 512, 768 of the last chain launch of the instrumented pass"
    # ---- the SAME step replayed as a hipGraph (graphs.GraphedStep), timed the same way; the faster mode is reported.
    # Eager: 19 launches through the C ABI per step, ~0.41 ms of host work on a quiet box (hidden by the 0.58 ms of GPU work)
    # but up to 0.8 ms on a loaded host -- and two collectives more per step when distributed -- which then bounds the step.
    # Replay has no host work.  Distributed steps are captured WITH their RCCL collectives (capture_error_mode =
    # "thread_local": the process-group watchdog may touch HIP during the capture; tests/test_gpu_rccl.py).  A watchdog
    # timer (capture_watchdog) leaves with the eager result and exit code 0 if a capture or a replay ever hangs.
    # At world > 1 rank 0 prints (and flushes) the EAGER line before the leg starts: whatever happens to a capture that
    # contains RCCL collectives on more than one rank (it has never run on hardware), the run has a valid
    # measurement on stdout.  If the leg completes, the final line follows and supersedes it.  DSS2_BENCH_DIST_GRAPH=0 skips
    # the leg at world > 1.
    # at world > 1 the capture races with torch's process-group watchdog thread and ABORTS the process about one run in ten
    # (hipErrorCapturedEvent, tests/test_gpu_rccl.py) -- an abort on any of 8 ranks would lose everything measured after the early line.
    # So the distributed capture leg is OPT-IN (DSS2_BENCH_DIST_GRAPH=1); the host-free leg at world > 1 is the segmented launch plan.
    dist_graph = os.environ.get("DSS2_BENCH_DIST_GRAPH", "0") == "1"
    run_graph_leg = not args.no_graph and (world == 1 or dist_graph)
    if world > 1 and rank == 0 and run_graph_leg:
        early = dict(result)
        early["partial"] = "eager measurement, printed before the hipGraph leg; a later line (if any) supersedes it"
        print(json.dumps(early), flush=True)
    if run_graph_leg:
        timer = capture_watchdog(result, rank, world, args.graph_timeout, leg="hipGraph replay")
        try:
            graphed = pkg.graphs.GraphedStep(step, stream=work_stream, capture_error_mode="thread_local")
            for _ in range(max(args.warmup, 5)):
                graphed.replay()
            gw, _ = timed_windows(graphed.replay)
            gdt = sorted(gw)[len(gw) // 2]
            result["config"]["ms_per_step_by_mode"]["hipGraph replay"] = gdt / args.steps * 1e3
            result["legs"]["hipGraph replay"] = "ok"
            if gdt < dt:
                dt, ms, windows, mode = gdt, gdt / args.steps * 1e3, gw, "hipGraph replay"
                value = args.batch * world / (dt / args.steps)
                result.update(value=value, ms_per_step=ms)
                result["config"].update(mode=mode, timed_windows=len(gw),
                                        window_ms_min_median_max=[min(gw) * 1e3, gdt * 1e3, max(gw) * 1e3])
        except Exception as exc:          # capture not possible on this stack: the eager numbers stand
            result["config"]["hipgraph"] = f"failed: {type(exc).__name__}: {exc}"[:300]
            result["legs"]["hipGraph replay"] = "failed"
        timer.cancel()
    elif not args.no_graph:
        result["config"]["hipgraph"] = "skipped at world > 1 (opt in with DSS2_BENCH_DIST_GRAPH=1: a capture with RCCL inside can abort the process)"
        result["legs"]["hipGraph replay"] = "skipped"
    # ---- and as a launch plan (graphs.PlannedStep, include/dss2_hip.h "launch plans"): the library's own record of the step's launches
    # re-issued from ONE C call per step -- real launches of the same kernels on the same stream, no Python between them, and like the
    # hipGraph replay it carries the step's vminmax launch (no cached batch constants).  At world > 1 the step's collectives (loss
    # sums, gradient bucket) cut the plan into segments: one C call per segment, the collectives between them (graphs.plan_collective)
    # -- no Python around the launches and no RCCL inside a capture.  Behind the same watchdog as the hipGraph leg; rank 0 first
    # prints what it has (the hipGraph result, if that leg completed), so a hang here costs nothing.
    dist_plan = os.environ.get("DSS2_BENCH_DIST_PLAN", "1") != "0"
    if not args.no_graph and (world == 1 or dist_plan):
        if world > 1 and rank == 0:
            early = dict(result)
            early["partial"] = "printed before the launch-plan leg; a later line (if any) supersedes it"
            print(json.dumps(early), flush=True)
        ptimer = capture_watchdog(result, rank, world, args.graph_timeout, leg="launch plan")
        try:
            # (verify: one replay must reproduce the recorded step's gradients bit for bit -- a launch the plan does not carry would show here)
            planned = pkg.graphs.PlannedStep(step, stream=work_stream, verify=(lambda: [p.grad for p in params]) if world == 1 else None)
            for _ in range(max(args.warmup, 5)):
                planned.replay()
            pw, _ = timed_windows(planned.replay)
            pdt = sorted(pw)[len(pw) // 2]
            result["config"]["ms_per_step_by_mode"]["launch plan"] = pdt / args.steps * 1e3
            result["config"]["plan_launches"] = planned.n_launches
            result["legs"]["launch plan"] = "ok"
            if pdt < dt:
                dt, ms, windows, mode = pdt, pdt / args.steps * 1e3, pw, "launch plan (dss2_plan_run)"
                value = args.batch * world / (dt / args.steps)
                result.update(value=value, ms_per_step=ms)
                result["config"].update(mode=mode, timed_windows=len(pw),
                                        window_ms_min_median_max=[min(pw) * 1e3, pdt * 1e3, max(pw) * 1e3])
            if world > 1:
                result["config"]["plan_segments"] = len(planned.segments)
        except Exception as exc:
            result["config"]["launch_plan"] = f"failed: {type(exc).__name__}: {exc}"[:300]
            result["legs"]["launch plan"] = "failed"
        ptimer.cancel()

    # ---- the same step on the two other arithmetic routes of the H -> H layers: the headline runs the products as
    # f16x3 (two fp16 pieces per operand, THREE 16-bit MFMAs per fp32 product, fp32 accumulate); beside it bf16x6 (SIX) and the true
    # fp32-MFMA route (v_mfma_f32_32x32x2_f32), each as a replayed launch plan on the same resident batch -- times only, the headline stays
    # the default route.  (Their errors against fp64: cpu_baseline.accuracy_vs_fp64_oracle below.)
    if world == 1 and not args.no_graph and not args.no_routes:
        by_route = {route_name(pkg): result["ms_per_step"]}
        saved = {k: getattr(pkg.flags, k) for k in ("CHAIN_BF16", "WGRAD_BF16", "CHAIN_F16", "WGRAD_F16")}
        for rname, setting in ROUTES.items():
            if rname in by_route:
                continue
            rt = capture_watchdog(result, rank, world, args.graph_timeout, leg=f"route {rname}")
            try:
                for k, v in setting.items():
                    setattr(pkg.flags, k, v)
                for _ in range(3):
                    step()
                pl = pkg.graphs.PlannedStep(step, stream=work_stream)
                for _ in range(10):
                    pl.replay()
                torch.cuda.synchronize()
                n_r, t0 = 0, time.perf_counter()
                while time.perf_counter() - t0 < args.route_seconds:
                    for _ in range(args.steps):
                        pl.replay()
                    torch.cuda.synchronize()
                    n_r += args.steps
                by_route[rname] = (time.perf_counter() - t0) / n_r * 1e3
                del pl
            except Exception as exc:
                by_route[rname] = f"failed: {type(exc).__name__}: {exc}"[:200]
            finally:
                rt.cancel()
                for k, v in saved.items():
                    setattr(pkg.flags, k, v)
        for _ in range(2):
            step()          # (the default route's plans / caches are current again)
        torch.cuda.synchronize()
        result["config"]["ms_per_step_by_arithmetic_route"] = by_route
        if isinstance(by_route.get("fp32-mfma"), float):
            result["ms_per_step_fp32_mfma_route"] = by_route["fp32-mfma"]

    if rank == 0:
        # ---- standalone scatter-add (K6) against the HBM roofline (north_star asks for it separately), at two sizes:
        # the C2 batch (90.9 MB: sits inside the 256 MiB Infinity Cache) and B = 32768 graphs (msg 470 MB + out 252 MB:
        # cache-busting) -- the second one is the number to hold against HBM.
        def scatter_point(n_graphs):
            if n_graphs == args.batch:
                ei_s, n_s = ei, N
            else:
                reps = n_graphs // args.batch
                npg = N // args.batch
                ei_s = torch.cat([ei + k * N for k in range(reps)], 1)      # block-diagonal copies of the batch
                n_s = npg * args.batch * reps
            topo = pkg.topology.get_topology(ei_s, n_s)
            e2 = 2 * ei_s.shape[1]
            msg = torch.randn(e2, HID, device=dev)
            ent = topo.perm.to(torch.int32)
            for _ in range(5):
                pkg.networks.segment_sum(msg, topo.rowptr, ent, n_s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps_t = 50 if n_graphs == args.batch else 20
            e0.record()
            for _ in range(reps_t):
                pkg.networks.segment_sum(msg, topo.rowptr, ent, n_s)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / reps_t * 1e3
            byts = 4.0 * e2 * HID + 4.0 * n_s * HID + 4.0 * e2 + 4.0 * (n_s + 1)
            return {"graphs": n_graphs, "achieved": byts / us / 1e3, "frac": byts / us / 1e3 / HBM_PEAK_GBS,
                    "frac_of_measured_copy_rate_6290": byts / us / 1e3 / 6290.0, "avg_launch_us": us,
                    "algorithmic_bytes_per_launch": byts}
        small = scatter_point(args.batch)
        result["scatter_add"] = {"kernel": "dss2::segment_sum_kernel", "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "cache_resident": small, "frac_cache_resident": small["frac"],
                                 "standalone": "K6 is a standalone kernel by SURVEY 8's design (what MessagePassing.propagate runs for a user message()): "
                                               "the C2 training step never launches it -- its aggregations live inside edge16_fwd / edge16_bwd and the chains' "
                                               "Horner hops, priced in `roofline` (MFMA-bound), not here",
                                 "note": "cache_resident: the C2 batch (90.9 MB working set < 256 MiB Infinity Cache); "
                                         "cache_busting: B = 32768 graphs (722 MB), the HBM number"}
        try:
            big = scatter_point(8 * args.batch)
            try:    # HBM bytes per launch of this very launch shape from the committed PMC passes (a pointer, not a live counter)
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
                    rec = json.load(fh)["segment_sum_kernel (B=32768, cache-busting)"]
                if 8 * args.batch == 32768:
                    big["traffic"], big["traffic_source"] = rec["hbm_bytes_per_launch"], rec["source"]
            except Exception:
                pass
            result["scatter_add"].update(cache_busting=big, frac_cache_busting=big["frac"], achieved=big["achieved"],
                                         frac=big["frac"])
        except Exception as exc:      # e.g. a box short of memory: report the cache-resident point only
            result["scatter_add"].update(cache_busting=f"skipped: {exc}", achieved=small["achieved"], frac=small["frac"])
        torch.cuda.empty_cache()

        # ---- the other configurations on this GPU (not bench lines of their own: BASELINE.json's metric is quoted on C2)
        if world == 1 and not args.no_other_configs:
            others = {}
            for tag, grids, B_, cls, cargs, ne, blocks in OTHER_CONFIGS:
                try:
                    others[tag] = time_other_config(pkg, dev, work_stream, tag, grids, B_, cls, cargs, ne, blocks, args.other_seconds)
                except Exception as exc:      # e.g. a box short of memory at B = 32768
                    others[tag] = {"skipped": f"{type(exc).__name__}: {exc}"[:200]}
                    torch.cuda.empty_cache()
            if not args.no_train_epochs:
                try:
                    others["C5 shuffled: a new topology mix every step, B=4096 H=256 L=8"] = time_c5_fresh_batches(pkg, dev)
                except Exception as exc:
                    others["C5 shuffled: a new topology mix every step, B=4096 H=256 L=8"] = {"skipped": f"{type(exc).__name__}: {exc}"[:200]}
                    torch.cuda.empty_cache()
                for tag, grid, B_, cls, cargs, nbat in TRAIN_EPOCHS:
                    try:
                        others[tag] = time_train_epoch(pkg, dev, work_stream, tag, grid, B_, cls, cargs, nbat)
                    except Exception as exc:
                        others[tag] = {"skipped": f"{type(exc).__name__}: {exc}"[:200]}
                        torch.cuda.empty_cache()
            result["other_configs"] = others
            result["other_configs_note"] = ("forward + gsp_wls_edge + backward on a resident synthetic batch, one GPU, eager, as a hipGraph "
                                            "replay and as a launch plan (one C call per step, dss2_plan_run), >= %.1f s of timed work each; "
                                            "ms_per_step = the fastest of eager / replay / plan; TFLOP/s from SURVEY 8(d)'s "
                                            "algorithmic FLOPs (3 x forward); the driver line excludes the optimizer" % args.other_seconds)

        # ---- CPU baseline (SURVEY 8d / BASELINE.md 3): the oracle -- a port of the reference's PyTorch-eager path --
        # on this box's host cores, same batch.  BASELINE.md 3's protocol: >= 10 warm-up + >= 50 timed steps, MEDIAN step, at the best
        # thread count (found by a short sweep: 2 warm-ups, >= 3 steps within --cpu-seconds per count); >= 5 timed steps (2 warm-ups)
        # for the as-is form (with the reference's unused dense Laplacian, data.py:422-423: 15.1 GB at C2); the single-thread figure
        # from the sweep; configuration C1 at the full protocol.
        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import dss2_oracle as oracle
            cpu_model = oracle.MPN(8, 6, 2, HID, LAYERS, KHOPS, 0.0)
            cpu_model.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
            cpu_batch = {"x": batch["x"], "edge_index": batch["edge_index"], "edge_attr": batch["edge_attr"]}
            ncpu = os.cpu_count() or 1
            max_threads = torch.get_num_threads()

            def time_leg(mdl, bt, stats_, n_graphs, threads, as_is=False, budget=args.cpu_seconds, warm=2, min_steps=3, max_steps=50):
                """`warm` untimed steps, then steps until (>= min_steps and the budget is spent) or max_steps; the MEDIAN step is reported."""
                torch.set_num_threads(threads)
                for _ in range(warm):
                    oracle.train_step(mdl, bt, stats_, REG, as_is_laplacian=as_is)
                ts, t_start = [], time.perf_counter()
                while True:
                    t0 = time.perf_counter()
                    oracle.train_step(mdl, bt, stats_, REG, as_is_laplacian=as_is)
                    ts.append(time.perf_counter() - t0)
                    if (len(ts) >= min_steps and time.perf_counter() - t_start > budget) or len(ts) >= max_steps:
                        break
                med = sorted(ts)[len(ts) // 2]
                return {"threads": threads, "graphs_per_s": n_graphs / med, "ms_per_step": med * 1e3, "steps": len(ts), "warmup_steps": warm,
                        "ms_per_step_mean": sum(ts) / len(ts) * 1e3}

            sweep = [time_leg(cpu_model, cpu_batch, batch["stats"], args.batch, th, budget=min(args.cpu_seconds, 3.0))
                     for th in sorted({th for th in (1, 8, 16, 32, 64, max_threads) if th <= min(max(ncpu, 1), 64)})]
            best_th = max(sweep, key=lambda r: r["graphs_per_s"])["threads"]
            one = [r for r in sweep if r["threads"] == 1][0]
            full = args.cpu_protocol_steps
            best = time_leg(cpu_model, cpu_batch, batch["stats"], args.batch, best_th, budget=0.0, warm=min(10, full), min_steps=full, max_steps=full)
            cb = {
                "value": best["graphs_per_s"], "unit": "graphs/s", "cores": best["threads"], "kind": "port",
                "sample": f"the same B={args.batch} CIGRE-14 batch (fwd + gsp_wls_edge + bwd, PyTorch eager CPU fp32, unused dense Laplacian of "
                          f"data.py:422-423 stripped): {best['warmup_steps']} warm-up + {best['steps']} timed steps at {best_th} threads, median step "
                          f"(BASELINE.md 3); thread count = best of a short sweep {[r['threads'] for r in sweep]} (2 warm-ups, >= 3 steps each)",
                "threads_best": best["threads"], "ms_per_step": best["ms_per_step"], "steps": best["steps"], "warmup_steps": best["warmup_steps"],
                "protocol_leg": best, "threads_1": one, "thread_sweep": sweep, "host_cpus": ncpu,
            }
            # as-is: with the dead dense Laplacian (N^2 fp32 zero-fill per step), needs ~2x 15.1 GB of free RAM
            need = 2.2 * 4.0 * N * N
            try:
                import psutil
                avail = psutil.virtual_memory().available
            except Exception:
                avail = 0
            if avail > need:
                cb["as_is_with_dense_laplacian"] = time_leg(cpu_model, cpu_batch, batch["stats"], args.batch, best_th, as_is=True,
                                                            budget=0.0, warm=2, min_steps=min(5, full), max_steps=min(5, full))
            else:
                cb["as_is_with_dense_laplacian"] = f"skipped: needs {need / 1e9:.0f} GB of free host RAM, {avail / 1e9:.0f} GB available"
            # C1: the reference's own CPU-runnable configuration (B = 64, H = 32, 1 layer), full protocol
            c1b = pkg.synthetic.make_batch(["cigre14"], 64, seed=2000)
            c1m = oracle.MPN(8, 6, 2, 32, 1, 2, 0.0)
            c1 = [time_leg(c1m, c1b, c1b["stats"], 64, th, budget=0.0, warm=10, min_steps=50, max_steps=50) for th in (1, min(8, ncpu))]
            cb["c1_cigre14_b64_h32_l1"] = max(c1, key=lambda r: r["graphs_per_s"])
            torch.set_num_threads(max_threads)

            # ---- what the three arithmetic routes cost in accuracy: each against the fp64 oracle (the checker) on the C2 MODEL and a 256-graph
            # batch -- max-normalised error of the output, relative error of the loss, worst max-normalised parameter gradient
            if not args.no_routes:
                try:
                    cb["accuracy_vs_fp64_oracle"] = route_accuracy(pkg, oracle, dev)
                except Exception as exc:
                    cb["accuracy_vs_fp64_oracle"] = f"failed: {type(exc).__name__}: {exc}"[:200]
            result["cpu_baseline"] = cb
            result["speedup_vs_cpu_baseline"] = value / cb["value"]
        print(json.dumps(result), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
