"""Child process of tests/test_gpu_rccl.py (never imported by pytest: the leading underscore keeps it out of collection).

Started as a FRESH process (RANK=0, WORLD_SIZE=1, MASTER_ADDR=127.0.0.1) before anything in it touches the GPU, it runs
the product's own data-parallel branches on the one GPU of the box, over the "nccl" backend (= RCCL on ROCm):

  * ``gsp_wls_edge(..., group=WORLD)``: the all-reduce of the 7 loss sums between the two loss kernels (data.py),
  * the flat-bucket gradient all-reduce hook at the end of every MPN block's backward (networks._MPNFn.backward,
    parallel.attach_grad_allreduce), blocking and ``async_op=True`` + ``wait_grad_allreduce``,
  * the same step captured into a hipGraph (graphs.GraphedStep) with the collectives inside the capture.

At world size 1 a SUM all-reduce is the identity, so loss and every gradient must be BITWISE equal to the
non-distributed step.  Prints one JSON object on the last line of stdout.

    python _rccl_child.py eager     blocking and asynchronous collectives
    python _rccl_child.py plan      the step as a segmented launch plan (graphs.PlannedStep: one C call per segment, collectives between)
    python _rccl_child.py epoch     runner.EpochTrainer over the process group (segmented plan) against the same epochs without one
    python _rccl_child.py loose     the coalesced "loose" gradients (MultiMPN) over RCCL
    python _rccl_child.py graph     the step with its collectives inside a hipGraph capture (a separate process: a
                                    failing capture can take the process down, which must not take the eager result
                                    with it)
"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "deep-statistical-solver-for-distribution-system-state-estimation_amd"
REG = {"mu_v": 1e-1, "mu_theta": 1e-1, "lam_v": 1e-4, "lam_p": 1e-8, "lam_pf": 1e-6, "lam_reg": 1e2}


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "eager"
    import torch
    import torch.distributed as dist
    pkg = importlib.import_module(PKG)
    pkg._lib.lib()
    env = pkg.parallel.init_from_env("nccl")
    assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
    dev = torch.device("cuda", env["local"])
    group = dist.group.WORLD
    res = {"backend": dist.get_backend(), "world": dist.get_world_size(), "cases": {}}

    if mode == "epoch":
        # a data-parallel TRAINING epoch without the interpreter (runner.EpochTrainer, mode "plan": the step's collectives cut the recorded
        # step into segments) against the same epoch without a process group: every parameter bitwise (SUM over one rank = identity)
        full = pkg.synthetic.make_batch(["cigre14"], 200, seed=4, violate=0.3)
        ds = pkg.dataset.DeviceDataset.from_batch(full, device=dev)
        stats = tuple(t_.to(dev) for t_ in full["stats"])
        out = {}
        finals = []
        for grp in (None, group):
            torch.manual_seed(1)
            m = pkg.MPN(8, 6, 2, 64, 3, 2, 0.0).to(dev)
            o = pkg.optim.FusedAdamax(m.parameters(), lr=3e-3, capturable=True)
            if grp is not None:
                pkg.parallel.attach_grad_allreduce(m, grp)
            tr = pkg.runner.EpochTrainer(m, o, stats, REG, ds, 64, shuffle=False, mode="plan", group=grp)
            for _ in range(2):
                tr.train_epoch()
            torch.cuda.synchronize()
            finals.append(([p.detach().clone() for p in m.parameters()], tr.mean_loss()))
            out["segments" if grp is not None else "segments_local"] = len(tr.steps[64][0].segments)
        out["bitwise"] = bool(all(torch.equal(a, b_) for a, b_ in zip(finals[0][0], finals[1][0])) and finals[0][1] == finals[1][1])
        res["cases"]["EpochTrainer"] = out
        dist.barrier(); dist.destroy_process_group()
        print(json.dumps(res), flush=True)
        return
    if mode == "loose":
        # gradients that come in no flat bucket (the MultiMPN family): ONE collective per backward (parallel.LooseGradCoalescer) over RCCL,
        # bitwise the non-distributed gradients; a backward on top of gradients already in place falls back to one collective per parameter
        import types
        b = pkg.synthetic.make_batch(["cigre14"], 64, seed=2)
        data = types.SimpleNamespace(x=b["x"].to(dev), edge_index=b["edge_index"].to(dev), edge_attr=b["edge_attr"].to(dev))
        data.x = data.x[:, :8].contiguous(); data.edge_attr = data.edge_attr[:, :6].contiguous()
        torch.manual_seed(0)
        m = pkg.MultiMPN(8, 6, 2, 32, 3, 2, 0.0).to(dev)
        gout = torch.randn(data.x.shape[0], 2, device=dev)

        def run():
            m(data).backward(gout)
            torch.cuda.synchronize()
            return [p.grad.detach().clone() for p in m.parameters()]
        g0 = run()
        for p in m.parameters():
            p.grad = None
        pkg.parallel.attach_grad_allreduce(m, group)
        co = m._dss2_loose_grads
        g1 = run()
        c = {"loose_parameters": len(co.params) if co is not None else 0, "collectives": co.collectives if co is not None else -1,
             "fallback": co.fallback_collectives if co is not None else -1,
             "bitwise": bool(all(torch.equal(a, q) for a, q in zip(g0, g1)))}
        g2 = run()          # (gradients in place: accumulation)
        c["accumulated_bitwise"] = bool(all(torch.equal(2 * a, q) for a, q in zip(g0, g2)))
        c["fallback_after_accumulation"] = co.fallback_collectives if co is not None else -1
        res["cases"]["MultiMPN"] = c
        dist.barrier(); dist.destroy_process_group()
        print(json.dumps(res), flush=True)
        return

    models = {
        "MPN_C2_model": lambda: pkg.MPN(8, 6, 2, 128, 4, 2, 0.0),          # BASELINE config C2's model
        "SkipPFN_5_blocks": lambda: pkg.SkipPFN(8, 6, 2, 32, 3, 2, 0.0, 5),   # 5 blocks -> 5 gradient collectives per step
    }
    for name, build in models.items():
        torch.manual_seed(0)
        b = pkg.synthetic.make_batch(["cigre14", "cigre14_reswitched"], 256, seed=3, violate=0.5)   # penalties active
        x, ei, ea = b["x"].to(dev), b["edge_index"].to(dev), b["edge_attr"].to(dev)
        st = tuple(s.to(dev) for s in b["stats"])
        model = build().to(dev)
        calls = []

        def step(grp):
            for p in model.parameters():
                p.grad = None
            out = model(x[:, :8], ei, ea[:, :6])
            loss = pkg.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=out, x_mean=st[0], x_std=st[1],
                                    edge_mean=st[2], edge_std=st[3], edge_index=ei, reg_coefs=REG, num_samples=None,
                                    node_param=x[:, 8:], edge_param=ea[:, 6:], group=grp)
            loss.backward()
            return loss

        def snapshot(loss):
            torch.cuda.synchronize()
            return loss.detach().clone(), [p.grad.detach().clone() for p in model.parameters()]

        c = {}
        if mode == "graph":
            # canonical capture order (graphs.GraphedStep): capture BEFORE any eager step has created AccumulateGrad
            # nodes on the default stream; the non-distributed reference runs afterwards on the capture stream
            pkg.parallel.attach_grad_allreduce(model, group)
            print(f"[{name}] capturing", file=sys.stderr, flush=True)
            gs = pkg.graphs.GraphedStep(lambda: step(group), capture_error_mode=os.environ.get("DSS2_CAPTURE_MODE", "thread_local"))
            print(f"[{name}] captured, replaying", file=sys.stderr, flush=True)
            l3, g3 = snapshot(gs.replay())
            for m in model.modules():
                if hasattr(m, "_grad_bucket_hook"):
                    m._grad_bucket_hook = None
            with torch.cuda.stream(gs.stream):
                l0, g0 = snapshot(step(None))
            c["graph_capture"] = "ok"
            c["graph_bitwise"] = bool(torch.equal(l0, l3) and all(torch.equal(a, q) for a, q in zip(g0, g3)))
            res["cases"][name] = c
            continue
        if mode == "plan":
            # the data-parallel step as a SEGMENTED launch plan (graphs.PlannedStep): the collectives cut the record, a replay issues
            # segment, collective, segment, ...; asynchronous bucket collectives joined by wait_grad_allreduce inside the step
            l0, g0 = snapshot(step(None))
            pkg.parallel.attach_grad_allreduce(model, group, async_op=(name == "MPN_C2_model" and os.environ.get("PLAN_ASYNC", "1") == "1"))

            def dstep():
                loss = step(group if os.environ.get("PLAN_GROUP", "1") == "1" else None)
                pkg.parallel.wait_grad_allreduce(model)
                return loss
            plan = pkg.graphs.PlannedStep(dstep)
            for p in model.parameters():
                p.grad.fill_(float("nan"))
            l4, g4 = snapshot(plan.replay())
            l5, g5 = snapshot(plan.replay())
            c["segments"], c["collectives"], c["launches"] = len(plan.segments), plan.n_collectives, plan.n_launches
            c["loss_equal"] = bool(torch.equal(l0, l4))
            c["grads_unequal"] = [i for i, (a, q) in enumerate(zip(g0, g4)) if not torch.equal(a, q)]
            c["worst_rel"] = max([float((a - q).abs().max() / q.abs().max().clamp_min(1e-30)) for a, q in zip(g0, g4)] + [0.0])
            c["nan_left"] = int(sum(int(torch.isnan(q).sum()) for q in g4))
            c["plan_bitwise"] = bool(torch.equal(l0, l4) and all(torch.equal(a, q) for a, q in zip(g0, g4))
                                     and torch.equal(l0, l5) and all(torch.equal(a, q) for a, q in zip(g0, g5)))
            del plan
            res["cases"][name] = c
            continue
        l0, g0 = snapshot(step(None))                                     # non-distributed reference
        print(f"[{name}] reference step done", file=sys.stderr, flush=True)
        # ---- blocking collectives
        n_hooks = pkg.parallel.attach_grad_allreduce(model, group)
        orig = pkg.parallel.allreduce_flat_grads

        def counting(flat, g=None, pending=None):
            calls.append(int(flat.numel()))
            return orig(flat, g, pending)
        pkg.parallel.allreduce_flat_grads = counting
        l1, g1 = snapshot(step(group))
        c["hooks_attached"], c["grad_allreduces_per_step"], c["bucket_elems"] = n_hooks, len(calls), list(calls)
        c["blocking_bitwise"] = bool(torch.equal(l0, l1) and all(torch.equal(a, q) for a, q in zip(g0, g1)))
        # ---- asynchronous collectives, joined before the gradients are read
        calls.clear()
        pkg.parallel.attach_grad_allreduce(model, group, async_op=True)
        loss = step(group)
        c["async_joined"] = pkg.parallel.wait_grad_allreduce(model)
        l2, g2 = snapshot(loss)
        c["async_bitwise"] = bool(torch.equal(l0, l2) and all(torch.equal(a, q) for a, q in zip(g0, g2)))
        pkg.parallel.allreduce_flat_grads = orig
        res["cases"][name] = c
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
