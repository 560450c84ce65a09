"""GPU: launch plans (include/dss2_hip.h "launch plans", graphs.PlannedStep) -- a training step recorded once as the library's own
launch list and re-issued from ONE C call (steps that cannot be captured into a hipGraph stop paying the Python around
every launch; the reference runs its loop from Python, /root/reference/dss2_run.py:134-144).

* replaying the plan gives bit for bit the eager step: loss, outputs, every parameter gradient -- MPN on the chained kernels (the C2
  model), the C1 configuration on the whole-stack kernels, a SkipPFN with in-kernel dropout (device-side random state);
* N planned training steps with the fused Adamax (capturable: device-side step count) = N eager steps, bit for bit in every parameter;
* a plan that is still recording cannot be run; recording twice at once is refused;
* tensors allocated between two replays are not written by the second one (every thread's allocations of the recorded step stay in the plan's pool).
"""
import ctypes as C
import importlib

import pytest
import torch

from conftest import PKG_NAME

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REG = {"mu_v": 1e-1, "mu_theta": 1e-1, "lam_v": 1e-4, "lam_p": 1e-8, "lam_pf": 1e-6, "lam_reg": 1e2}


@pytest.fixture(scope="module")
def pkg():
    return importlib.import_module(PKG_NAME)


def _make(pkg, cls, cargs, grids, B, seed=0):
    torch.manual_seed(seed)
    b = pkg.synthetic.make_batch(grids, B, seed=seed, violate=0.3)
    x, ei, ea = b["x"].to(DEV), b["edge_index"].to(DEV), b["edge_attr"].to(DEV)
    st = tuple(s.to(DEV) for s in b["stats"])
    model = getattr(pkg, cls)(*cargs).to(DEV)
    params = list(model.parameters())

    def step(opt=None):
        for p in params:
            p.grad = None
        out = model(x[:, :8], ei, ea[:, :6])
        loss = pkg.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=out, x_mean=st[0], x_std=st[1], edge_mean=st[2], edge_std=st[3],
                                edge_index=ei, reg_coefs=REG, num_samples=None, node_param=x[:, 8:], edge_param=ea[:, 6:])
        loss.backward(pkg.data.unit_grad(loss))
        if opt is not None:
            opt.step()
        return loss
    return model, params, step


@pytest.mark.parametrize("cls,cargs,grids,B", [
    ("MPN", (8, 6, 2, 128, 4, 2, 0.0), ["cigre14"], 256),                         # the C2 model: chained layers, batched weight gradients
    ("MPN", (8, 6, 2, 32, 1, 2, 0.0), ["cigre14"], 64),                           # BASELINE C1: the whole-stack kernels
    ("MPN", (8, 6, 2, 128, 4, 2, 0.0), ["cigre14", "cigre14_reswitched"], 100),   # mixed topologies
    ("SkipPFN", (8, 6, 2, 32, 3, 2, 0.0, 3), ["cigre14"], 128),                   # a stack, one autograd node
])
def test_plan_replay_is_bitwise_the_eager_step(pkg, cls, cargs, grids, B):
    model, params, step = _make(pkg, cls, cargs, grids, B)
    loss_e = step().detach().clone()
    grads_e = [p.grad.detach().clone() for p in params]
    torch.cuda.synchronize()
    plan = pkg.graphs.PlannedStep(step)
    assert plan.n_launches >= 5
    for p in params:          # the recording step left its gradients in the plan's pool: poison them, the replay must rewrite every one
        p.grad.fill_(float("nan"))
    plan.loss.detach().fill_(float("nan"))
    loss_p = plan.replay()
    torch.cuda.synchronize()
    assert torch.equal(loss_p, loss_e), (loss_p.item(), loss_e.item())
    for p, g in zip(params, grads_e):
        assert torch.equal(p.grad, g)
    # ... again (the plan is reusable), with tensors allocated in between: nothing the recorded step allocated -- its backward runs on the
    # autograd engine's thread -- may have gone back to the general allocator, or this replay writes its slabs into the clones
    clones = [p.grad.clone() for p in params] + [torch.full((1 << 20,), 7.0, device=DEV) for _ in range(24)]
    plan.replay()
    torch.cuda.synchronize()
    for c, g in zip(clones, grads_e):
        assert torch.equal(c, g)
    assert all(bool((c == 7.0).all()) for c in clones[len(params):])
    # ... and against a fresh eager step
    loss_e2 = step().detach().clone()
    torch.cuda.synchronize()
    assert torch.equal(loss_e2, loss_e)


def test_planned_training_steps_with_fused_adamax_equal_eager_steps(pkg):
    cargs = (8, 6, 2, 32, 3, 2, 0.3, 2)      # SkipPFN with dropout: the masks come from the device-side state, advanced by every replay
    m1, p1, step1 = _make(pkg, "SkipPFN", cargs, ["cigre14"], 64, seed=3)
    m2, p2, step2 = _make(pkg, "SkipPFN", cargs, ["cigre14"], 64, seed=3)
    m2.load_state_dict(m1.state_dict())
    o1 = pkg.optim.FusedAdamax(p1, lr=1e-3, capturable=True)
    o2 = pkg.optim.FusedAdamax(p2, lr=1e-3, capturable=True)
    # dropout needs the SAME random stream in both runs: both models draw their seeds from torch's generator at first use
    torch.manual_seed(11)
    plan = pkg.graphs.PlannedStep(lambda: step1(o1), warmup=1)       # warm-up 1 + recording 1 = 2 real steps
    for _ in range(3):
        plan.replay()
    torch.manual_seed(11)
    g = pkg.graphs.GraphedStep(lambda: step2(o2), warmup=1)          # the same schedule through a hipGraph: 1 warm-up, capture (no execution) ...
    for _ in range(4):                                               # ... so 4 replays = the plan's recording step + 3 replays
        g.replay()
    torch.cuda.synchronize()
    for a, b in zip(p1, p2):
        assert torch.equal(a, b), (a - b).abs().max().item()


def test_plan_api_refuses_misuse(pkg):
    L = pkg._lib.lib()
    h1, h2 = C.c_void_p(), C.c_void_p()
    assert L.dss2_plan_begin(C.byref(h1)) == 0
    try:
        assert L.dss2_plan_begin(C.byref(h2)) != 0                   # one plan records at a time
        assert L.dss2_plan_run(h1, None) != 0                        # ... and a recording plan cannot run
    finally:
        assert L.dss2_plan_end(h1) == 0
    assert L.dss2_plan_size(h1) == 0
    assert L.dss2_plan_run(h1, None) == 0                            # an empty plan runs (nothing)
    assert L.dss2_plan_end(h1) != 0                                  # not recording any more
    L.dss2_plan_destroy(h1)
