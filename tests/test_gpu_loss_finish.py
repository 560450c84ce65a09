"""GPU: the loss's fused finish (the last workgroup of the partials launch sums the workgroup partials) against the two-launch path,
under stress: many launches back to back with a CHANGING number of workgroups on one counter word -- 1 ... 16 workgroups
(the default use: release arrival + acquire by the last workgroup, by the HIP memory model) and, opted in, hundreds (relaxed agent-scope
arrivals behind write-through stores: rests on gfx950 behaviour) -- must give the two-launch path's loss and sums bit for bit, every time."""
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


def _cases(pkg, sizes):
    out = []
    for i, B in enumerate(sizes):
        b = pkg.synthetic.make_batch(["cigre14"], B, seed=10 + i, violate=0.4)
        x, ei, ea = b["x"].to(DEV), b["edge_index"].to(DEV), b["edge_attr"].to(DEV)
        st = tuple(s.to(DEV) for s in b["stats"])
        torch.manual_seed(i)
        y = torch.randn(x.shape[0], 2, device=DEV) * 0.5
        out.append((x, ei, ea, st, y))
    return out


def _loss(pkg, case):
    x, ei, ea, st, y = case
    return pkg.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=y.clone(), x_mean=st[0], x_std=st[1], edge_mean=st[2], edge_std=st[3],
                            edge_index=ei, reg_coefs=REG, num_samples=None, node_param=x[:, 8:], edge_param=ea[:, 6:])


@pytest.mark.parametrize("sizes,opt_in", [
    ([1, 17, 18, 35, 100, 137, 200, 273, 16, 256, 69, 3], False),     # 1 ... 16 workgroups of 256 nodes (15 nodes per graph): the default fused finish
    ([300, 1000, 4096, 512, 2000, 274], True),                        # 18 ... 240 workgroups: fused only with DSS2_WLS_FUSED_FINISH=1
])
def test_fused_finish_under_a_changing_number_of_workgroups(pkg, sizes, opt_in):
    cases = _cases(pkg, sizes)
    saved = (pkg.flags.WLS_FUSED_FINISH, pkg.flags.WLS_FUSED_FINISH_SMALL)
    try:
        pkg.flags.WLS_FUSED_FINISH, pkg.flags.WLS_FUSED_FINISH_SMALL = False, False
        want = [_loss(pkg, c).detach().clone() for c in cases]                       # two launches: partials, then a one-workgroup finish
        pkg.flags.WLS_FUSED_FINISH, pkg.flags.WLS_FUSED_FINISH_SMALL = opt_in, True
        order = torch.randint(0, len(cases), (400,), generator=torch.Generator().manual_seed(0)).tolist()
        got = [(_k, _loss(pkg, cases[_k]).detach()) for _k in order]                 # 400 launches back to back, no synchronisation between them
        torch.cuda.synchronize()
        for k, l in got:
            assert torch.equal(l, want[k]), (k, sizes[k], l.item(), want[k].item())
    finally:
        pkg.flags.WLS_FUSED_FINISH, pkg.flags.WLS_FUSED_FINISH_SMALL = saved
