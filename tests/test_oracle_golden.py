"""CPU: the oracle (oracle/dss2_oracle.py) against the golden vectors produced by the reference
itself (tests/golden/make_goldens.py) and against pandapower's stored power-flow results."""
import numpy as np
import pytest
import torch

from conftest import MULTI_CASES, multi_case_data, CASES, LOSS_CASES, case_batch, case_grads, case_state_dict, golden, rel_err, t, tagconv_known_answers

# fp32 noise floor of the reference path itself is ~6e-7 (out), ~1.2e-6 (loss), ~2e-5 (grads)
TOL_OUT, TOL_LOSS, TOL_GRAD = 2e-6, 5e-6, 5e-5


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_model_matches_reference(oracle, name):
    cls, args, with_loss = CASES[name]
    g = golden(f"case_{name}.npz")
    model = getattr(oracle, cls)(*args)
    missing = model.load_state_dict(case_state_dict(g), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    b = case_batch(g)
    x, ei, ea = b["x"], b["edge_index"], b["edge_attr"]
    out = model(x[:, :8], ei, ea[:, :6])
    assert rel_err(out, t(g["out"])) < TOL_OUT
    if with_loss:
        st = b["stats"]
        loss = oracle.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=out, x_mean=st[0], x_std=st[1],
                                   edge_mean=st[2], edge_std=st[3], edge_index=ei,
                                   reg_coefs=oracle.DEFAULT_REG_COEFS, num_samples=None,
                                   node_param=x[:, 8:], edge_param=ea[:, 6:])
        assert abs(loss.item() - float(g["loss"])) <= TOL_LOSS * abs(float(g["loss"]))
        # gsp_wls_edge mutates the model output in place (data.py:413)
        assert torch.equal(out.detach(), t(g["out_after_loss"])) or rel_err(out, t(g["out_after_loss"])) < TOL_OUT
        loss.backward()
    else:
        out.backward(t(g["gout"]))
    grads = case_grads(g)
    for k, p in model.named_parameters():
        assert rel_err(p.grad, grads[k]) < TOL_GRAD, k


@pytest.mark.parametrize("name", LOSS_CASES)
def test_oracle_loss_matches_reference(oracle, name):
    g = golden(f"case_{name}.npz")
    b = case_batch(g)
    x, ei, ea, st = b["x"], b["edge_index"], b["edge_attr"], b["stats"]
    o_leaf = t(g["output"]).clone().requires_grad_(True)
    o = o_leaf * 1.0
    loss = oracle.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=o, x_mean=st[0], x_std=st[1],
                               edge_mean=st[2], edge_std=st[3], edge_index=ei, reg_coefs=oracle.DEFAULT_REG_COEFS,
                               num_samples=None, node_param=x[:, 8:], edge_param=ea[:, 6:])
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) <= TOL_LOSS * abs(float(g["loss"]))
    assert rel_err(o, t(g["output_after"])) < 1e-7
    assert rel_err(o_leaf.grad, t(g["grad_output"])) < TOL_GRAD
    yv = torch.cat([o.detach()[:, 0:1] * st[1][:1] + st[0][:1], o.detach()[:, 1:]], 1)
    flows = torch.stack(oracle.get_pflow(yv, ei, x[:, 8:], ea[:, 6:]), 1)
    assert rel_err(flows, t(g["pflow"])) < TOL_OUT
    # the partial-sum decomposition used for the data-parallel loss reproduces the same scalar
    sums = oracle.wls_partial_sums(x[:, :8], ea[:, :6], o.detach(), st[0], st[1], st[2], st[3], ei,
                                   x[:, 8:], ea[:, 6:], oracle.DEFAULT_REG_COEFS)
    l2 = oracle.loss_from_sums(sums, x.shape[0], ei.shape[1], oracle.DEFAULT_REG_COEFS["lam_reg"])
    assert abs(l2.item() - float(g["loss"])) <= 2e-5 * abs(float(g["loss"]))


def test_penalty_terms_active_in_violating_cases(oracle):
    """The violating fixtures must actually exercise J_v / J_theta / J_loading."""
    for name in ["loss_violate_cigre", "loss_violate_ober"]:
        g = golden(f"case_{name}.npz")
        b = case_batch(g)
        x, ei, ea, st = b["x"], b["edge_index"], b["edge_attr"], b["stats"]
        sums = oracle.wls_partial_sums(x[:, :8], ea[:, :6], t(g["output_after"]), st[0], st[1], st[2], st[3], ei,
                                       x[:, 8:], ea[:, 6:], oracle.DEFAULT_REG_COEFS)
        assert (sums[2:] > 0).all(), (name, sums)


def test_physics_known_answers(oracle):
    """get_pflow fed with pandapower's solved bus voltages must reproduce pandapower's stored
    branch flows and bus injections (independent of PyG and of the reference's Python)."""
    g = golden("physics_known.npz")
    for k in range(5):
        y = t(g[f"s{k}/labels"], torch.float64)
        ei = t(g[f"s{k}/edge_index"])
        npar = t(g[f"s{k}/node_param"], torch.float64)
        epar = t(g[f"s{k}/edge_param"], torch.float64)
        br = t(g[f"s{k}/branch"], torch.float64)
        ll, lt, pf, qf, pt, qt, i_f, i_t = oracle.get_pflow(y, ei, npar, epar)
        for mine, col, tol in [(pf, 0, 4e-5), (qf, 1, 4e-5), (pt, 2, 4e-5), (qt, 3, 4e-5), (i_f, 4, 1e-4), (i_t, 5, 1e-4)]:
            assert rel_err(mine, br[:, col]) < tol, (k, col)
        n = y.shape[0]
        p_i = -oracle.scatter_sum(pt, ei[1], n) - oracle.scatter_sum(pf, ei[0], n)
        assert rel_err(p_i, t(g[f"s{k}/bus_pq"], torch.float64)[:, 0]) < 2e-5
        # reference-behaviour quirk: line loading equals pandapower's loading_percent/100 on lines
        lines = epar[:, 5] == 0
        assert rel_err(ll[lines], br[lines, 6] / 100.0) < 1e-3


def test_structural_facts(oracle):
    f = golden("facts.npz")
    real = golden("cigre14_real64.npz")
    ei = t(real["edge_index"])
    assert oracle.is_directed(ei) == bool(f["is_directed_real"]) is True
    und = torch.cat([ei[:, :56], ei[:, :56].flip(0)], 1)
    assert oracle.is_directed(und) == bool(f["is_directed_undirected_input"]) is False
    assert float(f["trafo_pos_cigre_max"]) == 1.0 and float(f["trafo_pos_ober_max"]) == 3.0
    assert bool(f["eval_dropout_differs"])
    assert list(oracle.MPN(8, 6, 2, 32, 2, 2, 0.0).state_dict().keys()) == list(f["state_dict_keys_mpn"])
    assert list(oracle.SkipPFN(8, 6, 2, 32, 2, 2, 0.0, 2).state_dict().keys()) == list(f["state_dict_keys_skippfn"])
    m = oracle.MPN(8, 6, 2, 32, 2, 2, 0.3).eval()
    x, ea = t(real["x"]), t(real["edge_attr"])
    with torch.no_grad():
        assert (m(x[:, :8], ei, ea[:, :6]) - m(x[:, :8], ei, ea[:, :6])).abs().max() > 0


def test_oracle_get_pflow_with_phase_shift_matches_reference(oracle):
    """get_pflow(..., phase_shift=False): shift = edge_param[:, 5] (data.py:364-365), golden from the reference itself."""
    gs = golden("case_pflow_shift.npz")
    for name in LOSS_CASES:
        g = golden(f"case_{name}.npz")
        b = case_batch(g)
        x, ei, ea, st = b["x"], b["edge_index"], b["edge_attr"], b["stats"]
        o = t(g["output_after"])
        yv = torch.cat([o[:, 0:1] * st[1][:1] + st[0][:1], o[:, 1:]], 1)
        flows = torch.stack(oracle.get_pflow(yv, ei, x[:, 8:], ea[:, 6:], phase_shift=False), 1)
        assert rel_err(flows, t(gs[f"{name}/pflow_shift"])) < TOL_OUT
        assert rel_err(flows, t(g["pflow"])) > 1e-3          # the shift changes the flows on the trafo branches


def test_oracle_tagconv_matches_hand_derived_known_answers(oracle):
    """An anchor for PyG's TAGConv / gcn_norm semantics that does not come from the oracle's author's stand-in: exact
    paper-and-pencil outputs on a 3-node path graph (degrees 1/2/1, all weights 1/sqrt2) and on a directed 3-node DAG
    that separates in-degree from out-degree normalisation and exercises the deg = 0 -> 0 rule, K = 0..3, with bias."""
    x, lins, bias, cases = tagconv_known_answers()
    for name, (ei, exp) in cases.items():
        for K, want in exp.items():
            conv = oracle.TAGConv(2, 2, K).double()
            with torch.no_grad():
                conv.bias.copy_(bias)
                for k in range(K + 1):
                    conv.lins[k].weight.copy_(lins[k])
            got = conv(x, ei)
            assert (got - want).abs().max() < 1e-13, (name, K)


@pytest.mark.parametrize("name", list(MULTI_CASES))
def test_oracle_multi_variants_match_reference(oracle, name):
    """MaskEmbdMPN / MultiMPN / MaskEmbdMultiMPN / MaskEmbdMultiMPN_NoMP (networks.py:390-735): the oracle's restatement
    against goldens produced by the reference's own classes (forward(data) + backward of a stored output gradient)."""
    cls, args = MULTI_CASES[name]
    g = golden(f"case_{name}.npz")
    model = getattr(oracle, cls)(*args)
    res = model.load_state_dict(case_state_dict(g), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    out = model(multi_case_data(g))
    assert rel_err(out, t(g["out"])) < TOL_OUT
    out.backward(t(g["gout"]))
    grads = case_grads(g)
    for k, p in model.named_parameters():
        assert rel_err(p.grad, grads[k]) < TOL_GRAD, k
