"""CPU: numpy transliterations of the arithmetic inside the HIP kernels (not the kernels
themselves) against the oracle, in fp64.  They pin the formulas the kernels implement - the
analytic backward of gsp_wls_edge/get_pflow (csrc/dss2_loss.hip), the Horner form of TAGConv
and its transposed data-gradient / weight-gradient (csrc/dss2_gemm_prop.hip, dss2_wgrad.hip) and
the aggregate-then-Linear form of EdgeAggregation (csrc/dss2_edge.hip) - before any GPU run."""
import numpy as np
import pytest
import torch

from conftest import LOSS_CASES, case_batch, golden, load_pkg, t

import dss2_topology_oracle as topo_oracle   # structure oracle (CSR / incidence layout of include/dss2_hip.h)


def _emulate_wls(topo, x, ea, out, st, reg):
    """Transliteration of wls_partials_kernel + wls_grad_kernel (node-centric, incidence CSR)."""
    x, ea, out = x.double().numpy(), ea.double().numpy(), out.double().numpy().copy()
    xm, xs, em, es = (s.double().numpy() for s in st)
    N, E = x.shape[0], ea.shape[0]
    npar, epar = x[:, 8:], ea[:, 6:]
    vlv, vhv = npar[:, 0].min(), npar[:, 0].max()
    ef, et = topo.efrom.numpy(), topo.eto.numpy()
    rp, ent = topo.inc_rowptr.numpy(), topo.inc_ent.numpy()
    mi = 1.0 - npar[:, 1]
    out[:, 1] *= mi
    v, th = out[:, 0] * xs[0] + xm[0], out[:, 1]

    def meas(row, mean, std, n):
        Z = np.zeros(n); R = np.zeros(n)
        for c in range(n):
            z, r = row[2 * c], row[2 * c + 1]
            Z[c] = (z * std[2 * c] + mean[2 * c]) if z != 0 else 0.0
            R[c] = (r * std[2 * c + 1] + mean[2 * c + 1]) if r != 0 else 0.0
        return Z, R

    def flow(vf, vt, thf, tht, ep):
        G, B, Gs, Bs, tp, imax = ep[0], ep[1], ep[2], ep[3], np.ceil(ep[5]), ep[6]
        d = thf - tht; s, c = np.sin(d), np.cos(d); kk = vlv * vlv
        gg, bb = G + Gs / 2, B + Bs / 2
        pf = (-vf * vt * (G * c + B * s) + gg * vf * vf) * kk
        qf = (vf * vt * (-G * s + B * c) - bb * vf * vf) * kk
        pt = (-vf * vt * (G * c - B * s) + gg * vt * vt) * kk
        qt = (vf * vt * (G * s + B * c) - bb * vt * vt) * kk
        ratio = vhv / vlv; s3 = float(np.sqrt(np.float32(3)))  # data.py:378 takes sqrt of an int tensor -> f32
        i_f = np.hypot(pf, qf) / (vf * vlv * s3) / (1 - tp * (1 - ratio))
        i_t = np.hypot(pt, qt) / (vt * vlv * s3)
        ll = (1 - tp) * max(i_f, i_t) / imax
        lt = tp * max(i_f * vhv, i_t * vlv) / imax
        return dict(pf=pf, qf=qf, pt=pt, qt=qt, i_f=i_f, i_t=i_t, ll=ll, lt=lt, d=d, s=s, c=c, tp=tp, imax=imax,
                    G=G, B=B, gg=gg, bb=bb)

    sums = np.zeros(5); apq = np.zeros((N, 2))
    lam = [reg["lam_v"], reg["lam_v"], reg["lam_p"], reg["lam_p"]]
    for i in range(N):
        Z, R = meas(x[i], xm, xs, 4)
        p_i = q_i = 0.0
        for k in range(rp[i], rp[i + 1]):
            en = int(ent[k]); e = en & 0x7fffffff; to_end = en < 0
            o = ef[e] if to_end else et[e]
            f = flow(v[o], v[i], th[o], th[i], epar[e]) if to_end else flow(v[i], v[o], th[i], th[o], epar[e])
            if to_end:
                p_i -= f["pt"]; q_i -= f["qt"]
            else:
                p_i -= f["pf"]; q_i -= f["qf"]
                eZ, eR = meas(ea[e], em, es, 2)
                sums[1] += (eZ[0] - f["pf"]) ** 2 * eR[0] * reg["lam_pf"] + (eZ[1] - f["qf"]) ** 2 * eR[1] * reg["lam_pf"]
                sums[3] += max(abs(f["d"]) - 0.5, 0.0)
                sums[4] += max(f["ll"] + f["lt"] - 1.5, 0.0)
        h = [v[i], th[i], p_i, q_i]
        sums[0] += sum((Z[c] - h[c]) ** 2 * R[c] * lam[c] for c in range(4))
        apq[i] = [-2 * (Z[2] - p_i) * R[2] * reg["lam_p"], -2 * (Z[3] - q_i) * R[3] * reg["lam_p"]]
        sums[2] += max(v[i] - 1.1, 0) + max(0.9 - v[i], 0)
    lr = reg["lam_reg"]
    mv, mt, ml = sums[2] / N, sums[3] / E, sums[4] / E
    loss = sums[0] / N + sums[1] / E + lr * (mv * mv + mt * mt + ml * ml)
    m_v, m_t, m_l = 2 * lr * mv / N, 2 * lr * mt / E, 2 * lr * ml / E
    grad = np.zeros((N, 2)); kk = vlv * vlv; s3 = float(np.sqrt(np.float32(3))); ratio = vhv / vlv
    for i in range(N):
        Z, R = meas(x[i], xm, xs, 4)
        gv = (-2 * (Z[0] - v[i]) * R[0] * reg["lam_v"]) / N + m_v * ((v[i] > 1.1) * 1.0 - (v[i] < 0.9) * 1.0)
        gth = (-2 * (Z[1] - th[i]) * R[1] * reg["lam_v"]) / N
        for k in range(rp[i], rp[i + 1]):
            en = int(ent[k]); e = en & 0x7fffffff; to_end = en < 0
            o = ef[e] if to_end else et[e]
            vf, vt = (v[o], v[i]) if to_end else (v[i], v[o])
            thf, tht = (th[o], th[i]) if to_end else (th[i], th[o])
            apf, aqf = (apq[o] / N) if to_end else (apq[i] / N)
            apt, aqt = (apq[i] / N) if to_end else (apq[o] / N)
            f = flow(vf, vt, thf, tht, epar[e])
            eZ, eR = meas(ea[e], em, es, 2)
            uPf = (-2 * (eZ[0] - f["pf"]) * eR[0] * reg["lam_pf"]) / E - apf
            uQf = (-2 * (eZ[1] - f["qf"]) * eR[1] * reg["lam_pf"]) / E - aqf
            uPt, uQt = -apt, -aqt
            dvf_d = dvt_d = 0.0
            tp, imax = f["tp"], f["imax"]
            if f["ll"] + f["lt"] > 1.5 and m_l != 0:
                a = 1.0 if f["i_f"] >= f["i_t"] else 0.0
                b = 1.0 if f["i_f"] * vhv >= f["i_t"] * vlv else 0.0
                gIf = m_l * ((1 - tp) * a + tp * vhv * b) / imax
                gIt = m_l * ((1 - tp) * (1 - a) + tp * vlv * (1 - b)) / imax
                cf, ct = vlv * s3 * (1 - tp * (1 - ratio)), vlv * s3
                Af, At = np.hypot(f["pf"], f["qf"]), np.hypot(f["pt"], f["qt"])
                if Af > 0:
                    uPf += gIf * f["pf"] / (Af * vf * cf); uQf += gIf * f["qf"] / (Af * vf * cf)
                if At > 0:
                    uPt += gIt * f["pt"] / (At * vt * ct); uQt += gIt * f["qt"] / (At * vt * ct)
                dvf_d, dvt_d = -gIf * f["i_f"] / vf, -gIt * f["i_t"] / vt
            gd = m_t * np.sign(f["d"]) if abs(f["d"]) > 0.5 else 0.0
            G, B, gg, bb, c, s = f["G"], f["B"], f["gg"], f["bb"], f["c"], f["s"]
            a1, a2, a3, a4 = G * c + B * s, -G * s + B * c, G * c - B * s, G * s + B * c
            vv = vf * vt
            if not to_end:
                gv += (uPf * (-vt * a1 + 2 * gg * vf) + uQf * (vt * a2 - 2 * bb * vf) + uPt * (-vt * a3) + uQt * (vt * a4)) * kk + dvf_d
            else:
                gv += (uPf * (-vf * a1) + uQf * (vf * a2) + uPt * (-vf * a3 + 2 * gg * vt) + uQt * (vf * a4 - 2 * bb * vt)) * kk + dvt_d
            dd = (uPf * (-vv * a2) + uQf * (vv * (-a1)) + uPt * (vv * a4) + uQt * (vv * a3)) * kk + gd
            gth += -dd if to_end else dd
        grad[i] = [gv * xs[0], gth * mi[i]]
    return loss, grad, out, sums


@pytest.mark.parametrize("name", LOSS_CASES)
def test_loss_kernel_math(oracle, name):
    pkg = load_pkg()
    g = golden(f"case_{name}.npz")
    b = case_batch(g)
    x, ei, ea, st = b["x"], b["edge_index"], b["edge_attr"], b["stats"]
    n_sub = 45 if "ober" not in name else 70          # a few whole graphs keep the python loops short
    e_sub = int((ei[0] < n_sub).sum())
    x, ea, ei = x[:n_sub], ea[:e_sub], ei[:, :e_sub]
    out0 = t(g["output"])[:n_sub]
    topo = topo_oracle.TopologyOracle(ei, n_sub)
    loss, grad, out_after, _ = _emulate_wls(topo, x, ea, out0, st, oracle.DEFAULT_REG_COEFS)
    d = torch.float64
    o_leaf = out0.to(d).clone().requires_grad_(True)
    o = o_leaf * 1.0
    ref = oracle.gsp_wls_edge(input=x[:, :8].to(d), edge_input=ea[:, :6].to(d), output=o, x_mean=st[0].to(d),
                              x_std=st[1].to(d), edge_mean=st[2].to(d), edge_std=st[3].to(d), edge_index=ei,
                              reg_coefs=oracle.DEFAULT_REG_COEFS, num_samples=None, node_param=x[:, 8:].to(d),
                              edge_param=ea[:, 6:].to(d))
    ref.backward()
    assert abs(loss - ref.item()) <= 1e-10 * abs(ref.item())
    np.testing.assert_allclose(out_after, o.detach().numpy(), rtol=0, atol=1e-12)
    gref = o_leaf.grad.numpy()
    assert np.abs(grad - gref).max() <= 1e-9 * np.abs(gref).max()


def _dense_ahat(topo):
    A = torch.zeros(topo.N, topo.N, dtype=torch.float64)
    rp, col, w = topo.rowptr.numpy(), topo.col.numpy(), topo.w.double().numpy()
    for i in range(topo.N):
        for k in range(rp[i], rp[i + 1]):
            A[i, col[k]] += w[k]
    return A


def test_horner_tagconv_and_its_gradients(oracle):
    """out = G0 + A(G1 + A G2); dh = sum_m (A^T)^m (g W_m); dW_m = (A^T^m g)^T h; and the CSR by
    source really is A^T."""
    pkg = load_pkg()
    torch.manual_seed(0)
    b = pkg.synthetic.make_batch(["cigre14", "cigre14_reswitched"], 4, seed=3)
    N = b["x"].shape[0]
    topo = topo_oracle.TopologyOracle(b["edge_index"], N)
    A = _dense_ahat(topo)
    AT = torch.zeros_like(A)
    rp, col, w = topo.rowptrT.numpy(), topo.colT.numpy(), topo.wT.double().numpy()
    for i in range(N):
        for k in range(rp[i], rp[i + 1]):
            AT[i, col[k]] += w[k]
    assert torch.allclose(AT, A.t())   # fp32 gcn_norm weights on both sides
    conv = oracle.TAGConv(16, 8, K=2).double()
    conv.bias.data.uniform_(-1, 1)
    h = torch.randn(N, 16, dtype=torch.float64, requires_grad=True)
    ei2, _ = oracle.undirect_graph(b["edge_index"], b["edge_attr"][:, :6])
    ref = conv(h, ei2)
    W = [l.weight.detach() for l in conv.lins]
    G = [h.detach() @ Wm.t() for Wm in W]
    out = G[0] + A @ (G[1] + A @ G[2]) + conv.bias.detach()
    assert torch.allclose(out, ref.detach(), atol=2e-6)
    g = torch.randn_like(ref)
    ref.backward(g)
    D = [g @ Wm for Wm in W]
    dh = D[0] + AT @ (D[1] + AT @ D[2])
    assert torch.allclose(dh, h.grad, atol=2e-6)
    Z = [g, AT @ g, AT @ (AT @ g)]
    for m in range(3):
        assert torch.allclose(Z[m].t() @ h.detach(), conv.lins[m].weight.grad, atol=2e-6)
    assert torch.allclose(g.sum(0), conv.bias.grad, atol=2e-6)


def test_edge_aggregation_aggregate_then_linear(oracle):
    """sum_e (W2 relu(z_e) + b2) == W2 (sum_e relu(z_e)) + deg * b2, with the CSR's flip flags."""
    pkg = load_pkg()
    torch.manual_seed(1)
    b = pkg.synthetic.make_batch(["cigre14"], 3, seed=5)
    N = b["x"].shape[0]
    topo = topo_oracle.TopologyOracle(b["edge_index"], N)
    ea_mod = oracle.EdgeAggregation(8, 6, 16, 16).double()
    x, ea = b["x"][:, :8].double(), b["edge_attr"][:, :6].double()
    ei2, ea2 = oracle.undirect_graph(b["edge_index"], ea)
    ref = ea_mod(x, ei2, ea2).detach()
    W1, b1 = ea_mod.edge_aggr[0].weight.detach(), ea_mod.edge_aggr[0].bias.detach()
    W2, b2 = ea_mod.edge_aggr[2].weight.detach(), ea_mod.edge_aggr[2].bias.detach()
    rp, col, ent = topo.rowptr.numpy(), topo.col.numpy(), topo.ent.numpy()
    S = torch.zeros(N, 16, dtype=torch.float64)
    for i in range(N):
        for k in range(rp[i], rp[i + 1]):
            en = int(ent[k]); eid = en & 0x7fffffff; sgn = -1.0 if en < 0 else 1.0
            a = ea[eid].clone(); a[0] *= sgn; a[2] *= sgn
            S[i] += torch.relu(W1 @ torch.cat([x[i], x[col[k]], a]) + b1)
    out = S @ W2.t() + topo.deg.double()[:, None] * b2
    assert torch.allclose(out, ref, atol=1e-12)
