#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself.

Runs ONLY in the build container (needs /root/reference).  It imports the reference's own,
unmodified ``networks.py`` and ``data.py`` from /root/reference; because ``torch_geometric`` is
not installed there, the nine PyG names those files import come from the stand-in in
``tests/golden/_pyg_standin`` (written from PyG's published definitions).  Outputs are data
only: inputs, explicit weights and the reference's outputs / loss / gradients, as .npz.

    python tests/golden/make_goldens.py

Fixtures written:
    <pkg>/grids.npz                     grid parameter tables (bus_param / edge_param) of the 3 grids
    tests/golden/cigre14_real64.npz     64 real CIGRE-14 samples through data_from_pickles (seeded)
    tests/golden/physics_known.npz      pandapower branch/bus results for 5 samples (known answers)
    tests/golden/case_<name>.npz        model / loss cases (weights, inputs, reference outputs)
    tests/golden/dataset64.npz          raw tables of 64 real CIGRE-14 samples, the standard-normal draws
                                        the reference consumed, and its data_from_pickles output on them
                                        (python tests/golden/make_goldens.py dataset64 writes only this one)
    tests/golden/case_mpn_{c2model,ober_h128,ober179_h128}.npz   the C2 model at the three tile heights
                                        (python tests/golden/make_goldens.py flagship)
"""
import os
import pickle
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
PKG = os.path.join(ROOT, "deep-statistical-solver-for-distribution-system-state-estimation_amd")

sys.path.insert(0, os.path.join(HERE, "_pyg_standin"))
sys.path.insert(0, REF)
import networks as ref_networks  # noqa: E402  (the reference's file, unmodified)
import data as ref_data          # noqa: E402  (the reference's file, unmodified)
from torch_geometric.loader import collate  # noqa: E402  (stand-in)

sys.path.insert(0, PKG)
import synthetic  # noqa: E402

REG = {"mu_v": 1e-1, "mu_theta": 1e-1, "lam_v": 1e-4, "lam_p": 1e-8, "lam_pf": 1e-6, "lam_reg": 1e2}


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


def grids():
    out = {}
    for g in ["cigre14", "cigre14_reswitched", "ober_sub"]:
        bp = pickle.load(open(f"{REF}/data/{g}/bus_param", "rb"))
        ep = pickle.load(open(f"{REF}/data/{g}/edge_param", "rb"))
        out[f"{g}/bus_param"] = bp.values.astype(np.float64)
        out[f"{g}/edge_param"] = ep.values.astype(np.float64)
    np.savez_compressed(os.path.join(PKG, "grids.npz"), **out)
    print("wrote grids.npz")


def real_cigre():
    np.random.seed(0)
    torch.manual_seed(0)
    meas_v, meas_pf = np.array([0, 1, 12, 7, 11, 14]), np.array([0, 10])
    ds, x_mean, x_std, e_mean, e_std = ref_data.data_from_pickles(f"{REF}/data/cigre14/", 8, 6, 4, 2, meas_v, meas_pf)
    b = collate(ds[:64])
    save("cigre14_real64.npz", x=b.x, edge_index=b.edge_index, edge_attr=b.edge_attr, y=b.y,
         x_mean=x_mean, x_std=x_std, edge_mean=e_mean, edge_std=e_std)
    return dict(x=b.x, edge_index=b.edge_index, edge_attr=b.edge_attr, y=b.y,
                stats=(x_mean, x_std, e_mean, e_std))


def dataset64():
    """data_from_pickles on the first 64 CIGRE-14 samples (pickles truncated into a temp folder), with the
    raw tables and the standard-normal draws saved beside the reference's output."""
    import tempfile
    src = f"{REF}/data/cigre14/"
    tmp = tempfile.mkdtemp(prefix="dss2_ds64_") + "/"
    tabs = {}
    for name in ["nodes", "edges", "labels"]:
        tabs[name] = pickle.load(open(src + name, "rb"))[:64]
        pickle.dump(tabs[name], open(tmp + name, "wb"))
    noise_df = pickle.load(open(src + "noise_param", "rb"))
    pickle.dump(noise_df, open(tmp + "noise_param", "wb"))
    meas_v, meas_pf = np.array([0, 1, 12, 7, 11, 14]), np.array([0, 10])
    np.random.seed(7)
    ds, x_mean, x_std, e_mean, e_std = ref_data.data_from_pickles(tmp, 8, 6, 4, 2, meas_v, meas_pf)
    b = collate(ds)
    # the draws the reference consumed: per sample normal(0, |x_std|) [n, 4], then normal(0, |e_std|) [e, 2]
    np.random.seed(7)
    zn, ze = [], []
    for i in range(64):
        zn.append(np.random.standard_normal([15, 4]))
        ze.append(np.random.standard_normal([14, 2]))
    ncols = ["vm_pu", "va_rad", "p_mw", "q_mvar", "vn_kv", "bool_slack", "bool_zero_inj"]
    ecols = ["from_bus", "to_bus", "p_from_mw", "q_from_mvar", "G", "B", "Gs", "Bs", "closed line", "phase shift",
             "imax or sn"]
    nodes = np.stack([t[ncols].values.astype(np.float64) for t in tabs["nodes"]])
    edges = np.stack([t[t["closed line"] == 1.0][ecols].values.astype(np.float64) for t in tabs["edges"]])
    labels = np.stack([t.values.astype(np.float64) for t in tabs["labels"]])
    noise = {k: float(noise_df[k].values[0]) for k in ["p_noise", "v_noise", "pm_noise", "zero_inj_coef"]}
    # the restatement must reproduce the reference from these inputs (draw order / shapes are right)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dss2_dataset_oracle as dso
    o = dso.data_from_tables(nodes, edges, labels, noise, meas_v, meas_pf, np.stack(zn), np.stack(ze))
    assert torch.equal(o["x"], b.x) and torch.equal(o["edge_attr"], b.edge_attr) and torch.equal(o["y"], b.y)
    assert torch.equal(o["x_mean"], x_mean) and torch.equal(o["edge_std"], e_std)
    save("dataset64.npz", nodes=nodes, edges=edges, labels=labels, z_nodes=np.stack(zn), z_edges=np.stack(ze),
         meas_v=meas_v, meas_pflow=meas_pf, noise_keys=np.array(list(noise.keys())),
         noise_vals=np.array(list(noise.values())), x=b.x, edge_index=b.edge_index, edge_attr=b.edge_attr, y=b.y,
         x_mean=x_mean, x_std=x_std, edge_mean=e_mean, edge_std=e_std)


def physics_known():
    nodes = pickle.load(open(f"{REF}/data/cigre14/nodes", "rb"))
    edges = pickle.load(open(f"{REF}/data/cigre14/edges", "rb"))
    labels = pickle.load(open(f"{REF}/data/cigre14/labels", "rb"))
    idx = [0, 1, 100, 400, 719]
    arr = {}
    for k, i in enumerate(idx):
        ce = edges[i][edges[i]["closed line"] == 1.0]
        arr[f"s{k}/labels"] = labels[i].values
        arr[f"s{k}/node_param"] = nodes[i][["vn_kv", "bool_slack", "bool_zero_inj"]].values
        arr[f"s{k}/bus_pq"] = nodes[i][["p_mw", "q_mvar"]].values
        arr[f"s{k}/edge_index"] = ce[["from_bus", "to_bus"]].values.astype(np.int64).T
        arr[f"s{k}/edge_param"] = ce[["G", "B", "Gs", "Bs", "closed line", "phase shift", "imax or sn"]].values
        arr[f"s{k}/branch"] = ce[["p_from_mw", "q_from_mvar", "p_to_mw", "q_to_mvar", "i_from_ka", "i_to_ka",
                                   "loading_percent"]].values
    save("physics_known.npz", **arr)


def run_case(name, model, batch, with_loss=True, seed=0):
    """Reference forward (+ gsp_wls_edge + backward) with explicit weights; saves everything."""
    torch.manual_seed(seed)
    with torch.no_grad():  # non-trivial TAGConv biases (PyG initialises them to zero)
        for n, p in model.named_parameters():
            if n.endswith("bias") and "convs" in n:
                p.uniform_(-0.2, 0.2)
    x, ei, ea = batch["x"], batch["edge_index"], batch["edge_attr"]
    st = batch["stats"]
    arrays = {f"param/{k}": v.clone() for k, v in model.state_dict().items()}
    arrays.update(x=x, edge_index=ei, edge_attr=ea, x_mean=st[0], x_std=st[1], edge_mean=st[2], edge_std=st[3])
    out = model(x[:, :8], ei, ea[:, :6])
    arrays["out"] = out.detach().clone()  # before gsp_wls_edge mutates it
    if with_loss:
        loss = ref_data.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=out, x_mean=st[0], x_std=st[1],
                                     edge_mean=st[2], edge_std=st[3], edge_index=ei, reg_coefs=REG,
                                     num_samples=None, node_param=x[:, 8:], edge_param=ea[:, 6:])
        arrays["out_after_loss"] = out.detach().clone()  # theta zeroed at the slack (data.py:413)
        arrays["loss"] = loss.detach().clone()
        loss.backward()
    else:
        torch.manual_seed(seed + 1)
        gout = torch.randn_like(out)
        arrays["gout"] = gout
        out.backward(gout)
    for k, p in model.named_parameters():
        arrays[f"grad/{k}"] = p.grad.clone()
    save(f"case_{name}.npz", **arrays)


def loss_case(name, batch, seed=0):
    """gsp_wls_edge alone on a given `output`; expects loss, d loss/d output and the mutation."""
    torch.manual_seed(seed)
    x, ei, ea, st = batch["x"], batch["edge_index"], batch["edge_attr"], batch["stats"]
    # model-like output: normalised V around 0, theta in rad
    o_leaf = torch.stack([torch.randn(x.shape[0]) * 1.5, batch["y"][:, 1] * 3 + 0.02 * torch.randn(x.shape[0])], 1)
    o_leaf.requires_grad_(True)
    o = o_leaf * 1.0
    loss = ref_data.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=o, x_mean=st[0], x_std=st[1],
                                 edge_mean=st[2], edge_std=st[3], edge_index=ei, reg_coefs=REG,
                                 num_samples=None, node_param=x[:, 8:], edge_param=ea[:, 6:])
    loss.backward()
    yv = torch.cat([o.detach()[:, 0:1] * st[1][:1] + st[0][:1], o.detach()[:, 1:]], 1)
    flows = ref_data.get_pflow(yv, ei, x[:, 8:], ea[:, 6:])
    save(f"case_{name}.npz", x=x, edge_index=ei, edge_attr=ea, x_mean=st[0], x_std=st[1], edge_mean=st[2],
         edge_std=st[3], output=o_leaf.detach(), output_after=o.detach(), loss=loss.detach(),
         grad_output=o_leaf.grad, pflow=torch.stack(flows, 1))


def extras():
    """Fixtures added after the first set (the first set is not regenerated: python make_goldens.py extras).
    case_pflow_shift.npz: the reference's get_pflow(..., phase_shift=False) (shift = edge_param[:, 5], data.py:364-365) on
    the inputs of the three committed loss cases."""
    arrays = {}
    for name in ["loss_real", "loss_violate_cigre", "loss_violate_ober"]:
        z = np.load(os.path.join(HERE, f"case_{name}.npz"))
        x, ei, ea = torch.from_numpy(z["x"]), torch.from_numpy(z["edge_index"]), torch.from_numpy(z["edge_attr"])
        o = torch.from_numpy(z["output_after"])
        yv = torch.cat([o[:, 0:1] * torch.from_numpy(z["x_std"])[:1] + torch.from_numpy(z["x_mean"])[:1], o[:, 1:]], 1)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")      # data.py:365 wraps a tensor in torch.tensor(...)
            flows = ref_data.get_pflow(yv, ei, x[:, 8:], ea[:, 6:], phase_shift=False)
        arrays[f"{name}/pflow_shift"] = torch.stack(flows, 1)
    save("case_pflow_shift.npz", **arrays)
    # ---- the per-layer-interleaved variants (networks.py:390-735): forward(data) + backward of a random output gradient
    import types
    z = np.load(os.path.join(HERE, "cigre14_real64.npz"))
    x11, ei, ea13 = torch.from_numpy(z["x"]), torch.from_numpy(z["edge_index"]), torch.from_numpy(z["edge_attr"])
    feats = x11[:, :8]
    torch.manual_seed(40)
    types_onehot = torch.nn.functional.one_hot(torch.randint(0, 4, (x11.shape[0],)), 4).float()
    x_masked = torch.cat([types_onehot, feats, (feats != 0).float()], dim=1)           # [N, 4 + 8 + 8]
    feats8 = torch.cat([types_onehot, feats, (feats != 0).float()], dim=1)
    N = ref_networks
    variants = {
        "multimpn": (N.MultiMPN, (8, 6, 2, 32, 3, 2, 0.0), feats),
        "multimpn_h128": (N.MultiMPN, (8, 6, 2, 128, 2, 2, 0.0), feats),
        "maskembdmpn": (N.MaskEmbdMPN, (8, 6, 2, 32, 3, 2, 0.0), x_masked),
        "maskembdmultimpn": (N.MaskEmbdMultiMPN, (8, 6, 2, 32, 2, 2, 0.0), x_masked),
        "maskembdmultimpn_nomp": (N.MaskEmbdMultiMPN_NoMP, (8, 6, 2, 8, 3, 2, 0.0), x_masked),
    }
    for k, (name, (cls, args, xin)) in enumerate(variants.items()):
        torch.manual_seed(50 + k)
        model = cls(*args)
        with torch.no_grad():
            for n_, p_ in model.named_parameters():
                if n_.endswith("bias") and ("convs" in n_ or ("layers" in n_ and "edge_aggr" not in n_)):
                    p_.uniform_(-0.2, 0.2)          # non-trivial TAGConv biases (PyG initialises them to zero)
        data = types.SimpleNamespace(x=xin, edge_index=ei, edge_attr=ea13[:, :6])
        out = model(data)
        torch.manual_seed(60 + k)
        gout = torch.randn_like(out)
        out.backward(gout)
        arr = {f"param/{kk}": v.clone() for kk, v in model.state_dict().items()}
        arr.update({f"grad/{kk}": p_.grad.clone() for kk, p_ in model.named_parameters()})
        arr.update(x=xin, edge_index=ei, edge_attr=ea13[:, :6], out=out.detach(), gout=gout)
        save(f"case_{name}.npz", **arr)


def flagship():
    """Reference-generated goldens at the flagship shapes (python make_goldens.py flagship): the C2 model
    (dim_hid 128, 4 layers) on the 64 real CIGRE-14 graphs (64-row tiles: split-plane layer chain, bf16x6 weight gradient),
    on ober_sub graphs (96-row tiles) and on the synthetic 179-bus feeder (192-row tiles)."""
    N = ref_networks
    z = np.load(os.path.join(HERE, "cigre14_real64.npz"))
    real = dict(x=torch.from_numpy(z["x"]), edge_index=torch.from_numpy(z["edge_index"]), edge_attr=torch.from_numpy(z["edge_attr"]),
                stats=tuple(torch.from_numpy(z[k]) for k in ("x_mean", "x_std", "edge_mean", "edge_std")))
    torch.manual_seed(31)
    run_case("mpn_c2model", N.MPN(8, 6, 2, 128, 4, 2, 0.0), real)
    torch.manual_seed(32)
    run_case("mpn_ober_h128", N.MPN(8, 6, 2, 128, 4, 2, 0.0), synthetic.make_batch(["ober_sub"], 6, seed=32))
    torch.manual_seed(33)
    run_case("mpn_ober179_h128", N.MPN(8, 6, 2, 128, 4, 2, 0.0), synthetic.make_batch(["ober179"], 4, seed=33))


def main():
    grids()
    real = real_cigre()
    physics_known()
    N = ref_networks
    torch.manual_seed(1)
    run_case("mpn_c1", N.MPN(8, 6, 2, 32, 1, 2, 0.0), real)
    torch.manual_seed(2)
    run_case("mpn_h64_l3", N.MPN(8, 6, 2, 64, 3, 2, 0.0), real)
    torch.manual_seed(3)
    run_case("skipmpn", N.SkipMPN(8, 6, 8, 32, 2, 2, 0.0), real, with_loss=False)
    torch.manual_seed(4)
    run_case("pfn", N.PFN(8, 6, 2, 32, 2, 2, 0.0, 2), real)
    torch.manual_seed(5)
    run_case("skippfn", N.SkipPFN(8, 6, 2, 32, 2, 2, 0.0, 3), real)
    resw = synthetic.make_batch(["cigre14_reswitched"], 8, seed=11)
    torch.manual_seed(6)
    run_case("mpn_resw_k3", N.MPN(8, 6, 2, 32, 2, 3, 0.0), resw)
    ober = synthetic.make_batch(["ober_sub"], 4, seed=12)
    torch.manual_seed(7)
    run_case("mpn_ober", N.MPN(8, 6, 2, 32, 2, 2, 0.0), ober)
    mixed = synthetic.make_batch(["cigre14", "cigre14_reswitched"], 16, seed=13)
    torch.manual_seed(8)
    run_case("mpn_mixed", N.MPN(8, 6, 2, 64, 2, 2, 0.0), mixed)
    # already-undirected input: is_directed() is False, no doubling, no sign flips
    und = dict(real)
    ei, ea = real["edge_index"][:, :14 * 4], real["edge_attr"][:14 * 4]
    und.update(x=real["x"][:60], y=real["y"][:60], edge_index=torch.cat([ei, ei.flip(0)], 1),
               edge_attr=torch.cat([ea, ea], 0))
    torch.manual_seed(9)
    run_case("mpn_undirected_input", N.MPN(8, 6, 2, 32, 2, 2, 0.0), und, with_loss=False)
    # loss-only cases, including states that activate the three penalty terms
    loss_case("loss_real", real, seed=20)
    loss_case("loss_violate_cigre", synthetic.make_batch(["cigre14"], 32, seed=21, violate=0.5), seed=21)
    loss_case("loss_violate_ober", synthetic.make_batch(["ober_sub"], 6, seed=22, violate=0.5), seed=22)
    # structural facts
    m = N.MPN(8, 6, 2, 32, 2, 2, 0.0)
    facts = {
        "is_directed_real": bool(m.is_directed(real["edge_index"])),
        "is_directed_undirected_input": bool(m.is_directed(und["edge_index"])),
        "trafo_pos_cigre_max": float(torch.ceil(real["edge_attr"][:, 11]).max()),
        "trafo_pos_ober_max": float(torch.ceil(ober["edge_attr"][:, 11]).max()),
    }
    # eval-time dropout quirk (networks.py:268): two eval() calls differ when p > 0
    md = N.MPN(8, 6, 2, 32, 2, 2, 0.3).eval()
    with torch.no_grad():
        a = md(real["x"][:, :8], real["edge_index"], real["edge_attr"][:, :6])
        b = md(real["x"][:, :8], real["edge_index"], real["edge_attr"][:, :6])
    facts["eval_dropout_differs"] = bool((a - b).abs().max() > 0)
    facts["state_dict_keys_mpn"] = np.array(list(N.MPN(8, 6, 2, 32, 2, 2, 0.0).state_dict().keys()))
    facts["state_dict_keys_skippfn"] = np.array(list(N.SkipPFN(8, 6, 2, 32, 2, 2, 0.0, 2).state_dict().keys()))
    save("facts.npz", **facts)
    print(facts)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "dataset64":
        dataset64()
    elif len(sys.argv) > 1 and sys.argv[1] == "extras":
        extras()
    elif len(sys.argv) > 1 and sys.argv[1] == "flagship":
        flagship()
    else:
        main()
