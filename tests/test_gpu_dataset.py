"""GPU: device-side measurement model, masked z-score, collation and loader (dataset.py over the C ABI) against
the reference's data_from_pickles output (golden dataset64.npz) and the dataset oracle."""
import numpy as np
import pytest
import torch

from conftest import golden, t

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _noise(g):
    return {str(k): float(v) for k, v in zip(g["noise_keys"], g["noise_vals"])}


@pytest.fixture(scope="module")
def built(pkg):
    g = golden("dataset64.npz")
    ds, *stats = pkg.dataset.data_from_tables(g["nodes"], g["edges"], g["labels"], _noise(g), 8, 6, g["meas_v"],
                                              g["meas_pflow"], device=DEV, z_nodes=g["z_nodes"], z_edges=g["z_edges"])
    return g, ds, stats


def test_device_pipeline_matches_reference_data_from_pickles(pkg, built):
    g, ds, stats = built
    x, ea = ds.x.reshape(-1, 11).cpu(), ds.edge_attr.reshape(-1, 13).cpu()
    gx, gea = t(g["x"]), t(g["edge_attr"])
    # raw parameter columns and labels are copies: exact
    assert torch.equal(x[:, 8:], gx[:, 8:]) and torch.equal(ea[:, 6:], gea[:, 6:]) and torch.equal(ds.y.reshape(-1, 2).cpu(), t(g["y"]))
    # A column whose measured entries are all the same number has zero variance; the reference's fp32 column sum
    # (960 rows, torch's CPU summation order) is off by one ulp there, so its "std" is that ulp and its z-scores
    # are +-1 of pure rounding noise (cov_theta at the slack bus, column 3).  The device path sums exactly and
    # returns 0 for such a column (std 0 -> nan -> 0, the reference's own rule, data.py:181-182).
    degenerate = [c for c in range(8) if len(torch.unique(gx[:, c][gx[:, c] != 0])) == 1]
    assert degenerate == [3]
    assert (x[:, 3] == 0).all() and stats[1][3].item() == 0.0
    keep = [c for c in range(8) if c not in degenerate]
    # zeros ("not measured") are preserved exactly; measured entries agree to fp32 rounding of the statistics
    assert torch.equal(x[:, keep] == 0, gx[:, keep] == 0) and torch.equal(ea[:, :6] == 0, gea[:, :6] == 0)
    assert torch.allclose(x[:, keep], gx[:, keep], rtol=1e-5, atol=1e-5)
    assert torch.allclose(ea[:, :6], gea[:, :6], rtol=1e-5, atol=1e-6)
    for mine, key in zip(stats, ("x_mean", "x_std", "edge_mean", "edge_std")):
        sel = keep if key == "x_std" else slice(None)
        assert torch.allclose(mine.cpu()[sel], t(g[key])[sel], rtol=1e-6, atol=0), key


def test_measurement_kernels_are_bit_exact_before_normalisation(pkg, built, oracle):
    """The un-normalised features are float64 arithmetic rounded once: identical bits to numpy."""
    import ctypes as C
    import dss2_dataset_oracle as dso
    g = golden("dataset64.npz")
    S, n, e = 64, 15, 14
    L = pkg._lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    nd, zn = t(g["nodes"], device=DEV).reshape(-1, 7), t(g["z_nodes"], device=DEV).reshape(-1, 4)
    ed, ze = t(g["edges"], device=DEV).reshape(-1, 11), t(g["z_edges"], device=DEV).reshape(-1, 2)
    mv = torch.zeros(n, dtype=torch.uint8); mv[t(g["meas_v"])] = 1
    mp = torch.zeros(e, dtype=torch.uint8); mp[t(g["meas_pflow"])] = 1
    x = torch.empty(S * n, 11, device=DEV); ea = torch.empty(S * e, 13, device=DEV)
    nz = _noise(g)
    pkg._lib.check(L.dss2_measure_nodes(nd.data_ptr(), mv.to(DEV).data_ptr(), n, zn.data_ptr(), nz["v_noise"], nz["pm_noise"],
                                        nz["p_noise"], nz["zero_inj_coef"], x.data_ptr(), S * n, st), "measure_nodes")
    pkg._lib.check(L.dss2_measure_edges(ed.data_ptr(), mp.to(DEV).data_ptr(), e, ze.data_ptr(), nz["p_noise"], ea.data_ptr(),
                                        S * e, st), "measure_edges")
    for i in (0, 17, 63):
        xo = dso.measure_nodes(g["nodes"][i, :, 0:4], g["nodes"][i, :, 5], g["nodes"][i, :, 6], g["meas_v"], nz, g["z_nodes"][i])
        eo = dso.measure_edges(g["edges"][i, :, 2:4], g["edges"][i, :, 4:6], g["meas_pflow"], nz, g["z_edges"][i])
        assert torch.equal(x[i * n:(i + 1) * n, :8].cpu(), xo)
        assert torch.equal(ea[i * e:(i + 1) * e, :6].cpu(), eo)


def test_masked_zscore_kernel_edge_cases(pkg):
    a = torch.tensor([[0.0, 2.0, 5.0], [0.0, 4.0, 5.0], [0.0, 0.0, 5.0]], device=DEV)
    out, mean, std = pkg.dataset.masked_zscore(a.clone(), 2)
    assert mean.tolist() == [0.0, 3.0] and std.tolist() == [0.0, 1.0]
    assert out.cpu().tolist() == [[0.0, -1.0, 5.0], [0.0, 1.0, 5.0], [0.0, 0.0, 5.0]]
    # large, ragged row count: against torch in float64
    torch.manual_seed(0)
    b = torch.randn(100_003, 13, device=DEV) * (torch.rand(100_003, 13, device=DEV) > 0.3)
    o, m, s = pkg.dataset.masked_zscore(b.clone(), 6)
    b64, mask = b.double()[:, :6], (b[:, :6] != 0)
    m64 = (b64 * mask).sum(0) / mask.sum(0)
    s64 = torch.sqrt((((b64 - m64) ** 2) * mask).sum(0) / mask.sum(0))
    assert torch.allclose(m.double(), m64, rtol=1e-6, atol=1e-9) and torch.allclose(s.double(), s64, rtol=1e-6)
    assert torch.allclose(o[:, :6].double(), (b64 - m64) * mask / s64, rtol=1e-5, atol=1e-6)
    assert torch.equal(o[:, 6:], b[:, 6:])
    o2, _, _ = pkg.dataset.masked_zscore(b.clone(), 6)
    assert torch.equal(o, o2)                                   # deterministic


def test_loader_collates_like_the_reference_loader(pkg, built):
    import dss2_dataset_oracle as dso
    g, ds, _ = built
    n, e = 15, 14
    loader = pkg.dataset.DataLoader(ds, batch_size=24, shuffle=False)
    assert len(loader) == 3
    sizes, first_ei = [], None
    for k, b in enumerate(loader):
        B = b.num_graphs
        sizes.append(B)
        lo = k * 24
        xs = [ds.x[s].cpu() for s in range(lo, lo + B)]
        eis = [ds.edge_index[s].cpu() for s in range(lo, lo + B)]
        x, ei, ea, y = dso.collate(xs, eis, [ds.edge_attr[s].cpu() for s in range(lo, lo + B)], [ds.y[s].cpu() for s in range(lo, lo + B)])
        assert torch.equal(b.x.cpu(), x) and torch.equal(b.edge_index.cpu(), ei)
        assert torch.equal(b.edge_attr.cpu(), ea) and torch.equal(b.y.cpu(), y)
        if k == 0:
            first_ei = b.edge_index
        elif B == 24:
            assert b.edge_index is first_ei          # one topology: the batch edge list is built once per batch size
    assert sizes == [24, 24, 16]
    # the full collated set equals the reference's own collated batch
    full = next(iter(pkg.dataset.DataLoader(ds, batch_size=64)))
    assert torch.equal(full.edge_index.cpu(), t(g["edge_index"]))
    # shuffling: every sample exactly once per epoch, a different order
    gen = torch.Generator(device=DEV); gen.manual_seed(3)
    rows = torch.cat([b.y for b in pkg.dataset.DataLoader(ds, batch_size=10, shuffle=True, generator=gen)]).cpu()
    ref = ds.y.reshape(-1, 2).cpu()
    assert rows.shape == ref.shape and not torch.equal(rows, ref)
    w = torch.rand(30, dtype=torch.float64, generator=torch.Generator().manual_seed(0))
    key = lambda a: a.reshape(64, -1)[torch.argsort(a.reshape(64, -1).double() @ w)]     # canonical sample order
    assert torch.equal(key(rows), key(ref))
    # list-like use of the dataset (dss2_run.py:59-66)
    sub = ds.shuffled(gen)[0:57]
    assert len(sub) == 57 and len(ds[57:]) == 7 and ds[3].x.shape == (15, 11)


def test_training_from_the_device_loader(pkg, built):
    """dss2_run.py:131-147 on batches that never left the device."""
    g, ds, stats = built
    torch.manual_seed(0)
    model = pkg.MPN(8, 6, 2, 32, 2, 2, 0.0).to(DEV)
    opt = pkg.FusedAdamax(model.parameters(), lr=3e-3)
    losses = [pkg.runner.train_epoch(model, opt, pkg.dataset.DataLoader(ds, batch_size=16, shuffle=True), tuple(stats),
                                     pkg.runner.REG_COEFS) for _ in range(6)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    m = pkg.runner.evaluate(model, pkg.dataset.DataLoader(ds, batch_size=32), tuple(stats))
    assert all(np.isfinite(v) for v in m.values())


def test_device_side_evaluation_metrics(pkg, oracle, built):
    """dss2_run.py:178-208: the ten per-batch test metrics accumulated on the device (dss2_eval_batch) against the
    oracle's torch restatement, on real CIGRE-14 samples."""
    g, ds, stats = built
    torch.manual_seed(1)
    model = pkg.MPN(8, 6, 2, 32, 2, 2, 0.0).to(DEV)
    acc = torch.zeros(10, dtype=torch.float64, device=DEV)
    want = {k: 0.0 for k in pkg.data.EVAL_METRICS}
    n = 0
    with torch.no_grad():
        for b in pkg.dataset.DataLoader(ds, batch_size=24):
            out = model(b.x[:, :8], b.edge_index, b.edge_attr[:, :6])
            yhat = pkg.data.eval_batch(out, b.y, b.x, b.edge_index, b.edge_attr, stats[0], stats[1], acc)
            m, yh = oracle.eval_batch_metrics(out.cpu().double(), b.y.cpu().double(), b.x.cpu().double(), b.edge_index.cpu(),
                                              b.edge_attr.cpu().double(), stats[0].cpu().double(), stats[1].cpu().double())
            assert torch.allclose(yhat.cpu().double(), yh, rtol=1e-6, atol=1e-7)
            for k in want:
                want[k] += m[k]
            n += 1
    got = dict(zip(pkg.data.EVAL_METRICS, acc.cpu().tolist()))
    for k in want:
        assert abs(got[k] - want[k]) <= 2e-5 * abs(want[k]) + 1e-9, (k, got[k], want[k])
    ev = pkg.runner.evaluate(model, pkg.dataset.DataLoader(ds, batch_size=24), tuple(stats))
    for k in want:
        assert abs(ev[k] - want[k] / n) <= 2e-5 * abs(want[k] / n) + 1e-9, k


def _mixed(pkg, S=96, seed=5):
    """Two cases with the same buses and different closed-branch sets (14 / 15 branches), normalised with common stats."""
    full = pkg.synthetic.make_batch(["cigre14", "cigre14_reswitched"], 64, seed=seed)
    parts_h = [pkg.synthetic.make_batch([g], S, seed=seed + 1 + k, stats=full["stats"]) for k, g in enumerate(["cigre14", "cigre14_reswitched"])]
    parts = [pkg.dataset.DeviceDataset.from_batch(p, device=DEV) for p in parts_h]
    return parts_h, parts, tuple(s.to(DEV) for s in full["stats"])


def test_mixed_topology_loader_collates_without_host_sync(pkg, oracle):
    """BASELINE config C5's defining property: a new mix of 14- and 15-branch graphs in every batch.  The host picks the
    composition, the device moves the data and builds the graph structure; nothing is read back (sync-debug "error")."""
    parts_h, parts, stats = _mixed(pkg)
    ds = pkg.dataset.MixedDataset(parts)
    assert len(ds) == 192 and [p.e for p in parts] == [14, 15] and all(p.directed for p in parts)
    gen = torch.Generator().manual_seed(3)
    loader = pkg.dataset.DataLoader(ds, batch_size=80, shuffle=True, generator=gen)
    model = pkg.MPN(8, 6, 2, 32, 3, 2, 0.0).to(DEV)
    first = next(iter(loader))
    model(first.x[:, :8], first.edge_index, first.edge_attr[:, :6]).sum().backward()      # warm-up (plans, allocator)
    torch.cuda.synchronize()
    seen, batches = [], []
    torch.cuda.set_sync_debug_mode("error")
    try:
        for bt in loader:
            out = model(bt.x[:, :8], bt.edge_index, bt.edge_attr[:, :6])
            loss = pkg.gsp_wls_edge(input=bt.x[:, :8], edge_input=bt.edge_attr[:, :6], output=out, x_mean=stats[0], x_std=stats[1],
                                    edge_mean=stats[2], edge_std=stats[3], edge_index=bt.edge_index,
                                    reg_coefs=oracle.DEFAULT_REG_COEFS, num_samples=bt.num_graphs, node_param=bt.x[:, 8:],
                                    edge_param=bt.edge_attr[:, 6:])
            loss.backward()
            batches.append(bt)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    assert [b.num_graphs for b in batches] == [80, 80, 32]
    # every batch is the PyG collation of its samples: rebuild it on the host from the per-sample stores
    rng = np.random.default_rng(int(gen.initial_seed()))
    rng.permutation(np.arange(192))                      # the warm-up iteration above drew the first permutation
    ids_all = rng.permutation(np.arange(192))
    n = 15
    for k, bt in enumerate(batches):
        ids = ids_all[80 * k:80 * k + 80]
        xs, eas, ys, eis = [], [], [], []
        for slot, g in enumerate(ids):
            p, s = (0, g) if g < 96 else (1, g - 96)
            h, e = parts_h[p], parts[p].e
            xs.append(h["x"][s * n:(s + 1) * n]); ys.append(h["y"][s * n:(s + 1) * n])
            eas.append(h["edge_attr"][s * e:(s + 1) * e])
            eis.append(h["edge_index"][:, s * e:(s + 1) * e] - s * n + slot * n)
        assert torch.equal(bt.x.cpu(), torch.cat(xs)) and torch.equal(bt.y.cpu(), torch.cat(ys))
        assert torch.equal(bt.edge_attr.cpu(), torch.cat(eas)) and torch.equal(bt.edge_index.cpu(), torch.cat(eis, 1))
        seen.extend(ids.tolist())
        # the structure attached to the batch equals a fresh un-hinted build of the same edge list
        topo = pkg.topology.get_topology(bt.edge_index, bt.x.shape[0])
        assert topo.hint is not None and topo.stats()["error"] == 0
        ref = pkg.topology.Topology(bt.edge_index.clone(), bt.x.shape[0])
        for f in ["rowptr", "col", "ent", "w", "rowptrT", "colT", "entT", "wT", "inc_rowptr", "inc_ent", "tile_start", "ell_tiles",
                  "ellT_tiles", "ell_ent_tiles", "ellT_ent_tiles"]:
            assert torch.equal(getattr(topo, f), getattr(ref, f)), f
    assert sorted(seen) == list(range(192))


def test_mixed_topology_batch_step_matches_the_oracle(pkg, oracle):
    parts_h, parts, stats = _mixed(pkg, S=24)
    ds = pkg.dataset.MixedDataset(parts)
    bt = next(iter(pkg.dataset.DataLoader(ds, batch_size=48, shuffle=True, generator=torch.Generator().manual_seed(9))))
    torch.manual_seed(0)
    ref = oracle.MPN(8, 6, 2, 64, 3, 2, 0.0).double()          # fp64 referee, like every parity test
    mine = pkg.MPN(8, 6, 2, 64, 3, 2, 0.0)
    mine.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    mine = mine.to(DEV)
    hb = {"x": bt.x.cpu().double(), "edge_index": bt.edge_index.cpu(), "edge_attr": bt.edge_attr.cpu().double()}
    out_r, loss_r = oracle.train_step(ref, hb, tuple(s.cpu().double() for s in stats))
    out = mine(bt.x[:, :8], bt.edge_index, bt.edge_attr[:, :6])
    loss = pkg.gsp_wls_edge(input=bt.x[:, :8], edge_input=bt.edge_attr[:, :6], output=out, x_mean=stats[0], x_std=stats[1],
                            edge_mean=stats[2], edge_std=stats[3], edge_index=bt.edge_index, reg_coefs=oracle.DEFAULT_REG_COEFS,
                            num_samples=None, node_param=bt.x[:, 8:], edge_param=bt.edge_attr[:, 6:])
    loss.backward()
    assert (out.detach().cpu().double() - out_r).abs().max() <= 1e-5 * out_r.abs().max()
    # the loss of this small mixed batch is ill-conditioned (large terms cancelling): the reference's own fp32 evaluation is
    # 1-2e-5 away from the fp64 one, so the bound is "1e-5, or no worse than twice the fp32 reference's own distance"
    ref32 = oracle.MPN(8, 6, 2, 64, 3, 2, 0.0)
    ref32.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    hb32 = {"x": bt.x.cpu(), "edge_index": bt.edge_index.cpu(), "edge_attr": bt.edge_attr.cpu()}
    with torch.no_grad():
        o32 = ref32(hb32["x"][:, :8], hb32["edge_index"], hb32["edge_attr"][:, :6])
        st32 = tuple(s.cpu() for s in stats)
        l32 = oracle.gsp_wls_edge(input=hb32["x"][:, :8], edge_input=hb32["edge_attr"][:, :6], output=o32, x_mean=st32[0], x_std=st32[1],
                                  edge_mean=st32[2], edge_std=st32[3], edge_index=hb32["edge_index"], reg_coefs=oracle.DEFAULT_REG_COEFS,
                                  num_samples=None, node_param=hb32["x"][:, 8:], edge_param=hb32["edge_attr"][:, 6:])
    assert abs(loss.item() - loss_r.item()) <= max(1e-5 * abs(loss_r.item()), 2.0 * abs(l32.item() - loss_r.item()))
    tol = max(1e-4, 3.0 / bt.x.shape[0])        # un-pinned ReLU gates: one flipped gate moves a gradient row by ~1/N_nodes
    for (n_, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        assert (p.grad.cpu().double() - q.grad).abs().max() <= tol * q.grad.abs().max(), n_
