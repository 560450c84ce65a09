"""CPU: host-side logic of the product package that needs no GPU - the C-ABI library loads and
exports every declared symbol, topology building (CSR, incidence, tiles), interface parity
(state_dict keys), loud failure without a GPU."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, golden, load_pkg, t


def test_library_exports_every_declared_symbol():
    pkg = load_pkg()
    lib = pkg._lib.lib()
    hdr = open(os.path.join(ROOT, "include", "dss2_hip.h")).read()
    declared = set(re.findall(r"\b(dss2_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(pkg._lib.EXPORTED_SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.dss2_version() >= 1
    # pure host helpers may be called without a GPU
    assert lib.dss2_gemm_prop_lds_bytes(2, 3, 128, 4, 120, 3) < 160 * 1024
    assert lib.dss2_gemm_prop_lds_bytes(8, 3, 256, 8, 500, 0) > 160 * 1024


def test_struct_layouts_match_the_header_sizes():
    """ctypes mirrors of the C structs: same sizes as the C compiler's layout (probe compiled with gcc)."""
    import ctypes
    import subprocess
    import tempfile
    pkg = load_pkg()
    src = ('#include <stdio.h>\n#include "dss2_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(dss2_pack_desc), '
           'sizeof(dss2_gemm_prop_args), sizeof(dss2_wgrad_args), sizeof(dss2_wls_args), sizeof(dss2_csr_build_args), '
           'sizeof(dss2_ell_build_args), sizeof(dss2_csr_axpy_args), sizeof(dss2_stack_dims), sizeof(dss2_stack_args), sizeof(dss2_chain_head));return 0;}\n')
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "p.c"), "-o", os.path.join(d, "p")])
        sizes = [int(v) for v in subprocess.check_output([os.path.join(d, "p")]).split()]
    L = pkg._lib
    assert sizes == [ctypes.sizeof(L.PackDesc), ctypes.sizeof(L.GemmPropArgs), ctypes.sizeof(L.WgradArgs), ctypes.sizeof(L.WlsArgs),
                     ctypes.sizeof(L.CsrBuildArgs), ctypes.sizeof(L.EllBuildArgs), ctypes.sizeof(L.CsrAxpyArgs), ctypes.sizeof(L.StackDims),
                     ctypes.sizeof(L.StackArgs), ctypes.sizeof(L.ChainHead)]
    assert pkg.networks._DESC_DTYPE.itemsize == sizes[0]


@pytest.fixture(scope="module")
def topo_oracle():
    import dss2_topology_oracle
    return dss2_topology_oracle


@pytest.mark.parametrize("grids,B", [(["cigre14"], 9), (["cigre14", "cigre14_reswitched"], 33), (["ober_sub"], 5), (["ober179"], 3)])
def test_topology_oracle_structure(oracle, topo_oracle, grids, B):
    """The structure oracle (oracle/dss2_topology_oracle.py: what the device-side build must reproduce bit for bit,
    tests/test_gpu_topology.py) against the path oracle's PyG restatement: doubling, row order, gcn_norm weights."""
    pkg = load_pkg()
    b = pkg.synthetic.make_batch(grids, B, seed=7)
    ei, N = b["edge_index"], b["x"].shape[0]
    E = ei.shape[1]
    topo = topo_oracle.TopologyOracle(ei, N)
    assert topo.directed is True and oracle.is_directed(ei) is True and topo.E2 == 2 * E
    ei2, _ = oracle.undirect_graph(ei, b["edge_attr"][:, :6])
    src, tgt = ei2[0].numpy(), ei2[1].numpy()
    # CSR by target reproduces the doubled edge list; inside a row entries ascend in directed edge id
    rp, col, ent, perm = topo.rowptr.numpy(), topo.col.numpy(), topo.ent.numpy(), topo.perm.numpy()
    rows = np.repeat(np.arange(N), np.diff(rp))
    assert rp[0] == 0 and rp[-1] == topo.E2
    assert (tgt[perm] == rows).all() and (src[perm] == col).all()
    same_row = rows[1:] == rows[:-1]
    assert (np.diff(perm)[same_row] > 0).all()
    assert ((ent & 0x7fffffff) == perm % E).all() and ((ent < 0) == (perm >= E)).all()
    # gcn_norm weights and degrees equal the oracle's, bit for bit
    w_ref = oracle.gcn_norm_no_self_loops(ei2, N, torch.float32)
    assert torch.equal(topo.w, w_ref[topo.perm.long()])
    assert torch.equal(topo.deg, oracle.degree(ei2[1], N))
    # transposed CSR holds the same edges grouped by source
    rpT, colT = topo.rowptrT.numpy(), topo.colT.numpy()
    assert sorted(zip(np.repeat(np.arange(N), np.diff(rpT)).tolist(), colT.tolist())) == sorted(zip(src.tolist(), tgt.tolist()))
    # incidence CSR: every stored edge appears once per end
    irp, ient = topo.inc_rowptr.numpy(), topo.inc_ent.numpy()
    irows = np.repeat(np.arange(N), np.diff(irp))
    e_id, to_end = ient & 0x7fffffff, ient < 0
    assert len(ient) == 2 * E and (np.where(to_end, ei[1].numpy()[e_id], ei[0].numpy()[e_id]) == irows).all()
    # tiles: whole graphs only, within the row budget, covering all rows
    ts = topo.tile_start.numpy()
    assert ts[0] == 0 and ts[-1] == N and (np.diff(ts) > 0).all() and np.diff(ts).max() <= 32 * topo.nrb
    tile_of = np.searchsorted(ts, np.arange(N), side="right") - 1
    assert (tile_of[src] == tile_of[tgt]).all()
    assert topo.max_nnz == max(rp[ts[1:]] - rp[ts[:-1]])


def test_topology_edge_cases(topo_oracle):
    pkg = load_pkg()
    ei = torch.tensor([[0, 1, 1, 2], [1, 0, 2, 1]])      # already undirected: no doubling, no flips
    topo = topo_oracle.TopologyOracle(ei, 4)              # node 3 is isolated
    assert topo.directed is False and topo.E2 == 4 and (topo.ent.numpy() >= 0).all()
    assert topo.deg.tolist() == [1.0, 2.0, 1.0, 0.0] and torch.isfinite(topo.w).all()
    n = 400                                                # a component larger than the biggest tile has no tile part
    chain = torch.stack([torch.arange(n - 1), torch.arange(1, n)])
    assert topo_oracle.TopologyOracle(chain, n).tiled is False
    # the product class validates its input and has no CPU path
    with pytest.raises(ValueError):
        pkg.topology.Topology(torch.zeros(2, 0, dtype=torch.int64), 3)       # empty
    with pytest.raises(ValueError):
        pkg.topology.Topology(torch.zeros(2, 3, dtype=torch.int32), 3)       # wrong dtype
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg.topology.Topology(ei, 4)
    assert pkg.topology.reference_is_directed(ei) is False and pkg.topology.reference_is_directed(chain) is True


def test_interface_parity_and_loud_failure():
    pkg = load_pkg()
    f = golden("facts.npz")
    assert list(pkg.MPN(8, 6, 2, 32, 2, 2, 0.0).state_dict().keys()) == list(f["state_dict_keys_mpn"])
    assert list(pkg.SkipPFN(8, 6, 2, 32, 2, 2, 0.0, 2).state_dict().keys()) == list(f["state_dict_keys_skippfn"])
    g = golden("case_mpn_c1.npz")
    m = pkg.MPN(8, 6, 2, 32, 1, 2, 0.0)
    res = m.load_state_dict({k[6:]: t(v) for k, v in g.items() if k.startswith("param/")})
    assert not res.missing_keys and not res.unexpected_keys
    with pytest.raises(ValueError):
        pkg.SkipMPN(8, 6, 2, 32, 2, 2, 0.0)          # dim_out must equal dim_featn
    x, ei, ea = t(g["x"]), t(g["edge_index"]), t(g["edge_attr"])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(x[:, :8], ei, ea[:, :6])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=torch.zeros(x.shape[0], 2), x_mean=t(g["x_mean"]),
                         x_std=t(g["x_std"]), edge_mean=t(g["edge_mean"]), edge_std=t(g["edge_std"]), edge_index=ei,
                         reg_coefs={"lam_v": 1, "lam_p": 1, "lam_pf": 1, "lam_reg": 1}, num_samples=None,
                         node_param=x[:, 8:], edge_param=ea[:, 6:])
    with pytest.raises(RuntimeError):
        pkg.get_pflow(torch.zeros(x.shape[0], 2), ei, x[:, 8:], ea[:, 6:])


def test_product_package_never_imports_the_oracle():
    pkg_dir = os.path.join(ROOT, "deep-statistical-solver-for-distribution-system-state-estimation_amd")
    for fn in os.listdir(pkg_dir):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg_dir, fn)).read()
            assert "dss2_oracle" not in src and "import oracle" not in src, fn


def test_no_kernel_in_the_library_spills_registers():
    """Every kernel compiled into libdss2_hip.so stays inside its register budget (0 spilled VGPRs, 0 bytes of scratch):
    instantiations that miss it are excluded at compile time (e.g. wgrad_spills in csrc/dss2_wgrad.hip) and their shapes
    are served by a narrower one.  Re-compiles the kernel sources with -Rpass-analysis=kernel-resource-usage (hipcc
    cross-compiles without a GPU; the files are compiled in parallel)."""
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    csrc = os.path.join(ROOT, "deep-statistical-solver-for-distribution-system-state-estimation_amd", "csrc")
    files = sorted(f for f in os.listdir(csrc) if f.endswith(".hip"))
    assert len(files) >= 9

    def scan(fn):
        out = subprocess.run(["python", os.path.join(ROOT, "tools", "kernel_resources.py"), os.path.join(csrc, fn)],
                             capture_output=True, text=True, timeout=900).stdout
        rows = [ln.split() for ln in out.splitlines()[1:] if ln.strip()]
        return fn, [(" ".join(r[:-6]), int(r[-4]), int(r[-3])) for r in rows if r[-4].isdigit()]

    with ThreadPoolExecutor(max_workers=4) as ex:
        results = list(ex.map(scan, files))
    n_kernels = sum(len(k) for _, k in results)
    assert n_kernels > 100, n_kernels
    # ONE documented exception: the whole-stack backward (csrc/dss2_stack.hip).  Its three persistent weight-gradient
    # accumulator slots (48 VGPRs) live across the whole tile loop of a 256-register kernel (8 waves per workgroup); the
    # allocator saves ~50 loop-carried / invariant registers around the per-tile staging section (a few dozen scratch
    # accesses per tile and block, none inside the unit loop's phases: DESIGN.md section 4.7).  Bounded here so it cannot grow.
    # A second one: the 192-row split-plane chain (csrc/dss2_gemm_chain_sp6.hip).  Six row blocks x NMAT accumulator blocks take
    # 192 / 288 of the wave's 512 registers; the backward form's 96 prefetched ReLU-gate registers are parked in scratch memory
    # between the last hop and the epilogue, once per layer and outside the GEMM loop (the header of that file has the measurement).
    # (Its 96-row instantiation runs two workgroups per CU on 256 registers with 144 of them accumulators; what spills there are
    # loop invariants around the layer loop and the fp32-gate fallback of the hop phase -- the model path gates with bit words.)
    allowed = {"dss2::stack_bwd_kernel": 64}
    # (the direction-specialised instantiations <NRB, NMAT, 1 | 2> that the models run spill 0-16 registers at K = 2, none at K = 1;
    #  the generic ones <., ., 0> -- mask tensor, residual, fp32 gates: tests and outside callers -- carry every feature at once)
    for nrb, nmat, d, cap in ((3, 3, 1, 16), (3, 3, 2, 16), (6, 3, 1, 24), (6, 3, 2, 24), (3, 2, 0, 16), (3, 3, 0, 96), (6, 2, 0, 80), (6, 3, 0, 144)):
        allowed[f"void dss2::gemm_chain_sp6_kernel<{nrb}, {nmat}, {d}, 0, false>"] = cap
    for nrb, cap in ((3, 16), (6, 24)):      # <., 3, 2, 2>: the data-gradient chain with the head's gradient built in its staging (same body)
        allowed[f"void dss2::gemm_chain_sp6_kernel<{nrb}, 3, 2, 2, false>"] = cap
    # ... and their f16x3 forms (round 5: <., ., ., ., true>; two planes instead of three: fewer at 96 rows).  <6, 3, 2, 0, true> -- the
    # 192-row data-gradient chain WITHOUT the fused head, which the models do not run -- keeps its 24 input row pieces in registers while
    # the tile's maximum is formed and parks most of them in scratch memory once per tile, before the first layer.
    # (per-column weight exponents are per-lane values: three more registers across the hops; the 192-row forms, at 512 of 512 registers
    #  with K = 2, pay for them in parked loop invariants -- cfgbench: 179-bus step 1.504 -> 1.506 ms with them)
    for nrb, d, hm, cap in ((3, 1, 0, 16), (3, 2, 0, 8), (3, 2, 2, 8), (6, 1, 0, 64), (6, 2, 2, 32), (6, 2, 0, 128)):
        allowed[f"void dss2::gemm_chain_sp6_kernel<{nrb}, 3, {d}, {hm}, true>"] = cap
    # A third, chosen: the edge MLP's bf16x6 forward (csrc/dss2_edge16.hip) is held to 128 registers (four waves per SIMD,
    # amdgpu_waves_per_eu) because it waits on a dependent staging chain; the 2 / 7 registers that costs at 64- / 96-row tiles
    # are spilled once per tile outside the slot loop (C2: 18.5 -> 15.6 us with them).
    allowed["void dss2::edge16_fwd_kernel<2>"] = 4
    allowed["void dss2::edge16_fwd_kernel<3>"] = 8
    # A fourth, chosen: the 64-row split-plane chain on 16x16x32 MFMAs at K = 2 (csrc/dss2_gemm_chain_sp.hip, MS = 1).  96 accumulator
    # registers + the four row blocks' plane fragments (48) + two sets of weight fragments (72) leave 40 for everything else; 20-24 loop
    # invariants live in scratch memory around the layer loop, none inside the k-step loop.  Measured with them: forward / backward chain
    # 109.4 / 108.9 -> 100.7 / 101.7 us at C2 (the 32x32x16 form, MS = 0, does not spill).
    for nw in (4, 8):
        for hm in (0, 1, 2):
            allowed[f"void dss2::gemm_chain_sp_kernel<3, {nw}, {hm}, 1>"] = 32
            # (its f16x3 form, MS = 2, has 40 fragment registers fewer: 8-12 values parked around the layer loop -- the folded layer's
            #  eight row-scale vectors requested together at the top of the epilogue and the per-column weight exponents)
            allowed[f"void dss2::gemm_chain_sp_kernel<3, {nw}, {hm}, 2>"] = 16
    # (round 6: the tall-tile chain carries a sixth template argument, the number of ACTIVE row pieces per lane; the full-height
    #  instantiations <.., 4 NRB> are the kernels named above, the reduced one -- <3, 3, ., ., true, 9> -- does not spill at all)
    import re as _re
    for key in [k for k in allowed if "gemm_chain_sp6_kernel<" in k]:
        nrb_ = int(_re.search(r"kernel<(\d+),", key).group(1))
        allowed[key[:-1] + f", {4 * nrb_}>"] = allowed[key]
    bad = [(fn, name, sp, scr) for fn, ks in results for name, sp, scr in ks
           if (sp or scr) and not (name.strip() in allowed and sp <= allowed[name.strip()])]
    assert not bad, bad


def test_shape_queries_route_the_round_4_kernels():
    """Host-side shape queries of the C ABI (no GPU needed): which tile heights get the bf16x6 weight gradient, how many workgroups it
    puts on a tile-list slice, which head modes and gate-bit words the layer chains offer.  The Python layer routes on these answers."""
    from conftest import load_pkg
    L = load_pkg()._lib.lib()
    # weight gradient: 32-row form (two workgroups per CU), 64-row form, tall tiles (one 8-wave workgroup per CU); K = 3 and H <= 32 fall back
    lds = lambda nrb, nmat, h, ell, b16: int(L.dss2_wgrad_lds_bytes_ex(nrb, nmat, h, h, 8 * 32 * nrb, ell, b16))
    assert lds(1, 3, 128, 3, 1) <= 80 * 1024 < lds(2, 3, 128, 3, 1) <= 160 * 1024
    for nrb in (3, 4, 5, 6):
        assert 80 * 1024 < lds(nrb, 3, 128, 3, 1) <= 160 * 1024 and (nrb == 5 or lds(nrb, 3, 128, 3, 1) != lds(nrb, 3, 128, 3, 0))
    assert lds(3, 4, 128, 3, 1) == lds(3, 4, 128, 3, 0) and lds(3, 3, 32, 3, 1) == lds(3, 3, 32, 3, 0) and lds(3, 3, 128, 9, 1) == lds(3, 3, 128, 9, 0)
    ys = lambda nrb, h: int(L.dss2_wgrad_y_slices(nrb, 3, h, h, 3, 1, 0))
    assert (ys(1, 128), ys(2, 128), ys(3, 128), ys(6, 256), ys(2, 256)) == (2, 1, 2, 8, 4)
    # fused narrow head: 64-row split-plane chain both directions (mask 3), its tall forms the backward only (2), wide heads none
    head = lambda nrb, h, nout: int(L.dss2_gemm_prop_chain_head_supported(nrb, 3, h, h, 3, nout))
    assert (head(2, 128, 2), head(3, 128, 2), head(6, 128, 4), head(2, 128, 8), head(2, 64, 2), head(4, 128, 2)) == (3, 2, 2, 0, 0, 0)
    # ReLU gates as bit words: one 32-bit word per lane and eight row pieces, in 64-bit words per tile
    gw = lambda nrb, h: int(L.dss2_gemm_prop_chain_gate_words(nrb, 3, h, h, 3))
    assert (gw(2, 128), gw(3, 128), gw(6, 128), gw(2, 256), gw(2, 64), gw(1, 128)) == (128, 256, 384, 256, 0, 0)


def test_fused_adamax_checkpoint_has_one_step_tensor_per_parameter():
    """internally the parameters of a group share one step tensor; a checkpoint must not export that aliasing
    (torch.optim.Adamax would then advance the count once per PARAMETER per iteration)."""
    import copy
    pkg = load_pkg()
    ps = [torch.nn.Parameter(torch.ones(3)), torch.nn.Parameter(torch.ones(2, 2))]
    opt = pkg.FusedAdamax(ps, lr=3e-3)
    opt.init_state()
    opt.param_groups[0]["_step"] += 5.0                       # as after five steps
    sd = copy.deepcopy(opt.state_dict())
    steps = [sd["state"][i]["step"] for i in range(2)]
    assert steps[0] is not steps[1] and steps[0].data_ptr() != steps[1].data_ptr()
    assert all(float(s) == 5.0 for s in steps)
    assert "_step" not in sd["param_groups"][0]
    ref = torch.optim.Adamax([torch.nn.Parameter(p.detach().clone()) for p in ps], lr=3e-3)
    ref.load_state_dict(sd)
    for p in ref.param_groups[0]["params"]:
        p.grad = torch.ones_like(p)
    ref.step()
    assert [float(ref.state[p]["step"]) for p in ref.param_groups[0]["params"]] == [6.0, 6.0]
    # and the export did not detach the optimizer's own shared counter
    assert opt.state[ps[0]]["step"] is opt.state[ps[1]]["step"]


def test_device_dataset_rejects_out_of_range_node_ids():
    """the hinted structure build never reads its error flag back, so the dataset checks node ids once on the host."""
    import pytest
    import torch
    pkg = load_pkg()
    S, n, e = 3, 5, 4
    x, ea, y = torch.zeros(S, n, 11), torch.zeros(S, e, 13), torch.zeros(S, n, 2)
    ei = torch.tensor([[0, 1, 2, 3], [1, 2, 3, 4]]).expand(S, 2, e).clone()
    ds = pkg.dataset.DeviceDataset(x, ea, y, ei)
    assert ds.max_degree_asis == 1 and ds.hint().nodes_per_graph == n
    bad = ei.clone()
    bad[1, 1, 2] = n          # one endpoint of one sample outside [0, n)
    with pytest.raises(ValueError, match="outside"):
        pkg.dataset.DeviceDataset(x, ea, y, bad)
