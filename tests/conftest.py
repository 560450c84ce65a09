import importlib
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_NAME = "deep-statistical-solver-for-distribution-system-state-estimation_amd"
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "oracle") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    """The package directory name has hyphens, so it is imported through importlib."""
    return importlib.import_module(PKG_NAME)


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def oracle():
    import dss2_oracle
    return dss2_oracle


def golden(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def t(a, dtype=None, device=None):
    x = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None and x.is_floating_point():
        x = x.to(dtype)
    return x.to(device) if device is not None else x


def case_batch(g, dtype=torch.float32, device=None):
    return dict(x=t(g["x"], dtype, device), edge_index=t(g["edge_index"], device=device),
                edge_attr=t(g["edge_attr"], dtype, device),
                stats=tuple(t(g[k], dtype, device) for k in ("x_mean", "x_std", "edge_mean", "edge_std")))


def case_state_dict(g, dtype=torch.float32, device=None):
    return {k[len("param/"):]: t(v, dtype, device) for k, v in g.items() if k.startswith("param/")}


def case_grads(g):
    return {k[len("grad/"):]: t(v) for k, v in g.items() if k.startswith("grad/")}


def rel_err(a, b):
    """max |a-b| / max |b|  (max-normalised relative error, the metric of SURVEY.md section 4)."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    d = (a - b).abs().max().item()
    return d / max(b.abs().max().item(), 1e-30)


# model constructors for the golden cases: name -> (class name, ctor args, with_loss)
CASES = {
    "mpn_c1": ("MPN", (8, 6, 2, 32, 1, 2, 0.0), True),
    "mpn_h64_l3": ("MPN", (8, 6, 2, 64, 3, 2, 0.0), True),
    "skipmpn": ("SkipMPN", (8, 6, 8, 32, 2, 2, 0.0), False),
    "pfn": ("PFN", (8, 6, 2, 32, 2, 2, 0.0, 2), True),
    "skippfn": ("SkipPFN", (8, 6, 2, 32, 2, 2, 0.0, 3), True),
    "mpn_resw_k3": ("MPN", (8, 6, 2, 32, 2, 3, 0.0), True),
    "mpn_ober": ("MPN", (8, 6, 2, 32, 2, 2, 0.0), True),
    "mpn_mixed": ("MPN", (8, 6, 2, 64, 2, 2, 0.0), True),
    "mpn_undirected_input": ("MPN", (8, 6, 2, 32, 2, 2, 0.0), False),
    # the C2 model at the three tile heights (64 / 96 / 192 rows): make_goldens.py flagship
    "mpn_c2model": ("MPN", (8, 6, 2, 128, 4, 2, 0.0), True),
    "mpn_ober_h128": ("MPN", (8, 6, 2, 128, 4, 2, 0.0), True),
    "mpn_ober179_h128": ("MPN", (8, 6, 2, 128, 4, 2, 0.0), True),
}
LOSS_CASES = ["loss_real", "loss_violate_cigre", "loss_violate_ober"]


def tagconv_known_answers():
    """tests/golden/tagconv_known_answers.json: (x, lins, bias, {case: (edge_index, {K: expected out})}); expected entries
    are stored as exact pairs [a, b] = a + b / sqrt(2) (hand-derived, see the file)."""
    import json
    with open(os.path.join(GOLDEN, "tagconv_known_answers.json")) as fh:
        z = json.load(fh)
    x = torch.tensor(z["x"], dtype=torch.float64)
    lins = [torch.tensor(w, dtype=torch.float64) for w in z["lins"]]
    bias = torch.tensor(z["bias"], dtype=torch.float64)
    cases = {}
    for name, c in z["cases"].items():
        exp = {}
        for K, v in c["out"].items():
            a = torch.tensor(v, dtype=torch.float64)
            exp[int(K)] = a[..., 0] + a[..., 1] / (2.0 ** 0.5)
        cases[name] = (torch.tensor(c["edge_index"], dtype=torch.int64), exp)
    return x, lins, bias, cases


# the per-layer-interleaved variants (networks.py:390-735): golden name -> (class name, ctor args)
MULTI_CASES = {
    "multimpn": ("MultiMPN", (8, 6, 2, 32, 3, 2, 0.0)),
    "multimpn_h128": ("MultiMPN", (8, 6, 2, 128, 2, 2, 0.0)),
    "maskembdmpn": ("MaskEmbdMPN", (8, 6, 2, 32, 3, 2, 0.0)),
    "maskembdmultimpn": ("MaskEmbdMultiMPN", (8, 6, 2, 32, 2, 2, 0.0)),
    "maskembdmultimpn_nomp": ("MaskEmbdMultiMPN_NoMP", (8, 6, 2, 8, 3, 2, 0.0)),
}


def multi_case_data(g, dtype=torch.float32, device=None):
    import types
    return types.SimpleNamespace(x=t(g["x"], dtype, device), edge_index=t(g["edge_index"], device=device),
                                 edge_attr=t(g["edge_attr"], dtype, device))
