"""The build's own runner for the hot path: what /root/reference/dss2_run.py does around it.

Reproduces the training loop (dss2_run.py:131-147) and the per-epoch evaluation
(dss2_run.py:165-224) with the reference's knobs as arguments and its defaults:
feature slicing ``x[:, :8] / edge_attr[:, :6] / x[:, 8:] / edge_attr[:, 6:]`` (:138,140),
``reg_coefs`` (:104-112), Adamax lr 3e-3 (:91-92), batch_size 64, the hyper-parameter dict (:72-82).
The model is the MPN / SkipMPN / PFN / SkipPFN line (:88), not the default GAT (out of scope).
Data: either a folder in the reference's layout (``--data-folder``: ``dataset.data_from_pickles`` ->
shuffle -> 0.9 split -> ``dataset.DataLoader``, dss2_run.py:56-69, measurement model / z-score / collation
on the device) or synthetic batches from ``synthetic.make_batch`` (same layout), collated once and kept
resident on the GPU (re-used tensors take the no-sync topology fast path).

    python tools/train.py --case cigre14 --model SkipPFN --epochs 5
    python tools/train.py --data-folder /path/to/data/cigre14/ --model SkipPFN --epochs 5
"""
from __future__ import annotations

import argparse
import time
from typing import Dict, List

import torch

from . import data as dss2_data
from . import networks, synthetic
from .optim import FusedAdamax

REG_COEFS = {"mu_v": 1e-1, "mu_theta": 1e-1, "lam_v": 1e-4, "lam_p": 1e-8, "lam_pf": 1e-6, "lam_reg": 1e2}
HYPER = {"dim_nodes": 8, "dim_lines": 6, "dim_out": 2, "dim_hid": 32, "gnn_layers": 8, "K": 2, "dropout_rate": 0.3, "L": 5}


def build_model(name: str, hp: Dict) -> torch.nn.Module:
    cls = getattr(networks, name)
    a = (hp["dim_nodes"], hp["dim_lines"], hp["dim_out"], hp["dim_hid"], hp["gnn_layers"], hp["K"], hp["dropout_rate"])
    return cls(*a, hp["L"]) if name in ("PFN", "SkipPFN") else cls(*a)


def make_loaders(case: str, n_graphs: int, batch_size: int, device, seed: int = 0, split: float = 0.9):
    """Pre-collated, device-resident batches (train / test) with one set of normalisation statistics."""
    full = synthetic.make_batch([case], n_graphs, seed=seed)
    stats = tuple(s.to(device) for s in full["stats"])
    n_train = int(split * n_graphs)

    def batches(g0, g1, seed_off):
        out = []
        for b0 in range(g0, g1, batch_size):
            nb = min(batch_size, g1 - b0)
            b = synthetic.make_batch([case], nb, seed=seed + 1 + seed_off + b0, stats=full["stats"])
            out.append({k: (v.to(device) if torch.is_tensor(v) else v) for k, v in b.items() if k != "stats"})
        return out

    return batches(0, n_train, 0), batches(n_train, n_graphs, 10_000), stats


def _fields(data):
    """A batch is a dict (synthetic loaders) or a dataset.Batch (device loader): (x, edge_index, edge_attr, y, B)."""
    if isinstance(data, dict):
        return data["x"], data["edge_index"], data["edge_attr"], data.get("y"), data["num_graphs"]
    return data.x, data.edge_index, data.edge_attr, data.y, data.num_graphs


def train_epoch(model, opt, loader, stats, reg_coefs, group=None) -> float:
    model.train()
    total = torch.zeros((), device=stats[0].device)
    for data in loader:                                              # dss2_run.py:134-144
        opt.zero_grad()
        x, ei, ea, _, num_graphs = _fields(data)
        out = model(x[:, :8], ei, ea[:, :6])
        loss = dss2_data.gsp_wls_edge(input=x[:, :8], edge_input=ea[:, :6], output=out, x_mean=stats[0], x_std=stats[1],
                                      edge_mean=stats[2], edge_std=stats[3], edge_index=ei, reg_coefs=reg_coefs,
                                      num_samples=num_graphs, node_param=x[:, 8:], edge_param=ea[:, 6:], group=group)
        loss.backward(dss2_data.unit_grad(loss))                      # (= loss.backward(), without autograd's ones_like fill kernel)
        opt.step()
        total += loss.detach()
    return float(total / len(loader))                                # one host sync per epoch (:147)


class GraphedTrainer:
    """The training step of ``train_epoch`` -- zero_grad, forward, gsp_wls_edge, backward, optimizer step -- captured ONCE per
    batch shape into a hipGraph (graphs.GraphedStep) and replayed on static input buffers: at the driver's batch size the eager
    step is host-bound (~2.4 ms of Python / autograd dispatch around 0.33 ms of kernels for the SkipPFN line), the replay is
    not.  Valid when every batch of a shape has the SAME graph structure (one topology per dataset, e.g. the reference's
    cigre14 folder): the structure is part of the captured launches, so only x and edge_attr are copied per step.  In-kernel
    dropout draws new masks on every replay; the optimizer must be ``FusedAdamax(capturable=True)``."""

    def __init__(self, model, opt, stats, reg_coefs, group=None):
        if not getattr(opt, "capturable", False):
            raise ValueError("GraphedTrainer needs FusedAdamax(capturable=True): the step count must live on the device")
        self.model, self.opt, self.stats, self.reg, self.group = model, opt, stats, reg_coefs, group
        self.graphs = {}
        self.params = list(model.parameters())

    def _build(self, x, ei, ea):
        from .graphs import GraphedStep
        sx, sea, sei = x.clone(), ea.clone(), ei.clone()
        st, reg, model, opt, params, group = self.stats, self.reg, self.model, self.opt, self.params, self.group
        eager_losses = []

        def step_fn():
            for p in params:
                p.grad = None
            out = model(sx[:, :8], sei, sea[:, :6])
            loss = dss2_data.gsp_wls_edge(input=sx[:, :8], edge_input=sea[:, :6], output=out, x_mean=st[0], x_std=st[1],
                                          edge_mean=st[2], edge_std=st[3], edge_index=sei, reg_coefs=reg, num_samples=None,
                                          node_param=sx[:, 8:], edge_param=sea[:, 6:], group=group)
            loss.backward(dss2_data.unit_grad(loss))
            opt.step()
            if not torch.cuda.is_current_stream_capturing():
                eager_losses.append(loss.detach().clone())
            return loss
        # the warm-up step of the capture is a REAL training step on this batch (it advances the weights and the optimizer); the
        # capture runs on the caller's current stream when that is not the default one (the autograd state of earlier eager
        # steps belongs to it), else on a stream of its own
        cur = torch.cuda.current_stream()
        g = GraphedStep(step_fn, warmup=1, capture_error_mode=("thread_local" if group is not None else "global"),
                        stream=(cur if cur != torch.cuda.default_stream() else None))
        return g, sx, sea, eager_losses[-1]

    def step(self, x, ei, ea) -> torch.Tensor:
        key = (tuple(x.shape), tuple(ea.shape), tuple(ei.shape))
        hit = self.graphs.get(key)
        if hit is None:
            g, sx, sea, first_loss = self._build(x, ei, ea)
            self.graphs[key] = (g, sx, sea)
            return first_loss           # (the capture's warm-up step has trained on this batch)
        g, sx, sea = hit
        sx.copy_(x)
        sea.copy_(ea)
        return g.replay().detach()


def train_epoch_graphed(trainer: GraphedTrainer, loader) -> float:
    total = torch.zeros((), device=trainer.stats[0].device)
    for data in loader:
        x, ei, ea, _, _ = _fields(data)
        total += trainer.step(x, ei, ea)
    return float(total / len(loader))


class EpochTrainer:
    """The training loop of dss2_run.py:131-147 -- ``for data in loader: zero_grad; forward; gsp_wls_edge; backward; step`` -- with NO host
    work per batch beyond one C call: the loader's collation writes straight into the step's static input buffers as the FIRST launch of
    the recorded step (``dss2_collate_cursor``: the batch's position in the epoch's sample permutation lives on the device and is moved
    forward by the launch behind the gather), the optimizer step and the running loss sum (``dss2_accum_scalar``) are its last ones.  An
    epoch is ``len(dataset) // batch_size`` replays of one recorded step (+ one replay of a second, smaller step for the last
    batch, ``drop_last=False`` as torch_geometric's loader at dss2_run.py:68-69); the only per-epoch host work is drawing the
    permutation (on the device) and resetting two words.  Nothing is read back until the caller asks for the epoch's mean loss.

    ``mode``: "plan" (graphs.PlannedStep: the library's own launch list, one ``dss2_plan_run`` per step; also at world > 1, where the
    step's collectives cut it into segments) or "graph" (graphs.GraphedStep: a hipGraph).  Needs a single-topology ``DeviceDataset``
    (the graph structure is part of the recorded launches) and ``FusedAdamax(capturable=True)``.  Recording a step runs it (plans:
    three times, graphs: once), so the model, the optimizer state and the epoch position are saved before and restored after: the first
    ``train_epoch()`` starts from exactly the state the trainer was given."""

    def __init__(self, model, opt, stats, reg_coefs, dataset, batch_size: int, shuffle: bool = True, mode: str = "plan", group=None,
                 generator=None):
        from . import _lib
        from . import dataset as dss2_dataset
        if not isinstance(dataset, dss2_dataset.DeviceDataset) or not dataset.shared_topology:
            raise ValueError("EpochTrainer needs a single-topology DeviceDataset (per-batch structures: DataLoader / PrefetchLoader + train_epoch)")
        if not getattr(opt, "capturable", False):
            raise ValueError("EpochTrainer needs FusedAdamax(capturable=True): the step count must live on the device")
        if mode not in ("plan", "graph"):
            raise ValueError("mode: 'plan' or 'graph'")
        self._lib = _lib
        self.model, self.opt, self.stats, self.reg, self.group = model, opt, stats, reg_coefs, group
        self.ds, self.B, self.shuffle, self.mode, self.generator = dataset, int(batch_size), bool(shuffle), mode, generator
        self.params = list(model.parameters())
        dev = dataset.device
        n = len(dataset)
        if n < 1 or self.B < 1:
            raise ValueError("empty dataset / batch")
        self.n_full, self.rem = divmod(n, self.B)
        self.base_ids = dataset.ids.contiguous()
        self.ids = self.base_ids.clone()                                   # this epoch's order (static: the recorded steps read it)
        self.cursor = torch.tensor([0, n], dtype=torch.int64, device=dev)  # {position, epoch length}
        self._cursor0 = self.cursor.clone()
        self.acc = torch.zeros(2, dtype=torch.float64, device=dev)         # {sum of the step losses, steps}
        self.steps = {}
        opt.init_state()
        saved = self._snapshot()
        for nb in ([self.B] if self.n_full else []) + ([self.rem] if self.rem else []):
            self.steps[nb] = self._record(nb)
        self._restore(saved)

    # ---- state that recording a step consumes
    def _state_tensors(self):
        ts = list(self.params)
        for g in self.opt.param_groups:
            if torch.is_tensor(g.get("_step")):
                ts.append(g["_step"])
            for p in g["params"]:
                st = self.opt.state.get(p, {})
                ts += [st[k] for k in ("exp_avg", "exp_inf") if k in st]
        return ts

    def _snapshot(self):
        return [t.detach().clone() for t in self._state_tensors()]

    @torch.no_grad()
    def _restore(self, saved):
        for t, s in zip(self._state_tensors(), saved):
            t.copy_(s)
        self.cursor.copy_(self._cursor0)
        self.acc.zero_()

    def _record(self, nb: int):
        from . import graphs
        ds, st, reg, model, opt, params, group = self.ds, self.stats, self.reg, self.model, self.opt, self.params, self.group
        dev = ds.device
        sx = torch.empty(nb * ds.n, ds.x.size(2), dtype=torch.float32, device=dev)
        sea = torch.empty(nb * ds.e, ds.edge_attr.size(2), dtype=torch.float32, device=dev)
        ei, _ = ds.batch_structure(nb)
        descs = ds.collate_descs(sx, sea)
        ids, cursor, acc, L = self.ids, self.cursor, self.acc, self._lib

        def step_fn():
            ds.collate_into(descs, ids, nb, cursor=cursor, advance=True)
            for p in params:
                p.grad = None
            out = model(sx[:, :8], ei, sea[:, :6])
            loss = dss2_data.gsp_wls_edge(input=sx[:, :8], edge_input=sea[:, :6], output=out, x_mean=st[0], x_std=st[1],
                                          edge_mean=st[2], edge_std=st[3], edge_index=ei, reg_coefs=reg, num_samples=None,
                                          node_param=sx[:, 8:], edge_param=sea[:, 6:], group=group)
            loss.backward(dss2_data.unit_grad(loss))
            opt.step()
            L.check(L.lib().dss2_accum_scalar(acc.data_ptr(), loss.data_ptr(), L.stream_ptr(dev)), "dss2_accum_scalar")
            return loss
        cur = torch.cuda.current_stream(dev)
        if self.mode == "plan":
            rec = graphs.PlannedStep(step_fn, stream=cur)
        else:
            rec = graphs.GraphedStep(step_fn, warmup=1, capture_error_mode=("thread_local" if group is not None else "global"),
                                     stream=(cur if cur != torch.cuda.default_stream(dev) else None))
        return rec, sx, sea, descs

    def train_epoch(self) -> torch.Tensor:
        """One pass over the dataset.  Returns the DEVICE tensor {sum of the step losses, steps} of this epoch (``mean_loss()`` reads
        it: the epoch's single host synchronisation, dss2_run.py:147)."""
        if self.shuffle:
            perm = torch.randperm(self.base_ids.numel(), device=self.base_ids.device, generator=self.generator)
            torch.index_select(self.base_ids, 0, perm, out=self.ids)
        self.cursor.copy_(self._cursor0)
        self.acc.zero_()
        if self.n_full:
            replay = self.steps[self.B][0].replay
            for _ in range(self.n_full):
                replay()
        if self.rem:
            self.steps[self.rem][0].replay()
        return self.acc

    def mean_loss(self) -> float:
        s, k = self.acc.tolist()
        return s / max(k, 1.0)


@torch.no_grad()
def evaluate(model, loader, stats) -> Dict[str, float]:
    """dss2_run.py:165-224: RMSE / MAE of V and theta and of the line / trafo loadings from get_pflow, and the
    std ratios, averaged over the test batches.  The ten per-batch quantities are accumulated on the device
    (``data.eval_batch``); one device-to-host copy per evaluation."""
    model.eval()                                                     # dropout stays active, as in the reference
    acc = torch.zeros(10, dtype=torch.float64, device=stats[0].device)
    n = 0
    for data in loader:
        x, ei, ea, y, _ = _fields(data)
        out = model(x[:, :8], ei, ea[:, :6])
        dss2_data.eval_batch(out, y, x, ei, ea, stats[0], stats[1], acc)
        n += 1
    vals = (acc / max(n, 1)).cpu().tolist()
    return dict(zip(dss2_data.EVAL_METRICS, vals))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--case", default="cigre14", choices=["cigre14", "cigre14_reswitched", "ober_sub", "ober179"])
    ap.add_argument("--model", default="SkipPFN", choices=["MPN", "SkipMPN", "PFN", "SkipPFN"])
    ap.add_argument("--graphs", type=int, default=720)
    ap.add_argument("--batch-size", type=int, default=64)
    ap.add_argument("--epochs", type=int, default=600)
    ap.add_argument("--lr", type=float, default=3e-3)
    for k, v in HYPER.items():
        ap.add_argument("--" + k.replace("_", "-"), type=type(v), default=v)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--save", default="")
    ap.add_argument("--data-folder", default="", help="folder with the reference's pickles (nodes, edges, labels, noise_param)")
    ap.add_argument("--graph", type=int, default=-1, help="1: replay the training step as a hipGraph (GraphedTrainer; needs ONE graph "
                    "structure per batch shape), 0: eager, -1 (default): graph when the data has a single topology")
    a = ap.parse_args(argv)
    hp = {k: getattr(a, k) for k in HYPER}
    if a.model == "SkipMPN":
        hp["dim_out"] = hp["dim_nodes"]
    dev = torch.device("cuda")
    torch.manual_seed(a.seed)
    if a.data_folder:                                                # dss2_run.py:47-69
        import numpy as np
        from . import dataset as dss2_dataset
        cigre = "cigre" in a.data_folder
        meas_v = np.array([0, 1, 12, 7, 11, 14] if cigre else [35, 16, 52, 47, 6, 48, 59, 27, 37, 56])
        meas_pf = np.array([0, 10] if cigre else [40, 43, 11, 21, 54, 57])
        ds, *stats = dss2_dataset.data_from_pickles(a.data_folder, 8, 6, 4, 2, meas_v, meas_pf, device=dev)
        ds = ds.shuffled()
        n_train = int(0.9 * len(ds))
        train_loader = dss2_dataset.DataLoader(ds[0:n_train], batch_size=a.batch_size, shuffle=True)
        test_loader = dss2_dataset.DataLoader(ds[n_train:], batch_size=a.batch_size, shuffle=False)
        stats = tuple(stats)
    else:
        train_loader, test_loader, stats = make_loaders(a.case, a.graphs, a.batch_size, dev, a.seed)
    model = build_model(a.model, hp).to(dev)
    single_topology = (not a.data_folder) or bool(getattr(train_loader.dataset, "shared_topology", False))
    use_graph = (a.graph == 1) or (a.graph == -1 and single_topology and hp["dim_out"] == 2)
    opt = FusedAdamax(model.parameters(), lr=a.lr, capturable=use_graph)
    # a device-resident data folder with ONE graph structure: whole epochs without the interpreter (EpochTrainer: the loader's collation is the
    # first launch of the recorded step); pre-collated synthetic batches: the step replayed on copied inputs (GraphedTrainer); else eager
    epoch_trainer = None
    if use_graph and a.data_folder and single_topology:
        epoch_trainer = EpochTrainer(model, opt, stats, REG_COEFS, train_loader.dataset, a.batch_size, shuffle=True, mode="graph")
    trainer = GraphedTrainer(model, opt, stats, REG_COEFS) if (use_graph and epoch_trainer is None) else None
    how = "whole epochs as replays of one recorded step (EpochTrainer)" if epoch_trainer is not None else ("hipGraph replay" if use_graph else "eager")
    print(f"device:{dev}  train batches {len(train_loader)}  test batches {len(test_loader)}  model {a.model} {hp}  step: {how}")
    for epoch in range(a.epochs):
        t0 = time.perf_counter()
        if epoch_trainer is not None:
            epoch_trainer.train_epoch()
            tl = epoch_trainer.mean_loss()
        else:
            tl = train_epoch_graphed(trainer, train_loader) if use_graph else train_epoch(model, opt, train_loader, stats, REG_COEFS)
        m = evaluate(model, test_loader, stats) if hp["dim_out"] == 2 else {}
        torch.cuda.synchronize()
        print(f"epoch {epoch:4d}  train_loss {tl:.6g}  " + "  ".join(f"{k} {v:.4g}" for k, v in m.items()) +
              f"  ({time.perf_counter() - t0:.2f} s)", flush=True)
    if a.save:                                                       # dss2_run.py:240-247
        torch.save({"epoch": a.epochs - 1, "model_state_dict": model.state_dict(), "optimizer_state_dict": opt.state_dict()},
                   a.save)


if __name__ == "__main__":
    main()
