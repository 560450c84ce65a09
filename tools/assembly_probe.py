#!/usr/bin/env python3
"""GPU: batch assembly alone for the C5 shuffled case (ragged collate of a fresh cigre14 / reswitched mix + device-side
CSR / tile / ELL build), 30 batches of 4096 graphs -- for rocprofv3 --kernel-trace --stats."""
import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
dev = torch.device("cuda:0")
B, S = 4096, 8192
full = pkg.synthetic.make_batch(["cigre14", "cigre14_reswitched"], 256, seed=1)
parts = [pkg.dataset.DeviceDataset.from_batch(pkg.synthetic.make_batch([g], S, seed=2 + k, stats=full["stats"]), device=dev)
         for k, g in enumerate(["cigre14", "cigre14_reswitched"])]
ds = pkg.dataset.MixedDataset(parts)
rng = np.random.default_rng(0)
for _ in range(30):
    bt = ds.collate(rng.choice(2 * S, size=B, replace=False))
    pkg.topology.get_topology(bt.edge_index, bt.x.shape[0]).nrb
torch.cuda.synchronize()
