#!/usr/bin/env python3
"""GPU: device-side batch assembly (dss2_collate) against the HBM roofline, and one epoch from the device loader."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
dev = torch.device("cuda:0")
S, n, e, B = 65536, 15, 14, 4096
x = torch.randn(S, n, 11, device=dev); ea = torch.randn(S, e, 13, device=dev); y = torch.randn(S, n, 2, device=dev)
ei = torch.stack([torch.arange(e), torch.arange(1, e + 1)]).to(dev).repeat(S, 1, 1).contiguous()
ds = pkg.dataset.DeviceDataset(x, ea, y, ei)
ids = torch.randperm(S, device=dev)[:B].contiguous()
for _ in range(10):
    ds.collate(ids)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 200
e0.record()
for _ in range(reps):
    ds.collate(ids)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
byts = 2 * 4 * B * (n * 11 + e * 13 + n * 2)
print(f"collate B={B} (x, edge_attr, y gathered from {S} resident samples; edge_index cached): {us:.1f} us per batch "
      f"(incl. 3 torch.empty), {byts / 1e6:.2f} MB moved -> {byts / us / 1e3:.0f} GB/s ({100 * byts / us / 1e3 / 8000:.1f} % of 8 TB/s; launch-bound)")
loader = pkg.dataset.DataLoader(ds, batch_size=B, shuffle=True)
nb = sum(1 for _ in loader)                 # warm-up epoch (round 5 timed ONE cold epoch of 16 batches: 557 us per batch, mostly first-use costs)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    nb = sum(1 for _ in loader)
th = time.perf_counter() - t0
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"five shuffled epochs of {S} samples in {nb} batches each through dataset.DataLoader (allocating collate): host {th / (5 * nb) * 1e6:.0f} us per batch to enqueue, "
      f"{dt / (5 * nb) * 1e6:.0f} us per batch wall, no host sync")
# ... and straight into static buffers with the device-side cursor (what runner.EpochTrainer's recorded step starts with)
xs = torch.empty(B * n, 11, device=dev); eas = torch.empty(B * e, 13, device=dev)
descs = ds.collate_descs(xs, eas)
cursor = torch.tensor([0, S], dtype=torch.int64, device=dev)
perm = torch.randperm(S, device=dev)
for _ in range(10):
    ds.collate_into(descs, perm, B, cursor=cursor)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5 * nb):
    ds.collate_into(descs, perm, B, cursor=cursor)
th = time.perf_counter() - t0
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"collate_into + device cursor (no allocation, two launches): host {th / (5 * nb) * 1e6:.1f} us per batch to enqueue, {dt / (5 * nb) * 1e6:.1f} us per batch wall")
