import importlib, os, sys, time, torch
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
dev = torch.device("cuda:0")
REG = {"mu_v": 1e-1, "mu_theta": 1e-1, "lam_v": 1e-4, "lam_p": 1e-8, "lam_pf": 1e-6, "lam_reg": 1e2}
batch = pkg.synthetic.make_batch(["cigre14"], int(os.environ.get("B", 4096)), seed=1000)
x, ei, ea = batch["x"].to(dev), batch["edge_index"].to(dev), batch["edge_attr"].to(dev)
stats = tuple(s.to(dev) for s in batch["stats"])
torch.manual_seed(0)
model = pkg.MPN(8, 6, 2, 128, 4, 2, 0.0).to(dev)
xin, ein, npar, epar = x[:, :8], ea[:, :6], x[:, 8:], ea[:, 6:]
def step():
    for p in model.parameters(): p.grad = None
    out = model(xin, ei, ein)
    loss = pkg.gsp_wls_edge(input=xin, edge_input=ein, output=out, x_mean=stats[0], x_std=stats[1], edge_mean=stats[2],
                            edge_std=stats[3], edge_index=ei, reg_coefs=REG, num_samples=None, node_param=npar, edge_param=epar)
    loss.backward()
for _ in range(1500): step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(300): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0)/300:.3f} ms/step   total {1e3*(t2-t0)/300:.3f} ms/step")
