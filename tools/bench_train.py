#!/usr/bin/env python3
"""Developer timing: regressor training, HIP kernels vs the same loop in plain torch on the host CPU (what the reference runs)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_assisted_data_labeling_amd.train_driver import FcTrainer, cosine_warm_restarts_lr

n, d, hidden, bs, epochs = 9400, 1536, [264, 128, 64], int(sys.argv[1]) if len(sys.argv) > 1 else 16, 5
g = torch.Generator().manual_seed(0)
X = torch.randn(n, d, generator=g); T = torch.rand(n, generator=g)
sizes = [d] + hidden + [1]
layers = []
for i in range(len(sizes) - 1):
    layers.append(torch.nn.Linear(sizes[i], sizes[i + 1]))
    if i < len(sizes) - 2:
        layers += [torch.nn.LeakyReLU(), torch.nn.Dropout(0.5)]
layers.append(torch.nn.Sigmoid())
model = torch.nn.Sequential(*layers)
lin = [m for m in model if isinstance(m, torch.nn.Linear)]
dev = torch.device("cuda", 0)
tr = FcTrainer([m.weight for m in lin], [m.bias for m in lin], 0.01, dev)
Xd, Td = X.to(dev), T.to(dev)
rs = np.random.RandomState(0)
tr.epoch(Xd, Td, rs.permutation(n), bs, 2e-4, 6e-4, 0.5, 1); torch.cuda.synchronize()
t = time.perf_counter()
for ep in range(epochs):
    loss = tr.epoch(Xd, Td, rs.permutation(n), bs, cosine_warm_restarts_lr(2e-4, 1e-6, 10, ep), 6e-4, 0.5, 1)
torch.cuda.synchronize()
gpu_s = (time.perf_counter() - t) / epochs
print(f"HIP: {gpu_s * 1e3:.1f} ms per epoch of {n} samples at batch {bs} ({(n + bs - 1) // bs} Adam steps, {gpu_s / ((n + bs - 1) // bs) * 1e6:.1f} us/step); last train mse {float(loss.mean()):.4f}")
torch.set_num_threads(min(32, os.cpu_count()))
opt = torch.optim.Adam(model.parameters(), lr=2e-4, weight_decay=6e-4)
crit = torch.nn.MSELoss()
model.train()
t = time.perf_counter()
for ep in range(2):
    order = torch.from_numpy(rs.permutation(n))
    for b0 in range(0, n, bs):
        idx = order[b0:b0 + bs]
        opt.zero_grad(); l = crit(model(X[idx]).squeeze(), T[idx]); l.backward(); opt.step()
cpu_s = (time.perf_counter() - t) / 2
print(f"torch on the host CPU ({torch.get_num_threads()} threads): {cpu_s * 1e3:.0f} ms per epoch  -> x{cpu_s / gpu_s:.1f}")
