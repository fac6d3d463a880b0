#!/usr/bin/env python3
"""GPU: where does the PPO update spend its time?  (diagnostic, writes to stdout)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-collision-avoidance_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from mrca.net import CNNPolicy  # noqa: E402

dev = "cuda"
pol = CNNPolicy(3, 2).to(dev)
opt = torch.optim.Adam(pol.parameters(), lr=5e-5)


def sync():
    torch.cuda.synchronize()


for bs in (1024, 4096, 16384):
    x = torch.rand(bs, 3, 512, device=dev) - 0.5
    g = torch.rand(bs, 2, device=dev)
    s = torch.rand(bs, 2, device=dev)
    a = torch.rand(bs, 2, device=dev)
    for mode in ("fp32", "bf16"):
        def step():
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(mode == "bf16")):
                v, lp, ent = pol.evaluate_actions(x, g, s, a)
                loss = lp.float().mean() + v.float().pow(2).mean() - 0.01 * ent.float()
            opt.zero_grad()
            loss.backward()
            opt.step()
        t0 = time.perf_counter(); step(); sync(); first = time.perf_counter() - t0
        for _ in range(3):
            step()
        sync()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        sync()
        dt = (time.perf_counter() - t0) / 10
        with torch.no_grad():
            sync(); t0 = time.perf_counter()
            for _ in range(10):
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(mode == "bf16")):
                    pol.mean_value(x, g, s)
            sync(); fw = (time.perf_counter() - t0) / 10
        print(f"batch {bs:6d} {mode}: first step {first * 1e3:8.1f} ms, fwd+bwd+adam {dt * 1e3:7.2f} ms "
              f"({bs / dt / 1e6:.2f} M samples/s), inference fwd {fw * 1e3:6.2f} ms ({bs / fw / 1e6:.2f} M samples/s)")

# gather cost
big = torch.rand(128 * 4096, 3, 512, device=dev)
idx = torch.randperm(big.shape[0], device=dev)[:16384]
sync(); t0 = time.perf_counter()
for _ in range(10):
    y = big[idx]
sync(); print(f"gather 16384 rows of [3,512] from 524288: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms")
from torch.profiler import profile, ProfilerActivity  # noqa: E402
x = torch.rand(16384, 3, 512, device=dev) - 0.5
g = torch.rand(16384, 2, device=dev); s = torch.rand(16384, 2, device=dev); a = torch.rand(16384, 2, device=dev)
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(3):
        v, lp, ent = pol.evaluate_actions(x, g, s, a)
        loss = lp.mean() + v.pow(2).mean() - 0.01 * ent
        opt.zero_grad(); loss.backward(); opt.step()
    sync()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
