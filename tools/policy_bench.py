#!/usr/bin/env python3
"""GPU experiment: inference throughput of the lidar policy's towers under different formulations of
the two 1-D convolutions (same parameters, same maths up to fp32 summation order)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-collision-avoidance_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from mrca.net import CNNPolicy  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
pol = CNNPolicy(3, 2).to(dev).eval()
N = 4096
x = torch.rand(N, 3, 512, device=dev) - 0.5
g = torch.rand(N, 2, device=dev)
s = torch.rand(N, 2, device=dev)


def tower_ref(tw, x):
    h = torch.relu(getattr(pol, f"{tw}_fea_cv1")(x))
    h = torch.relu(getattr(pol, f"{tw}_fea_cv2")(h))
    return torch.relu(getattr(pol, f"{tw}_fc1")(h.flatten(1)))


def tower_cl(tw, x):  # conv2d, channels_last
    c1, c2 = getattr(pol, f"{tw}_fea_cv1"), getattr(pol, f"{tw}_fea_cv2")
    h = x.unsqueeze(2).contiguous(memory_format=torch.channels_last)
    h = torch.relu(F.conv2d(h, c1.weight.unsqueeze(2), c1.bias, stride=(1, 2), padding=(0, 1)))
    h = torch.relu(F.conv2d(h, c2.weight.unsqueeze(2), c2.bias, stride=(1, 2), padding=(0, 1)))
    return torch.relu(getattr(pol, f"{tw}_fc1")(h.contiguous().flatten(1)))


def tower_unfold(tw, x):  # im2col + GEMM
    c1, c2 = getattr(pol, f"{tw}_fea_cv1"), getattr(pol, f"{tw}_fea_cv2")
    n = x.shape[0]
    p = F.pad(x, (1, 1)).unfold(2, 5, 2)                      # [N,3,255,5]
    h = torch.relu(p.permute(0, 2, 1, 3).reshape(n * 255, 15) @ c1.weight.reshape(32, 15).t() + c1.bias)
    h = h.view(n, 255, 32)
    p = F.pad(h, (0, 0, 1, 1)).unfold(1, 3, 2)                # [N,128,32,3]
    h = torch.relu(p.reshape(n * 128, 96) @ c2.weight.reshape(32, 96).t() + c2.bias)
    h = h.view(n, 128, 32).transpose(1, 2).reshape(n, 4096)
    return torch.relu(getattr(pol, f"{tw}_fc1")(h))


def both_fused_fc(fn):
    def run(x):
        return fn("act", x), fn("crt", x)
    return run


def bench(name, fn, autocast=False):
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        for _ in range(3):
            fn(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
    print(f"{name:<34} {dt * 1e3:7.3f} ms / {N} samples  = {N / dt / 1e6:6.2f} M samples/s")


with torch.no_grad():
    a = tower_ref("act", x)
    print("max |cl - ref|    ", float((tower_cl("act", x) - a).abs().max()))
    print("max |unfold - ref|", float((tower_unfold("act", x) - a).abs().max()))
for nm, fn in (("conv1d (current)", tower_ref), ("conv2d channels_last", tower_cl), ("unfold + GEMM", tower_unfold)):
    bench(nm + " fp32", both_fused_fc(fn))
    bench(nm + " bf16 autocast", both_fused_fc(fn), autocast=True)
with torch.no_grad():
    for _ in range(3):
        pol.mean_value(x, g, s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        pol.mean_value(x, g, s)
    torch.cuda.synchronize(); print(f"full mean_value fp32: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
