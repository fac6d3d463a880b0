#!/usr/bin/env python3
"""GPU experiment: where the lidar policy's inference time goes at 4096 robots -- per layer, and for the fused
formulations of the twin towers (conv1 as one 3->64 convolution, conv2 grouped, fc1 as one batched GEMM)."""
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
N = int(os.environ.get("N", "4096"))
x = torch.rand(N, 3, 512, device=dev) - 0.5
g = torch.rand(N, 2, device=dev)
s = torch.rand(N, 2, device=dev)


def bench(name, fn, flops=None, iters=30):
    with torch.no_grad():
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
    extra = f"  {flops / dt / 1e12:6.1f} TFLOP/s" if flops else ""
    print(f"{name:<58} {dt * 1e6:9.1f} us{extra}", flush=True)
    return dt


for bm in (False, True):
    torch.backends.cudnn.benchmark = bm
    print(f"--- torch.backends.cudnn.benchmark = {bm}")
    c1, c2, f1 = pol.act_fea_cv1, pol.act_fea_cv2, pol.act_fc1
    xcl = x.unsqueeze(2).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        h1 = torch.relu(F.conv2d(xcl, c1.weight.unsqueeze(2), c1.bias, stride=(1, 2), padding=(0, 1)))
        h2 = torch.relu(F.conv2d(h1, c2.weight.unsqueeze(2), c2.bias, stride=(1, 2), padding=(0, 1)))
        flat = h2.contiguous().flatten(1)
        h1n = torch.relu(c1(x))
    bench("conv1 (one tower) conv2d channels_last fp32", lambda: F.conv2d(xcl, c1.weight.unsqueeze(2), c1.bias, stride=(1, 2), padding=(0, 1)), N * 32 * 255 * 15 * 2)
    bench("conv1 (one tower) conv1d NCL fp32", lambda: c1(x), N * 32 * 255 * 15 * 2)
    bench("conv2 (one tower) conv2d channels_last fp32", lambda: F.conv2d(h1, c2.weight.unsqueeze(2), c2.bias, stride=(1, 2), padding=(0, 1)), N * 32 * 128 * 96 * 2)
    bench("conv2 (one tower) conv1d NCL fp32", lambda: c2(h1n), N * 32 * 128 * 96 * 2)
    bench("h2.contiguous().flatten (layout change)", lambda: h2.contiguous().flatten(1))
    bench("fc1 (one tower) fp32 [N,4096]x[4096,256]", lambda: f1(flat), N * 4096 * 256 * 2)
    w2 = torch.stack([pol.act_fc1.weight.t(), pol.crt_fc1.weight.t()]).contiguous()
    flat2 = torch.stack([flat, flat])
    bench("fc1 both towers as bmm fp32", lambda: torch.bmm(flat2, w2), 2 * N * 4096 * 256 * 2)
    fb, wb = flat.bfloat16(), pol.act_fc1.weight.t().contiguous().bfloat16()
    bench("fc1 (one tower) bf16", lambda: fb @ wb, N * 4096 * 256 * 2)
    # twin towers fused: conv1 3->64, conv2 grouped
    w1f = torch.cat([pol.act_fea_cv1.weight, pol.crt_fea_cv1.weight]).unsqueeze(2).contiguous()
    b1f = torch.cat([pol.act_fea_cv1.bias, pol.crt_fea_cv1.bias])
    w2f = torch.cat([pol.act_fea_cv2.weight, pol.crt_fea_cv2.weight]).unsqueeze(2).contiguous()
    b2f = torch.cat([pol.act_fea_cv2.bias, pol.crt_fea_cv2.bias])
    with torch.no_grad():
        h1f = torch.relu(F.conv2d(xcl, w1f, b1f, stride=(1, 2), padding=(0, 1)))
    bench("conv1 both towers as 3->64 channels_last fp32", lambda: F.conv2d(xcl, w1f, b1f, stride=(1, 2), padding=(0, 1)), 2 * N * 32 * 255 * 15 * 2)
    bench("conv2 both towers grouped(2) channels_last fp32", lambda: F.conv2d(h1f, w2f, b2f, stride=(1, 2), padding=(0, 1), groups=2), 2 * N * 32 * 128 * 96 * 2)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        bench("conv1 both towers 3->64 channels_last bf16", lambda: F.conv2d(xcl, w1f, b1f, stride=(1, 2), padding=(0, 1)), 2 * N * 32 * 255 * 15 * 2)
        h1b = torch.relu(F.conv2d(xcl, w1f, b1f, stride=(1, 2), padding=(0, 1)))
        bench("conv2 both towers grouped(2) channels_last bf16", lambda: F.conv2d(h1b, w2f, b2f, stride=(1, 2), padding=(0, 1), groups=2), 2 * N * 32 * 128 * 96 * 2)
    bench("full mean_value fp32", lambda: pol.mean_value(x, g, s), N * 6.4e6)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        bench("full mean_value bf16 autocast", lambda: pol.mean_value(x, g, s), N * 6.4e6)
bench("relu on [N,32,255]", lambda: torch.relu(h1))
bench("copy obs [N,3,512]", lambda: x.clone())
