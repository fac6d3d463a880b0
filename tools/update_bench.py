#!/usr/bin/env python3
"""GPU: the PPO update's forward + backward through the conv front end -- the hand-written HIP kernels against the
stock PyTorch layers (MIOpen).  Prints, per minibatch size: the forward and the backward kernel alone (HIP events over
20 launches; achieved fp32 MFMA TFLOP/s against the 157.3 TFLOP/s peak), and fwd + bwd + Adam of the whole policy with
`fused_train` on / off; then a torch.profiler table of the fused step."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-collision-avoidance_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as G  # noqa: E402
G.build()
from mrca import policy_ops  # noqa: E402
from mrca.net import CNNPolicy  # noqa: E402

dev = "cuda"
PEAK = 157.3
FWD_FLOP = 2 * (32 * 15 * 255 + 32 * 96 * 128)                     # per (sample, tower), useful
BWD_FLOP = 2 * (32 * 15 * 255 + 2 * 32 * 96 * 128 + 32 * 15 * 255)   # conv1 recompute + conv2 wgrad + dgrad + conv1 wgrad


def timed(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3


torch.manual_seed(0)
pol = CNNPolicy(3, 2).to(dev)
opt = torch.optim.Adam(pol.parameters(), lr=5e-5)
sizes = [int(a) for a in sys.argv[1:]] or [4096, 16384]
for bs in sizes:
    x = torch.rand(bs, 3, 512, device=dev) - 0.5
    g = torch.rand(bs, 2, device=dev)
    s = torch.rand(bs, 2, device=dev)
    a = torch.rand(bs, 2, device=dev)
    rc = pol.refresh_rollout_cache()
    feat = policy_ops.lidar_features(x, rc["w1"], rc["b1"], rc["w2"], rc["b2"])
    gfeat = torch.randn_like(feat)
    t_f = timed(lambda: policy_ops.lidar_features(x, rc["w1"], rc["b1"], rc["w2"], rc["b2"], out=feat))
    t_b = timed(lambda: policy_ops.lidar_features_backward(x, rc["w1"], rc["b1"], rc["w2"], feat, gfeat[0], gfeat[1]))
    out = {"minibatch": bs,
           "forward_kernel_us": t_f * 1e6, "forward_tflops": 2 * bs * FWD_FLOP / t_f / 1e12,
           "forward_frac_of_fp32_mfma_peak": 2 * bs * FWD_FLOP / t_f / 1e12 / PEAK,
           "backward_kernels_us": t_b * 1e6, "backward_tflops": 2 * bs * BWD_FLOP / t_b / 1e12,
           "backward_frac_of_fp32_mfma_peak": 2 * bs * BWD_FLOP / t_b / 1e12 / PEAK}
    for fused in (True, False):
        pol.fused_train = fused

        def step():
            v, lp, ent = pol.evaluate_actions(x, g, s, a)
            loss = lp.mean() + v.pow(2).mean() - 0.01 * ent
            opt.zero_grad()
            loss.backward()
            opt.step()
        out["fwd_bwd_adam_ms_" + ("fused" if fused else "stock")] = timed(step, n=10) * 1e3
    # the update as mrca.ppo runs it since round 4: HIP front end forward / backward, the PPO loss tail as ONE launch
    # (policy_ops.ppo_loss: values + gradients), Adam as one multi-tensor launch (fused=True)
    pol.fused_train = True
    opt_f = torch.optim.Adam(pol.parameters(), lr=5e-5, fused=True)
    old_lp = torch.randn(bs, 1, device=dev) * 0.1 - 2.0
    adv = torch.randn(bs, 1, device=dev)
    tgt = torch.randn(bs, 1, device=dev)

    def step_round4():
        mean, v = pol.mean_value(x, g, s)
        loss, _stats = policy_ops.ppo_loss(mean, v, pol.logstd, a, old_lp, adv, tgt, 0.1, 20.0, 5e-4)
        opt_f.zero_grad()
        loss.backward()
        opt_f.step()
    out["fwd_bwd_adam_ms_fused_front_end_loss_kernel_fused_adam"] = timed(step_round4, n=10) * 1e3
    print(json.dumps(out))

from torch.profiler import ProfilerActivity, profile  # noqa: E402
bs = sizes[-1]
x = torch.rand(bs, 3, 512, device=dev) - 0.5
g = torch.rand(bs, 2, device=dev); s = torch.rand(bs, 2, device=dev); a = torch.rand(bs, 2, device=dev)
pol.fused_train = True
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(3):
        v, lp, ent = pol.evaluate_actions(x, g, s, a)
        loss = lp.mean() + v.pow(2).mean() - 0.01 * ent
        opt.zero_grad(); loss.backward(); opt.step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=16, max_name_column_width=70))
