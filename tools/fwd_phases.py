#!/usr/bin/env python3
"""GPU, profiling build: where a wave of lidar_features_kernel spends a robot's time -- s_memtime ticks (shader clocks) per
phase against the MFMA work of the phase (64 clocks per v_mfma_f32_32x32x2_f32: tools/mfma_probe.hip)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-collision-avoidance_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as G  # noqa: E402

G.build()
from mrca import _lib  # noqa: E402

lib = _lib.load(_lib.PROFILING_LIB_PATH)
g = torch.Generator(device="cuda").manual_seed(0)
for N in [int(a) for a in sys.argv[1:]] or [4096, 16384]:
    obs = torch.rand(N, 3, 512, device="cuda", generator=g) - 0.5
    w1 = torch.randn(2, 32, 3, 5, device="cuda", generator=g) * 0.3
    b1 = torch.randn(2, 32, device="cuda", generator=g) * 0.1
    w2 = torch.randn(2, 32, 32, 3, device="cuda", generator=g) * 0.1
    b2 = torch.randn(2, 32, device="cuda", generator=g) * 0.1
    out = torch.empty(2, N, 4096, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    args = (obs.data_ptr(), None, 0, N, 3, 512, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), out.data_ptr(), st)
    for _ in range(3):
        _lib.check(lib.mrca_lidar_features(*args), "mrca_lidar_features")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        _lib.check(lib.mrca_lidar_features(*args), "mrca_lidar_features")
    e1.record()
    torch.cuda.synchronize()
    t = (C.c_double * 8)()
    _lib.check(lib.mrca_debug_fwd_stamps(t), "mrca_debug_fwd_stamps")
    names = ("conv1 pair 0 (+ previous robot's output)", "conv1 pair 1", "conv1 pair 2", "conv1 pair 3", "conv2 pair 0 (+ next robot's scan)",
             "conv2 pair 1 (+ pair 0's output)")
    work = (16, 16, 16, 16, 96, 96)
    total = sum(t[k] for k in range(6))
    print(f"{N} rows: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per launch (stamped build), {t[6]:.1f} robots per wave, {total:.0f} ticks per robot "
          f"(MFMA work: {256 * 64}); shader clock during the loop {t[7]:.3f} GHz")
    for k in range(6):
        print(f"   {names[k]:<44} {t[k]:8.0f} ticks   {work[k]:3d} MFMAs = {work[k] * 64:5d}   x{t[k] / (work[k] * 64):.2f}")
