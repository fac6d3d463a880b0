#!/usr/bin/env python3
"""GPU, profiling build: where a wave of lidar_features_bwd_kernel spends an item's time -- s_memtime ticks (shader clocks) per
phase against the MFMA work of the phase (64 clocks per v_mfma_f32_32x32x2_f32: tools/mfma_probe.hip).  VERDICT r05 item 6a."""
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

PRODUCT = "--product" in sys.argv          # time the product build (no stamps) instead
sys.argv = [a for a in sys.argv if a != "--product"]
lib = _lib.load(_lib.LIB_PATH if PRODUCT else _lib.PROFILING_LIB_PATH)
g = torch.Generator(device="cuda").manual_seed(0)
for N in [int(a) for a in sys.argv[1:]] or [16384]:
    obs = torch.rand(N, 3, 512, device="cuda", generator=g) - 0.5
    w1 = torch.randn(2, 32, 3, 5, device="cuda", generator=g) * 0.3
    b1 = torch.randn(2, 32, device="cuda", generator=g) * 0.1
    w2 = torch.randn(2, 32, 32, 3, device="cuda", generator=g) * 0.1
    b2 = torch.randn(2, 32, device="cuda", generator=g) * 0.1
    feat = torch.empty(2, N, 4096, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.mrca_lidar_features(obs.data_ptr(), None, 0, N, 3, 512, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                       feat.data_ptr(), st), "mrca_lidar_features")
    ga = torch.randn(N, 4096, device="cuda", generator=g)
    gc = torch.randn(N, 4096, device="cuda", generator=g)
    dw1, db1, dw2, db2 = torch.empty_like(w1), torch.empty_like(b1), torch.empty_like(w2), torch.empty_like(b2)
    nb = C.c_size_t()
    _lib.check(lib.mrca_lidar_features_backward_scratch(C.byref(nb)), "scratch")
    scratch = torch.empty(nb.value, dtype=torch.uint8, device="cuda")
    args = (obs.data_ptr(), N, 3, 512, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), feat.data_ptr(), ga.data_ptr(), gc.data_ptr(),
            dw1.data_ptr(), db1.data_ptr(), dw2.data_ptr(), db2.data_ptr(), scratch.data_ptr(), scratch.numel(), st)
    for _ in range(3):
        _lib.check(lib.mrca_lidar_features_backward(*args), "mrca_lidar_features_backward")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        _lib.check(lib.mrca_lidar_features_backward(*args), "mrca_lidar_features_backward")
    e1.record()
    torch.cuda.synchronize()
    if PRODUCT:
        print(f"{N} rows: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per launch incl. finalize (product build)")
        continue
    t = (C.c_double * 10)()
    _lib.check(lib.mrca_debug_bwd_stamps(t), "mrca_debug_bwd_stamps")
    names = ("scan staged (per item)", "gradient rows staged + next requested (x2)", "conv1 recompute (x2)", "conv2 wgrad (x2)",
             "conv2 dgrad (x4)", "ReLU mask (x4)", "conv1 wgrad (x4)")
    work = (0, 0, 64, 192, 192, 0, 128)
    total = sum(t[k] for k in range(7))
    print(f"{N} rows: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per launch incl. finalize (stamped build), {t[8]:.1f} items per wave, {total:.0f} ticks per item "
          f"(MFMA work: {576 * 64}); shader clock during the loop {t[9]:.3f} GHz")
    for k in range(7):
        w = work[k] * 64
        print(f"   {names[k]:<46} {t[k]:8.0f} ticks   {work[k]:3d} MFMAs = {w:5d}" + (f"   x{t[k] / w:.2f}" if w else ""))
