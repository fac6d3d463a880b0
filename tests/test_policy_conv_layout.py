"""The fused conv front end (csrc/mrca_policy.hip) is two GEMMs on implicit im2col operands read from a de-interleaved
LDS image, through the fixed lane layouts of v_mfma_f32_32x32x2_f32, with the K index enumerated in an order that turns
every operand address into "lane-constant base + immediate" and an epilogue that transposes the accumulators through
LDS.  This test re-enacts one wavefront's data movement in NumPy -- same staging, same steps, the K orders and address
formulas taken from the kernel's own header (mrca_policy_layout.h, compiled into the host harness) -- and compares with
torch's Conv1d, so that an indexing slip is caught on the CPU before any GPU time is spent.  (The GPU test,
tests/test_gpu_policy_ops.py, checks the real kernel against the PyTorch layers to 1e-5.)"""
import ctypes as C
import os
import re

import numpy as np
import torch
import torch.nn.functional as F

import util as U

SRC = open(os.path.join(U.ROOT, "rl-collision-avoidance_amd", "csrc", "mrca_policy.hip")).read()
L = U.emul_lib()
_c = (C.c_int * 8)()
L.pf_constants(_c)
XP, HP, kXE, kXO, kH1E, kH1O, WAVE_FLOATS, WAVES = list(_c)
LANE = np.arange(64)
COL, HL = LANE & 31, LANE >> 5
ROW0 = [(r & 3) + 8 * (r >> 2) for r in range(16)]
ROW = np.stack([np.array(ROW0[r]) + 4 * HL for r in range(16)], 1)       # [lane, reg]


def mfma(a, b, acc):
    """v_mfma_f32_32x32x2_f32: lane l holds A[i = l & 31][k = l >> 5], B[k = l >> 5][j = l & 31]; acc[lane][reg]."""
    A = np.zeros((32, 2), np.float64)
    B = np.zeros((2, 32), np.float64)
    A[COL, HL] = a
    B[HL, COL] = b
    acc += (A @ B)[ROW, COL[:, None]]


def test_lds_budget_and_alignment():
    assert WAVES * WAVE_FLOATS * 4 <= 160 * 1024
    assert HP % 4 == 0 and kH1E % 4 == 0 and WAVE_FLOATS % 4 == 0      # float4 rows of the epilogue
    for name in ("conv1_kk(", "conv1_step_off(", "conv2_ci(", "conv2_tap(", "conv2_step_off(", "rowmap("):
        assert name in SRC
    assert len(re.findall(r"MRCA_MFMA\(", SRC)) == 5                    # the macro + conv1 (2) + conv2 (2)


def test_k_orders_enumerate_every_k_once():
    k1 = sorted(L.pf_conv1_kk(s, hl) for s in range(8) for hl in range(2))
    assert k1 == list(range(16))
    k2 = sorted(L.pf_conv2_ci(s, hl) * 3 + L.pf_conv2_tap(s, hl) for s in range(48) for hl in range(2))
    assert k2 == list(range(96))


def test_kernel_data_movement_reproduces_conv1d_relu_conv1d_relu():
    rng = np.random.default_rng(0)
    x = rng.uniform(-0.5, 0.5, (3, 512))
    w1 = rng.normal(0, 0.3, (32, 3, 5))
    b1 = rng.normal(0, 0.1, 32)
    w2 = rng.normal(0, 0.1, (32, 32, 3))
    b2 = rng.normal(0, 0.1, 32)
    lds = np.full(WAVE_FLOATS, np.nan)       # NaN everywhere: any read of an unwritten word shows up
    # constant parts
    for ci in range(3):
        lds[kXO + ci * XP] = 0
        lds[kXE + ci * XP + 256: kXE + (ci + 1) * XP] = 0
        lds[kXO + ci * XP + 257: kXO + (ci + 1) * XP] = 0
    for c in range(32):
        lds[kH1O + c * HP] = 0
        lds[kH1O + c * HP + 128] = 0
    # fragments
    kk1 = [np.array([L.pf_conv1_kk(s, int(h)) for h in HL]) for s in range(8)]
    a1 = [np.where(kk1[s] < 15, w1.reshape(32, 15)[COL, np.minimum(kk1[s], 14)], b1[COL]) for s in range(8)]
    a2 = [w2[COL, np.array([L.pf_conv2_ci(s, int(h)) for h in HL]), np.array([L.pf_conv2_tap(s, int(h)) for h in HL])]
          for s in range(48)]
    bias2 = b2[ROW]                                                     # [lane, reg]
    x1, x2, x3 = COL + HL, COL + HL * XP, COL
    hst = np.where(COL & 1, kH1O + (COL + 1) // 2, kH1E + COL // 2) + 4 * HL * HP
    ha, hb = COL + HL, COL + HL * HP
    oe = kH1E + 4 * HL * HP + COL
    orow = kH1E + (LANE >> 4) * HP + 4 * (LANE & 15)
    gofs = (LANE >> 4) * 128 + 4 * (LANE & 15)
    # staging
    for idx in range(384):
        ci, m = idx >> 7, idx & 127
        v = x[ci, 4 * m: 4 * m + 4]
        xe, xo = kXE + ci * XP + 2 * m, kXO + ci * XP + 2 * m + 1
        lds[xe], lds[xo], lds[xe + 1], lds[xo + 1] = v
    # conv1
    for T in range(0, 8, 2):
        acca, accb = np.zeros((64, 16)), np.zeros((64, 16))
        for s in range(8):
            base = {1: x1, 2: x2, 3: x3}[L.pf_conv1_family(s)]
            ba, bb = lds[base + L.pf_conv1_step_off(s) + 32 * T], lds[base + L.pf_conv1_step_off(s) + 32 * T + 32]
            if s == 7:
                ba, bb = np.where(HL == 1, 1.0, ba), np.where(HL == 1, 1.0, bb)
            mfma(a1[s], ba, acca)
            mfma(a1[s], bb, accb)
        # the store base is the header's h1_store_off(p) + row * pitch
        assert np.array_equal(hst + 16 * T, np.array([L.pf_h1_store_off(32 * T + int(c)) for c in COL]) + 4 * HL * HP)
        for r in range(16):
            lds[hst + 16 * T + ROW0[r] * HP] = np.maximum(acca[:, r], 0)
            keep = np.ones(64, bool) if T < 6 else COL != 31
            lds[(hst + 16 * T + 16 + ROW0[r] * HP)[keep]] = np.maximum(accb[:, r], 0)[keep]
    assert np.all(lds[kH1O + np.arange(32) * HP + 128] == 0)             # h1[.][255] stayed the padding
    # conv2 + epilogue
    out = np.full(32 * 128, np.nan)
    done = None
    for P in range(2):
        acc0, acc1 = bias2.copy(), bias2.copy()
        for ch in range(12):
            if P == 1 and ch == 1:
                for r in range(16):
                    lds[oe + ROW0[r] * HP] = np.maximum(done[0][:, r], 0)
                    lds[oe + ROW0[r] * HP + 32] = np.maximum(done[1][:, r], 0)
            if P == 1 and ch in (3, 5):
                for q in range(4 * (ch == 5), 4 * (ch == 5) + 4):
                    for e in range(4):
                        out[gofs + 4 * q * 128 + e] = lds[orow + 4 * q * HP + e]
            for k in range(4):
                s = 4 * ch + k
                base = ha if s < 32 else hb
                off = L.pf_conv2_step_off(s) + 64 * P
                mfma(a2[s], lds[base + off], acc0)
                mfma(a2[s], lds[base + off + 32], acc1)
        if P == 0:
            done = (acc0, acc1)
    for r in range(16):
        lds[oe + ROW0[r] * HP + 64] = np.maximum(acc0[:, r], 0)
        lds[oe + ROW0[r] * HP + 96] = np.maximum(acc1[:, r], 0)
    for q in range(8):
        for e in range(4):
            out[gofs + 4 * q * 128 + 64 + e] = lds[orow + 4 * q * HP + 64 + e]
    xt = torch.from_numpy(x)[None]
    h1 = torch.relu(F.conv1d(xt, torch.from_numpy(w1), torch.from_numpy(b1), stride=2, padding=1))
    h2 = torch.relu(F.conv1d(h1, torch.from_numpy(w2), torch.from_numpy(b2), stride=2, padding=1))
    want = h2.flatten(1)[0].numpy()
    assert not np.isnan(out).any()
    assert np.abs(out - want).max() < 1e-12, np.abs(out - want).max()
