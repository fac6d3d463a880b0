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


def _relu_bits(v):
    """the kernel's ReLU: max(as_int(x), 0) as float (one v_max_i32)."""
    return np.maximum(np.asarray(v, np.float32).view(np.int32), 0).view(np.float32).astype(np.float64)


def test_relu_as_integer_max():
    v = np.array([0.0, -0.0, 1.5, -1.5, 1e-40, -1e-40, np.inf, -np.inf, 3.4e38, -3.4e38], np.float32)
    assert np.array_equal(_relu_bits(v), np.maximum(v, 0).astype(np.float64))
    assert "max(__float_as_int(x), 0)" in SRC


def test_kernel_data_movement_reproduces_conv1d_relu_conv1d_relu():
    """Three robots through ONE wave, every LDS / register event in the kernel's program order (the LDS operations of a
    wave complete in order, so program order is the hazard rule): the software pipeline overlaps the previous robot's
    output, this robot's two convolutions and the next robot's staging in one LDS image."""
    rng = np.random.default_rng(0)
    xs = [rng.uniform(-0.5, 0.5, (3, 512)) for _ in range(3)]
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
    for T in range(8):      # the store base is the header's h1_store_off(p) + row * pitch
        assert np.array_equal(hst + 16 * T, np.array([L.pf_h1_store_off(32 * T + int(c)) for c in COL]) + 4 * HL * HP)

    def stage(x, q):                      # MRCA_STAGE_SCAN(q): float4 q of every lane
        for idx in range(q * 64, q * 64 + 64):
            ci, m = idx >> 7, idx & 127
            v = x[ci, 4 * m: 4 * m + 4]
            xe, xo = kXE + ci * XP + 2 * m, kXO + ci * XP + 2 * m + 1
            lds[xe], lds[xo], lds[xe + 1], lds[xo + 1] = v

    def conv1_load(T):
        ba, bb = [], []
        for s in range(8):
            base = {1: x1, 2: x2, 3: x3}[L.pf_conv1_family(s)]
            a, b = lds[base + L.pf_conv1_step_off(s) + 32 * T], lds[base + L.pf_conv1_step_off(s) + 32 * T + 32]
            if s == 7:
                a, b = np.where(HL == 1, 1.0, a), np.where(HL == 1, 1.0, b)
            ba.append(a)
            bb.append(b)
        return ba, bb

    def conv1_out(acc, tpp, r):
        lds[hst + 32 * tpp + ROW0[r] * HP] = _relu_bits(acc[0][:, r])
        lds[hst + 32 * tpp + 16 + ROW0[r] * HP] = _relu_bits(acc[1][:, r])

    def conv2_load_k(g, k):
        s = 4 * (g % 12) + k
        base = ha if s < 32 else hb
        off = L.pf_conv2_step_off(s) + 64 * (g // 12)
        return lds[base + off], lds[base + off + 32]

    def conv2_out(acc, P, r):
        lds[oe + ROW0[r] * HP + 64 * P] = _relu_bits(acc[0][:, r])
        lds[oe + ROW0[r] * HP + 64 * P + 32] = _relu_bits(acc[1][:, r])

    def row_read(q, col0):
        return np.stack([lds[orow + 4 * q * HP + col0 + e] for e in range(4)], 1)

    def row_store(out, q, col0, v):
        for e in range(4):
            out[gofs + 4 * q * 128 + col0 + e] = v[:, e]

    outs = []
    for q in range(6):
        stage(xs[0], q)
    opb = [None, None]
    opb[0] = conv1_load(0)
    p1 = (np.zeros((64, 16)), np.zeros((64, 16)))
    for i, x in enumerate(xs):
        nxt_x = xs[i + 1] if i + 1 < len(xs) else None
        out = np.full(32 * 128, np.nan)
        c1 = [None, None]
        row = [None] * 8
        b2buf = [[None] * 4, [None] * 4]
        for tp in range(4):
            cur, nxt = tp & 1, (tp & 1) ^ 1
            if tp < 3:
                opb[nxt] = conv1_load(2 * tp + 2)
            acc = (np.zeros((64, 16)), np.zeros((64, 16)))
            for s in range(8):
                mfma(a1[s], opb[cur][0][s], acc[0])
                mfma(a1[s], opb[cur][1][s], acc[1])
                if tp > 0:
                    conv1_out(c1[nxt], tp - 1, 2 * s)
                    conv1_out(c1[nxt], tp - 1, 2 * s + 1)
                if tp == 0:
                    conv2_out(p1, 1, 2 * s)
                    conv2_out(p1, 1, 2 * s + 1)
                if tp == 1:
                    row[s] = row_read(s, 64)
                if tp == 2 and outs:
                    row_store(outs[-1], s, 64, row[s])
                if tp == 3 and s == 5:
                    for k in range(4):
                        b2buf[0][k] = conv2_load_k(0, k)
            c1[cur] = acc
        done = None
        for g in range(24):
            P, ch, cur, nxt = g // 12, g % 12, g & 1, (g & 1) ^ 1
            if ch == 0:
                acc0, acc1 = bias2.copy(), bias2.copy()
            for k in range(4):
                s = 4 * ch + k
                mfma(a2[s], b2buf[cur][k][0], acc0)
                mfma(a2[s], b2buf[cur][k][1], acc1)
                if g < 23:
                    b2buf[nxt][k] = conv2_load_k(g + 1, k)
                if P == 0:
                    if ch < 8:
                        if k < 2:
                            conv1_out(c1[1], 3, 2 * ch + k)
                    elif ch < 11:
                        if k < 2 and nxt_x is not None:
                            stage(nxt_x, 2 * (ch - 8) + k)
                        if ch == 8 and k == 2:
                            lds[kH1O + COL * HP + 128] = 0.0
                else:
                    if ch < 4:
                        conv2_out(done, 0, 4 * ch + k)
                    elif ch < 8:
                        if k < 2:
                            row[2 * (ch - 4) + k] = row_read(2 * (ch - 4) + k, 0)
                    else:
                        if k < 2:
                            row_store(out, 2 * (ch - 8) + k, 0, row[2 * (ch - 8) + k])
                        if ch == 11 and k == 2:
                            opb[0] = conv1_load(0)
            if g == 11:
                done = (acc0.copy(), acc1.copy())
        p1 = (acc0, acc1)
        outs.append(out)
    for r in range(16):
        conv2_out(p1, 1, r)
    for q in range(8):
        row_store(outs[-1], q, 64, row_read(q, 64))
    for x, out in zip(xs, outs):
        xt = torch.from_numpy(x)[None]
        h1 = torch.relu(F.conv1d(xt, torch.from_numpy(w1), torch.from_numpy(b1), stride=2, padding=1))
        h2 = torch.relu(F.conv1d(h1, torch.from_numpy(w2), torch.from_numpy(b2), stride=2, padding=1))
        want = h2.flatten(1)[0].numpy()
        assert not np.isnan(out).any()
        # the ReLU goes through fp32 bit patterns here: fp32 rounding of the fp64 emulation
        assert np.abs(out - want).max() < 1e-6, np.abs(out - want).max()


def test_weight_staging_reaches_the_same_fragments():
    """The prologue fetches W2 / W1 with coalesced float4 loads into padded LDS rows ([32][97], [32][17]) and picks the
    MFMA A fragments up from there: same values as indexing the weight tensors directly."""
    rng = np.random.default_rng(1)
    w1 = rng.normal(0, 0.3, (32, 3, 5))
    w2 = rng.normal(0, 0.1, (32, 32, 3))
    W2L, W1L = kH1E, kH1E + 32 * 97
    assert W1L + 32 * 17 <= WAVE_FLOATS and "kW1L = kH1E + 32 * 97" in SRC
    lds = np.full(WAVE_FLOATS, np.nan)
    f2, f1 = w2.reshape(-1), w1.reshape(-1)
    for q in range(12):
        for lane in range(64):
            f = q * 64 + lane
            d = W2L + (f // 24) * 97 + 4 * (f % 24)
            lds[d: d + 4] = f2[4 * f: 4 * f + 4]
    for q in range(2):
        for lane in range(64):
            if q == 0 or lane < 56:
                for j in range(4):
                    e = 4 * (q * 64 + lane) + j
                    lds[W1L + (e // 15) * 17 + e % 15] = f1[e]
    for s in range(48):
        ci = np.array([L.pf_conv2_ci(s, int(h)) for h in HL])
        tap = np.array([L.pf_conv2_tap(s, int(h)) for h in HL])
        assert np.array_equal(lds[W2L + COL * 97 + ci * 3 + tap], w2[COL, ci, tap])
    for s in range(8):
        kk = np.array([L.pf_conv1_kk(s, int(h)) for h in HL])
        m = kk < 15
        assert np.array_equal(lds[W1L + COL[m] * 17 + kk[m]], w1.reshape(32, 15)[COL[m], kk[m]])
