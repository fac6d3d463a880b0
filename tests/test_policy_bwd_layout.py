"""The backward kernel of the policy's conv front end (csrc/mrca_policy_bwd.hip) chains four MFMA contractions through
one LDS image and the fixed lane layouts of v_mfma_f32_32x32x2_f32 (conv1 recompute -> conv2 wgrad -> conv2 dgrad whose
accumulators ARE the B operand of conv1's wgrad).  This test re-enacts one wavefront's data movement in NumPy -- the same
steps in the same order, the address formulas taken from the kernel's own header (mrca_policy_layout.h, compiled into
the host harness) -- and compares the partial sums with torch.autograd through Conv1d -> ReLU -> Conv1d -> ReLU, so an
indexing slip is caught on the CPU before any GPU time is spent.  The GPU test (tests/test_gpu_policy_bwd.py) checks
the real kernel."""
import ctypes as C
import os
import re

import numpy as np
import torch
import torch.nn.functional as F

import util as U

SRC = open(os.path.join(U.ROOT, "rl-collision-avoidance_amd", "csrc", "mrca_policy_bwd.hip")).read()
L = U.emul_lib()
_c = (C.c_int * 16)()
L.pl_constants(_c)
(XP, GP, HP, kXE, kXO, kG2, kH1E, kH1O, WAVE_FLOATS, WAVES, P_DW2, P_DW1, P_DB1, P_DB2, P_FLOATS, HALF) = list(_c)

LANE = np.arange(64)
COL, HL = LANE & 31, LANE >> 5


def rowmap(r, hl):
    return np.vectorize(lambda a, b: L.pl_rowmap(int(a), int(b)))(r, hl)


def xbase(kk):
    return np.vectorize(lambda k: L.pl_x_operand_base(int(k)))(kk)


def h1_store(p, h):
    return np.vectorize(lambda q: L.pl_h1_store_off(int(q), int(h)))(p)


ROW = np.stack([rowmap(np.full(64, r), HL) for r in range(16)], 1)      # [lane, reg]


def mfma(a, b, acc):
    """v_mfma_f32_32x32x2_f32: lane l holds A[i = l & 31][k = l >> 5], B[k = l >> 5][j = l & 31]; acc[lane][reg] is
    C[row = rowmap(reg, l >> 5)][col = l & 31]."""
    A = np.zeros((32, 2), np.float64)
    B = np.zeros((2, 32), np.float64)
    A[COL, HL] = a
    B[HL, COL] = b
    Cm = A @ B
    acc += Cm[ROW, COL[:, None]]


def test_lds_budget_and_layout_constants():
    assert WAVES * WAVE_FLOATS * 4 <= 160 * 1024
    assert GP % 2 == 1 and HP % 2 == 1          # rows read with lanes over channels: odd pitches are conflict-free
    assert kXO == 3 * XP and kG2 == 6 * XP and kH1E == kG2 + 32 * GP and kH1O == kH1E + 32 * HP
    assert WAVE_FLOATS == kH1O + 32 * HP
    # the kernel uses the header's formulas, not private copies
    for name in ("x_operand_base(", "h1_store_off(", "rowmap(", "conv1_pstart("):
        assert name in SRC
    assert len(re.findall(r"__builtin_amdgcn_mfma_f32_32x32x2f32", SRC)) == 9


def reenact(x, w1, b1, w2, feat, gfeat):
    """One wave's work on the items x[n] of ONE tower; returns its partial sums (dw2, dw1, db1, db2)."""
    lds = np.full(WAVE_FLOATS, np.nan)                      # NaN everywhere: a read of an unwritten word shows up
    a1 = [np.where(2 * s + HL < 15, w1.reshape(32, 15)[COL, np.minimum(2 * s + HL, 14)], b1[COL]) for s in range(8)]
    w2f = [[w2[2 * s + HL, COL, tap] for s in range(16)] for tap in range(3)]
    for ci in range(3):
        lds[kXO + ci * XP] = 0
        lds[kXE + ci * XP + 256: kXE + (ci + 1) * XP] = 0
        lds[kXO + ci * XP + 257: kXO + (ci + 1) * XP] = 0
    lds[kG2 + np.arange(32) * GP + 128] = 0
    acc2 = [np.zeros((64, 16)) for _ in range(3)]
    acc1 = np.zeros((64, 16))
    db2p = np.zeros(64)
    xb1 = [xbase(np.minimum(2 * s + HL, 14)) for s in range(8)]
    xbw = xbase(np.where(COL < 15, COL, 0))
    ones_row = COL >= 15
    for n in range(x.shape[0]):
        for idx in range(384):
            ci, m = idx >> 7, idx & 127
            v = x[n, ci, 4 * m: 4 * m + 4]
            xe, xo = kXE + ci * XP + 2 * m, kXO + ci * XP + 2 * m + 1
            lds[xe], lds[xo], lds[xe + 1], lds[xo + 1] = v
        for idx in range(1024):
            c, m = idx >> 5, idx & 31
            g, f = gfeat[n, c * 128 + 4 * m: c * 128 + 4 * m + 4], feat[n, c * 128 + 4 * m: c * 128 + 4 * m + 4]
            lds[kG2 + c * GP + 4 * m: kG2 + c * GP + 4 * m + 4] = np.where(f > 0, g, 0)
        for h in range(2):
            lds[kH1O + np.arange(32) * HP + (HALF if h else 0)] = 0
            pstart = L.pl_conv1_pstart(h)
            for T in range(4):
                acc = np.zeros((64, 16))
                for s in range(8):
                    b = lds[xb1[s] + pstart + 32 * T + COL]
                    if s == 7:
                        b = np.where(HL == 1, 1.0, b)
                    mfma(a1[s], b, acc)
                dst = h1_store(pstart + 32 * T + COL, h)
                for r in range(16):
                    lds[dst + ROW[:, r] * HP] = np.maximum(acc[:, r], 0)
            for s in range(32):
                i = 2 * s + HL
                a = lds[kG2 + COL * GP + HALF * h + i]
                b0 = lds[kH1O + COL * HP + i]
                b1v = lds[kH1E + COL * HP + i]
                b2 = lds[kH1O + COL * HP + i + 1]
                db2p += a
                mfma(a, b0, acc2[0])
                mfma(a, b1v, acc2[1])
                mfma(a, b2, acc2[2])
            for u in range(2):
                L0 = HALF * h + 32 * u
                accE, accO = np.zeros((64, 16)), np.zeros((64, 16))
                for s in range(16):
                    g = kG2 + (2 * s + HL) * GP + L0 + COL
                    ae, as_ = lds[g], lds[g + 1]
                    mfma(ae, w2f[1][s], accE)
                    mfma(ae, w2f[2][s], accO)
                    mfma(as_, w2f[0][s], accO)
                for r in range(16):
                    i = 32 * u + ROW[:, r]
                    accE[:, r] = np.where(lds[kH1E + COL * HP + i] > 0, accE[:, r], 0)
                    accO[:, r] = np.where(lds[kH1O + COL * HP + i + 1] > 0, accO[:, r], 0)
                for r in range(16):
                    p = 2 * (L0 + ROW[:, r])
                    xe, xo = lds[xbw + p], lds[xbw + p + 1]
                    mfma(np.where(ones_row, 1.0, xe), accE[:, r], acc1)
                    mfma(np.where(ones_row, 1.0, xo), accO[:, r], acc1)
    P = np.full(P_FLOATS, np.nan)
    for tap in range(3):
        for r in range(16):
            P[P_DW2 + (ROW[:, r] * 32 + COL) * 3 + tap] = acc2[tap][:, r]
    for r in range(16):
        i = ROW[:, r]
        m = i < 15
        P[P_DW1 + COL[m] * 15 + i[m]] = acc1[m, r]
        m = i == 15
        P[P_DB1 + COL[m]] = acc1[m, r]
    P[P_DB2 + COL[:32]] = db2p[:32] + db2p[32:]
    assert not np.isnan(P).any()
    return (P[P_DW2:P_DW1].reshape(32, 32, 3), P[P_DW1:P_DB1].reshape(32, 3, 5), P[P_DB1:P_DB2], P[P_DB2:])


def test_kernel_data_movement_reproduces_the_autograd_gradients():
    rng = np.random.default_rng(0)
    N = 2
    x = rng.uniform(-0.5, 0.5, (N, 3, 512))
    w1 = rng.normal(0, 0.3, (32, 3, 5))
    b1 = rng.normal(0, 0.1, 32)
    w2 = rng.normal(0, 0.1, (32, 32, 3))
    b2 = rng.normal(0, 0.1, 32)
    t = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)      # noqa: E731
    tw1, tb1, tw2, tb2 = t(w1), t(b1), t(w2), t(b2)
    h1 = torch.relu(F.conv1d(torch.tensor(x), tw1, tb1, stride=2, padding=1))
    feat = torch.relu(F.conv1d(h1, tw2, tb2, stride=2, padding=1)).flatten(1)
    gfeat = torch.tensor(rng.normal(0, 1, (N, 4096)))
    feat.backward(gfeat)
    assert 0.2 < float((feat > 0).double().mean()) < 0.8 and 0.2 < float((h1 > 0).double().mean()) < 0.8
    dw2, dw1, db1, db2 = reenact(x, w1, b1, w2, feat.detach().numpy(), gfeat.numpy())
    for got, want, name in ((dw2, tw2.grad, "dw2"), (dw1, tw1.grad, "dw1"), (db1, tb1.grad, "db1"), (db2, tb2.grad, "db2")):
        err = np.abs(got - want.numpy()).max()
        assert err < 1e-9 * max(1.0, float(want.abs().max())), (name, err)
