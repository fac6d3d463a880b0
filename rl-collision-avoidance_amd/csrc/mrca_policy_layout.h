// mrca_policy_layout.h -- LDS images and operand address formulas of the FORWARD (csrc/mrca_policy.hip, namespace
// mrca_pfwd) and BACKWARD (csrc/mrca_policy_bwd.hip, namespace mrca_pbwd) kernels of the policy's lidar front end.  Plain
// integer functions, shared by the gfx950 kernels and by the host harness (tests/host_emul/emul.cpp exports them;
// tests/test_policy_conv_layout.py and tests/test_policy_bwd_layout.py re-enact the kernels' data movement with them on
// the CPU before any GPU time is spent).
#pragma once

#if defined(__HIPCC__)
#define MRCA_PL_HD __host__ __device__ inline
#else
#define MRCA_PL_HD inline
#endif

namespace mrca_pbwd {

constexpr int kBeams = 512, kFrames = 3, kCh = 32;
constexpr int kL1 = 255;             // conv1 output length (512 + 2 - 5) / 2 + 1
constexpr int kL2 = 128;             // conv2 output length (255 + 2 - 3) / 2 + 1
constexpr int kHalf = 64;            // conv2 positions per half of an item

// LDS image of one wave (floats).  A wave owns one (sample, tower) at a time:
//   XE[3][kXPitch], XO[3][kXPitch]  the scan de-interleaved: XE[ci][m] = x[ci][2m], XO[ci][m + 1] = x[ci][2m + 1], XO[ci][0] = x[ci][-1] = 0
//   G2[32][kGPitch]                 ONE HALF of g2 = gfeat * (feat > 0): G2[c][i] = g2[c][64h + i], i in [0, 64]
//                                   (conv2's dgrad reads l + 1: column 64 is g2[c][64] for h = 0 and 0 for h = 1)
//   H1E[32][kHPitch], H1O[32][kHPitch]   ONE HALF of relu(conv1) de-interleaved.  Half h covers conv2 positions
//                                   l in [64h, 64h + 64), i.e. h1 positions p in [128h - 1, 128h + 127]:
//                                     H1E[c][i] = h1[c][2 (64h + i)]          i in [0, 64)
//                                     H1O[c][i] = h1[c][2 (64h + i) - 1]      i in [0, 64]   (h = 0: i = 0 is the left padding; h = 1: i = 64 is the right padding)
// Odd row pitches for everything that is read with lanes running over ROWS (channels): conflict-free ds_read_b32.
constexpr int kXPitch = 260;
constexpr int kGPitch = 65;
constexpr int kHPitch = 65;
constexpr int kXE = 0, kXO = 3 * kXPitch;
constexpr int kG2 = 6 * kXPitch;
constexpr int kH1E = kG2 + kCh * kGPitch;
constexpr int kH1O = kH1E + kCh * kHPitch;
constexpr int kWaveFloats = kH1O + kCh * kHPitch;
constexpr int kWavesPerBlock = 4;
// behind the four waves' images: conv2's weights of both towers, tap-major, W2L[tower][tap][c][ci] (dgrad's B operand)
constexpr int kW2LFloats = 2 * 3 * kCh * kCh;
constexpr int kBlockFloats = kWavesPerBlock * kWaveFloats + kW2LFloats;
static_assert(kBlockFloats * 4 <= 160 * 1024, "one workgroup of 4 waves per CU");

// per-wave partial sums handed to the finalize kernel: dw2[32][32][3] | dw1[32][3][5] | db1[32] | db2[32]
constexpr int kPartDw2 = 0, kPartDw1 = 3072, kPartDb1 = 3072 + 480, kPartDb2 = kPartDb1 + 32, kPartFloats = kPartDb2 + 32;

// C/D layout of v_mfma_f32_32x32x2_f32: lane holds column (lane & 31), register r holds row rowmap(r, lane >> 5)
MRCA_PL_HD int rowmap(int reg, int hl) { return (reg & 3) + 8 * (reg >> 2) + 4 * hl; }

// x[ci][2p + tap - 1] (the conv1 operand of output position p, kk = ci * 5 + tap < 15) lives at x_operand_base(kk) + p:
//   tap 0 -> XO[ci][p]   tap 1 -> XE[ci][p]   tap 2 -> XO[ci][p + 1]   tap 3 -> XE[ci][p + 1]   tap 4 -> XO[ci][p + 2]
MRCA_PL_HD int x_operand_base(int kk) {
    const int ci = kk / 5, tap = kk % 5;
    return ((tap & 1) ? kXE : kXO) + ci * kXPitch + tap / 2;
}

// first h1 position half h recomputes: 128 positions p = conv1_pstart(h) + 0..127  (h = 0: 0..127, h = 1: 127..254)
MRCA_PL_HD int conv1_pstart(int h) { return h ? 127 : 0; }

// where h1[.][p] goes in the half-h image (add channel * kHPitch)
MRCA_PL_HD int h1_store_off(int p, int h) {
    return (p & 1) ? (kH1O + ((p + 1) >> 1) - kHalf * h) : (kH1E + (p >> 1) - kHalf * h);
}

}  // namespace mrca_pbwd

namespace mrca_pfwd {

constexpr int kBeams = 512, kFrames = 3, kCh = 32;
constexpr int kL1 = 255, kL2 = 128;

// LDS image of one wave (floats):
//   XE[3][kXPitch], XO[3][kXPitch]     the scan de-interleaved (as in mrca_pbwd)
//   H1E[32][kHPitch], H1O[32][kHPitch] relu(conv1) de-interleaved: H1E[c][j] = h1[c][2j], H1O[c][j + 1] = h1[c][2j + 1],
//                                      H1O[c][0] = h1[c][-1] = 0, H1O[c][128] = h1[c][255] = 0 (conv2's paddings)
//   H1E[c][64 t' .. 64 t' + 63] doubles as the transposition buffer of the epilogue of conv2's tile pair t' once that
//   pair's MFMAs have read it: the accumulators (positions on lanes) go in, rows of 64 positions come out as float4.
// kHPitch is a multiple of 4 floats so that those rows are 16-byte aligned.
constexpr int kXPitch = 260;
constexpr int kHPitch = 132;
constexpr int kXE = 0, kXO = 3 * kXPitch;
constexpr int kH1E = 6 * kXPitch;
constexpr int kH1O = kH1E + kCh * kHPitch;
constexpr int kWaveFloats = kH1O + kCh * kHPitch;
constexpr int kWavesPerBlock = 4;
static_assert(kWavesPerBlock * kWaveFloats * 4 <= 160 * 1024, "one workgroup of 4 waves per CU");
static_assert(kHPitch % 4 == 0 && kH1E % 4 == 0 && kWaveFloats % 4 == 0, "float4 rows of the epilogue");

MRCA_PL_HD int rowmap(int reg, int hl) { return (reg & 3) + 8 * (reg >> 2) + 4 * hl; }

// The K index of a contraction may be enumerated in any order as long as both operands agree.  The orders below pair
// the two k of one MFMA step (hl = 0 / 1) so that their LDS operands differ by a CONSTANT address offset: the lane's hl
// then folds into one base pointer per family and the step's offset is an instruction immediate (with kk = 2s + hl
// every step needed an address register of its own).
//
// conv1, 8 steps: kk = ci * 5 + tap of (step s, hl), 15 = the bias (B operand 1.0)
//   s = 2 ci     : taps 0 | 2  ->  XO[ci][p] | XO[ci][p + 1]          family 1: base = col + hl
//   s = 2 ci + 1 : taps 1 | 3  ->  XE[ci][p] | XE[ci][p + 1]          family 1
//   s = 6        : tap 4 of ci = 0 | 1 -> XO[0][p + 2] | XO[1][p + 2] family 2: base = col + hl * kXPitch
//   s = 7        : tap 4 of ci = 2 | bias -> XO[2][p + 2] | 1.0       family 3: base = col
MRCA_PL_HD int conv1_kk(int s, int hl) {
    if (s < 6) return (s >> 1) * 5 + ((s & 1) ? (hl ? 3 : 1) : (hl ? 2 : 0));
    if (s == 6) return (hl ? 1 : 0) * 5 + 4;
    return hl ? 15 : 2 * 5 + 4;
}
MRCA_PL_HD int conv1_family(int s) { return s < 6 ? 1 : (s == 6 ? 2 : 3); }
MRCA_PL_HD int conv1_step_off(int s) {           // added to the family base and the position p
    if (s < 6) return ((s & 1) ? kXE : kXO) + (s >> 1) * kXPitch;
    if (s == 6) return kXO + 2;
    return kXO + 2 * kXPitch + 2;
}
// conv2, 48 steps: (ci, tap) of (step s, hl); the operand of position l is h1[ci][2l + tap - 1]
//   s < 32  : ci = s, taps 0 | 2          ->  H1O[ci][l] | H1O[ci][l + 1]       family A: base = col + hl
//   s >= 32 : ci = 2 (s - 32) + hl, tap 1 ->  H1E[ci][l]                        family B: base = col + hl * kHPitch
MRCA_PL_HD int conv2_ci(int s, int hl) { return s < 32 ? s : 2 * (s - 32) + hl; }
MRCA_PL_HD int conv2_tap(int s, int hl) { return s < 32 ? (hl ? 2 : 0) : 1; }
MRCA_PL_HD int conv2_step_off(int s) { return s < 32 ? kH1O + s * kHPitch : kH1E + 2 * (s - 32) * kHPitch; }

// where h1[.][p] goes (add channel * kHPitch)
MRCA_PL_HD int h1_store_off(int p) { return (p & 1) ? (kH1O + ((p + 1) >> 1)) : (kH1E + (p >> 1)); }

}  // namespace mrca_pfwd
