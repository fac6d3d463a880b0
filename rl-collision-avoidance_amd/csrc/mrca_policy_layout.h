// mrca_policy_layout.h -- LDS image and operand address formulas of the BACKWARD kernel of the policy's lidar front end
// (csrc/mrca_policy_bwd.hip).  Plain integer functions, shared by the gfx950 kernel and by the host harness
// (tests/host_emul/emul.cpp exports them, tests/test_policy_bwd_layout.py re-enacts the kernel's data movement with
// them on the CPU before any GPU time is spent).
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
//   G2[32][kGPitch]                 g2[c][l] = gfeat * (feat > 0) for l < 128, column 128 = 0 (conv2's dgrad reads l + 1)
//   H1E[32][kHPitch], H1O[32][kHPitch]   ONE HALF of relu(conv1) de-interleaved.  Half h covers conv2 positions
//                                   l in [64h, 64h + 64), i.e. h1 positions p in [128h - 1, 128h + 127]:
//                                     H1E[c][i] = h1[c][2 (64h + i)]          i in [0, 64)
//                                     H1O[c][i] = h1[c][2 (64h + i) - 1]      i in [0, 64]   (h = 0: i = 0 is the left padding; h = 1: i = 64 is the right padding)
// Odd row pitches for everything that is read with lanes running over ROWS (channels): conflict-free ds_read_b32.
constexpr int kXPitch = 260;
constexpr int kGPitch = 129;
constexpr int kHPitch = 65;
constexpr int kXE = 0, kXO = 3 * kXPitch;
constexpr int kG2 = 6 * kXPitch;
constexpr int kH1E = kG2 + kCh * kGPitch;
constexpr int kH1O = kH1E + kCh * kHPitch;
constexpr int kWaveFloats = kH1O + kCh * kHPitch;
constexpr int kWavesPerBlock = 4;
static_assert(kWavesPerBlock * kWaveFloats * 4 <= 160 * 1024, "one workgroup of 4 waves per CU");

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
