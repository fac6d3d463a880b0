// mrca_policy_bwd.hip -- backward pass of the lidar front end of the actor-critic (model/net.py:19-25,37-49: two
// Conv1d + ReLU per tower) as ONE fused gfx950 kernel for the PPO update (model/ppo.py:158-192 differentiates through
// these layers for every minibatch; through MIOpen that is 85 % of an update: profiles/r01/r01_f_ppo_update_profile.txt).
//
//   given  gfeat_t[n][c*128 + l] = dLoss / dfeat of tower t  (feat = the forward kernel's output, mrca_policy.hip)
//   g2[c][l]   = gfeat * (feat > 0)                                                       l < 128
//   dw2[c][ci][k] = sum_{n,l} g2[c][l] * h1[ci][2l + k - 1]          db2[c] = sum_{n,l} g2[c][l]
//   dh1[ci][p] = sum_{c,k : 2l + k - 1 = p} w2[c][ci][k] * g2[c][l]  g1 = dh1 * (h1 > 0)  p < 255
//   dw1[c][ci][k] = sum_{n,p} g1[c][p] * x[ci][2p + k - 1]           db1[c] = sum_{n,p} g1[c][p]
//   (no gradient with respect to the scan: it is data)
//
// One wavefront owns one (sample, tower) at a time, persistent over the minibatch; every contraction is an fp32 MFMA
// (v_mfma_f32_32x32x2_f32, exact f32) whose operands are read from a de-interleaved LDS image by address (implicit
// im2col, mrca_policy_layout.h) or are registers already:
//   conv1 recompute  H1[32 ch][pos]      = W1[32][16] x X1[16][pos]         (h1 is not saved by the forward: 32 kB per item)
//   conv2 wgrad      DW2_k[32 c][32 ci] += G2[32 c][l] x H1_k[l][32 ci]     k = 0, 1, 2    contraction over positions
//   conv2 dgrad      D'[pos][32 ci]      = G2^T[pos][32 c] x W2_k[32 c][32 ci]             positions x channels: the C layout
//                                          puts channels on lanes and positions in registers ...
//   conv1 wgrad      DW1'[16 (ci,k)][32 c] += X1[16][pos] x D'[pos][32 c]   ... which IS the B operand layout of the
//                                          contraction over positions: dgrad's accumulators feed conv1's wgrad in place,
//                                          g1 never leaves the registers.  Row 15 of X1 is ones: that row of DW1' is db1.
// An item is processed in two halves of 64 conv2 positions so that the h1 and g2 images of a wave are 8.3 kB each and four
// waves fit the CU's 160 kB of LDS (31.2 kB each).  The tower's weights live in registers for the wave's life; the next
// half's gfeat / feat rows (and the next item's scan) are requested from HBM while the current half computes, and every
// LDS operand is requested a few MFMAs ahead of its use.  Per-wave partial sums go to a scratch buffer; a second kernel
// adds them in a fixed order (deterministic, no float atomics).
// MFMAs per (sample, tower): 64 + 192 + 192 + 128 = 576 (2.4 MFLOP); 16 384 x 2 items: 0.49 ms at the fp32 MFMA peak.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mrca_env.h"
#include "mrca_hostutil.h"
#include "mrca_policy_layout.h"

namespace mrca_pbwd {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// relu as ONE integer instruction (as the forward kernel forms it, mrca_policy.hip): max(bits, 0) -- every negative float, -0
// included, has a negative bit pattern; `x > 0 ? x : 0` compiled to a canonicalising max and a max
__device__ __forceinline__ float relu(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

__device__ inline f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.0f;
    return z;
}

// HBM inputs in flight.  The scan of an item: 6 float4 per lane.  One HALF of an item's gfeat / feat rows (32 rows x 64
// columns each): 8 + 8 float4 per lane, plus -- for half 0 -- column 64, which conv2's dgrad reads as "l + 1".
struct ScanLoads {
    float4 x[6];
};
struct GradLoads {
    float4 g[8], f[8];
    float ge, fe;
};

// `n` is WAVE-UNIFORM (the kernel makes it so with readfirstlane): an item's rows then start at a scalar base, the lane's share
// of the address is one loop-invariant 32-bit offset and the rest an immediate -- written with per-lane 64-bit pointers the 38
// loads of an item kept 76 address registers alive, the kernel ran out of VGPRs and moved ~440 values per item through AGPRs
// (profiles/r06_ak_*).
// rows != NULL: `obs` is a matrix of frames [*, 512] and rows[3 n + f] the row of item n's frame f (deque order) -- the rollout
// buffer's one-frame-per-tick store read in place (mrca/ppo.py FrameRows) instead of a gathered [n, 3, 512] copy of it
__device__ inline void request_scan(ScanLoads& ld, const float* __restrict__ obs, const int32_t* __restrict__ rows, int n, int lane) {
    if (rows) {
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            const float* xs = obs + (size_t)rows[3 * n + f] * kBeams + 4 * lane;
            ld.x[2 * f] = *reinterpret_cast<const float4*>(xs);
            ld.x[2 * f + 1] = *reinterpret_cast<const float4*>(xs + 256);
        }
        return;
    }
    const float* xs = obs + (size_t)n * (kFrames * kBeams) + 4 * lane;
#pragma unroll
    for (int q = 0; q < 6; ++q) ld.x[q] = *reinterpret_cast<const float4*>(xs + q * 256);
}

__device__ inline void request_grad(GradLoads& ld, const float* __restrict__ feat_t, const float* __restrict__ gfeat_t,
                                    int n, int h, int lane) {
    const size_t row = (size_t)n * (kCh * kL2) + kHalf * h;
    // float4 index idx = q * 64 + lane: c = idx / 16 = 4q + lane / 16, m = idx % 16 = lane % 16 -> [c][64h + 4m .. 4m+3]
    const int lo = (lane >> 4) * kL2 + 4 * (lane & 15);
    const float* gb = gfeat_t + row + lo;
    const float* fb = feat_t + row + lo;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        ld.g[q] = *reinterpret_cast<const float4*>(gb + q * 4 * kL2);
        ld.f[q] = *reinterpret_cast<const float4*>(fb + q * 4 * kL2);
    }
    ld.ge = 0.0f;
    ld.fe = 0.0f;
    if (h == 0 && lane < kCh) {
        ld.ge = gfeat_t[row + (size_t)lane * kL2 + kHalf];
        ld.fe = feat_t[row + (size_t)lane * kL2 + kHalf];
    }
}

__device__ inline void stage_scan(float* lds, const ScanLoads& ld, int lane) {
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int idx = q * 64 + lane;           // float4 index: ci = idx / 128, m = idx % 128 -> x[ci][4m .. 4m+3]
        const float4 v = ld.x[q];
        const int ci = idx >> 7, m = idx & 127;
        float* xe = lds + kXE + ci * kXPitch + 2 * m;
        float* xo = lds + kXO + ci * kXPitch + 2 * m + 1;
        xe[0] = v.x;
        xo[0] = v.y;
        xe[1] = v.z;
        xo[1] = v.w;
    }
}

__device__ inline void stage_grad(float* lds, const GradLoads& ld, int lane) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int idx = q * 64 + lane;
        const float4 g = ld.g[q], f = ld.f[q];
        float* dst = lds + kG2 + (idx >> 4) * kGPitch + 4 * (idx & 15);
        dst[0] = f.x > 0.0f ? g.x : 0.0f;
        dst[1] = f.y > 0.0f ? g.y : 0.0f;
        dst[2] = f.z > 0.0f ? g.z : 0.0f;
        dst[3] = f.w > 0.0f ? g.w : 0.0f;
    }
    if (lane < kCh) lds[kG2 + lane * kGPitch + kHalf] = ld.fe > 0.0f ? ld.ge : 0.0f;
}

#define MRCA_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
// Pins the order "operands of the NEXT chunk requested, then the MFMAs of THIS chunk": left to itself hipcc (ROCm 7.2)
// sinks the LDS reads next to their uses in half of this kernel's blocks ("ds_read, s_waitcnt lgkmcnt(0), v_mfma" --
// every MFMA group then waits a full LDS round trip; measured 41 % of the MFMA peak)
#define MRCA_PIN() __builtin_amdgcn_sched_barrier(0)

#if defined(MRCA_PROFILING)
// profiling build only: s_memtime ticks a wave spends in each phase of an item (summed over its items) + item count; reading
// the clock drains the LDS queue, so the stamped kernel runs a few per cent slower than the product (tools/bwd_phases.py)
constexpr int kBwdStamps = 10, kBwdStampWaves = 1024;
__device__ unsigned long long g_bwd_stamps[kBwdStamps][kBwdStampWaves];
#define MRCA_BSTAMP(k)                                              \
    {                                                               \
        MRCA_PIN();                                                 \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        bst[k] += t_ - bprev;                                       \
        bprev = t_;                                                 \
        MRCA_PIN();                                                 \
    }
#else
#define MRCA_BSTAMP(k)
#endif

__global__ __launch_bounds__(64 * kWavesPerBlock) void lidar_features_bwd_kernel(
    const float* __restrict__ obs, const int32_t* __restrict__ rows, int n_items, const float* __restrict__ w1,
    const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ feat, const float* __restrict__ gfeat_act,
    const float* __restrict__ gfeat_crt, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // scalar: and with it the item index and every row base
    float* lds = lds_all + wave * kWaveFloats;
    const int gwave = blockIdx.x * kWavesPerBlock + wave;
    const int nwaves = gridDim.x * kWavesPerBlock;
    const int tower = gwave & 1;                 // waves come in (actor, critic) pairs on the same samples
    const int col = lane & 31, hl = lane >> 5;
    const float* __restrict__ feat_t = feat + (size_t)tower * n_items * (kCh * kL2);
    const float* __restrict__ gfeat_t = tower ? gfeat_crt : gfeat_act;

    // --- the tower's weights in MFMA fragment form, for the wave's whole life
    // conv1 (A operand, A[i = out channel][k]): kk = 2s + hl; kk = 15 is the bias (its B operand is 1)
    float a1[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int kk = 2 * s + hl;
        a1[s] = kk < 15 ? w1[tower * 480 + col * 15 + kk] : b1[tower * 32 + col];
    }
    // conv2 dgrad's B operand B[k = c][j = ci] = w2[c][ci][tap]: both towers' weights tap-major in LDS behind the four
    // waves' images, W2L[tower][tap][c][ci] (24 kB per workgroup; 48 registers per lane if kept in fragments instead --
    // with them the kernel spilled)
    float* w2l_all = lds_all + kWavesPerBlock * kWaveFloats;
    for (int k = threadIdx.x; k < 2 * 3072; k += 64 * kWavesPerBlock) {
        const int t = k / 3072, rem = k % 3072, c = rem / 96, ci = (rem % 96) / 3, tap = rem % 3;
        w2l_all[t * 3072 + (tap * 32 + c) * 32 + ci] = w2[k];
    }
    __syncthreads();
    const float* w2l = w2l_all + tower * 3072 + hl * 32 + col;      // + (tap * 32 + 2s) * 32

    // constant parts of the LDS image: x[ci][-1] = 0 and the row tails conv1's wgrad reads for the non-existent h1
    // position 255 (its g1 is 0, but 0 x garbage could be NaN)
    if (lane < 3) {
        lds[kXO + lane * kXPitch] = 0.0f;
#pragma unroll
        for (int k = 256; k < kXPitch; ++k) lds[kXE + lane * kXPitch + k] = 0.0f;
#pragma unroll
        for (int k = 257; k < kXPitch; ++k) lds[kXO + lane * kXPitch + k] = 0.0f;
    }

    f32x16 acc2[3] = {zero16(), zero16(), zero16()};     // dw2[c = rowmap][ci = col][tap]
    f32x16 acc1e = zero16(), acc1o = zero16();           // rows (ci, tap) = rowmap < 15 and db1 (row 15), column c = col;
                                                         // even / odd h1 positions in chains of their own
    float db2p = 0.0f;                                   // sum of this lane's g2[c = col][l] over its l

    // lane-constant operand addresses
    // Every LDS access below is written as  <lane-constant base pointer>[<compile-time offset>]  with the lane's hl
    // folded into the base: the offset then rides in the instruction's immediate field.  (Written as lds[f(hl) + ...]
    // hipcc materialised -- and hoisted out of the item loop -- one address REGISTER per access: 64 of them for conv1's
    // h1 stores alone, and spilled.)
    const float* xrow[8];                                // conv1 B operand of step s: x[ci][2p + tap - 1], kk = 2s + hl
#pragma unroll
    for (int s = 0; s < 8; ++s) xrow[s] = lds + x_operand_base((2 * s + hl) < 15 ? (2 * s + hl) : 14) + col;
    // conv1's h1 stores of position p = pstart(h) + 32 T + col, channel rowmap(r, hl): hst[h][16 T + rowmap(r, 0) * kHPitch]
    float* hst[2];
    hst[0] = lds + ((col & 1) ? kH1O + (col + 1) / 2 : kH1E + col / 2) + 4 * hl * kHPitch;
    hst[1] = lds + ((col & 1) ? kH1E + (col - 1) / 2 : kH1O + col / 2) + 4 * hl * kHPitch;
    const float* xw = lds + x_operand_base(col < 15 ? col : 0) + 8 * hl;   // conv1 wgrad A operand: row (ci, tap) = col
    const bool ones_row = col >= 15;                     // row 15 = ones (db1); rows 16..31 are not stored
    const float* g2row = lds + kG2 + col * kGPitch + hl; // lanes over channels c, position 2s + hl
    const float* g2col = lds + kG2 + hl * kGPitch + col; // lanes over positions, rows c = 2s + hl
    const float* h1e = lds + kH1E + col * kHPitch + hl;  // lanes over channels ci, position 2s + hl
    const float* h1o = lds + kH1O + col * kHPitch + hl;
    const float* m1e = lds + kH1E + col * kHPitch + 4 * hl;   // the same rows at position rowmap(r, hl)
    const float* m1o = lds + kH1O + col * kHPitch + 4 * hl;

    const int stride = nwaves >> 1;
    int n = gwave >> 1;
    ScanLoads sx;
    GradLoads sg;
    if (n < n_items) {
        request_scan(sx, obs, rows, n, lane);
        request_grad(sg, feat_t, gfeat_t, n, 0, lane);
    }

#if defined(MRCA_PROFILING)
    unsigned long long bst[kBwdStamps] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long breal0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long bprev = __builtin_amdgcn_s_memtime();
#endif
    for (; n < n_items; n += stride) {
        stage_scan(lds, sx, lane);
        MRCA_PIN();
        MRCA_BSTAMP(0)      // scan staged
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            stage_grad(lds, sg, lane);
            MRCA_PIN();
            // --- h1 paddings of this half: h = 0: h1[-1] (H1O[c][0]); h = 1: h1[255] (H1O[c][64])
            if (lane < kCh) lds[kH1O + lane * kHPitch + (h ? kHalf : 0)] = 0.0f;
            MRCA_PIN();
            MRCA_BSTAMP(1)      // gradient rows staged, the next half's requested
            // --- conv1 recompute: 128 positions p = pstart + 32 T + col, two tiles at a time, bias through the K padding
            const int pstart = conv1_pstart(h);
            // The second tile pair's MFMAs carry the first pair's leave (accumulator -> ReLU -> H1 image, 32 rows = ~100 vector and
            // LDS instructions) between them, two rows per MFMA pair: left behind its own MFMAs the leave stood with the matrix
            // pipe idle, ~900 clocks per tile pair (tools/bwd_phases.py, profiles/r06_ak_*).
            {
                float ba[8], bb[8], bc[8], bd[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    ba[s] = xrow[s][pstart];
                    bb[s] = xrow[s][pstart + 32];
                }
                ba[7] = hl ? 1.0f : ba[7];
                bb[7] = hl ? 1.0f : bb[7];
                MRCA_PIN();
                f32x16 acca = zero16(), accb = zero16();
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    acca = MRCA_MFMA(a1[s], ba[s], acca);
                    accb = MRCA_MFMA(a1[s], bb[s], accb);
                }
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    bc[s] = xrow[s][pstart + 64];
                    bd[s] = xrow[s][pstart + 96];
                }
                bc[7] = hl ? 1.0f : bc[7];
                bd[7] = hl ? 1.0f : bd[7];
                MRCA_PIN();
                f32x16 accc = zero16(), accd = zero16();
                // h1_store_off(pstart + 32 T + col, h) + rowmap(r, hl) * kHPitch, see hst above
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    accc = MRCA_MFMA(a1[s], bc[s], accc);
                    accd = MRCA_MFMA(a1[s], bd[s], accd);
#pragma unroll
                    for (int r = 2 * s; r < 2 * s + 2; ++r) {
                        hst[h][rowmap(r, 0) * kHPitch] = relu(acca[r]);
                        hst[h][16 + rowmap(r, 0) * kHPitch] = relu(accb[r]);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
                    MRCA_PIN();
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    hst[h][32 + rowmap(r, 0) * kHPitch] = relu(accc[r]);
                    hst[h][48 + rowmap(r, 0) * kHPitch] = relu(accd[r]);
                }
            }
            MRCA_BSTAMP(2)      // conv1 recomputed
            // --- conv2 wgrad: contraction over this half's 64 positions, two per MFMA (i = 2s + hl), operands of four
            //     steps requested ahead of the MFMAs that use them
            {
                float a[2][4], b0[2][4], b1v[2][4], b2[2][4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = 2 * k;
                    a[0][k] = g2row[i];
                    b0[0][k] = h1o[i];
                    b1v[0][k] = h1e[i];
                    b2[0][k] = h1o[i + 1];
                }
                MRCA_PIN();
#pragma unroll
                for (int ch = 0; ch < 8; ++ch) {
                    const int cur = ch & 1, nxt = cur ^ 1;
                    if (ch < 7) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int i = 2 * (4 * (ch + 1) + k);
                            a[nxt][k] = g2row[i];
                            b0[nxt][k] = h1o[i];
                            b1v[nxt][k] = h1e[i];
                            b2[nxt][k] = h1o[i + 1];
                        }
                    }
                    MRCA_PIN();
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        db2p += a[cur][k];
                        acc2[0] = MRCA_MFMA(a[cur][k], b0[cur][k], acc2[0]);
                        acc2[1] = MRCA_MFMA(a[cur][k], b1v[cur][k], acc2[1]);
                        acc2[2] = MRCA_MFMA(a[cur][k], b2[cur][k], acc2[2]);
                    }
                    MRCA_PIN();
                }
            }
            MRCA_BSTAMP(3)      // conv2 wgrad
            // --- conv2 dgrad -> ReLU mask -> conv1 wgrad, 32 conv2 positions (64 h1 positions) at a time
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                // the inputs of the next half travel while the second quarter of this one computes (~7000 clocks: three HBM round
                // trips).  Requested at the half's start they held 90 registers through its conv2 wgrad and first dgrad -- the
                // phases with the most operands in flight -- and the kernel spilled into AGPRs there (profiles/r06_ak_*).
                if (u == 1) {
                    if (h == 0) {
                        request_grad(sg, feat_t, gfeat_t, n, 1, lane);
                    } else if (n + stride < n_items) {
                        request_scan(sx, obs, rows, n + stride, lane);
                        request_grad(sg, feat_t, gfeat_t, n + stride, 0, lane);
                    }
                    MRCA_PIN();
                }
                const int I0 = 32 * u;                        // position inside the half; l = 64h + I0 + row
                f32x16 accE = zero16(), accO = zero16();     // dh1 at p = 2l and 2l + 1
                float ge[2][4], gs[2][4], wa[2][4], wb[2][4], wc[2][4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    ge[0][k] = g2col[2 * k * kGPitch + I0];
                    gs[0][k] = g2col[2 * k * kGPitch + I0 + 1];
                    wa[0][k] = w2l[(0 * 32 + 2 * k) * 32];
                    wb[0][k] = w2l[(1 * 32 + 2 * k) * 32];
                    wc[0][k] = w2l[(2 * 32 + 2 * k) * 32];
                }
                MRCA_PIN();
                float me[16], mo[16];
                float xe[2][4], xo[2][4];
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) {
                    const int cur = ch & 1, nxt = cur ^ 1;
                    if (ch == 3) {
                        // the mask source and conv1 wgrad's first scan operands: requested in front of dgrad's LAST chunk, whose
                        // twelve MFMAs cover the round trip (behind them it was the first thing the idle matrix pipe waited for)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            me[r] = m1e[I0 + rowmap(r, 0)];
                            mo[r] = m1o[I0 + rowmap(r, 0) + 1];
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int p = 2 * (kHalf * h + I0 + rowmap(k, 0));
                            xe[0][k] = xw[p];
                            xo[0][k] = xw[p + 1];
                        }
                    }
                    if (ch < 3) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int s = 4 * (ch + 1) + k;
                            ge[nxt][k] = g2col[2 * s * kGPitch + I0];
                            gs[nxt][k] = g2col[2 * s * kGPitch + I0 + 1];
                            wa[nxt][k] = w2l[(0 * 32 + 2 * s) * 32];
                            wb[nxt][k] = w2l[(1 * 32 + 2 * s) * 32];
                            wc[nxt][k] = w2l[(2 * 32 + 2 * s) * 32];
                        }
                    }
                    MRCA_PIN();
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        accO = MRCA_MFMA(ge[cur][k], wc[cur][k], accO);
                        accE = MRCA_MFMA(ge[cur][k], wb[cur][k], accE);
                        accO = MRCA_MFMA(gs[cur][k], wa[cur][k], accO);
                    }
                    MRCA_PIN();
                }
                MRCA_BSTAMP(4)      // conv2 dgrad
                // g1 = dh1 where h1 > 0, formed FOUR ROWS AHEAD of the MFMAs that take it as their B operand and between those of the
                // chunk before: masked all at once in front of the first MFMA, the 32 rows (accumulator read, compare, select, write
                // back: 128 instructions) stood with the matrix pipe idle, 1730 clocks per quarter item -- 13 % of the kernel
                // (tools/bwd_phases.py, profiles/r06_ak_*)
                float gE[4], gO[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    gE[k] = me[k] > 0.0f ? accE[k] : 0.0f;
                    gO[k] = mo[k] > 0.0f ? accO[k] : 0.0f;
                    xe[0][k] = ones_row ? 1.0f : xe[0][k];
                    xo[0][k] = ones_row ? 1.0f : xo[0][k];
                }
                MRCA_BSTAMP(5)      // ReLU mask (of the first four rows)
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) {
                    const int cur = ch & 1, nxt = cur ^ 1;
                    if (ch < 3) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int p = 2 * (kHalf * h + I0 + rowmap(4 * (ch + 1) + k, 0));
                            xe[nxt][k] = xw[p];
                            xo[nxt][k] = xw[p + 1];
                        }
                    }
                    MRCA_PIN();
                    float nE[4], nO[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        // two matrix instructions, then -- while they run -- the vector instructions of a row pair of the next chunk
                        acc1e = MRCA_MFMA(xe[cur][k], gE[k], acc1e);
                        acc1o = MRCA_MFMA(xo[cur][k], gO[k], acc1o);
                        if (ch < 3) {
                            const int r = 4 * (ch + 1) + k;
                            nE[k] = me[r] > 0.0f ? accE[r] : 0.0f;
                            nO[k] = mo[r] > 0.0f ? accO[r] : 0.0f;
                            xe[nxt][k] = ones_row ? 1.0f : xe[nxt][k];
                            xo[nxt][k] = ones_row ? 1.0f : xo[nxt][k];
                            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
                        }
                        MRCA_PIN();
                    }
                    if (ch < 3) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            gE[k] = nE[k];
                            gO[k] = nO[k];
                        }
                    }
                }
                MRCA_BSTAMP(6)      // conv1 wgrad
            }
        }
#if defined(MRCA_PROFILING)
        bst[8] += 1;
#endif
    }
#if defined(MRCA_PROFILING)
    bst[9] = __builtin_amdgcn_s_memrealtime() - breal0;
    if (lane == 0 && gwave < kBwdStampWaves)
        for (int k = 0; k < kBwdStamps; ++k) g_bwd_stamps[k][gwave] = bst[k];
#endif

    // --- this wave's partial sums
    float* P = partial + (size_t)gwave * kPartFloats;
#pragma unroll
    for (int tap = 0; tap < 3; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) P[kPartDw2 + (rowmap(r, hl) * 32 + col) * 3 + tap] = acc2[tap][r];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = rowmap(r, hl);
        const float v = acc1e[r] + acc1o[r];
        if (i < 15) P[kPartDw1 + col * 15 + i] = v;
        if (i == 15) P[kPartDb1 + col] = v;
    }
    const float other = __shfl_xor(db2p, 32);
    if (hl == 0) P[kPartDb2 + col] = db2p + other;
}

// out[t][k] = sum over the waves of tower t (gwave & 1 == t) in a FIXED order: a block owns 64 consecutive outputs of one
// tower; its 16 wavefronts each add every 16th wave's partial (coalesced 256-byte rows), then the 16 sums are added in
// index order.  The partials of a weight gradient cancel heavily (|sum| << sum of |terms|), so they are added in float64:
// 3.7 M additions, and the result carries the rounding of the per-wave fp32 MFMA chains only.
constexpr int kFinGroups = 16;
__global__ __launch_bounds__(64 * kFinGroups) void lidar_features_bwd_finalize(
    const float* __restrict__ partial, int nwaves, float* __restrict__ dw1, float* __restrict__ db1,
    float* __restrict__ dw2, float* __restrict__ db2) {
    __shared__ double part[kFinGroups][64];
    const int j = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int chunks = (kPartFloats + 63) / 64;
    const int t = blockIdx.x / chunks, k = (blockIdx.x % chunks) * 64 + j;
    double s = 0.0;
    if (k < kPartFloats)
        for (int w = t + 2 * grp; w < nwaves; w += 2 * kFinGroups) s += (double)partial[(size_t)w * kPartFloats + k];
    part[grp][j] = s;
    __syncthreads();
    if (grp != 0 || k >= kPartFloats) return;
    double acc = 0.0;
#pragma unroll
    for (int gidx = 0; gidx < kFinGroups; ++gidx) acc += part[gidx][j];
    const float tot = (float)acc;
    if (k < kPartDw1) dw2[t * 3072 + k] = tot;
    else if (k < kPartDb1) dw1[t * 480 + (k - kPartDw1)] = tot;
    else if (k < kPartDb2) db1[t * 32 + (k - kPartDb1)] = tot;
    else db2[t * 32 + (k - kPartDb2)] = tot;
}

struct DeviceInfo {
    int cus = 0;
    bool attr_set = false;
};
static DeviceInfo g_dev[64];

// per DEVICE: CU count and the dynamic-LDS attribute (a second GPU in the same process needs its own)
static int prepare_device(int* cus_out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        return mrca::set_error(MRCA_ERR_HIP, "mrca_lidar_features_backward: hipGetDevice failed");
    DeviceInfo& d = g_dev[dev];
    if (d.cus == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        d.cus = cus;
    }
    if (!d.attr_set) {
        const size_t lds = (size_t)kBlockFloats * sizeof(float);
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lidar_features_bwd_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return mrca::set_error(MRCA_ERR_HIP, "mrca_lidar_features_backward: %zu B of dynamic LDS refused: %s", lds,
                                   hipGetErrorString(e));
        d.attr_set = true;
    }
    *cus_out = d.cus;
    return MRCA_OK;
}

}  // namespace mrca_pbwd

#if defined(MRCA_PROFILING)
// Profiling build only: where the waves of the LAST mrca_lidar_features_backward launch spent their time.  out[0..6] = s_memtime
// ticks per item in: scan staged | gradient rows staged | conv1 recompute | conv2 wgrad | conv2 dgrad | ReLU mask | conv1 wgrad,
// averaged over the waves that had work; out[7] = 0; out[8] = items per wave; out[9] = the shader clock during the loop [GHz].
// Synchronises the device.  tools/bwd_phases.py.
extern "C" int mrca_debug_bwd_stamps(double* out /* [10] */) {
    using namespace mrca_pbwd;
    if (!out) return mrca::set_error(MRCA_ERR_INVALID, "mrca_debug_bwd_stamps: NULL");
    if (hipDeviceSynchronize() != hipSuccess) return mrca::set_error(MRCA_ERR_HIP, "mrca_debug_bwd_stamps: sync failed");
    static unsigned long long h[kBwdStamps][kBwdStampWaves];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_bwd_stamps), sizeof(h), 0, hipMemcpyDeviceToHost) != hipSuccess)
        return mrca::set_error(MRCA_ERR_HIP, "mrca_debug_bwd_stamps: copy failed");
    double sum[kBwdStamps] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int waves = 0;
    for (int w = 0; w < kBwdStampWaves; ++w) {
        if (h[8][w] == 0) continue;
        ++waves;
        double ticks = 0.0;
        for (int k = 0; k < 8; ++k) {
            sum[k] += (double)h[k][w] / (double)h[8][w];
            ticks += (double)h[k][w];
        }
        sum[8] += (double)h[8][w];
        sum[9] += ticks / ((double)h[9][w] * 10.0);
    }
    for (int k = 0; k < kBwdStamps; ++k) out[k] = waves ? sum[k] / waves : 0.0;
    return MRCA_OK;
}
#endif

extern "C" int mrca_lidar_features_backward_scratch(size_t* bytes_out) {
    using namespace mrca_pbwd;
    if (!bytes_out) return mrca::set_error(MRCA_ERR_INVALID, "mrca_lidar_features_backward_scratch: bytes_out is NULL");
    int cus = 0;
    const int rc = prepare_device(&cus);
    if (rc != MRCA_OK) return rc;
    *bytes_out = (size_t)cus * kWavesPerBlock * kPartFloats * sizeof(float);
    return MRCA_OK;
}

static int lidar_features_backward_impl(const char* who, const float* obs_dev, const int32_t* rows_dev, int32_t n_robots, int32_t frames,
                                        int32_t beams, const float* w1_dev, const float* b1_dev, const float* w2_dev, const float* feat_dev,
                                        const float* gfeat_act_dev, const float* gfeat_crt_dev, float* dw1_dev, float* db1_dev,
                                        float* dw2_dev, float* db2_dev, void* scratch_dev, size_t scratch_bytes, void* stream) {
    using namespace mrca_pbwd;
    if (!obs_dev || !w1_dev || !b1_dev || !w2_dev || !feat_dev || !gfeat_act_dev || !gfeat_crt_dev || !dw1_dev || !db1_dev || !dw2_dev ||
        !db2_dev || !scratch_dev)
        return mrca::set_error(MRCA_ERR_INVALID, "%s: NULL pointer", who);
    if (frames != kFrames || beams != kBeams || n_robots < 1)
        return mrca::set_error(MRCA_ERR_UNSUPPORTED, "%s: frames %d beams %d samples %d (needs 3 x 512, >= 1)", who, frames, beams, n_robots);
    mrca::DeviceGuard guard(mrca::device_of(obs_dev));     // launch where the buffers live
    int cus = 0;
    const int rc = prepare_device(&cus);
    if (rc != MRCA_OK) return rc;
    const int blocks = cus;                 // persistent: one workgroup of 4 waves per CU
    const int nwaves = blocks * kWavesPerBlock;
    if (scratch_bytes < (size_t)nwaves * kPartFloats * sizeof(float))
        return mrca::set_error(MRCA_ERR_INVALID, "%s: scratch of %zu B < %zu B", who, scratch_bytes, (size_t)nwaves * kPartFloats * sizeof(float));
    const size_t lds = (size_t)kBlockFloats * sizeof(float);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(lidar_features_bwd_kernel, dim3(blocks), dim3(64 * kWavesPerBlock), lds, st, obs_dev, rows_dev, n_robots,
                       w1_dev, b1_dev, w2_dev, feat_dev, gfeat_act_dev, gfeat_crt_dev, static_cast<float*>(scratch_dev));
    hipLaunchKernelGGL(lidar_features_bwd_finalize, dim3(2 * ((kPartFloats + 63) / 64)), dim3(64 * kFinGroups), 0, st,
                       static_cast<const float*>(scratch_dev), nwaves, dw1_dev, db1_dev, dw2_dev, db2_dev);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mrca::set_error(MRCA_ERR_HIP, "%s launch: %s", who, hipGetErrorString(e));
    return MRCA_OK;
}

extern "C" int mrca_lidar_features_backward(const float* obs_dev, int32_t n_robots, int32_t frames, int32_t beams,
                                            const float* w1_dev, const float* b1_dev, const float* w2_dev,
                                            const float* feat_dev, const float* gfeat_act_dev,
                                            const float* gfeat_crt_dev, float* dw1_dev, float* db1_dev, float* dw2_dev,
                                            float* db2_dev, void* scratch_dev, size_t scratch_bytes, void* stream) {
    return lidar_features_backward_impl("mrca_lidar_features_backward", obs_dev, nullptr, n_robots, frames, beams, w1_dev, b1_dev, w2_dev,
                                        feat_dev, gfeat_act_dev, gfeat_crt_dev, dw1_dev, db1_dev, dw2_dev, db2_dev, scratch_dev,
                                        scratch_bytes, stream);
}

extern "C" int mrca_lidar_features_backward_rows(const float* frames_dev, const int32_t* rows_dev, int32_t n_samples, int32_t frames,
                                                 int32_t beams, const float* w1_dev, const float* b1_dev, const float* w2_dev,
                                                 const float* feat_dev, const float* gfeat_act_dev, const float* gfeat_crt_dev,
                                                 float* dw1_dev, float* db1_dev, float* dw2_dev, float* db2_dev, void* scratch_dev,
                                                 size_t scratch_bytes, void* stream) {
    if (!rows_dev) return mrca::set_error(MRCA_ERR_INVALID, "mrca_lidar_features_backward_rows: rows_dev is NULL");
    return lidar_features_backward_impl("mrca_lidar_features_backward_rows", frames_dev, rows_dev, n_samples, frames, beams, w1_dev, b1_dev,
                                        w2_dev, feat_dev, gfeat_act_dev, gfeat_crt_dev, dw1_dev, db1_dev, dw2_dev, db2_dev, scratch_dev,
                                        scratch_bytes, stream);
}
