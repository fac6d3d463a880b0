// mrca_policy_bwd.hip -- backward pass of the lidar front end of the actor-critic (model/net.py:19-25,37-49: two
// Conv1d + ReLU per tower) as ONE fused gfx950 kernel for the PPO update (model/ppo.py:158-192 differentiates through
// these layers for every minibatch; through MIOpen that is 85 % of an update: profiles/r01_f_ppo_update_profile.txt).
//
//   given  gfeat[t][n][c*128 + l] = dLoss / dfeat   (feat = the forward kernel's output, mrca_policy.hip)
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
// An item is processed in two halves of 64 conv2 positions so that the h1 image of a wave is 16.6 kB and four waves fit
// the CU's 160 kB of LDS (39.4 kB each).  The tower's weights live in registers for the wave's life; the next item's
// scan / gfeat / feat are requested from HBM before the second half of the current item.  Per-wave partial sums go to a
// scratch buffer; a second kernel adds them in a fixed order (deterministic, no float atomics).
// MFMAs per (sample, tower): 64 + 192 + 192 + 128 = 576 (2.4 MFLOP); 16 384 x 2 items: 0.49 ms at the fp32 MFMA peak.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mrca_env.h"
#include "mrca_hostutil.h"
#include "mrca_policy_layout.h"

namespace mrca_pbwd {

using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ inline f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.0f;
    return z;
}

struct ItemLoads {           // one item's HBM inputs in flight: the scan (6 float4 per lane), gfeat and feat (16 each)
    float4 x[6], g[16], f[16];
};

__device__ inline void request_item(ItemLoads& ld, const float* __restrict__ obs, const float* __restrict__ feat,
                                    const float* __restrict__ gfeat, int n_items, int tower, int n, int lane) {
    const float4* xs = reinterpret_cast<const float4*>(obs + (size_t)n * kFrames * kBeams);
    const size_t row = ((size_t)tower * n_items + n) * (kCh * kL2);
    const float4* gs = reinterpret_cast<const float4*>(gfeat + row);
    const float4* fs = reinterpret_cast<const float4*>(feat + row);
#pragma unroll
    for (int q = 0; q < 6; ++q) ld.x[q] = xs[q * 64 + lane];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        ld.g[q] = gs[q * 64 + lane];
        ld.f[q] = fs[q * 64 + lane];
    }
}

__global__ __launch_bounds__(64 * kWavesPerBlock) void lidar_features_bwd_kernel(
    const float* __restrict__ obs, int n_items, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ feat, const float* __restrict__ gfeat,
    float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float* lds = lds_all + wave * kWaveFloats;
    const int gwave = blockIdx.x * kWavesPerBlock + wave;
    const int nwaves = gridDim.x * kWavesPerBlock;
    const int tower = gwave & 1;                 // waves come in (actor, critic) pairs on the same samples
    const int col = lane & 31, hl = lane >> 5;

    // --- the tower's weights in MFMA fragment form, for the wave's whole life
    // conv1 (A operand, A[i = out channel][k]): kk = 2s + hl; kk = 15 is the bias (its B operand is 1)
    float a1[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int kk = 2 * s + hl;
        a1[s] = kk < 15 ? w1[tower * 480 + col * 15 + kk] : b1[tower * 32 + col];
    }
    // conv2 dgrad (B operand, B[k = c][j = ci]): w2f[tap][s] = w2[c = 2s + hl][ci = col][tap]
    float w2f[3][16];
#pragma unroll
    for (int tap = 0; tap < 3; ++tap)
#pragma unroll
        for (int s = 0; s < 16; ++s) w2f[tap][s] = w2[tower * 3072 + ((2 * s + hl) * 32 + col) * 3 + tap];

    // constant parts of the LDS image: x[ci][-1] = 0, the row tails conv1's wgrad reads for the non-existent h1
    // position 255 (its g1 is 0, but 0 x garbage could be NaN), g2[c][128] = 0
    if (lane < 3) {
        lds[kXO + lane * kXPitch] = 0.0f;
#pragma unroll
        for (int k = 256; k < kXPitch; ++k) lds[kXE + lane * kXPitch + k] = 0.0f;
#pragma unroll
        for (int k = 257; k < kXPitch; ++k) lds[kXO + lane * kXPitch + k] = 0.0f;
    }
    if (lane < kCh) lds[kG2 + lane * kGPitch + kL2] = 0.0f;

    f32x16 acc2[3] = {zero16(), zero16(), zero16()};     // dw2[c = rowmap][ci = col][tap]
    f32x16 acc1 = zero16();                              // rows (ci, tap) = rowmap < 15 and db1 (row 15), column c = col
    float db2p = 0.0f;                                   // sum of this lane's g2[c = col][l] over its l

    // lane-constant operand bases
    int xb1[8];                                          // conv1 B operand base of step s for this lane's hl
#pragma unroll
    for (int s = 0; s < 8; ++s) xb1[s] = x_operand_base((2 * s + hl) < 15 ? (2 * s + hl) : 14);
    const int xbw = x_operand_base(col < 15 ? col : 0);  // conv1 wgrad A operand base: row (ci, tap) = col
    const bool ones_row = col >= 15;                     // row 15 = ones (db1); rows 16..31 are not stored

    const int stride = nwaves >> 1;
    int n = gwave >> 1;
    ItemLoads ld;
    if (n < n_items) request_item(ld, obs, feat, gfeat, n_items, tower, n, lane);

    for (; n < n_items; n += stride) {
        // --- stage the scan de-interleaved and g2 = gfeat * (feat > 0)
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
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int idx = q * 64 + lane;           // float4 index: c = idx / 32, m = idx % 32 -> [c][4m .. 4m+3]
            const float4 g = ld.g[q], f = ld.f[q];
            float* dst = lds + kG2 + (idx >> 5) * kGPitch + 4 * (idx & 31);
            dst[0] = f.x > 0.0f ? g.x : 0.0f;
            dst[1] = f.y > 0.0f ? g.y : 0.0f;
            dst[2] = f.z > 0.0f ? g.z : 0.0f;
            dst[3] = f.w > 0.0f ? g.w : 0.0f;
        }

#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && n + stride < n_items)      // the next item's inputs travel while this half computes
                request_item(ld, obs, feat, gfeat, n_items, tower, n + stride, lane);
            // --- h1 paddings of this half: h = 0: h1[-1] (H1O[c][0]); h = 1: h1[255] (H1O[c][64])
            if (lane < kCh) lds[kH1O + lane * kHPitch + (h ? kHalf : 0)] = 0.0f;
            // --- conv1 recompute: 128 positions p = pstart + 32 T + col, bias through the K padding
            const int pstart = conv1_pstart(h);
#pragma unroll
            for (int T = 0; T < 4; ++T) {
                f32x16 acc = zero16();
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    float b = lds[xb1[s] + pstart + 32 * T + col];
                    if (s == 7) b = hl ? 1.0f : b;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b, acc, 0, 0, 0);
                }
                const int dst = h1_store_off(pstart + 32 * T + col, h);
#pragma unroll
                for (int r = 0; r < 16; ++r) lds[dst + rowmap(r, hl) * kHPitch] = acc[r] > 0.0f ? acc[r] : 0.0f;
            }
            // --- conv2 wgrad: contraction over this half's 64 positions, two per MFMA (l = 64h + 2s + hl)
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                const int i = 2 * s + hl;
                const float a = lds[kG2 + col * kGPitch + kHalf * h + i];
                const float b0 = lds[kH1O + col * kHPitch + i];
                const float b1v = lds[kH1E + col * kHPitch + i];
                const float b2 = lds[kH1O + col * kHPitch + i + 1];
                db2p += a;
                acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc2[0], 0, 0, 0);
                acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1v, acc2[1], 0, 0, 0);
                acc2[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b2, acc2[2], 0, 0, 0);
            }
            // --- conv2 dgrad -> ReLU mask -> conv1 wgrad, 32 conv2 positions (64 h1 positions) at a time
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int L0 = kHalf * h + 32 * u;
                f32x16 accE = zero16(), accO = zero16();     // dh1 at p = 2 (L0 + row) and 2 (L0 + row) + 1
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const float* g = lds + kG2 + (2 * s + hl) * kGPitch + L0 + col;
                    const float ae = g[0], as = g[1];
                    accE = __builtin_amdgcn_mfma_f32_32x32x2f32(ae, w2f[1][s], accE, 0, 0, 0);
                    accO = __builtin_amdgcn_mfma_f32_32x32x2f32(ae, w2f[2][s], accO, 0, 0, 0);
                    accO = __builtin_amdgcn_mfma_f32_32x32x2f32(as, w2f[0][s], accO, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = 32 * u + rowmap(r, hl);
                    accE[r] = lds[kH1E + col * kHPitch + i] > 0.0f ? accE[r] : 0.0f;
                    accO[r] = lds[kH1O + col * kHPitch + i + 1] > 0.0f ? accO[r] : 0.0f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int p = 2 * (L0 + rowmap(r, hl));
                    const float xe = lds[xbw + p], xo = lds[xbw + p + 1];
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ones_row ? 1.0f : xe, accE[r], acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ones_row ? 1.0f : xo, accO[r], acc1, 0, 0, 0);
                }
            }
        }
    }

    // --- this wave's partial sums
    float* P = partial + (size_t)gwave * kPartFloats;
#pragma unroll
    for (int tap = 0; tap < 3; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) P[kPartDw2 + (rowmap(r, hl) * 32 + col) * 3 + tap] = acc2[tap][r];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = rowmap(r, hl);
        if (i < 15) P[kPartDw1 + col * 15 + i] = acc1[r];
        if (i == 15) P[kPartDb1 + col] = acc1[r];
    }
    const float other = __shfl_xor(db2p, 32);
    if (hl == 0) P[kPartDb2 + col] = db2p + other;
}

// out[t][k] = sum over the waves of tower t (gwave & 1 == t), in wave order
__global__ void lidar_features_bwd_finalize(const float* __restrict__ partial, int nwaves, float* __restrict__ dw1,
                                            float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= 2 * kPartFloats) return;
    const int t = o / kPartFloats, k = o % kPartFloats;
    float s = 0.0f;
    for (int w = t; w < nwaves; w += 2) s += partial[(size_t)w * kPartFloats + k];
    if (k < kPartDw1) dw2[t * 3072 + k] = s;
    else if (k < kPartDb1) dw1[t * 480 + (k - kPartDw1)] = s;
    else if (k < kPartDb2) db1[t * 32 + (k - kPartDb1)] = s;
    else db2[t * 32 + (k - kPartDb2)] = s;
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
        const size_t lds = (size_t)kWavesPerBlock * kWaveFloats * sizeof(float);
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

extern "C" int mrca_lidar_features_backward_scratch(size_t* bytes_out) {
    using namespace mrca_pbwd;
    if (!bytes_out) return mrca::set_error(MRCA_ERR_INVALID, "mrca_lidar_features_backward_scratch: bytes_out is NULL");
    int cus = 0;
    const int rc = prepare_device(&cus);
    if (rc != MRCA_OK) return rc;
    *bytes_out = (size_t)cus * kWavesPerBlock * kPartFloats * sizeof(float);
    return MRCA_OK;
}

extern "C" int mrca_lidar_features_backward(const float* obs_dev, int32_t n_robots, int32_t frames, int32_t beams,
                                            const float* w1_dev, const float* b1_dev, const float* w2_dev,
                                            const float* feat_dev, const float* gfeat_dev, float* dw1_dev,
                                            float* db1_dev, float* dw2_dev, float* db2_dev, void* scratch_dev,
                                            size_t scratch_bytes, void* stream) {
    using namespace mrca_pbwd;
    if (!obs_dev || !w1_dev || !b1_dev || !w2_dev || !feat_dev || !gfeat_dev || !dw1_dev || !db1_dev || !dw2_dev ||
        !db2_dev || !scratch_dev)
        return mrca::set_error(MRCA_ERR_INVALID, "mrca_lidar_features_backward: NULL pointer");
    if (frames != kFrames || beams != kBeams || n_robots < 1)
        return mrca::set_error(MRCA_ERR_UNSUPPORTED, "mrca_lidar_features_backward: frames %d beams %d samples %d (needs 3 x 512, >= 1)",
                               frames, beams, n_robots);
    mrca::DeviceGuard guard(mrca::device_of(obs_dev));     // launch where the buffers live
    int cus = 0;
    const int rc = prepare_device(&cus);
    if (rc != MRCA_OK) return rc;
    const int blocks = cus;                 // persistent: one workgroup of 4 waves per CU
    const int nwaves = blocks * kWavesPerBlock;
    if (scratch_bytes < (size_t)nwaves * kPartFloats * sizeof(float))
        return mrca::set_error(MRCA_ERR_INVALID, "mrca_lidar_features_backward: scratch of %zu B < %zu B", scratch_bytes,
                               (size_t)nwaves * kPartFloats * sizeof(float));
    const size_t lds = (size_t)kWavesPerBlock * kWaveFloats * sizeof(float);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(lidar_features_bwd_kernel, dim3(blocks), dim3(64 * kWavesPerBlock), lds, st, obs_dev, n_robots,
                       w1_dev, b1_dev, w2_dev, feat_dev, gfeat_dev, static_cast<float*>(scratch_dev));
    hipLaunchKernelGGL(lidar_features_bwd_finalize, dim3((2 * kPartFloats + 255) / 256), dim3(256), 0, st,
                       static_cast<const float*>(scratch_dev), nwaves, dw1_dev, db1_dev, dw2_dev, db2_dev);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mrca::set_error(MRCA_ERR_HIP, "mrca_lidar_features_backward launch: %s", hipGetErrorString(e));
    return MRCA_OK;
}
