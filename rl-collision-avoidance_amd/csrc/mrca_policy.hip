// mrca_policy.hip -- the lidar front end of the actor-critic (model/net.py:19-25,37-49: two Conv1d + ReLU per tower)
// as ONE fused gfx950 kernel: the rollout's inference and the forward half of the PPO update (its backward pass is
// mrca_policy_bwd.hip).
//
//   feat[t][n][c*128 + l] = relu(b2[t][c] + sum_{ci<32,k<3} w2[t][c][ci][k] * h1[t][ci][2l+k-1])          l < 128
//   h1[t][c][l]           = relu(b1[t][c] + sum_{ci<3, k<5} w1[t][c][ci][k] * obs[n][ci][2l+k-1])         l < 255
//   (Conv1d(3,32,k5,s2,p1) -> ReLU -> Conv1d(32,32,k3,s2,p1) -> ReLU, towers t = actor, critic; zero padding.)
//
// Why a kernel: the stock path runs these two tiny-channel convolutions through MIOpen at ~10 TFLOP/s -- two thirds
// of a rollout tick at 4096 robots -- and round-trips the 32 x 255 intermediate (268 MB per tick for both towers)
// through HBM.  Here one wavefront owns one (robot, tower): the scan is staged de-interleaved in LDS (stride-2
// convolutions become unit-stride reads), both convolutions are fp32 MFMA GEMMs (v_mfma_f32_32x32x2_f32: exact f32,
// a k-ordered fmaf chain) on implicit im2col operands read straight from LDS, the intermediate never leaves the CU,
// and the weights of the tower live in registers for the wave's whole life (persistent waves walk the robots).
//   conv1:  C[32 ch][256 pos] = W1[32][16] x X1[16][256]     (K = 3*5 = 15, padded to 16)   64 MFMAs
//   conv2:  C[32 ch][128 pos] = W2[32][96] x X2[96][128]     (K = 32*3)                     192 MFMAs
// 256 MFMAs x 64 cycles per (robot, tower): 55 us at 4096 robots if every SIMD issued back to back.
//
// fp32 in, fp32 accumulate: the result differs from the PyTorch layers only by summation order (tested to 1e-5).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mrca_env.h"
#include "mrca_hostutil.h"

namespace mrca_policy {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kBeams = 512, kFrames = 3, kCh = 32;
constexpr int kL1 = 255;            // conv1 output length: (512 + 2 - 5) / 2 + 1
constexpr int kL2 = 128;            // conv2 output length: (255 + 2 - 3) / 2 + 1
constexpr int kXPitch = 260;        // floats per de-interleaved scan row (259 used)
constexpr int kHPitch = 130;        // floats per de-interleaved h1 row (129 used)
// LDS per wave (floats): XE[3][kXPitch] XO[3][kXPitch] | H1E[32][kHPitch] H1O[32][kHPitch] | 256 zeros (the K-padding
// operand of conv1, one per position of the 8 tiles)
constexpr int kXE = 0, kXO = 3 * kXPitch, kH1E = 6 * kXPitch, kH1O = kH1E + kCh * kHPitch;
constexpr int kZero = kH1O + kCh * kHPitch;
constexpr int kWaveFloats = kZero + 256;
constexpr int kWavesPerBlock = 4;

// x[ci][2l + tap - 1] for conv1 position l, from the de-interleaved rows:
//   XE[ci][j] = x[ci][2j],  XO[ci][j + 1] = x[ci][2j + 1],  XO[ci][0] = x[ci][-1] = 0 (left padding)
//   tap 0 -> XO[ci][l]   tap 1 -> XE[ci][l]   tap 2 -> XO[ci][l+1]   tap 3 -> XE[ci][l+1]   tap 4 -> XO[ci][l+2]
__device__ __host__ inline int conv1_operand_base(int kk) {
    if (kk >= 15) return kZero;   // K padding: reads zeros (the matching weight is zero as well)
    const int ci = kk / 5, tap = kk % 5;
    const int row = ((tap & 1) ? kXE : kXO) + ci * kXPitch;
    return row + (tap + 1) / 2 - ((tap & 1) ? 1 : 0);     // tap 0,1 -> +0 ; tap 2,3 -> +1 ; tap 4 -> +2
}

// h1[ci][2l + tap - 1] for conv2 position l:
//   H1E[ci][j] = h1[ci][2j],  H1O[ci][j + 1] = h1[ci][2j + 1],  H1O[ci][0] = h1[ci][-1] = 0,  h1[ci][255] = 0
//   tap 0 -> H1O[ci][l]   tap 1 -> H1E[ci][l]   tap 2 -> H1O[ci][l+1]
__device__ __host__ inline int conv2_operand_base(int kk) {
    const int ci = kk / 3, tap = kk % 3;
    return (tap == 1 ? kH1E : kH1O) + ci * kHPitch + (tap == 2 ? 1 : 0);
}

// C/D layout of v_mfma_f32_32x32x2_f32: lane holds column (lane & 31), rows (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
__device__ __host__ inline int mfma_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

__global__ __launch_bounds__(64 * kWavesPerBlock) void lidar_features_kernel(
    const float* __restrict__ obs, int n_robots, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ feat) {
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float* lds = lds_all + wave * kWaveFloats;
    const int gwave = blockIdx.x * kWavesPerBlock + wave;
    const int nwaves = gridDim.x * kWavesPerBlock;
    const int tower = gwave & 1;                 // waves come in (actor, critic) pairs on the same robots
    const int col = lane & 31, half = lane >> 5;

    // --- the tower's weights as MFMA A fragments, for the wave's whole life: A[i = lane & 31][k = lane >> 5]
    float a1[8], a2[48];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int kk = 2 * s + half;
        a1[s] = kk < 15 ? w1[tower * 480 + col * 15 + kk] : 0.0f;
    }
#pragma unroll
    for (int s = 0; s < 48; ++s) a2[s] = w2[tower * 3072 + col * 96 + 2 * s + half];
    float bias1[16], bias2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        bias1[r] = b1[tower * 32 + mfma_row(r, lane)];
        bias2[r] = b2[tower * 32 + mfma_row(r, lane)];
    }
    // constant parts of this wave's LDS image: left paddings, the right padding of h1, the zero block
    for (int ci = lane; ci < 3; ci += 64) lds[kXO + ci * kXPitch] = 0.0f;
    for (int ci = lane; ci < kCh; ci += 64) {
        lds[kH1O + ci * kHPitch] = 0.0f;              // h1[ci][-1]
        lds[kH1O + ci * kHPitch + 128] = 0.0f;        // h1[ci][255]
    }
    for (int k = lane; k < 256; k += 64) lds[kZero + k] = 0.0f;
    // also clear the tails the padded conv1 tile may read (x[ci][512 .. 515])
    for (int ci = lane; ci < 3; ci += 64) {
        lds[kXE + ci * kXPitch + 256] = 0.0f;
        lds[kXE + ci * kXPitch + 257] = 0.0f;
        lds[kXO + ci * kXPitch + 257] = 0.0f;
        lds[kXO + ci * kXPitch + 258] = 0.0f;
    }

    for (int n = gwave >> 1; n < n_robots; n += nwaves >> 1) {
        // --- stage the scan de-interleaved: 3 x 512 floats = 384 float4, 6 per lane
        const float4* src = reinterpret_cast<const float4*>(obs + (size_t)n * kFrames * kBeams);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int idx = q * 64 + lane;            // float4 index: ci = idx / 128, m = idx % 128 -> x[ci][4m .. 4m+3]
            const float4 v = src[idx];
            const int ci = idx >> 7, m = idx & 127;
            float* xe = lds + kXE + ci * kXPitch + 2 * m;
            float* xo = lds + kXO + ci * kXPitch + 2 * m + 1;
            xe[0] = v.x;
            xo[0] = v.y;
            xe[1] = v.z;
            xo[1] = v.w;
        }
        // --- conv1: 8 position tiles of 32, four at a time (independent accumulators keep the MFMA pipe full)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            f32x16 acc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = bias1[r];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int base = half ? conv1_operand_base(2 * s + 1) : conv1_operand_base(2 * s);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float b = lds[base + (g * 4 + t) * 32 + col];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b, acc[t], 0, 0, 0);
                }
            }
            // ReLU, then to LDS de-interleaved (position 255 does not exist: it is conv2's right padding)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int l = (g * 4 + t) * 32 + col;
                const int dst = (l & 1) ? (kH1O + (l >> 1) + 1) : (kH1E + (l >> 1));
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[t][r] > 0.0f ? acc[t][r] : 0.0f;
                    if (l < kL1) lds[dst + mfma_row(r, lane) * kHPitch] = v;
                }
            }
        }
        // --- conv2: 4 position tiles of 32, all at once
        {
            f32x16 acc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = bias2[r];
#pragma unroll
            for (int s = 0; s < 48; ++s) {
                const int base = half ? conv2_operand_base(2 * s + 1) : conv2_operand_base(2 * s);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float b = lds[base + t * 32 + col];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[s], b, acc[t], 0, 0, 0);
                }
            }
            float* out = feat + ((size_t)tower * n_robots + n) * (kCh * kL2);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[t][r] > 0.0f ? acc[t][r] : 0.0f;
                    out[mfma_row(r, lane) * kL2 + t * 32 + col] = v;     // flatten order of [32, 128]: c * 128 + l
                }
        }
    }
}

}  // namespace mrca_policy

namespace mrca_policy {
struct DeviceInfo {
    int cus = 0;
    bool attr_set = false;
};
static DeviceInfo g_dev[64];     // per DEVICE: CU count and the dynamic-LDS attribute (a second GPU needs its own)
}  // namespace mrca_policy

extern "C" int mrca_lidar_features(const float* obs_dev, int32_t n_robots, int32_t frames, int32_t beams,
                                   const float* w1_dev, const float* b1_dev, const float* w2_dev, const float* b2_dev,
                                   float* feat_dev, void* stream) {
    using namespace mrca_policy;
    if (!obs_dev || !w1_dev || !b1_dev || !w2_dev || !b2_dev || !feat_dev)
        return mrca::set_error(MRCA_ERR_INVALID, "mrca_lidar_features: NULL pointer");
    if (frames != kFrames || beams != kBeams || n_robots < 1)
        return mrca::set_error(MRCA_ERR_UNSUPPORTED, "mrca_lidar_features: frames %d beams %d robots %d (needs 3 x 512, >= 1)",
                               frames, beams, n_robots);
    mrca::DeviceGuard guard(mrca::device_of(obs_dev));     // launch where the buffers live
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        return mrca::set_error(MRCA_ERR_HIP, "mrca_lidar_features: hipGetDevice failed");
    DeviceInfo& d = g_dev[dev];
    const size_t lds = (size_t)kWavesPerBlock * kWaveFloats * sizeof(float);
    if (d.cus == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        d.cus = cus;
    }
    if (!d.attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lidar_features_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return mrca::set_error(MRCA_ERR_HIP, "mrca_lidar_features: %zu B of dynamic LDS refused: %s", lds,
                                   hipGetErrorString(e));
        d.attr_set = true;
    }
    // persistent waves: one workgroup of 4 waves per CU (159 kB of LDS), each wave pair walks every (#pairs)-th robot
    int blocks = d.cus;
    const int pairs_needed = (n_robots + 1) / 2;          // a block holds two (actor, critic) pairs
    if (blocks > pairs_needed) blocks = pairs_needed;
    hipLaunchKernelGGL(lidar_features_kernel, dim3(blocks), dim3(64 * kWavesPerBlock), lds,
                       static_cast<hipStream_t>(stream), obs_dev, n_robots, w1_dev, b1_dev, w2_dev, b2_dev, feat_dev);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mrca::set_error(MRCA_ERR_HIP, "mrca_lidar_features launch: %s", hipGetErrorString(e));
    return MRCA_OK;
}
