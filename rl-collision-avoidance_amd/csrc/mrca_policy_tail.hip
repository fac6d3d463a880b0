// mrca_policy_tail.hip -- everything of the actor-critic's rollout inference behind fc1, in ONE gfx950 kernel
// (model/net.py:41-55,61-70 and model/ppo.py:57-82): ReLU of fc1's output, the concatenation with the local goal and the
// speed, fc2 + ReLU of both towers, the three heads (sigmoid / tanh means, value), the Gaussian sample, its log-density and
// the clip to the action bounds.  In PyTorch that tail was ~35 launches and 140 of a rollout tick's 400 us at 4096
// robots (profiles/r02/r02_e_rollout_kernel_stats.csv); fc1 itself stays a library GEMM (137 TFLOP/s in hipBLASLt).
//
// One workgroup of 4 wavefronts owns 32 robots of ONE tower (blockIdx & 1): the robots' 260 inputs are staged in LDS
// transposed (H[k][robot]; fc1's bias -- when the GEMM left it out: a plain bmm instead of a baddbmm, whose broadcast of
// the bias into its output is a 7 us copy of its own per tick -- and the ReLU applied on the way), wave q computes units
// [32q, 32q + 32) of fc2 as
//     C[32 units][32 robots] = W2^T[32 units][260] x H[260][32 robots]        130 x v_mfma_f32_32x32x2_f32 (exact fp32)
// with the weights as A fragments in registers, then bias + ReLU, its share of the head dot products, a reduction over
// the four waves through LDS, and lane = robot finishes: sigmoid / tanh, a = mean + exp(logstd) * noise, logprob, clip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mrca_env.h"
#include "mrca_hostutil.h"

namespace mrca_ptail {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kIn = 260, kHid = 128, kFc1 = 256, kTile = 32;
constexpr int kHPitch = 33;                      // H[k][robot]: lanes run over robots, odd pitch for the transposed staging
constexpr int kHFloats = kIn * kHPitch;
constexpr int kRedFloats = 4 * kTile * 3;        // per-wave partial head sums [wave][robot][3]
constexpr float kHalfLog2Pi = 0.91893853320467274178f;

__device__ __forceinline__ int rowmap(int reg, int hl) { return (reg & 3) + 8 * (reg >> 2) + 4 * hl; }

__global__ __launch_bounds__(256) void policy_tail_kernel(
    const float* __restrict__ h1, const float* __restrict__ fc1_b, const float* __restrict__ goal,
    const float* __restrict__ speed, int n_robots, const float* __restrict__ fc2_w, const float* __restrict__ fc2_b,
    const float* __restrict__ head_w, const float* __restrict__ head_b, const float* __restrict__ critic_w, const float* __restrict__ critic_b,
    const float* __restrict__ logstd, const float* __restrict__ noise, const float* __restrict__ lo,
    const float* __restrict__ hi, float* __restrict__ value, float* __restrict__ action, float* __restrict__ logprob,
    float* __restrict__ scaled, float* __restrict__ mean_out) {
    __shared__ float H[kHFloats];
    __shared__ float red[kRedFloats];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, hl = lane >> 5;
    const int tower = blockIdx.x & 1;
    const int n0 = (blockIdx.x >> 1) * kTile;

    // --- this wave's 32 units of fc2 as A fragments A[i = unit][k = 2s + hl] (fc2_w is [tower][in = 260][out = 128])
    const int unit = 32 * wave + col;
    float a[kIn / 2];
#pragma unroll
    for (int s = 0; s < kIn / 2; ++s) a[s] = fc2_w[((size_t)tower * kIn + 2 * s + hl) * kHid + unit];

    // --- stage H[k][j] = relu(fc1 output) for k < 256, then goal x, goal y, speed v, speed w (cat order of model/net.py:45)
    const float* src = h1 + ((size_t)tower * n_robots + n0) * kFc1;
    // (rows past the batch are clamped to its last robot: every load is unconditional, so all eight are in flight together;
    // their results are never stored)
    float4 hv[kTile * kFc1 / 4 / 256];                            // 8 float4 per thread
    const int last = n_robots - 1 - n0;
    // (a thread stages the same four inputs of every robot row it touches: one bias quad)
    const float4 bq = fc1_b ? *reinterpret_cast<const float4*>(fc1_b + tower * kFc1 + (tid & 63) * 4)
                            : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int q = 0; q < kTile * kFc1 / 4 / 256; ++q) {
        const int idx = q * 256 + tid;
        const int j = idx >> 6, k4 = (idx & 63) * 4;             // robot row, first of four inputs
        hv[q] = *reinterpret_cast<const float4*>(src + (size_t)(j < last ? j : last) * kFc1 + k4);
    }
#pragma unroll
    for (int q = 0; q < kTile * kFc1 / 4 / 256; ++q) {
        const int idx = q * 256 + tid;
        const int j = idx >> 6, k4 = (idx & 63) * 4;
        const float4 v = make_float4(hv[q].x + bq.x, hv[q].y + bq.y, hv[q].z + bq.z, hv[q].w + bq.w);
        H[(k4 + 0) * kHPitch + j] = v.x > 0.0f ? v.x : 0.0f;
        H[(k4 + 1) * kHPitch + j] = v.y > 0.0f ? v.y : 0.0f;
        H[(k4 + 2) * kHPitch + j] = v.z > 0.0f ? v.z : 0.0f;
        H[(k4 + 3) * kHPitch + j] = v.w > 0.0f ? v.w : 0.0f;
    }
    if (tid < kTile * 4) {
        const int j = tid >> 2, c = tid & 3;
        const int jj = n0 + (j < last ? j : last);
        H[(kFc1 + c) * kHPitch + j] = c < 2 ? goal[(size_t)jj * 2 + c] : speed[(size_t)jj * 2 + c - 2];
    }
    __syncthreads();

    // --- fc2: 130 MFMA steps, B[k = 2s + hl][j = robot] read from LDS a chunk (10 steps) ahead
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const float* hb = H + hl * kHPitch + col;
    constexpr int kChunk = 10;
    float b[2][kChunk];
#pragma unroll
    for (int k = 0; k < kChunk; ++k) b[0][k] = hb[2 * k * kHPitch];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ch = 0; ch < kIn / 2 / kChunk; ++ch) {
        const int cur = ch & 1, nxt = cur ^ 1;
        if (ch + 1 < kIn / 2 / kChunk) {
#pragma unroll
            for (int k = 0; k < kChunk; ++k) b[nxt][k] = hb[2 * ((ch + 1) * kChunk + k) * kHPitch];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < kChunk; ++k)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ch * kChunk + k], b[cur][k], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }

    // --- bias + ReLU, then this wave's share of the head dot products: register r of lane (robot col, hl) is unit
    //     32 wave + rowmap(r, hl)
    float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int u = 32 * wave + rowmap(r, hl);
        float h = acc[r] + fc2_b[tower * kHid + u];
        h = h > 0.0f ? h : 0.0f;
        if (tower == 0) {
            p0 += h * head_w[u * 2 + 0];
            p1 += h * head_w[u * 2 + 1];
        } else {
            p0 += h * critic_w[u];
        }
    }
    p0 += __shfl_xor(p0, 32);
    p1 += __shfl_xor(p1, 32);
    if (hl == 0) {
        red[(wave * kTile + col) * 3 + 0] = p0;
        red[(wave * kTile + col) * 3 + 1] = p1;
    }
    __syncthreads();
    if (tid >= kTile) return;
    const int n = n0 + tid;
    if (n >= n_robots) return;
    float d0 = 0.0f, d1 = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {                 // fixed order: deterministic
        d0 += red[(q * kTile + tid) * 3 + 0];
        d1 += red[(q * kTile + tid) * 3 + 1];
    }
    if (tower == 1) {
        value[n] = d0 + critic_b[0];
        return;
    }
    // mean = [sigmoid(actor1), tanh(actor2)] (model/net.py:49-51); a ~ N(mean, exp(logstd)) (model/ppo.py:73-75)
    const float m0 = 1.0f / (1.0f + expf(-(d0 + head_b[0])));
    const float m1 = tanhf(d1 + head_b[1]);
    const float ls0 = logstd[0], ls1 = logstd[1];
    const float sd0 = expf(ls0), sd1 = expf(ls1);
    const float z0 = noise ? noise[(size_t)n * 2 + 0] : 0.0f, z1 = noise ? noise[(size_t)n * 2 + 1] : 0.0f;
    const float a0 = m0 + sd0 * z0, a1 = m1 + sd1 * z1;
    // log N(a; mean, std) summed over the action dimension (model/utils.py:90-97)
    const float e0 = a0 - m0, e1 = a1 - m1;
    const float lp = (-(e0 * e0) / (2.0f * (sd0 * sd0)) - kHalfLog2Pi - ls0) + (-(e1 * e1) / (2.0f * (sd1 * sd1)) - kHalfLog2Pi - ls1);
    mean_out[(size_t)n * 2 + 0] = m0;
    mean_out[(size_t)n * 2 + 1] = m1;
    action[(size_t)n * 2 + 0] = a0;
    action[(size_t)n * 2 + 1] = a1;
    logprob[n] = lp;
    scaled[(size_t)n * 2 + 0] = fminf(fmaxf(a0, lo[0]), hi[0]);
    scaled[(size_t)n * 2 + 1] = fminf(fmaxf(a1, lo[1]), hi[1]);
}

}  // namespace mrca_ptail

extern "C" int mrca_policy_tail(const float* h1_dev, const float* fc1_b_dev, const float* goal_dev, const float* speed_dev,
                                int32_t n_robots,
                                const float* fc2_w_dev, const float* fc2_b_dev, const float* head_w_dev,
                                const float* head_b_dev, const float* critic_w_dev, const float* critic_b_dev,
                                const float* logstd_dev, const float* noise_dev, const float* lo_dev, const float* hi_dev,
                                float* value_dev, float* action_dev, float* logprob_dev, float* scaled_dev,
                                float* mean_dev, void* stream) {
    using namespace mrca_ptail;
    if (!h1_dev || !goal_dev || !speed_dev || !fc2_w_dev || !fc2_b_dev || !head_w_dev || !head_b_dev || !critic_w_dev ||
        !critic_b_dev || !logstd_dev || !lo_dev || !hi_dev || !value_dev || !action_dev || !logprob_dev || !scaled_dev ||
        !mean_dev)
        return mrca::set_error(MRCA_ERR_INVALID, "mrca_policy_tail: NULL pointer");
    if (n_robots < 1) return mrca::set_error(MRCA_ERR_INVALID, "mrca_policy_tail: n_robots %d", n_robots);
    mrca::DeviceGuard guard(mrca::device_of(h1_dev));
    const int tiles = (n_robots + kTile - 1) / kTile;
    hipLaunchKernelGGL(policy_tail_kernel, dim3(2 * tiles), dim3(256), 0, static_cast<hipStream_t>(stream), h1_dev, fc1_b_dev,
                       goal_dev, speed_dev, n_robots, fc2_w_dev, fc2_b_dev, head_w_dev, head_b_dev, critic_w_dev, critic_b_dev,
                       logstd_dev, noise_dev, lo_dev, hi_dev, value_dev, action_dev, logprob_dev, scaled_dev, mean_dev);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mrca::set_error(MRCA_ERR_HIP, "mrca_policy_tail launch: %s", hipGetErrorString(e));
    return MRCA_OK;
}
