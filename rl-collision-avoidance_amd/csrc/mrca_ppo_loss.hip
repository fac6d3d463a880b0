// mrca_ppo_loss.hip -- the loss tail of the PPO update, forward AND backward, in ONE gfx950 launch.
//
// model/ppo.py:172-185 (ppo_update_stage1; :238-251 for stage 2) evaluates, per minibatch,
//     logp_i   = sum_j [ -(a_ij - mu_ij)^2 / (2 exp(ls_j)^2) - 0.5 log(2 pi) - ls_j ]         model/utils.py:90-97
//     r_i      = exp(logp_i - logp_old_i)
//     L_pi     = -mean_i min(r_i A_i, clamp(r_i, 1 - c, 1 + c) A_i)
//     L_v      = mean_i (v_i - target_i)^2
//     H        = mean_i sum_j (0.5 + 0.5 log(2 pi) + ls_j)                                      model/net.py:72-80
//     loss     = L_pi + 20 L_v - coeff_entropy H
// as ~30 element-wise / reduction launches over [B, 1] and [B, 2] tensors, and autograd replays as many on the way back:
// 2 300 tiny launches per 64-minibatch update, 0.25 ms of a 3.5 ms minibatch (profiles/r03/r03_s_train_kernel_stats.csv).
// Here one launch produces the five scalars AND the gradients of `loss` with respect to the network's outputs:
//     dloss/dv_i    = 20 * 2 (v_i - target_i) / B
//     dloss/dlogp_i = -(1 / B) A_i r_i g_i,    g_i = 1 where torch.min / torch.clamp route the gradient to r_i:
//                     lo <= r_i <= hi (both surrogates equal: half the gradient through each, the clamp passes its half),
//                     or r_i A_i < clamp(r_i) A_i outside the range; 0 otherwise            (autograd's tie rules)
//     dloss/dmu_ij  = dloss/dlogp_i * (a_ij - mu_ij) / var_j
//     dloss/dls_j   = sum_i dloss/dlogp_i * ((a_ij - mu_ij)^2 / var_j - 1)  -  coeff_entropy
// Sums are formed per workgroup in a fixed order and combined by the last workgroup to finish, again in a fixed order:
// the result is bit-identical from run to run.  tests/test_gpu_ppo_loss.py holds it against the PyTorch expression
// (values and autograd's gradients) and the learner goldens replay the reference's own updates through it.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mrca_env.h"
#include "mrca_hostutil.h"

namespace mrca_ppoloss {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 256;
constexpr int kSums = 6;     // sum min-term | sum (v - t)^2 | sum k3-KL | dls_0 | dls_1 | (unused)
constexpr float kHalfLog2Pi = 0.91893853320467274178f;

// scratch: [kMaxBlocks][kSums] doubles of partial sums + one ticket counter (uint32) behind them
constexpr size_t kScratchBytes = sizeof(double) * kMaxBlocks * kSums + 256;

__global__ __launch_bounds__(kThreads) void ppo_loss_kernel(
    const float* __restrict__ mean, const float* __restrict__ value, const float* __restrict__ logstd,
    const float* __restrict__ action, const float* __restrict__ old_logprob, const float* __restrict__ adv,
    const float* __restrict__ target, int n, float clip, float value_coef, float coeff_entropy,
    float* __restrict__ out /* [8]: loss, L_pi, L_v, H, kl, dls_0, dls_1, - */, float* __restrict__ gmean /* [n,2] */,
    float* __restrict__ gvalue /* [n] */, double* partial, unsigned int* ticket) {
    __shared__ double red[kSums][kThreads / 64];
    __shared__ bool is_last;
    const float ls0 = logstd[0], ls1 = logstd[1];
    const float sd0 = expf(ls0), sd1 = expf(ls1);
    const float var0 = sd0 * sd0, var1 = sd1 * sd1;
    const float inv_n = 1.0f / (float)n;
    const float lo = 1.0f - clip, hi = 1.0f + clip;
    double acc[kSums] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        const float2 mu = reinterpret_cast<const float2*>(mean)[i];
        const float2 a = reinterpret_cast<const float2*>(action)[i];
        const float d0 = a.x - mu.x, d1 = a.y - mu.y;
        // the same operation order as gaussian_logprob (mrca/net.py; model/utils.py:90-97)
        const float lp = (-(d0 * d0) / (2.0f * var0) - kHalfLog2Pi - ls0) + (-(d1 * d1) / (2.0f * var1) - kHalfLog2Pi - ls1);
        const float log_ratio = lp - old_logprob[i];
        const float r = expf(log_ratio);
        const float A = adv[i];
        const float s1 = r * A;
        const float rc = fminf(fmaxf(r, lo), hi);
        const float s2 = rc * A;
        const bool inside = r >= lo && r <= hi;
        const float g = (inside || s1 < s2) ? 1.0f : 0.0f;
        const float dlp = -inv_n * A * r * g;                    // dloss / dlogp_i
        const float dv = value[i] - target[i];
        reinterpret_cast<float2*>(gmean)[i] = make_float2(dlp * (d0 / var0), dlp * (d1 / var1));
        gvalue[i] = value_coef * 2.0f * dv * inv_n;
        // torch.min propagates a NaN surrogate (a diverged update must show up in the loss, not train on silently); fminf drops it
        acc[0] += (double)((s1 < s2 || s1 != s1) ? s1 : s2);
        acc[1] += (double)(dv * dv);
        acc[2] += (double)((r - 1.0f) - log_ratio);              // k3 estimator of KL(old || new)
        acc[3] += (double)(dlp * (d0 * d0 / var0 - 1.0f));
        acc[4] += (double)(dlp * (d1 * d1 / var1 - 1.0f));
    }
    // wave reduction (fixed butterfly order), then the block's four waves in order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kSums; ++k) {
        double v = acc[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) red[k][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < kSums) {
        double v = 0.0;
        for (int w = 0; w < kThreads / 64; ++w) v += red[threadIdx.x][w];
        partial[(size_t)blockIdx.x * kSums + threadIdx.x] = v;
    }
    // release (agent scope: the workgroups of a launch sit on eight XCDs with an L2 each), the write-back waited for in so
    // many words (MI355X_MICROARCH.md, "compiler hazard": the s_waitcnt behind buffer_wbl2 can be dropped), THEN the ticket
    __threadfence();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) is_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    __syncthreads();
    if (!is_last) return;
    __threadfence();      // acquire: the other workgroups' partial sums, whichever XCD wrote them
    if (threadIdx.x < kSums) {
        double v = 0.0;
        for (unsigned b = 0; b < gridDim.x; ++b)
            v += partial[(size_t)b * kSums + threadIdx.x];     // fixed order over workgroups
        red[threadIdx.x][0] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double L_pi = -red[0][0] / n, L_v = red[1][0] / n;
        const double H = (0.5 + (double)kHalfLog2Pi + (double)ls0) + (0.5 + (double)kHalfLog2Pi + (double)ls1);
        out[0] = (float)(L_pi + (double)value_coef * L_v - (double)coeff_entropy * H);
        out[1] = (float)L_pi;
        out[2] = (float)L_v;
        out[3] = (float)H;
        out[4] = (float)(red[2][0] / n);
        out[5] = (float)(red[3][0] - (double)coeff_entropy);
        out[6] = (float)(red[4][0] - (double)coeff_entropy);
        out[7] = 0.0f;
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
    }
}

}  // namespace mrca_ppoloss

extern "C" int mrca_ppo_loss_scratch(size_t* bytes_out) {
    if (!bytes_out) return mrca::set_error(MRCA_ERR_INVALID, "mrca_ppo_loss_scratch: bytes_out is NULL");
    *bytes_out = mrca_ppoloss::kScratchBytes;
    return MRCA_OK;
}

extern "C" int mrca_ppo_loss(const float* mean_dev, const float* value_dev, const float* logstd_dev, const float* action_dev,
                             const float* old_logprob_dev, const float* adv_dev, const float* target_dev, int32_t n,
                             float clip_value, float value_coef, float coeff_entropy, float* out_dev, float* gmean_dev,
                             float* gvalue_dev, void* scratch_dev, size_t scratch_bytes, void* stream) {
    using namespace mrca_ppoloss;
    if (!mean_dev || !value_dev || !logstd_dev || !action_dev || !old_logprob_dev || !adv_dev || !target_dev || !out_dev ||
        !gmean_dev || !gvalue_dev || !scratch_dev)
        return mrca::set_error(MRCA_ERR_INVALID, "mrca_ppo_loss: NULL pointer");
    if (n < 1) return mrca::set_error(MRCA_ERR_INVALID, "mrca_ppo_loss: n = %d", n);
    if (scratch_bytes < kScratchBytes)
        return mrca::set_error(MRCA_ERR_NOMEM, "mrca_ppo_loss: scratch of %zu bytes < %zu", scratch_bytes, kScratchBytes);
    mrca::DeviceGuard guard(mrca::device_of(mean_dev));       // launch where the buffers live
    int blocks = (n + kThreads - 1) / kThreads;
    if (blocks > kMaxBlocks) blocks = kMaxBlocks;
    double* partial = static_cast<double*>(scratch_dev);
    unsigned int* ticket = reinterpret_cast<unsigned int*>(static_cast<char*>(scratch_dev) + sizeof(double) * kMaxBlocks * kSums);
    // the ticket starts at 0 for THIS launch whatever became of the launch before (an aborted launch never resets it, and every
    // launch after it would have no "last" workgroup: `out` uninitialised): a 4-byte memset node on the same stream
    if (hipMemsetAsync(ticket, 0, sizeof(unsigned int), static_cast<hipStream_t>(stream)) != hipSuccess)
        return mrca::set_error(MRCA_ERR_HIP, "mrca_ppo_loss: hipMemsetAsync of the ticket failed");
    hipLaunchKernelGGL(ppo_loss_kernel, dim3(blocks), dim3(kThreads), 0, static_cast<hipStream_t>(stream), mean_dev, value_dev,
                       logstd_dev, action_dev, old_logprob_dev, adv_dev, target_dev, n, clip_value, value_coef, coeff_entropy,
                       out_dev, gmean_dev, gvalue_dev, partial, ticket);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mrca::set_error(MRCA_ERR_HIP, "mrca_ppo_loss launch: %s", hipGetErrorString(e));
    return MRCA_OK;
}
