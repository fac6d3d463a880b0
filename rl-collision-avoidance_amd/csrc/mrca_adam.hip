// mrca_adam.hip -- the optimiser step of the PPO update on ONE flat buffer, one gfx950 launch.
//
// The reference trains with torch.optim.Adam(policy.parameters(), lr) (ppo_stage1.py:176; betas 0.9 / 0.999, eps 1e-8, no
// weight decay, no amsgrad) and takes one step per minibatch (model/ppo.py:187-189).  The learner here already keeps the
// gradients of all 23 parameter tensors as views of one bucket (one RCCL all-reduce, mrca/ppo.py FlatGrads); with the
// parameters and the two moment estimates laid out the same way the step is a single element-wise pass over 2 172 101
// floats -- 61 MB of traffic -- instead of PyTorch's multi-tensor launch, which walks the tensor list in 64 k-element chunks
// on 34 workgroups (102 us per step, profiles/r05_z_train_kernel_stats.csv).
//
// Per element, in the operation order of torch.optim.Adam's single-tensor form (torch/optim/adam.py, _single_tensor_adam):
//     m <- m + (g - m) (1 - beta1)                       exp_avg.lerp_(grad, 1 - beta1)
//     v <- v beta2 + (1 - beta2) g g                     exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
//     p <- p - step_size * (m / (sqrt(v) / sqrt(1 - beta2^t) + eps)),     step_size = lr / (1 - beta1^t)
// fp32 throughout, the two bias corrections formed on the host in double as PyTorch does.  Compiled like the rest of the
// library with -ffp-contract=off and correctly rounded divide / sqrt: the same numbers on every run.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/mrca_env.h"
#include "mrca_hostutil.h"

namespace mrca_adam {

constexpr int kThreads = 256;

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float w1, float beta2, float w2,
                                         float step_size, float bc2_sqrt, float eps) {
    m = m + w1 * (g - m);
    v = v * beta2 + w2 * (g * g);
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - step_size * (m / denom);
}

// One float4 per thread (the buffers come from the caching allocator: 256-byte aligned), the last n % 4 elements by the
// first threads of the last workgroup.
__global__ __launch_bounds__(kThreads) void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                             float* __restrict__ m, float* __restrict__ v, long n,
                                                             float w1, float beta2, float w2, float step_size,
                                                             float bc2_sqrt, float eps) {
    const long quads = n >> 2;
    const long i = (long)blockIdx.x * kThreads + threadIdx.x;
    if (i < quads) {
        float4 pq = reinterpret_cast<float4*>(p)[i];
        const float4 gq = reinterpret_cast<const float4*>(g)[i];
        float4 mq = reinterpret_cast<float4*>(m)[i];
        float4 vq = reinterpret_cast<float4*>(v)[i];
        adam_one(pq.x, gq.x, mq.x, vq.x, w1, beta2, w2, step_size, bc2_sqrt, eps);
        adam_one(pq.y, gq.y, mq.y, vq.y, w1, beta2, w2, step_size, bc2_sqrt, eps);
        adam_one(pq.z, gq.z, mq.z, vq.z, w1, beta2, w2, step_size, bc2_sqrt, eps);
        adam_one(pq.w, gq.w, mq.w, vq.w, w1, beta2, w2, step_size, bc2_sqrt, eps);
        reinterpret_cast<float4*>(p)[i] = pq;
        reinterpret_cast<float4*>(m)[i] = mq;
        reinterpret_cast<float4*>(v)[i] = vq;
    } else {
        const long j = (quads << 2) + (i - quads);
        if (j < n) {
            float pj = p[j], mj = m[j], vj = v[j];
            adam_one(pj, g[j], mj, vj, w1, beta2, w2, step_size, bc2_sqrt, eps);
            p[j] = pj;
            m[j] = mj;
            v[j] = vj;
        }
    }
}

}  // namespace mrca_adam

extern "C" int mrca_adam_step(float* param_dev, const float* grad_dev, float* exp_avg_dev, float* exp_avg_sq_dev, int64_t n,
                              double lr, double beta1, double beta2, double eps, int32_t step, void* stream) {
    using namespace mrca_adam;
    if (!param_dev || !grad_dev || !exp_avg_dev || !exp_avg_sq_dev)
        return mrca::set_error(MRCA_ERR_INVALID, "mrca_adam_step: NULL pointer");
    if (n < 1 || step < 1) return mrca::set_error(MRCA_ERR_INVALID, "mrca_adam_step: n = %lld, step = %d", (long long)n, step);
    if (!(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0))
        return mrca::set_error(MRCA_ERR_INVALID, "mrca_adam_step: betas (%g, %g) / eps %g", beta1, beta2, eps);
    if ((reinterpret_cast<uintptr_t>(param_dev) | reinterpret_cast<uintptr_t>(grad_dev) | reinterpret_cast<uintptr_t>(exp_avg_dev) |
         reinterpret_cast<uintptr_t>(exp_avg_sq_dev)) & 15)
        return mrca::set_error(MRCA_ERR_INVALID, "mrca_adam_step: the four buffers must be 16-byte aligned");
    // the hyper-parameters arrive as doubles (Python floats) and the bias corrections are formed in double, as PyTorch forms
    // them; the kernel sees their fp32 roundings, as PyTorch's element-wise kernels see their scalars
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    const float step_size = (float)(lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    mrca::DeviceGuard guard(mrca::device_of(param_dev));
    const long quads = n >> 2, work = quads + (n & 3);
    const unsigned blocks = (unsigned)((work + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(adam_step_kernel, dim3(blocks), dim3(kThreads), 0, static_cast<hipStream_t>(stream), param_dev, grad_dev,
                       exp_avg_dev, exp_avg_sq_dev, (long)n, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), step_size, bc2_sqrt,
                       (float)eps);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mrca::set_error(MRCA_ERR_HIP, "mrca_adam_step launch: %s", hipGetErrorString(e));
    return MRCA_OK;
}
