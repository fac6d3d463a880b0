// mrca_policy_heads.hip -- the three output heads of the actor-critic in the PPO update, forward and backward, for gfx950.
//
// model/net.py:47-55,61-63: mean = [sigmoid(actor1(a)), tanh(actor2(a))], value = critic(c) with a, c the 128 features of
// the two towers -- three Linear(128, 1) layers.  As library GEMMs these are the worst shapes of the update: the forward is
// three [B x 128] x [128 x 1] products behind a bias copy each, the backward three rank-1 products for the feature
// gradients and three [1 x B] x [B x 128] products for the weight gradients -- a reduction over the whole minibatch into 128
// numbers, 32 - 35 us each on ONE workgroup whichever kernel TunableOp picks -- plus sigmoid / tanh, their backward, a cat and
// its backward, three bias sums and an add: ~25 launches and ~200 us of a 3.1 ms minibatch of 16 384 rows for 12 MFLOP
// (profiles/r05_z_update_profile.txt).  It is a memory-bound row operation: each of a and c is read once forward (16 MB) and
// once backward, each feature gradient written once.
//
//   heads_forward_kernel    half a wavefront per row: lane q holds columns 4q .. 4q+3 of the three weight rows, three dot
//                           products meet by five xor-shuffles inside the half, lane 0 applies bias, sigmoid / tanh
//   heads_backward_kernel   the same mapping: the row's three output gradients through sigmoid' / tanh' (as
//                           sigmoid_backward / tanh_backward form them: g (1 - y) y, g (1 - y y)), the feature gradients
//                           da = g0 w1 + g1 w2, dc = gv wc, and per-wave partial sums of the weight / bias gradients
//   heads_finalize_kernel   the partial sums added in a fixed order in float64: bit-identical from run to run
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mrca_env.h"
#include "mrca_hostutil.h"

namespace mrca_heads {

constexpr int kFeat = 128;                 // features per tower (act_fc2 / crt_fc2 outputs)
constexpr int kThreads = 256;
constexpr int kWavesPerBlock = kThreads / 64;
constexpr int kMaxBlocks = 256;            // 1024 waves: 16 rows each at 16 384 rows
constexpr int kRowsInFlight = 4;           // row pairs a wave loads before it computes on any of them
constexpr int kHeadOut = 3 * kFeat + 3;    // dW actor1, dW actor2, dW critic, db actor1, db actor2, db critic
// ... and, behind them, the column sums of da and dc: the gradients of act_fc2.bias / crt_fc2.bias (the layers that produced a
// and c) -- formed where da / dc are formed instead of by two `sum` launches over the 8 MB just written (model/ppo.py:186-188
// back-propagates through fc2; as autograd nodes those sums were 2 of the 4 reduce kernels of a minibatch)
constexpr int kOut = kHeadOut + 2 * kFeat;
constexpr int kPartialPitch = kOut + 1;
constexpr size_t kScratchBytes = sizeof(float) * kPartialPitch * kMaxBlocks;       // one record per workgroup

__device__ __forceinline__ float4 load4(const float* p) { return make_float4(p[0], p[1], p[2], p[3]); }
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float half_sum(float v) {          // over the 32 lanes of this lane's half of the wavefront
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

__device__ __forceinline__ float4 relu4(float4 v) {
    return make_float4(v.x > 0.0f ? v.x : 0.0f, v.y > 0.0f ? v.y : 0.0f, v.z > 0.0f ? v.z : 0.0f, v.w > 0.0f ? v.w : 0.0f);
}
__device__ __forceinline__ float4 mask4(float4 g, float4 z) {     // threshold_backward: the gradient where the input was > 0
    return make_float4(z.x > 0.0f ? g.x : 0.0f, z.y > 0.0f ? g.y : 0.0f, z.z > 0.0f ? g.z : 0.0f, z.w > 0.0f ? g.w : 0.0f);
}

// RELU: a and c are fc2's outputs BEFORE their ReLU; it is applied as they are loaded (and its mask to da, dc on the way back)
template <bool RELU>
__global__ __launch_bounds__(kThreads) void heads_forward_kernel(
    const float* __restrict__ a, const float* __restrict__ c, int n, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ wc, const float* __restrict__ bc,
    float* __restrict__ mean, float* __restrict__ value) {
    const int lane = threadIdx.x & 63, half = lane >> 5, q = lane & 31;
    const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6), nwaves = gridDim.x * kWavesPerBlock;
    // (the weight rows are slices of the optimiser's flat buffer: 4-byte aligned only)
    const float4 w1q = load4(w1 + 4 * q), w2q = load4(w2 + 4 * q), wcq = load4(wc + 4 * q);
    const float bias1 = b1[0], bias2 = b2[0], biasc = bc[0];
    // a wave owns a contiguous run of row PAIRS (one row per half); four pairs per pass so that eight 16-byte loads per lane
    // are in flight together -- one row pair per pass is one memory round trip per pass, 20 us for 16 384 rows
    const int pairs = (n + 1) >> 1;
    const int per_wave = (pairs + nwaves - 1) / nwaves;
    const int first = wave * per_wave, last = min(pairs, first + per_wave);
    for (int p0 = first; p0 < last; p0 += kRowsInFlight) {          // (wave-uniform bounds: the shuffles need every lane)
        float4 aq[kRowsInFlight], cq[kRowsInFlight];
#pragma unroll
        for (int j = 0; j < kRowsInFlight; ++j) {
            const int row = min(2 * min(p0 + j, last - 1) + half, n - 1);
            const size_t r = (size_t)row * (kFeat / 4) + q;
            aq[j] = reinterpret_cast<const float4*>(a)[r];
            cq[j] = reinterpret_cast<const float4*>(c)[r];
        }
#pragma unroll
        for (int j = 0; j < kRowsInFlight; ++j) {
            const int row = 2 * (p0 + j) + half;
            if (RELU) {
                aq[j] = relu4(aq[j]);
                cq[j] = relu4(cq[j]);
            }
            const float s0 = half_sum(dot4(aq[j], w1q)), s1 = half_sum(dot4(aq[j], w2q)), s2 = half_sum(dot4(cq[j], wcq));
            if (q == 0 && p0 + j < last && row < n) {
                mean[(size_t)row * 2 + 0] = 1.0f / (1.0f + expf(-(s0 + bias1)));
                mean[(size_t)row * 2 + 1] = tanhf(s1 + bias2);
                value[row] = s2 + biasc;
            }
        }
    }
}

template <bool RELU>
__global__ __launch_bounds__(kThreads) void heads_backward_kernel(
    const float* __restrict__ a, const float* __restrict__ c, const float* __restrict__ mean, const float* __restrict__ gmean,
    const float* __restrict__ gvalue, int n, const float* __restrict__ w1, const float* __restrict__ w2,
    const float* __restrict__ wc, float* __restrict__ da, float* __restrict__ dc, float* __restrict__ partial) {
    const int lane = threadIdx.x & 63, half = lane >> 5, q = lane & 31;
    const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6), nwaves = gridDim.x * kWavesPerBlock;
    // (the weight rows are slices of the optimiser's flat buffer: 4-byte aligned only)
    const float4 w1q = load4(w1 + 4 * q), w2q = load4(w2 + 4 * q), wcq = load4(wc + 4 * q);
    float4 acc1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), acc2 = acc1, accc = acc1, sda = acc1, sdc = acc1;
    float sb0 = 0.0f, sb1 = 0.0f, sbv = 0.0f;
    const int pairs = (n + 1) >> 1;
    const int per_wave = (pairs + nwaves - 1) / nwaves;
    const int first = wave * per_wave, last = min(pairs, first + per_wave);
    for (int p0 = first; p0 < last; p0 += kRowsInFlight) {
        float4 aq[kRowsInFlight], cq[kRowsInFlight];
        float g0[kRowsInFlight], g1[kRowsInFlight], gv[kRowsInFlight];
#pragma unroll
        for (int j = 0; j < kRowsInFlight; ++j) {                   // every load of the pass first ...
            const int row = 2 * (p0 + j) + half;
            const bool valid = p0 + j < last && row < n;
            const int rr = valid ? row : 0;
            const size_t r = (size_t)rr * (kFeat / 4) + q;
            aq[j] = reinterpret_cast<const float4*>(a)[r];
            cq[j] = reinterpret_cast<const float4*>(c)[r];
            const float m0 = mean[(size_t)rr * 2 + 0], m1 = mean[(size_t)rr * 2 + 1];
            const float u0 = gmean ? gmean[(size_t)rr * 2 + 0] : 0.0f, u1 = gmean ? gmean[(size_t)rr * 2 + 1] : 0.0f;
            const float uv = gvalue ? gvalue[rr] : 0.0f;
            g0[j] = valid ? u0 * (1.0f - m0) * m0 : 0.0f;           // sigmoid_backward: g (1 - y) y
            g1[j] = valid ? u1 * (1.0f - m1 * m1) : 0.0f;           // tanh_backward:    g (1 - y y)
            gv[j] = valid ? uv : 0.0f;
            if (!valid) aq[j] = cq[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);     // (0 x a non-finite stand-in row would be NaN)
        }
#pragma unroll
        for (int j = 0; j < kRowsInFlight; ++j) {                   // ... then the arithmetic and the two stores per row
            const int row = 2 * (p0 + j) + half;
            if (p0 + j < last && row < n) {
                const size_t r = (size_t)row * (kFeat / 4) + q;
                float4 dav = make_float4(g0[j] * w1q.x + g1[j] * w2q.x, g0[j] * w1q.y + g1[j] * w2q.y,
                                         g0[j] * w1q.z + g1[j] * w2q.z, g0[j] * w1q.w + g1[j] * w2q.w);
                float4 dcv = make_float4(gv[j] * wcq.x, gv[j] * wcq.y, gv[j] * wcq.z, gv[j] * wcq.w);
                if (RELU) {
                    dav = mask4(dav, aq[j]);
                    dcv = mask4(dcv, cq[j]);
                }
                reinterpret_cast<float4*>(da)[r] = dav;
                reinterpret_cast<float4*>(dc)[r] = dcv;
                sda.x += dav.x; sda.y += dav.y; sda.z += dav.z; sda.w += dav.w;
                sdc.x += dcv.x; sdc.y += dcv.y; sdc.z += dcv.z; sdc.w += dcv.w;
            }
            if (RELU) {
                aq[j] = relu4(aq[j]);
                cq[j] = relu4(cq[j]);
            }
            acc1.x += g0[j] * aq[j].x; acc1.y += g0[j] * aq[j].y; acc1.z += g0[j] * aq[j].z; acc1.w += g0[j] * aq[j].w;
            acc2.x += g1[j] * aq[j].x; acc2.y += g1[j] * aq[j].y; acc2.z += g1[j] * aq[j].z; acc2.w += g1[j] * aq[j].w;
            accc.x += gv[j] * cq[j].x; accc.y += gv[j] * cq[j].y; accc.z += gv[j] * cq[j].z; accc.w += gv[j] * cq[j].w;
            sb0 += g0[j];
            sb1 += g1[j];
            sbv += gv[j];
        }
    }
    // the two halves of the wave hold the same columns for different rows: lower half + upper half ...
    acc1.x += __shfl_xor(acc1.x, 32); acc1.y += __shfl_xor(acc1.y, 32); acc1.z += __shfl_xor(acc1.z, 32); acc1.w += __shfl_xor(acc1.w, 32);
    acc2.x += __shfl_xor(acc2.x, 32); acc2.y += __shfl_xor(acc2.y, 32); acc2.z += __shfl_xor(acc2.z, 32); acc2.w += __shfl_xor(acc2.w, 32);
    accc.x += __shfl_xor(accc.x, 32); accc.y += __shfl_xor(accc.y, 32); accc.z += __shfl_xor(accc.z, 32); accc.w += __shfl_xor(accc.w, 32);
    sda.x += __shfl_xor(sda.x, 32); sda.y += __shfl_xor(sda.y, 32); sda.z += __shfl_xor(sda.z, 32); sda.w += __shfl_xor(sda.w, 32);
    sdc.x += __shfl_xor(sdc.x, 32); sdc.y += __shfl_xor(sdc.y, 32); sdc.z += __shfl_xor(sdc.z, 32); sdc.w += __shfl_xor(sdc.w, 32);
    sb0 += __shfl_xor(sb0, 32);
    sb1 += __shfl_xor(sb1, 32);
    sbv += __shfl_xor(sbv, 32);
    // ... the four waves of the workgroup meet in LDS and are added in a fixed order: one record per workgroup
    __shared__ float red[kWavesPerBlock][kPartialPitch];
    float* mine = red[threadIdx.x >> 6];
    if (half == 0) {
        reinterpret_cast<float4*>(mine)[q] = acc1;
        reinterpret_cast<float4*>(mine + kFeat)[q] = acc2;
        reinterpret_cast<float4*>(mine + 2 * kFeat)[q] = accc;
        if (q == 0) {
            mine[3 * kFeat + 0] = sb0;
            mine[3 * kFeat + 1] = sb1;
            mine[3 * kFeat + 2] = sbv;
        }
        // (kHeadOut is not a multiple of 4: scalar stores)
        float* za = mine + kHeadOut + 4 * q;
        float* zc = mine + kHeadOut + kFeat + 4 * q;
        za[0] = sda.x; za[1] = sda.y; za[2] = sda.z; za[3] = sda.w;
        zc[0] = sdc.x; zc[1] = sdc.y; zc[2] = sdc.z; zc[3] = sdc.w;
    }
    __syncthreads();
    float* rec = partial + (size_t)blockIdx.x * kPartialPitch;
    for (int k = threadIdx.x; k < kOut; k += kThreads) rec[k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
}

// dw[k] = the workgroups' records added in a fixed order in float64: thread (k, group g) adds records g, g + 8, ...; the
// eight groups meet in LDS
constexpr int kFinK = 32, kFinGroups = 8;
__global__ __launch_bounds__(kFinK * kFinGroups) void heads_finalize_kernel(const float* __restrict__ partial, int nrec,
                                                                            float* __restrict__ dw, float* __restrict__ dzb) {
    __shared__ double sh[kFinGroups][kFinK];
    const int kk = threadIdx.x & (kFinK - 1), grp = threadIdx.x / kFinK;
    const int k = blockIdx.x * kFinK + kk;
    double s = 0.0;
    if (k < kOut) {
#pragma unroll 8
        for (int r = grp; r < nrec; r += kFinGroups) s += (double)partial[(size_t)r * kPartialPitch + k];
    }
    sh[grp][kk] = s;
    __syncthreads();
    if (grp == 0 && k < kOut) {
        double t = sh[0][kk];
#pragma unroll
        for (int g = 1; g < kFinGroups; ++g) t += sh[g][kk];
        if (k < kHeadOut) dw[k] = (float)t;
        else if (dzb) dzb[k - kHeadOut] = (float)t;      // [0, 128): act_fc2.bias, [128, 256): crt_fc2.bias
    }
}

// x2[n][260] = [relu(h1[n][256]), goal[n][2], speed[n][2]]  (model/net.py:43-45: F.relu(act_fc1(a)); torch.cat((a, goal,
// speed), dim=-1)) and its backward dh1 = gout[:, :256] where h1 > 0 -- one launch each instead of clamp_min + cat and
// narrow + copy + threshold_backward.  A thread owns one float4 of the output row (65 per row) / of h1 (64 per row).
constexpr int kFc1 = 256, kCat = 260;

__global__ __launch_bounds__(kThreads) void relu_cat_kernel(const float4* __restrict__ h1, const float2* __restrict__ goal,
                                                            const float2* __restrict__ speed, long n, float4* __restrict__ out) {
    const long total = n * (kCat / 4);
    const long stride = (long)gridDim.x * kThreads;
    for (long k = (long)blockIdx.x * kThreads + threadIdx.x; k < total; k += stride) {
        const long row = k / (kCat / 4);
        const int c4 = (int)(k - row * (kCat / 4));
        if (c4 < kFc1 / 4) {
            out[k] = relu4(h1[row * (kFc1 / 4) + c4]);
        } else {
            const float2 g = goal[row], v = speed[row];
            out[k] = make_float4(g.x, g.y, v.x, v.y);
        }
    }
}

__global__ __launch_bounds__(kThreads) void relu_cat_backward_kernel(const float4* __restrict__ h1, const float4* __restrict__ gout,
                                                                     long n, float4* __restrict__ dh1) {
    const long total = n * (kFc1 / 4);
    const long stride = (long)gridDim.x * kThreads;
    for (long k = (long)blockIdx.x * kThreads + threadIdx.x; k < total; k += stride) {
        const long row = k / (kFc1 / 4);
        const int c4 = (int)(k - row * (kFc1 / 4));
        dh1[k] = mask4(gout[row * (kCat / 4) + c4], h1[k]);
    }
}

// the same with the column sums of dh1 -- the gradient of fc1's bias (model/net.py:41: act_fc1 / crt_fc1 produced h1) -- formed on
// the way: a thread keeps ONE float4 column for all its rows (the grid's stride is a multiple of the 64 float4 of a row), the
// four row groups of a workgroup meet in LDS, one record of 256 sums per workgroup; colsum_finalize_kernel adds the records in a
// fixed order in float64.  As an autograd node that sum was a `reduce_kernel` over the 16 MB this kernel has just written.
constexpr int kBiasBlocks = 256;
__global__ __launch_bounds__(kThreads) void relu_cat_backward_bias_kernel(const float4* __restrict__ h1, const float4* __restrict__ gout,
                                                                          long n, float4* __restrict__ dh1, float* __restrict__ partial) {
    const int c4 = threadIdx.x & 63, grp = threadIdx.x >> 6;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (long row = (long)blockIdx.x * kWavesPerBlock + grp; row < n; row += (long)gridDim.x * kWavesPerBlock) {
        const float4 d = mask4(gout[row * (kCat / 4) + c4], h1[row * (kFc1 / 4) + c4]);
        dh1[row * (kFc1 / 4) + c4] = d;
        acc.x += d.x; acc.y += d.y; acc.z += d.z; acc.w += d.w;
    }
    __shared__ float4 red[kWavesPerBlock][kFc1 / 4];
    red[grp][c4] = acc;
    __syncthreads();
    if (grp == 0) {
        float4 t = red[0][c4];
#pragma unroll
        for (int g = 1; g < kWavesPerBlock; ++g) {
            t.x += red[g][c4].x; t.y += red[g][c4].y; t.z += red[g][c4].z; t.w += red[g][c4].w;
        }
        reinterpret_cast<float4*>(partial + (size_t)blockIdx.x * kFc1)[c4] = t;
    }
}

// out[k] = sum over the nrec records of partial[r][k], k < nout (pitch = nout), in a fixed order in float64
__global__ __launch_bounds__(kFinK * kFinGroups) void colsum_finalize_kernel(const float* __restrict__ partial, int nrec, int nout,
                                                                             float* __restrict__ out) {
    __shared__ double sh[kFinGroups][kFinK];
    const int kk = threadIdx.x & (kFinK - 1), grp = threadIdx.x / kFinK;
    const int k = blockIdx.x * kFinK + kk;
    double s = 0.0;
    if (k < nout) {
#pragma unroll 8
        for (int r = grp; r < nrec; r += kFinGroups) s += (double)partial[(size_t)r * nout + k];
    }
    sh[grp][kk] = s;
    __syncthreads();
    if (grp == 0 && k < nout) {
        double t = sh[0][kk];
#pragma unroll
        for (int g = 1; g < kFinGroups; ++g) t += sh[g][kk];
        out[k] = (float)t;
    }
}

static int blocks_for(int n) {
    int rows_pairs = (n + 1) / 2;
    int b = (rows_pairs + kWavesPerBlock - 1) / kWavesPerBlock;
    return b < 1 ? 1 : (b > kMaxBlocks ? kMaxBlocks : b);
}

}  // namespace mrca_heads

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int mrca_policy_heads(const float* a_dev, const float* c_dev, int32_t n, const float* w_actor1_dev,
                                 const float* b_actor1_dev, const float* w_actor2_dev, const float* b_actor2_dev,
                                 const float* w_critic_dev, const float* b_critic_dev, int32_t relu_inputs, float* mean_dev,
                                 float* value_dev, void* stream) {
    using namespace mrca_heads;
    if (!a_dev || !c_dev || !w_actor1_dev || !b_actor1_dev || !w_actor2_dev || !b_actor2_dev || !w_critic_dev || !b_critic_dev ||
        !mean_dev || !value_dev)
        return mrca::set_error(MRCA_ERR_INVALID, "mrca_policy_heads: NULL pointer");
    if (n < 1) return mrca::set_error(MRCA_ERR_INVALID, "mrca_policy_heads: n = %d", n);
    if (!aligned16(a_dev) || !aligned16(c_dev))
        return mrca::set_error(MRCA_ERR_INVALID, "mrca_policy_heads: the feature matrices must be 16-byte aligned");
    mrca::DeviceGuard guard(mrca::device_of(a_dev));
    if (relu_inputs)
        hipLaunchKernelGGL(heads_forward_kernel<true>, dim3(blocks_for(n)), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                           a_dev, c_dev, n, w_actor1_dev, b_actor1_dev, w_actor2_dev, b_actor2_dev, w_critic_dev, b_critic_dev,
                           mean_dev, value_dev);
    else
        hipLaunchKernelGGL(heads_forward_kernel<false>, dim3(blocks_for(n)), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                           a_dev, c_dev, n, w_actor1_dev, b_actor1_dev, w_actor2_dev, b_actor2_dev, w_critic_dev, b_critic_dev,
                           mean_dev, value_dev);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mrca::set_error(MRCA_ERR_HIP, "mrca_policy_heads launch: %s", hipGetErrorString(e));
    return MRCA_OK;
}

extern "C" int mrca_policy_heads_backward_scratch(size_t* bytes_out) {
    if (!bytes_out) return mrca::set_error(MRCA_ERR_INVALID, "mrca_policy_heads_backward_scratch: bytes_out is NULL");
    *bytes_out = mrca_heads::kScratchBytes;
    return MRCA_OK;
}

static int policy_heads_backward_impl(const float* a_dev, const float* c_dev, const float* mean_dev, const float* gmean_dev,
                                      const float* gvalue_dev, int32_t n, const float* w_actor1_dev, const float* w_actor2_dev,
                                      const float* w_critic_dev, int32_t relu_inputs, float* da_dev, float* dc_dev, float* dw_dev,
                                      float* dzb_dev, void* scratch_dev, size_t scratch_bytes, void* stream) {
    using namespace mrca_heads;
    if (!a_dev || !c_dev || !mean_dev || !w_actor1_dev || !w_actor2_dev || !w_critic_dev || !da_dev || !dc_dev || !dw_dev ||
        !scratch_dev)
        return mrca::set_error(MRCA_ERR_INVALID, "mrca_policy_heads_backward: NULL pointer");
    if (n < 1) return mrca::set_error(MRCA_ERR_INVALID, "mrca_policy_heads_backward: n = %d", n);
    if (scratch_bytes < kScratchBytes)
        return mrca::set_error(MRCA_ERR_NOMEM, "mrca_policy_heads_backward: scratch of %zu bytes < %zu", scratch_bytes, kScratchBytes);
    if (!aligned16(a_dev) || !aligned16(c_dev) || !aligned16(da_dev) || !aligned16(dc_dev) || !aligned16(scratch_dev))
        return mrca::set_error(MRCA_ERR_INVALID, "mrca_policy_heads_backward: features, their gradients and the scratch must be "
                                                 "16-byte aligned");
    mrca::DeviceGuard guard(mrca::device_of(a_dev));
    const int blocks = blocks_for(n);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (relu_inputs)
        hipLaunchKernelGGL(heads_backward_kernel<true>, dim3(blocks), dim3(kThreads), 0, s, a_dev, c_dev, mean_dev, gmean_dev,
                           gvalue_dev, n, w_actor1_dev, w_actor2_dev, w_critic_dev, da_dev, dc_dev, static_cast<float*>(scratch_dev));
    else
        hipLaunchKernelGGL(heads_backward_kernel<false>, dim3(blocks), dim3(kThreads), 0, s, a_dev, c_dev, mean_dev, gmean_dev,
                           gvalue_dev, n, w_actor1_dev, w_actor2_dev, w_critic_dev, da_dev, dc_dev, static_cast<float*>(scratch_dev));
    hipLaunchKernelGGL(heads_finalize_kernel, dim3(((dzb_dev ? kOut : kHeadOut) + kFinK - 1) / kFinK), dim3(kFinK * kFinGroups), 0, s,
                       static_cast<const float*>(scratch_dev), blocks, dw_dev, dzb_dev);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mrca::set_error(MRCA_ERR_HIP, "mrca_policy_heads_backward launch: %s", hipGetErrorString(e));
    return MRCA_OK;
}

extern "C" int mrca_policy_heads_backward(const float* a_dev, const float* c_dev, const float* mean_dev, const float* gmean_dev,
                                          const float* gvalue_dev, int32_t n, const float* w_actor1_dev,
                                          const float* w_actor2_dev, const float* w_critic_dev, int32_t relu_inputs,
                                          float* da_dev, float* dc_dev, float* dw_dev, void* scratch_dev, size_t scratch_bytes,
                                          void* stream) {
    return policy_heads_backward_impl(a_dev, c_dev, mean_dev, gmean_dev, gvalue_dev, n, w_actor1_dev, w_actor2_dev, w_critic_dev,
                                      relu_inputs, da_dev, dc_dev, dw_dev, nullptr, scratch_dev, scratch_bytes, stream);
}

extern "C" int mrca_policy_heads_backward_bias(const float* a_dev, const float* c_dev, const float* mean_dev, const float* gmean_dev,
                                               const float* gvalue_dev, int32_t n, const float* w_actor1_dev,
                                               const float* w_actor2_dev, const float* w_critic_dev, int32_t relu_inputs,
                                               float* da_dev, float* dc_dev, float* dw_dev, float* dz_bias_dev, void* scratch_dev,
                                               size_t scratch_bytes, void* stream) {
    if (!dz_bias_dev) return mrca::set_error(MRCA_ERR_INVALID, "mrca_policy_heads_backward_bias: dz_bias_dev is NULL");
    return policy_heads_backward_impl(a_dev, c_dev, mean_dev, gmean_dev, gvalue_dev, n, w_actor1_dev, w_actor2_dev, w_critic_dev,
                                      relu_inputs, da_dev, dc_dev, dw_dev, dz_bias_dev, scratch_dev, scratch_bytes, stream);
}

extern "C" int mrca_relu_cat(const float* h1_dev, const float* goal_dev, const float* speed_dev, int32_t n, float* out_dev,
                             void* stream) {
    using namespace mrca_heads;
    if (!h1_dev || !goal_dev || !speed_dev || !out_dev) return mrca::set_error(MRCA_ERR_INVALID, "mrca_relu_cat: NULL pointer");
    if (n < 1) return mrca::set_error(MRCA_ERR_INVALID, "mrca_relu_cat: n = %d", n);
    if (!aligned16(h1_dev) || !aligned16(out_dev) || (reinterpret_cast<uintptr_t>(goal_dev) & 7) || (reinterpret_cast<uintptr_t>(speed_dev) & 7))
        return mrca::set_error(MRCA_ERR_INVALID, "mrca_relu_cat: h1 / out must be 16-byte, goal / speed 8-byte aligned");
    mrca::DeviceGuard guard(mrca::device_of(h1_dev));
    long blocks = ((long)n * (kCat / 4) + kThreads - 1) / kThreads;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(relu_cat_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float4*>(h1_dev), reinterpret_cast<const float2*>(goal_dev),
                       reinterpret_cast<const float2*>(speed_dev), (long)n, reinterpret_cast<float4*>(out_dev));
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mrca::set_error(MRCA_ERR_HIP, "mrca_relu_cat launch: %s", hipGetErrorString(e));
    return MRCA_OK;
}

extern "C" int mrca_relu_cat_backward(const float* h1_dev, const float* gout_dev, int32_t n, float* dh1_dev, void* stream) {
    using namespace mrca_heads;
    if (!h1_dev || !gout_dev || !dh1_dev) return mrca::set_error(MRCA_ERR_INVALID, "mrca_relu_cat_backward: NULL pointer");
    if (n < 1) return mrca::set_error(MRCA_ERR_INVALID, "mrca_relu_cat_backward: n = %d", n);
    if (!aligned16(h1_dev) || !aligned16(gout_dev) || !aligned16(dh1_dev))
        return mrca::set_error(MRCA_ERR_INVALID, "mrca_relu_cat_backward: the three buffers must be 16-byte aligned");
    mrca::DeviceGuard guard(mrca::device_of(h1_dev));
    long blocks = ((long)n * (kFc1 / 4) + kThreads - 1) / kThreads;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(relu_cat_backward_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float4*>(h1_dev), reinterpret_cast<const float4*>(gout_dev), (long)n,
                       reinterpret_cast<float4*>(dh1_dev));
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mrca::set_error(MRCA_ERR_HIP, "mrca_relu_cat_backward launch: %s", hipGetErrorString(e));
    return MRCA_OK;
}

extern "C" int mrca_relu_cat_backward_bias_scratch(size_t* bytes_out) {
    if (!bytes_out) return mrca::set_error(MRCA_ERR_INVALID, "mrca_relu_cat_backward_bias_scratch: bytes_out is NULL");
    *bytes_out = sizeof(float) * mrca_heads::kBiasBlocks * mrca_heads::kFc1;
    return MRCA_OK;
}

extern "C" int mrca_relu_cat_backward_bias(const float* h1_dev, const float* gout_dev, int32_t n, float* dh1_dev, float* db_dev,
                                           void* scratch_dev, size_t scratch_bytes, void* stream) {
    using namespace mrca_heads;
    if (!h1_dev || !gout_dev || !dh1_dev || !db_dev || !scratch_dev)
        return mrca::set_error(MRCA_ERR_INVALID, "mrca_relu_cat_backward_bias: NULL pointer");
    if (n < 1) return mrca::set_error(MRCA_ERR_INVALID, "mrca_relu_cat_backward_bias: n = %d", n);
    if (scratch_bytes < sizeof(float) * kBiasBlocks * kFc1)
        return mrca::set_error(MRCA_ERR_NOMEM, "mrca_relu_cat_backward_bias: scratch of %zu bytes < %zu", scratch_bytes,
                               sizeof(float) * kBiasBlocks * kFc1);
    if (!aligned16(h1_dev) || !aligned16(gout_dev) || !aligned16(dh1_dev) || !aligned16(scratch_dev))
        return mrca::set_error(MRCA_ERR_INVALID, "mrca_relu_cat_backward_bias: the buffers and the scratch must be 16-byte aligned");
    mrca::DeviceGuard guard(mrca::device_of(h1_dev));
    int blocks = (n + kWavesPerBlock - 1) / kWavesPerBlock;
    if (blocks > kBiasBlocks) blocks = kBiasBlocks;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(relu_cat_backward_bias_kernel, dim3(blocks), dim3(kThreads), 0, s, reinterpret_cast<const float4*>(h1_dev),
                       reinterpret_cast<const float4*>(gout_dev), (long)n, reinterpret_cast<float4*>(dh1_dev),
                       static_cast<float*>(scratch_dev));
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((kFc1 + kFinK - 1) / kFinK), dim3(kFinK * kFinGroups), 0, s,
                       static_cast<const float*>(scratch_dev), blocks, kFc1, db_dev);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mrca::set_error(MRCA_ERR_HIP, "mrca_relu_cat_backward_bias launch: %s", hipGetErrorString(e));
    return MRCA_OK;
}
