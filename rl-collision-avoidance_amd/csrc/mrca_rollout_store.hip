// mrca_rollout_store.hip -- what a tick of the rollout leaves in the learner's buffer, in two launches.
//
// The reference appends (state, a, r_list, terminal_list, logprob, v) to a Python list every step and turns the list into
// arrays before the update (ppo_stage1.py:102-103, model/ppo.py:22-54 transform_buffer).  The buffer here is preallocated on
// the device (mrca/ppo.py RolloutBuffer) and keeps ONE lidar frame per tick plus, per tick and robot, which frame rows make
// up its stack (the stack at tick t shares two frames with the stack at t-1; a robot that restarted has all three pointing
// at its fresh scan, ppo_stage1.py:59-60).  Written through PyTorch a tick's bookkeeping was ~20 launches of 2 - 8 us (the
// row copy, five index_copy_, the cat / where chain of the row indices, the counter) -- 27 us of a 270 us training tick
// (profiles/r05_z_train_kernel_stats_after.csv).  The row of the tick comes from a DEVICE counter: nothing here depends on a
// host value, so the tick can be captured once as a hipGraph and replayed for every row of the horizon.
//
//   rollout_store_state_kernel    before the env steps: the newest frame x / 6 - 0.5 from the scan ring into
//                                 frames[t + F - 1], the stack's row indices, goal / speed (the env's fields) and
//                                 action / logprob / value (the policy's outputs) into row t
//   rollout_store_outcome_kernel  after the env stepped: reward / done into row t; the last workgroup to finish bumps t
#include "mrca_rollout_store.h"

#include "mrca_device.h"

namespace mrca {
namespace {

constexpr int kThreads = 256;

// A thread owns float4 columns of the frame row; the threads whose column is 0 also write their robot's small fields.
__global__ __launch_bounds__(kThreads) void rollout_store_state_kernel(
    int N, int F, int B, const float* __restrict__ scan_ring, const uint8_t* __restrict__ ring_head,
    const float* __restrict__ local_goal, const float* __restrict__ speed, const uint8_t* __restrict__ fresh,
    uint32_t* __restrict__ status, mrca_rollout_rows rows, const int64_t* __restrict__ tick, const float* __restrict__ action,
    const float* __restrict__ logprob, const float* __restrict__ value) {
    const long long t = *tick;
    if (t < 0 || t >= rows.horizon) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(status, kStatusBadRolloutRow);
        return;
    }
    const int fstride = B >> 2;
    const long long total = (long long)N * fstride;
    const long long row = t + F - 1;                                  // the newest frame of tick t lives in row t + F - 1
    const float4* ring = reinterpret_cast<const float4*>(scan_ring);
    float4* out = reinterpret_cast<float4*>(rows.frames) + row * total;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += stride) {
        const long long n = k / fstride;
        const int col = (int)(k - n * fstride);
        const int hd = ring_head[n];
        // (|x|: a source modifier, free -- ABI 4-5 kept what a beam hit in the ring's sign bit)
        const float4 v = ring[(n * F + hd) * fstride + col];
        out[k] = make_float4(norm_obs(fabsf(v.x)), norm_obs(fabsf(v.y)), norm_obs(fabsf(v.z)), norm_obs(fabsf(v.w)));
        if (col != 0) continue;
        // --- which rows make up robot n's stack at tick t.  Tick 0 of a horizon: rows 0 .. F-1 whatever the flags say (the
        //     buffer's begin_horizon copied the real older frames there); a robot that restarted: F times its fresh scan
        int64_t* cur = rows.cur + n * F;
        int64_t* fi = rows.fidx + ((long long)t * N + n) * F;
        const bool restarted = fresh[n] != 0;
        for (int f = 0; f < F; ++f) {
            const int64_t next = f + 1 < F ? cur[f + 1] : row;      // (read before cur[f + 1] is overwritten: f ascends)
            const int64_t r = t == 0 ? (int64_t)f : (restarted ? (int64_t)row : next);
            cur[f] = r;
            fi[f] = r;
        }
        const long long tn = (long long)t * N + n;
        reinterpret_cast<float2*>(rows.goal)[tn] = reinterpret_cast<const float2*>(local_goal)[n];
        reinterpret_cast<float2*>(rows.speed)[tn] = reinterpret_cast<const float2*>(speed)[n];
        reinterpret_cast<float2*>(rows.action)[tn] = reinterpret_cast<const float2*>(action)[n];
        rows.logprob[tn] = logprob[n];
        rows.value[tn] = value[n];
    }
}

__global__ __launch_bounds__(kThreads) void rollout_store_outcome_kernel(int N, const float* __restrict__ reward,
                                                                         const uint8_t* __restrict__ done,
                                                                         uint32_t* __restrict__ status, mrca_rollout_rows rows,
                                                                         int64_t* __restrict__ tick, uint32_t* __restrict__ ticket) {
    const long long t = *tick;                       // every thread reads the counter before its workgroup takes a ticket
    const bool ok = t >= 0 && t < rows.horizon;
    const int n = blockIdx.x * kThreads + threadIdx.x;
    if (ok && n < N) {
        rows.reward[t * N + n] = reward[n];
        rows.done[t * N + n] = done[n];
    }
    if (!ok && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(status, kStatusBadRolloutRow);
    __syncthreads();
    if (threadIdx.x == 0) {
        // the last workgroup to get here has seen every other one read `tick`: it alone moves the counter on and leaves the
        // ticket at zero for the next launch (same stream: launches do not overlap)
        const uint32_t got = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (got == gridDim.x - 1) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ok) __hip_atomic_store(tick, (int64_t)(t + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace

void launch_rollout_store_state(const EnvView& e, const mrca_rollout_rows& rows, const int64_t* tick, const float* action,
                                const float* logprob, const float* value, hipStream_t s) {
    const long long cols = (long long)e.N * (e.B >> 2);
    long long nb = (cols + kThreads - 1) / kThreads;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(rollout_store_state_kernel, dim3((int)nb), dim3(kThreads), 0, s, e.N, e.F, e.B, e.scan_ring, e.ring_head,
                       e.local_goal, e.speed, e.fresh, e.status, rows, tick, action, logprob, value);
}

void launch_rollout_store_outcome(const EnvView& e, const mrca_rollout_rows& rows, int64_t* tick, uint32_t* ticket, hipStream_t s) {
    const int nb = (e.N + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(rollout_store_outcome_kernel, dim3(nb), dim3(kThreads), 0, s, e.N, e.reward, e.done, e.status, rows, tick,
                       ticket);
}

}  // namespace mrca
