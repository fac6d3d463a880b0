// Which LDS access of lidar_features_kernel (csrc/mrca_policy.hip) produces its bank-conflict cycles?
//
// The forward kernel of the policy's front end counts 1.4 SQ_LDS_BANK_CONFLICT cycles per LDS instruction
// (profiles/r03/r03_zz_pmc_policy_sq_counters_final_kernels.txt).  Each kernel below issues ONE of its access patterns -- the
// address formulas of csrc/mrca_policy_layout.h (namespace mrca_pfwd), same lanes, same pitches -- kReps times per wave and
// nothing else, so that
//     rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE -- tools/_build/lds_conflict_probe
// attributes the counter pattern by pattern (tools/pmc_summary.py prints the per-kernel averages).
//
//   stage_scan        the scan's de-interleaving stores: lane m writes XE[ci][2m], XO[ci][2m + 1], ...  (ds_write_b32, stride 2)
//   conv1_operands    B operands of conv1's MFMAs: base[col + hl + off]                                    (ds_read_b32)
//   conv1_epilogue    relu(conv1) back to LDS: even positions to H1E[c][p / 2], odd ones to H1O[c][(p + 1) / 2] (ds_write_b32)
//   conv2_operands    B operands of conv2's MFMAs (families A and B)                                         (ds_read_b32)
//   conv2_epilogue_w  conv2's accumulators into the transposition buffer: oe[row * kHPitch + col]           (ds_write_b32)
//   conv2_epilogue_r  rows of 64 positions out as float4: lane -> (row = lane / 16, 16-byte column = lane % 16) (ds_read_b128)
//
//   hipcc --offload-arch=gfx950 -O3 tools/lds_conflict_probe.hip -o tools/_build/lds_conflict_probe
#include <hip/hip_runtime.h>

#include <cstdio>

#include "../rl-collision-avoidance_amd/csrc/mrca_policy_layout.h"

using namespace mrca_pfwd;

namespace mrca_ldsprobe {     // ("mrca" in the kernel names: tools/pmc_summary.py lists those)

constexpr int kReps = 4096;
// the compiler may not keep a loaded value across repetitions
#define PROBE_RELOAD() asm volatile("" ::: "memory")

#define PROBE_PROLOGUE                                                   \
    extern __shared__ __attribute__((aligned(16))) float lds_all[];     \
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;         \
    float* lds = lds_all + wave * kWaveFloats;                           \
    const int col = lane & 31, hl = lane >> 5;                           \
    (void)col; (void)hl;                                                 \
    float acc = 0.0f;

#define PROBE_EPILOGUE \
    if (acc == 12345.678f) out[threadIdx.x] = acc;

__global__ void stage_scan(float* out) {
    PROBE_PROLOGUE
    for (int r = 0; r < kReps; ++r) {
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int idx = q * 64 + lane;
            const int ci = idx >> 7, m = idx & 127;
            float* xe = lds + kXE + ci * kXPitch + 2 * m;
            float* xo = lds + kXO + ci * kXPitch + 2 * m + 1;
            xe[0] = (float)r;
            xo[0] = (float)q;
            xe[1] = (float)lane;
            xo[1] = 1.0f;
        }
        __builtin_amdgcn_s_waitcnt(0);
    }
    PROBE_EPILOGUE
}

__global__ void conv1_operands(float* out) {
    PROBE_PROLOGUE
    const float* x1 = lds + col + hl;
    const float* x2 = lds + col + hl * kXPitch;
    const float* x3 = lds + col;
    for (int r = 0; r < kReps; ++r) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float* base = conv1_family(s) == 1 ? x1 : (conv1_family(s) == 2 ? x2 : x3);
            acc += base[conv1_step_off(s) + 32 * (r & 3)] + base[conv1_step_off(s) + 32 * (r & 3) + 32];
        }
        PROBE_RELOAD();
    }
    PROBE_EPILOGUE
}

__global__ void conv1_epilogue(float* out) {
    PROBE_PROLOGUE
    float* hst = lds + ((col & 1) ? kH1O + (col + 1) / 2 : kH1E + col / 2) + 4 * hl * kHPitch;
    for (int r = 0; r < kReps; ++r) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            hst[32 * (r & 3) + rowmap(reg, 0) * kHPitch] = (float)reg;
            hst[32 * (r & 3) + 16 + rowmap(reg, 0) * kHPitch] = (float)r;
        }
        __builtin_amdgcn_s_waitcnt(0);
    }
    PROBE_EPILOGUE
}

__global__ void conv2_operands(float* out) {
    PROBE_PROLOGUE
    const float* ha = lds + col + hl;
    const float* hb = lds + col + hl * kHPitch;
    for (int r = 0; r < kReps / 4; ++r) {
#pragma unroll
        for (int s = 0; s < 48; ++s) {
            const float* base = s < 32 ? ha : hb;
            acc += base[conv2_step_off(s) + 64 * (r & 1)] + base[conv2_step_off(s) + 64 * (r & 1) + 32];
        }
        PROBE_RELOAD();
    }
    PROBE_EPILOGUE
}

__global__ void conv2_epilogue_w(float* out) {
    PROBE_PROLOGUE
    float* oe = lds + kH1E + col + 4 * hl * kHPitch;
    for (int r = 0; r < kReps; ++r) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            oe[rowmap(reg, 0) * kHPitch + 64 * (r & 1)] = (float)reg;
            oe[rowmap(reg, 0) * kHPitch + 64 * (r & 1) + 32] = (float)r;
        }
        __builtin_amdgcn_s_waitcnt(0);
    }
    PROBE_EPILOGUE
}

__global__ void conv2_epilogue_r(float* out) {
    PROBE_PROLOGUE
    for (int r = 0; r < kReps; ++r) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {          // 32 rows of 64 positions = 8 x (4 rows x 16 float4) per tile pair
            const int row = 4 * q + (lane >> 4);
            const float4 v = *reinterpret_cast<const float4*>(lds + kH1E + row * kHPitch + 64 * (r & 1) + 4 * (lane & 15));
            acc += v.x + v.y + v.z + v.w;
        }
        PROBE_RELOAD();
    }
    PROBE_EPILOGUE
}

}  // namespace mrca_ldsprobe

int main() {
    using namespace mrca_ldsprobe;
    float* out;
    if (hipMalloc(&out, 4096) != hipSuccess) return 1;
    const size_t lds = (size_t)kWavesPerBlock * kWaveFloats * sizeof(float);
    void (*kernels[])(float*) = {stage_scan, conv1_operands, conv1_epilogue, conv2_operands, conv2_epilogue_w, conv2_epilogue_r};
    const char* names[] = {"stage_scan", "conv1_operands", "conv1_epilogue", "conv2_operands", "conv2_epilogue_w", "conv2_epilogue_r"};
    for (int k = 0; k < 6; ++k) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernels[k]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
            return 1;
        hipEvent_t a, b;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        hipLaunchKernelGGL(kernels[k], dim3(256), dim3(64 * kWavesPerBlock), lds, 0, out);      // warm
        (void)hipEventRecord(a, 0);
        hipLaunchKernelGGL(kernels[k], dim3(256), dim3(64 * kWavesPerBlock), lds, 0, out);
        (void)hipEventRecord(b, 0);
        (void)hipEventSynchronize(b);
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, a, b);
        printf("%-18s %8.1f us for %d repetitions per wave (256 workgroups x 4 waves)\n", names[k], ms * 1e3, kReps);
    }
    return 0;
}
