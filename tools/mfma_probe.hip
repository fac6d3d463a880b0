// What the fp32 matrix pipe of one SIMD sustains, and what a wave can issue in an MFMA's shadow (gfx950).
// The policy kernels run ONE wave per SIMD (their LDS image decides that), so their roof is not "157 TFLOP/s" but
// whatever a single wave can keep in flight: this probe measures it.    hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip
//   A  two independent accumulator chains, nothing else            -> cycles per v_mfma_f32_32x32x2_f32
//   B  + 4 independent VALU instructions per MFMA
//   C  + 8 independent VALU instructions per MFMA
//   D  + 1 ds_read_b32 and 1 ds_write_b32 per MFMA
//   E  + 4 VALU, 1 ds_read, 1 ds_write per MFMA (the shape of the conv kernels' inner steps)
//   F  one chain (every MFMA depends on the previous one)           -> latency
// each with 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define PIN() __builtin_amdgcn_sched_barrier(0)
#define VOP(v) asm volatile("v_max_i32 %0, %0, %1" : "+v"(v) : "v"(lane))

template <int MODE>
__global__ __launch_bounds__(1024) void probe(float* out, int iters, unsigned long long* ticks, unsigned long long* real) {
    __shared__ float lds[4096];
    const int lane = threadIdx.x & 63;
    f32x16 c0, c1;
    for (int r = 0; r < 16; ++r) {
        c0[r] = 0.0f;
        c1[r] = 0.0f;
    }
    float a = 1.0f + lane, b = 0.5f;
    int v0 = lane, v1 = lane + 1, v2 = lane + 2, v3 = lane + 3, v4 = lane + 4, v5 = lane + 5, v6 = lane + 6, v7 = lane + 7;
    float l0 = 0.0f;
    lds[threadIdx.x] = a;
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();      // the constant 100 MHz counter
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            c0 = MFMA(a, b, c0);
            if (MODE == 1 || MODE == 2 || MODE == 4) {
                VOP(v0);
                VOP(v1);
            }
            if (MODE == 2) {
                VOP(v4);
                VOP(v5);
            }
            if (MODE == 3 || MODE == 4) l0 += lds[(lane + 64 * u) & 4095];
            PIN();
            if (MODE != 5) c1 = MFMA(a, b, c1);
            else c0 = MFMA(a, b, c0);
            if (MODE == 1 || MODE == 2 || MODE == 4) {
                VOP(v2);
                VOP(v3);
            }
            if (MODE == 2) {
                VOP(v6);
                VOP(v7);
            }
            if (MODE == 3 || MODE == 4) lds[(lane + 64 * u + 2048) & 4095] = a;
            PIN();
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = l0 + (float)(v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7);
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        ticks[blockIdx.x] = t1 - t0;
        real[blockIdx.x] = r1 - r0;
    }
}

template <int MODE>
static void run(const char* label, int waves_per_simd, int cus) {
    const int iters = 20000, threads = 256 * waves_per_simd;      // 4 SIMDs per CU
    float* out;
    unsigned long long *ticks, *real;
    hipMalloc(&out, sizeof(float) * (size_t)cus * threads);
    hipMalloc(&ticks, sizeof(unsigned long long) * cus);
    hipMalloc(&real, sizeof(unsigned long long) * cus);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(cus), dim3(threads), 0, 0, out, 100, ticks, real);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<MODE>, dim3(cus), dim3(threads), 0, 0, out, iters, ticks, real);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.0f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(cus), hr(cus);
    hipMemcpy(h.data(), ticks, sizeof(unsigned long long) * cus, hipMemcpyDeviceToHost);
    hipMemcpy(hr.data(), real, sizeof(unsigned long long) * cus, hipMemcpyDeviceToHost);
    double avg = 0.0, ghz = 0.0;
    for (int i = 0; i < cus; ++i) {
        avg += (double)h[i];
        ghz += (double)h[i] / ((double)hr[i] * 10.0);
    }
    avg /= cus;
    ghz /= cus;
    const double mfma_per_simd = 16.0 * iters * waves_per_simd;
    const double flops = mfma_per_simd * 4.0 * cus * 4096.0;
    printf("%-58s %d wave(s)/SIMD  %7.2f ns per MFMA per SIMD  %7.1f memtime ticks per MFMA  %6.1f TFLOP/s  clock %.3f GHz\n", label, waves_per_simd,
           ms * 1e6 / mfma_per_simd, avg / (16.0 * iters) / waves_per_simd, flops / (ms * 1e-3) / 1e12, ghz);
    hipFree(real);
    hipFree(out);
    hipFree(ticks);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs, clock %d kHz\n", p.name, cus, p.clockRate);
    for (int w : {1, 2, 4}) {
        run<0>("A  two chains, MFMA only", w, cus);
        run<1>("B  + 4 VALU per MFMA pair... (2 per MFMA)", w, cus);
        run<2>("C  + 4 VALU per MFMA", w, cus);
        run<3>("D  + 1 ds_read / 1 ds_write per MFMA pair", w, cus);
        run<4>("E  + 2 VALU per MFMA, 1 ds_read / 1 ds_write per pair", w, cus);
        run<5>("F  one chain (dependent MFMAs)", w, cus);
    }
    return 0;
}
