// valu_rate_probe.hip -- how many cycles a SIMD of gfx950 takes per wave64 VALU instruction (plain v_fma_f32 / v_add_u32 / v_cndmask),
// with 1, 2, 4 and 8 waves per SIMD: the "VALU issue" roof DESIGN.md prices the ray cast against.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate_probe.hip -o tools/_build/valu_rate_probe && tools/_build/valu_rate_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    float a[16];
    int b[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        a[k] = (float)(threadIdx.x + k);
        b[k] = threadIdx.x * 3 + k;
    }
    const float m = 1.0000001f, c = 1e-9f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {      // 16 independent chains: no dependent-issue stall
            if (KIND == 0) a[k] = __builtin_fmaf(a[k], m, c);
            if (KIND == 1) b[k] = b[k] + (b[k] >> 3);                       // v_ashrrev + v_add (2 instructions)
            if (KIND == 2) a[k] = a[k] > 0.5f ? a[k] * m : a[k] + c;        // v_cmp + v_mul + v_add + v_cndmask
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += a[k] + (float)b[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * cus * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    const char* names[3] = {"v_fma_f32 (1 instruction / element)", "v_ashrrev_i32 + v_add_u32 (2)", "v_cmp + v_mul + v_add + v_cndmask (4)"};
    const int per[3] = {1, 2, 4};
    printf("# %s: %d CUs, clock %d MHz (hipDeviceProp clockRate)\n", p.name, cus, p.clockRate / 1000);
    for (int kind = 0; kind < 3; ++kind)
        for (int wgs_per_cu = 1; wgs_per_cu <= 8; wgs_per_cu *= 2) {     // 256 threads = 4 waves = 1 wave per SIMD per workgroup
            const int grid = cus * wgs_per_cu;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                if (kind == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(256), 0, 0, out, iters);
                if (kind == 1) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(256), 0, 0, out, iters);
                if (kind == 2) hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(256), 0, 0, out, iters);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
            }
            float ms = 0.0f;
            hipEventElapsedTime(&ms, e0, e1);
            const double insts_per_simd = (double)wgs_per_cu * iters * 16 * per[kind];     // wave-instructions one SIMD issued
            const double ns_per_inst = ms * 1e6 / insts_per_simd;
            printf("%-44s %d wave(s) per SIMD: %8.3f ms  %6.3f ns per wave-instruction per SIMD = %5.2f cycles at 2.4 GHz, %5.2f at 2.1 GHz\n",
                   names[kind], wgs_per_cu, ms, ns_per_inst, ns_per_inst * 2.4, ns_per_inst * 2.1);
        }
    return 0;
}
