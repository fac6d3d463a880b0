// Exhaustive check (all 2^32 float bit patterns, on the GPU): is a short v_rcp_f32 + Newton sequence bit-identical to
// the correctly rounded IEEE reciprocal 1.0f / x the ray march's specification asks for?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt tools/check_rcp.hip -o /tmp/check_rcp
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

__device__ __forceinline__ float rcp1(float d) {   // one Newton step
    float r = __builtin_amdgcn_rcpf(d);
    float e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float rcp2(float d) {   // two Newton steps
    float r = rcp1(d);
    float e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}

struct Stat { unsigned long long bad; unsigned int lo_abs, hi_abs; };   // mismatches, |x| range (bits) where they occur

__global__ void check(Stat* st, unsigned int abs_lo, unsigned int abs_hi) {
    const unsigned long long tid = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long bad[2] = {0, 0};
    unsigned int lo[2] = {0xFFFFFFFFu, 0xFFFFFFFFu}, hi[2] = {0, 0};
    for (unsigned long long u = tid; u < (1ull << 32); u += stride) {
        const unsigned int bits = (unsigned int)u, a = bits & 0x7FFFFFFFu;
        if (a < abs_lo || a > abs_hi) continue;
        const float x = __uint_as_float(bits);
        const float want = 1.0f / x;
        const float got[2] = {rcp1(x), rcp2(x)};
        for (int k = 0; k < 2; ++k)
            if (__float_as_uint(got[k]) != __float_as_uint(want)) {
                ++bad[k];
                lo[k] = a < lo[k] ? a : lo[k];
                hi[k] = a > hi[k] ? a : hi[k];
            }
    }
    for (int k = 0; k < 2; ++k)
        if (bad[k]) {
            atomicAdd(&st[k].bad, bad[k]);
            atomicMin(&st[k].lo_abs, lo[k]);
            atomicMax(&st[k].hi_abs, hi[k]);
        }
}

int main() {
    Stat* d;
    hipMalloc(&d, 2 * sizeof(Stat));
    const unsigned int ranges[3][2] = {{0x00000000u, 0x7FFFFFFFu},      // everything
                                       {0x00800000u, 0x7F7FFFFFu},      // all normal numbers
                                       {0x0D800000u, 0x71800000u}};     // 2^-100 .. 2^100
    const char* names[3] = {"all bit patterns", "normal numbers", "2^-100 <= |x| <= 2^100"};
    for (int r = 0; r < 3; ++r) {
        Stat h[2] = {{0, 0xFFFFFFFFu, 0}, {0, 0xFFFFFFFFu, 0}};
        hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, d, ranges[r][0], ranges[r][1]);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        for (int k = 0; k < 2; ++k) {
            float flo, fhi;
            memcpy(&flo, &h[k].lo_abs, 4);
            memcpy(&fhi, &h[k].hi_abs, 4);
            printf("%-26s rcp + %d Newton step(s): %llu mismatches vs IEEE 1/x", names[r], k + 1, h[k].bad);
            if (h[k].bad) printf("  (|x| from %g to %g)", flo, fhi);
            printf("\n");
        }
    }
    return 0;
}
