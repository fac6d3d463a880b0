// Which pairs of HIP streams of one process can run kernels CONCURRENTLY on gfx950?  The runtime maps streams onto a few
// hardware queues; two streams that share one run their kernels one after the other whatever the program says.  This probe
// creates S non-blocking streams (index 0 = the NULL stream), launches a 20 us busy kernel on stream i and on stream j back
// to back from the host and reads the kernels' own start / end stamps: '#' = j started only after i had ended (serialised),
// '.' = they overlapped.       hipcc --offload-arch=gfx950 -O3 tools/queue_alias_probe.hip -o tools/_build/queue_alias_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void busy_kernel(unsigned long long ticks, unsigned long long* stamps, int slot) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        stamps[slot * 2 + 0] = t0;
        stamps[slot * 2 + 1] = wall_clock64();
    }
}

int main(int argc, char** argv) {
    const int S = argc > 1 ? atoi(argv[1]) : 12;
    const int high = argc > 2 ? atoi(argv[2]) : -1;      // index of a stream to create at HIGH priority (-1: none)
    const int masked_from = argc > 3 ? atoi(argv[3]) : 1 << 30;   // streams with index >= this: hipExtStreamCreateWithCUMask, all CUs
    std::vector<hipStream_t> st(S, nullptr);
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    for (int i = 1; i < S; ++i) {
        if (i >= masked_from) {
            uint32_t mask[8] = {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u};
            if (hipExtStreamCreateWithCUMask(&st[i], 8, mask) != hipSuccess) printf("stream %d: hipExtStreamCreateWithCUMask failed\n", i);
        } else if (i == high) (void)hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, hi);
        else (void)hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
    }
    unsigned long long* stamps;
    (void)hipMalloc(&stamps, 4 * sizeof(unsigned long long));
    unsigned long long h[4];
    unsigned long long ticks = 2000;
    int s0 = 0, s1 = 1;
    void* a0[] = {&ticks, &stamps, &s0};
    void* a1[] = {&ticks, &stamps, &s1};
    for (int i = 0; i < S; ++i) (void)hipLaunchKernel((const void*)busy_kernel, dim3(1), dim3(64), a0, 0, st[i]);
    (void)hipDeviceSynchronize();
    printf("%d streams (0 = NULL stream%s%s); '#': the pair ran one after the other, '.': overlapped\n    ", S,
           high > 0 ? ", one at high priority" : "", masked_from < S ? ", the upper ones created with a full CU mask" : "");
    for (int j = 0; j < S; ++j) printf("%2d ", j);
    printf("\n");
    for (int i = 0; i < S; ++i) {
        printf("%2d  ", i);
        for (int j = 0; j < S; ++j) {
            if (j == i) { printf(" \\ "); continue; }
            int serial = 0;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipLaunchKernel((const void*)busy_kernel, dim3(1), dim3(64), a0, 0, st[i]);
                (void)hipLaunchKernel((const void*)busy_kernel, dim3(1), dim3(64), a1, 0, st[j]);
                (void)hipDeviceSynchronize();
                (void)hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost);
                if (h[2] >= h[1]) ++serial;
            }
            printf(" %c ", serial >= 2 ? '#' : '.');
        }
        printf("\n");
    }
    return 0;
}
