// Can two launches on ONE stream overlap on gfx950?  hipExtLaunchKernel's flag hipExtAnyOrderLaunch clears the AQL packet's
// barrier bit, so the packet does not wait for the one before it on the same queue to finish (hip_ext.h notes the flag as
// "not supported on GFX9xx" for the module-launch entry point; this measures what the runtime in this image really does).
// Also measured, for the schedule of DESIGN 5.10: what a cross-stream dependency costs on the device -- event record on stream
// A, hipStreamWaitEvent on stream B -- from the end of A's kernel to the start of B's, and the host's cost per API call.
//
//   hipcc --offload-arch=gfx950 -O3 tools/anyorder_probe.hip -o tools/_build/anyorder_probe && tools/_build/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>

// every workgroup spins `ticks` of the 100 MHz constant counter and stamps its start / end
__global__ void busy_kernel(unsigned long long ticks, unsigned long long* stamps, int slot) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        atomicMin(&stamps[slot * 2 + 0], t0);
        atomicMax(&stamps[slot * 2 + 1], wall_clock64());
    }
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    unsigned long long* stamps;
    (void)hipMalloc(&stamps, 64 * sizeof(unsigned long long));
    std::vector<unsigned long long> h(64);
    auto reset = [&]() {
        for (int i = 0; i < 32; ++i) { h[2 * i] = ~0ull; h[2 * i + 1] = 0ull; }
        (void)hipMemcpy(stamps, h.data(), 64 * sizeof(unsigned long long), hipMemcpyHostToDevice);
    };
    auto fetch = [&]() { (void)hipMemcpy(h.data(), stamps, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost); };
    hipStream_t s0, s1;
    (void)hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
    (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    const unsigned long long T = 2000;      // 20 us
    unsigned long long ticks = T;
    int slot0 = 0, slot1 = 1, slot2 = 2;
    void* a0[] = {&ticks, &stamps, &slot0};
    void* a1[] = {&ticks, &stamps, &slot1};
    void* a2[] = {&ticks, &stamps, &slot2};
    for (int warm = 0; warm < 3; ++warm) {
        (void)hipExtLaunchKernel((const void*)busy_kernel, dim3(64), dim3(64), a0, 0, s0, nullptr, nullptr, 0);
        (void)hipExtLaunchKernel((const void*)busy_kernel, dim3(64), dim3(64), a1, 0, s1, nullptr, nullptr, 0);
    }
    (void)hipDeviceSynchronize();

    for (int flags = 0; flags <= 1; ++flags) {
        for (int rep = 0; rep < 3; ++rep) {
            reset();
            (void)hipExtLaunchKernel((const void*)busy_kernel, dim3(64), dim3(64), a0, 0, s0, nullptr, nullptr, 0);
            (void)hipExtLaunchKernel((const void*)busy_kernel, dim3(64), dim3(64), a1, 0, s0, nullptr, nullptr, flags);
            (void)hipExtLaunchKernel((const void*)busy_kernel, dim3(64), dim3(64), a2, 0, s0, nullptr, nullptr, 0);
            (void)hipDeviceSynchronize();
            fetch();
            printf("one stream, 2nd launch flags=%d (%s): k0 [0, %.2f] us  k1 starts %.2f ends %.2f  k2 (ordered) starts %.2f\n", flags,
                   flags ? "hipExtAnyOrderLaunch" : "in order", (h[1] - h[0]) / 100.0, (double)(long long)(h[2] - h[0]) / 100.0,
                   (double)(long long)(h[3] - h[0]) / 100.0, (double)(long long)(h[4] - h[0]) / 100.0);
        }
    }
    // cross-stream dependency: k0 on s0, event, s1 waits, k1 on s1
    hipEvent_t ev;
    (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    for (int rep = 0; rep < 5; ++rep) {
        reset();
        (void)hipExtLaunchKernel((const void*)busy_kernel, dim3(64), dim3(64), a0, 0, s0, nullptr, nullptr, 0);
        (void)hipEventRecord(ev, s0);
        (void)hipStreamWaitEvent(s1, ev, 0);
        (void)hipExtLaunchKernel((const void*)busy_kernel, dim3(64), dim3(64), a1, 0, s1, nullptr, nullptr, 0);
        (void)hipDeviceSynchronize();
        fetch();
        printf("two streams, event record + wait: k0 ends %.2f us, k1 starts %.2f us -> dependency latency %.2f us\n",
               (h[1] - h[0]) / 100.0, (double)(long long)(h[2] - h[0]) / 100.0, (double)(long long)(h[2] - h[1]) / 100.0);
    }
    // same-stream back to back, for comparison
    for (int rep = 0; rep < 3; ++rep) {
        reset();
        (void)hipExtLaunchKernel((const void*)busy_kernel, dim3(64), dim3(64), a0, 0, s0, nullptr, nullptr, 0);
        (void)hipExtLaunchKernel((const void*)busy_kernel, dim3(64), dim3(64), a1, 0, s0, nullptr, nullptr, 0);
        (void)hipDeviceSynchronize();
        fetch();
        printf("one stream back to back: gap %.2f us\n", (double)(long long)(h[2] - h[1]) / 100.0);
    }
    // host cost per call, queues kept busy by a long kernel so nothing completes meanwhile
    {
        unsigned long long longt = 300000;   // 3 ms
        int slot9 = 9;
        void* al[] = {&longt, &stamps, &slot9};
        unsigned long long zero = 0;
        void* az[] = {&zero, &stamps, &slot9};
        (void)hipExtLaunchKernel((const void*)busy_kernel, dim3(1), dim3(64), al, 0, s0, nullptr, nullptr, 0);
        (void)hipExtLaunchKernel((const void*)busy_kernel, dim3(1), dim3(64), al, 0, s1, nullptr, nullptr, 0);
        const int n = 200;
        double t0 = now_us();
        for (int i = 0; i < n; ++i) (void)hipExtLaunchKernel((const void*)busy_kernel, dim3(1), dim3(64), az, 0, s0, nullptr, nullptr, 0);
        double t1 = now_us();
        for (int i = 0; i < n; ++i) (void)hipEventRecord(ev, s0);
        double t2 = now_us();
        for (int i = 0; i < n; ++i) (void)hipStreamWaitEvent(s1, ev, 0);
        double t3 = now_us();
        for (int i = 0; i < n; ++i) {
            (void)hipExtLaunchKernel((const void*)busy_kernel, dim3(1), dim3(64), az, 0, s0, nullptr, nullptr, 0);
            (void)hipEventRecord(ev, s0);
            (void)hipStreamWaitEvent(s1, ev, 0);
            (void)hipExtLaunchKernel((const void*)busy_kernel, dim3(1), dim3(64), az, 0, s1, nullptr, nullptr, 0);
        }
        double t4 = now_us();
        (void)hipDeviceSynchronize();
        printf("host cost per call: hipExtLaunchKernel %.2f us, hipEventRecord %.2f us, hipStreamWaitEvent %.2f us, "
               "launch + record + wait + launch %.2f us\n", (t1 - t0) / n, (t2 - t1) / n, (t3 - t2) / n, (t4 - t3) / n);
    }
    return 0;
}
