// A cross-stream dependency through stream memory operations instead of an event: stream A runs a kernel, then
// hipStreamWriteValue32(flag, k); stream B hipStreamWaitValue32(flag >= k), then its kernel.  Measured like tools/anyorder_probe.hip:
// from the end of A's kernel to the start of B's (device clock), and what the two calls cost the host -- against event record +
// hipStreamWaitEvent (10 - 11 us, 0.76 + 4.0 us).
//   hipcc --offload-arch=gfx950 -O3 tools/wait_value_probe.hip -o tools/_build/wait_value_probe && tools/_build/wait_value_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void busy_kernel(unsigned long long ticks, unsigned long long* stamps, int slot) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        atomicMin(&stamps[slot * 2 + 0], t0);
        atomicMax(&stamps[slot * 2 + 1], wall_clock64());
    }
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    int can = 0;
    (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    unsigned long long* stamps;
    CK(hipMalloc(&stamps, 64 * sizeof(unsigned long long)));
    std::vector<unsigned long long> h(64);
    auto reset = [&]() {
        for (int i = 0; i < 32; ++i) { h[2 * i] = ~0ull; h[2 * i + 1] = 0ull; }
        (void)hipMemcpy(stamps, h.data(), 64 * sizeof(unsigned long long), hipMemcpyHostToDevice);
    };
    auto fetch = [&]() { (void)hipMemcpy(h.data(), stamps, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost); };
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    uint32_t* flag = nullptr;
    // signal memory: the documented kind for stream memory operations
    hipError_t em = hipExtMallocWithFlags(reinterpret_cast<void**>(&flag), 64, hipMallocSignalMemory);
    if (em != hipSuccess) { printf("hipMallocSignalMemory: %s -> plain hipMalloc\n", hipGetErrorString(em)); CK(hipMalloc(&flag, 64)); }
    CK(hipMemset(flag, 0, 64));
    unsigned long long T = 2000;
    for (int warm = 0; warm < 3; ++warm) {
        hipLaunchKernelGGL(busy_kernel, dim3(64), dim3(64), 0, s0, T, stamps, 0);
        hipLaunchKernelGGL(busy_kernel, dim3(64), dim3(64), 0, s1, T, stamps, 1);
    }
    CK(hipDeviceSynchronize());
    uint32_t seq = 0;
    for (int rep = 0; rep < 6; ++rep) {
        reset();
        ++seq;
        hipLaunchKernelGGL(busy_kernel, dim3(64), dim3(64), 0, s0, T, stamps, 0);
        CK(hipStreamWriteValue32(s0, flag, seq, 0));
        CK(hipStreamWaitValue32(s1, flag, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
        hipLaunchKernelGGL(busy_kernel, dim3(64), dim3(64), 0, s1, T, stamps, 1);
        CK(hipDeviceSynchronize());
        fetch();
        printf("write value + wait value: k0 ends %.2f us, k1 starts %.2f us -> dependency latency %.2f us\n", (h[1] - h[0]) / 100.0,
               (double)(long long)(h[2] - h[0]) / 100.0, (double)(long long)(h[2] - h[1]) / 100.0);
    }
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (int rep = 0; rep < 3; ++rep) {
        reset();
        hipLaunchKernelGGL(busy_kernel, dim3(64), dim3(64), 0, s0, T, stamps, 0);
        CK(hipEventRecord(ev, s0));
        CK(hipStreamWaitEvent(s1, ev, 0));
        hipLaunchKernelGGL(busy_kernel, dim3(64), dim3(64), 0, s1, T, stamps, 1);
        CK(hipDeviceSynchronize());
        fetch();
        printf("event record + wait event: dependency latency %.2f us\n", (double)(long long)(h[2] - h[1]) / 100.0);
    }
    {   // host cost, queues kept busy
        unsigned long long longt = 300000;
        hipLaunchKernelGGL(busy_kernel, dim3(1), dim3(64), 0, s0, longt, stamps, 9);
        hipLaunchKernelGGL(busy_kernel, dim3(1), dim3(64), 0, s1, longt, stamps, 9);
        const int n = 200;
        double t0 = now_us();
        for (int i = 0; i < n; ++i) (void)hipStreamWriteValue32(s0, flag, ++seq, 0);
        double t1 = now_us();
        for (int i = 0; i < n; ++i) (void)hipStreamWaitValue32(s1, flag, seq - n + i + 1, hipStreamWaitValueGte, 0xFFFFFFFFu);
        double t2 = now_us();
        for (int i = 0; i < n; ++i) (void)hipEventRecord(ev, s0);
        double t3 = now_us();
        for (int i = 0; i < n; ++i) (void)hipStreamWaitEvent(s1, ev, 0);
        double t4 = now_us();
        CK(hipDeviceSynchronize());
        printf("host cost per call: hipStreamWriteValue32 %.2f us, hipStreamWaitValue32 %.2f us; hipEventRecord %.2f us, hipStreamWaitEvent %.2f us\n",
               (t1 - t0) / n, (t2 - t1) / n, (t3 - t2) / n, (t4 - t3) / n);
    }
    return 0;
}
