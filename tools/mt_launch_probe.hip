// Do launches from two host threads, each to a stream of its own, proceed in parallel inside the HIP runtime?  (Would a helper
// thread that enqueues the second world range's ray casts shorten the host's share of a mrca_step_many region?)
// 2 x 300 launches + a stream wait every 4th on one thread against 300 + 300 on two threads; queues kept busy by a long kernel.
//   hipcc --offload-arch=gfx950 -O3 -pthread tools/mt_launch_probe.hip -o tools/_build/mt_launch_probe && tools/_build/mt_launch_probe
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>

struct Blob { unsigned v[22]; };
__global__ void sink_kernel(Blob b, unsigned* out) { if (b.v[0] == 0xFFFFFFFFu) out[0] = 1; }
__global__ void busy_kernel(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    hipStream_t s[2];
    hipEvent_t ev;
    unsigned* out;
    (void)hipMalloc(&out, 64);
    for (auto& x : s) (void)hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    Blob b{};
    const int n = 300;
    auto burst = [&](hipStream_t st) {
        for (int i = 0; i < n; ++i) {
            if ((i & 3) == 0) (void)hipStreamWaitEvent(st, ev, 0);
            hipLaunchKernelGGL(sink_kernel, dim3(2048), dim3(256), 0, st, b, out);
        }
    };
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(ev, s[0]);
        hipLaunchKernelGGL(busy_kernel, dim3(1), dim3(64), 0, s[0], 400000ull);
        hipLaunchKernelGGL(busy_kernel, dim3(1), dim3(64), 0, s[1], 400000ull);
        double t0 = now_us();
        burst(s[0]);
        burst(s[1]);
        double t1 = now_us();
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(ev, s[0]);
        hipLaunchKernelGGL(busy_kernel, dim3(1), dim3(64), 0, s[0], 400000ull);
        hipLaunchKernelGGL(busy_kernel, dim3(1), dim3(64), 0, s[1], 400000ull);
        std::atomic<int> go{0};
        std::thread helper([&]() { while (!go.load()) {} burst(s[1]); });
        double t2 = now_us();
        go.store(1);
        burst(s[0]);
        helper.join();
        double t3 = now_us();
        (void)hipDeviceSynchronize();
        if (rep) printf("one thread, two streams: %.2f us per launch (+ a wait every 4th); two threads, a stream each: %.2f us per launch of wall time\n",
                        (t1 - t0) / (2 * n), (t3 - t2) / (2 * n));
    }
    return 0;
}
