// What a launch costs the HOST as a function of the size of its kernel arguments (the env kernels take a ~700-byte EnvView by
// value): empty kernels with 16 / 104 / 704 bytes of arguments, enqueued on a stream kept busy by a long kernel, hipLaunchKernelGGL
// and hipModuleLaunchKernel-style (hipExtLaunchKernel) entry points.
//   hipcc --offload-arch=gfx950 -O3 tools/launch_cost_probe.hip -o tools/_build/launch_cost_probe && tools/_build/launch_cost_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>

template <int N>
struct Blob { unsigned v[N]; };

template <int N>
__global__ void sink_kernel(Blob<N> b, unsigned* out) {
    if (b.v[0] == 0xFFFFFFFFu && b.v[N - 1] == 7u) out[0] = 1;
}
__global__ void busy_kernel(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <int N>
static void run(hipStream_t s, unsigned* out, const char* what) {
    Blob<N> b{};
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(busy_kernel, dim3(1), dim3(64), 0, s, 200000ull);     // 2 ms: nothing completes meanwhile
        const int n = 300;
        const double t0 = now_us();
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(sink_kernel<N>, dim3(2048), dim3(256), 0, s, b, out);
        const double t1 = now_us();
        (void)hipStreamSynchronize(s);
        if (rep) printf("%s: %3d bytes of arguments, %.2f us per launch\n", what, (int)sizeof(b) + 8, (t1 - t0) / n);
    }
}

int main() {
    hipStream_t s;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned* out;
    (void)hipMalloc(&out, 64);
    run<2>(s, out, "hipLaunchKernelGGL");
    run<24>(s, out, "hipLaunchKernelGGL");
    run<30>(s, out, "hipLaunchKernelGGL");
    run<38>(s, out, "hipLaunchKernelGGL");
    run<46>(s, out, "hipLaunchKernelGGL");
    run<54>(s, out, "hipLaunchKernelGGL");
    run<62>(s, out, "hipLaunchKernelGGL");
    run<94>(s, out, "hipLaunchKernelGGL");
    run<174>(s, out, "hipLaunchKernelGGL");
    return 0;
}
