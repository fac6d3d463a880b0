// What a launch of the ray cast's SHAPE costs before it does anything (gfx950): 4096 workgroups x 256 threads, 5.6 kB of
// LDS, a 512-byte kernel argument block -- empty, and with only the ray cast's stores (2 x 2 kB per workgroup) -- and the
// move kernel's shape (128 x 256).  HIP events around every launch (an event pair reads ~2.4 us longer than rocprofv3's
// timestamps) and back-to-back launches (wall time / count: the ramp of one launch hides behind the tail of
// the one before).     hipcc --offload-arch=gfx950 -O3 tools/launch_probe.hip -o tools/_build/launch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct Args {
    float* out;
    int n;
    char pad[500];
};

__global__ void empty_kernel(Args a) {
    extern __shared__ float lds[];
    if (a.n < 0) a.out[threadIdx.x] = lds[threadIdx.x];      // never true: keeps the arguments and LDS alive
}

__global__ void store_kernel(Args a) {
    extern __shared__ float lds[];
    float* row = a.out + (size_t)blockIdx.x * 1024;
    row[threadIdx.x] = (float)threadIdx.x;                   // "scan": 512 floats
    row[threadIdx.x + 256] = 1.0f;
    row[threadIdx.x + 512] = (float)blockIdx.x;              // "newest frame": 512 floats
    row[threadIdx.x + 768] = 2.0f;
    if (a.n < 0) a.out[threadIdx.x] = lds[threadIdx.x];
}

template <class K>
static void run(const char* label, K kernel, int blocks, int threads, size_t lds, Args a) {
    const int reps = 300;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds, 0, a);
    (void)hipDeviceSynchronize();
    double each = 0.0;
    for (int i = 0; i < reps; ++i) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds, 0, a);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        each += ms;
    }
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds, 0, a);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-46s %5d x %3d, %5zu B LDS   event pair around one launch %6.2f us   back to back %6.2f us per launch\n", label,
           blocks, threads, lds, each / reps * 1e3, ms / reps * 1e3);
}

int main() {
    Args a{};
    (void)hipMalloc(&a.out, sizeof(float) * 4096 * 1024);
    a.n = 1;
    run("empty, the ray cast's shape", empty_kernel, 4096, 256, 5648, a);
    run("the ray cast's stores only", store_kernel, 4096, 256, 5648, a);
    run("empty, 8192 workgroups", empty_kernel, 8192, 256, 5648, a);
    run("empty, the move kernel's shape", empty_kernel, 128, 256, 20000, a);
    run("empty, one workgroup", empty_kernel, 1, 64, 0, a);
    return 0;
}
