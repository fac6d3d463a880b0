// The boundary between two DEPENDENT kernel dispatches on one stream of a gfx950, measured on the device itself.
//
// Round 3 divided the wall time of 300 eager launches by 300 and called the 4.8 us "the launch floor"; that figure mixes
// the HOST's submission rate with the device-side turn-around.  Here every workgroup of every launch stamps the constant
// 100 MHz counter (wall_clock64 = s_memrealtime) on entry and on exit into its launch's slot (atomicMin / atomicMax), so
//     duration_k = last exit_k - first entry_k        gap_k = first entry_{k+1} - last exit_k
// are what the DEVICE saw, whoever submitted the work and however fast: (i) eager launches, (ii) the same launches
// captured once into a hipGraph and replayed, (iii) the tick's shape -- a 128-workgroup launch followed by a
// 4096-workgroup launch, each spinning for a given time -- eager and as a graph.  Resolution 10 ns.
//
//     hipcc --offload-arch=gfx950 -O3 tools/launch_boundary.hip -o tools/_build/launch_boundary
//     tools/_build/launch_boundary                 (and under rocprofv3 --kernel-trace: tools/trace_gaps.py reads the csv)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                             \
            return 1;                                                                           \
        }                                                                                       \
    } while (0)

struct Args {
    unsigned long long* first_entry;   // [launches]
    unsigned long long* last_exit;     // [launches]
    int slot;
    int spin_ticks;                    // busy time of every workgroup in 10 ns units (0 = empty kernel)
    char pad[480];                     // the env kernels carry a ~512-byte argument block
};

__global__ void stamp_kernel(Args a) {
    extern __shared__ float lds[];
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) atomicMin(&a.first_entry[a.slot], t0);
    if (a.spin_ticks > 0)
        while (wall_clock64() - t0 < (unsigned long long)a.spin_ticks) __builtin_amdgcn_s_sleep(1);
    if (a.spin_ticks < 0) lds[threadIdx.x] = 1.0f;        // never: keeps the LDS allocation alive
    if (threadIdx.x == 0) atomicMax(&a.last_exit[a.slot], wall_clock64());
}

struct Shape {
    int blocks, threads;
    size_t lds;
    int spin;
};

static double median(std::vector<double> v) {
    std::sort(v.begin(), v.end());
    return v.empty() ? 0.0 : v[v.size() / 2];
}

// `pattern` is repeated `reps` times: launches = reps * pattern.size()
static int run(const char* label, const std::vector<Shape>& pattern, int reps, bool as_graph, hipStream_t s,
               unsigned long long* d_first, unsigned long long* d_last) {
    const int L = reps * (int)pattern.size();
    std::vector<unsigned long long> ones(L, ~0ull), zeros(L, 0ull), first(L), last(L);
    auto issue = [&]() {
        for (int k = 0; k < L; ++k) {
            const Shape& sh = pattern[k % pattern.size()];
            Args a{};
            a.first_entry = d_first;
            a.last_exit = d_last;
            a.slot = k;
            a.spin_ticks = sh.spin;
            hipLaunchKernelGGL(stamp_kernel, dim3(sh.blocks), dim3(sh.threads), sh.lds, s, a);
        }
    };
    hipGraphExec_t exec = nullptr;
    if (as_graph) {
        hipGraph_t g;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        issue();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
    }
    double wall_us = 0.0;
    for (int pass = 0; pass < 4; ++pass) {      // passes 0-2 warm up, pass 3 is reported
        CK(hipMemcpy(d_first, ones.data(), L * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_last, zeros.data(), L * 8, hipMemcpyHostToDevice));
        CK(hipStreamSynchronize(s));
        const auto t0 = std::chrono::steady_clock::now();
        if (as_graph) CK(hipGraphLaunch(exec, s));
        else issue();
        CK(hipStreamSynchronize(s));
        wall_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    CK(hipMemcpy(first.data(), d_first, L * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(last.data(), d_last, L * 8, hipMemcpyDeviceToHost));
    const size_t P = pattern.size();
    std::vector<std::vector<double>> dur(P), gap(P);
    for (int k = 8; k < L; ++k) {               // the first launches of a burst see a cold queue
        dur[k % P].push_back((double)(last[k] - first[k]) * 0.01);
        if (k + 1 < L) gap[k % P].push_back((double)((long long)(first[k + 1] - last[k])) * 0.01);
    }
    printf("%-58s %-6s %4d launches  host wall %7.2f us per launch   device span %7.2f us per launch\n", label,
           as_graph ? "graph" : "eager", L, wall_us / L, (double)(last[L - 1] - first[8]) * 0.01 / (L - 8));
    for (size_t p = 0; p < P; ++p)
        printf("    launch %zu of the pattern (%5d x %3d, spin %5.2f us): on-device duration median %6.2f us, gap to the NEXT "
               "launch's first workgroup median %6.2f us (min %6.2f, max %6.2f)\n",
               p, pattern[p].blocks, pattern[p].threads, pattern[p].spin * 0.01, median(dur[p]), median(gap[p]),
               gap[p].empty() ? 0.0 : *std::min_element(gap[p].begin(), gap[p].end()),
               gap[p].empty() ? 0.0 : *std::max_element(gap[p].begin(), gap[p].end()));
    if (exec) CK(hipGraphExecDestroy(exec));
    return 0;
}

int main() {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    unsigned long long *d_first, *d_last;
    CK(hipMalloc(&d_first, 8 * 4096));
    CK(hipMalloc(&d_last, 8 * 4096));
    const Shape ray{4096, 256, 5648, 0}, move{128, 256, 20000, 0}, one{1, 64, 0, 0};
    for (int g = 0; g < 2; ++g) {
        if (run("empty, the ray cast's shape", {ray}, 300, g, s, d_first, d_last)) return 1;
        if (run("empty, the move kernel's shape", {move}, 300, g, s, d_first, d_last)) return 1;
        if (run("empty, one workgroup", {one}, 300, g, s, d_first, d_last)) return 1;
        if (run("the tick's shape, both empty", {move, ray}, 150, g, s, d_first, d_last)) return 1;
        // the tick's shape with busy workgroups: every move workgroup 6 us, every ray-cast workgroup 8 us (two residency
        // rounds of 2048 workgroups -> ~16 us + ramp and tail): what a boundary costs BETWEEN REAL KERNELS
        if (run("the tick's shape, move 6 us / ray-cast workgroups 8 us",
                {Shape{128, 256, 20000, 600}, Shape{4096, 256, 5648, 800}}, 150, g, s, d_first, d_last))
            return 1;
    }
    return 0;
}
