// The boundary between two DEPENDENT kernel dispatches on one stream of a gfx950, measured on the device itself.
//
// Round 3 divided the wall time of 300 eager launches by 300 and called the 4.8 us "the launch floor"; that figure mixes
// the HOST's submission rate with the device-side turn-around.  Here every workgroup of every launch stamps the constant
// 100 MHz counter (wall_clock64 = s_memrealtime) on entry and on exit into a slot of its own, so
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

constexpr int kMaxBlocks = 4096;
struct Stamp {
    unsigned long long entry, exit;    // of the workgroup's first wave, 10 ns units
    unsigned int hw_id, xcc_id;        // where it ran (HW_REG_HW_ID, HW_REG_XCC_ID)
    unsigned int pad[2];
};

struct Args {
    Stamp* stamps;                     // [launches][kMaxBlocks]: one slot per workgroup (a shared min / max word made
                                       // 8192 same-address atomics per launch: 46 us of drain that looked like a "gap")
    int slot;
    int spin_ticks;                    // busy time of every workgroup in 10 ns units (0 = empty kernel)
    char pad[488];                     // the env kernels carry a ~512-byte argument block
};

__global__ void stamp_kernel(Args a) {
    extern __shared__ float lds[];
    const unsigned long long t0 = wall_clock64();
    if (a.spin_ticks > 0)
        while (wall_clock64() - t0 < (unsigned long long)a.spin_ticks) __builtin_amdgcn_s_sleep(1);
    if (a.spin_ticks < 0) lds[threadIdx.x] = 1.0f;        // never: keeps the LDS allocation alive
    if (threadIdx.x == 0) {
        Stamp st;
        st.entry = t0;
        st.exit = wall_clock64();
        st.hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID: wave / simd / cu / sh / se
        st.xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_XCC_ID
        st.pad[0] = st.pad[1] = 0;
        a.stamps[(size_t)a.slot * kMaxBlocks + blockIdx.x] = st;
    }
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
               Stamp* d_stamps, bool census = false) {
    const int L = reps * (int)pattern.size();
    std::vector<unsigned long long> first(L), last(L);
    std::vector<Stamp> host((size_t)L * kMaxBlocks);
    auto issue = [&]() {
        for (int k = 0; k < L; ++k) {
            const Shape& sh = pattern[k % pattern.size()];
            Args a{};
            a.stamps = d_stamps;
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
        CK(hipStreamSynchronize(s));
        const auto t0 = std::chrono::steady_clock::now();
        if (as_graph) CK(hipGraphLaunch(exec, s));
        else issue();
        CK(hipStreamSynchronize(s));
        wall_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    CK(hipMemcpy(host.data(), d_stamps, host.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
    for (int k = 0; k < L; ++k) {
        const int nb = pattern[k % pattern.size()].blocks;
        first[k] = ~0ull;
        last[k] = 0;
        for (int b = 0; b < nb; ++b) {
            const Stamp& st = host[(size_t)k * kMaxBlocks + b];
            first[k] = std::min(first[k], st.entry);
            last[k] = std::max(last[k], st.exit);
        }
    }
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
    if (census) {
        // the LAST launch of the pattern's last shape: when did its workgroups start, how many were resident at once,
        // how long did each live, on how many CUs
        const int k = L - 1, nb = pattern[k % P].blocks;
        const Stamp* st = &host[(size_t)k * kMaxBlocks];
        std::vector<double> starts, lives;
        std::vector<std::pair<double, int>> ev;
        std::vector<unsigned> cus;
        for (int b = 0; b < nb; ++b) {
            starts.push_back((double)(st[b].entry - first[k]) * 0.01);
            lives.push_back((double)(st[b].exit - st[b].entry) * 0.01);
            ev.push_back({(double)(st[b].entry - first[k]) * 0.01, +1});
            ev.push_back({(double)(st[b].exit - first[k]) * 0.01, -1});
            // HW_ID: [3:0] wave, [5:4] simd, [7:6] pipe, [11:8] cu, [12] sh, [15:13] se (gfx9); XCC_ID [3:0]
            cus.push_back(((st[b].xcc_id & 15u) << 8) | (((st[b].hw_id >> 13) & 7u) << 5) | (((st[b].hw_id >> 12) & 1u) << 4) |
                          ((st[b].hw_id >> 8) & 15u));
        }
        std::sort(ev.begin(), ev.end());
        int cur = 0, peak = 0;
        for (auto& e : ev) {
            cur += e.second;
            peak = std::max(peak, cur);
        }
        std::sort(cus.begin(), cus.end());
        const size_t ncu = std::unique(cus.begin(), cus.end()) - cus.begin();
        std::vector<double> ss = starts;
        std::sort(ss.begin(), ss.end());
        printf("    census of the last launch (%d workgroups): on %zu distinct CUs, at most %d resident at once; workgroup life "
               "median %.2f us; workgroup k started by: 10%% %.2f us, 25%% %.2f, 50%% %.2f, 75%% %.2f, 100%% %.2f; launch %.2f us\n",
               nb, ncu, peak, median(lives), ss[nb / 10], ss[nb / 4], ss[nb / 2], ss[3 * nb / 4], ss[nb - 1],
               (double)(last[k] - first[k]) * 0.01);
    }
    if (exec) CK(hipGraphExecDestroy(exec));
    return 0;
}

int main() {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    Stamp* d_stamps;
    CK(hipMalloc(&d_stamps, sizeof(Stamp) * 300 * (size_t)kMaxBlocks));
    const Shape ray{4096, 256, 5648, 0}, move{128, 256, 20000, 0}, one{1, 64, 0, 0};
    for (int g = 0; g < 2; ++g) {
        if (run("empty, the ray cast's shape", {ray}, 300, g, s, d_stamps)) return 1;
        if (run("empty, the move kernel's shape", {move}, 300, g, s, d_stamps)) return 1;
        if (run("empty, one workgroup", {one}, 300, g, s, d_stamps)) return 1;
        if (run("the tick's shape, both empty", {move, ray}, 150, g, s, d_stamps)) return 1;
        // the tick's shape with busy workgroups: every move workgroup 6 us, every ray-cast workgroup 8 us (two residency
        // rounds of 2048 workgroups -> ~16 us + ramp and tail): what a boundary costs BETWEEN REAL KERNELS
        if (run("the tick's shape, move 6 us / ray-cast workgroups 8 us",
                {Shape{128, 256, 20000, 600}, Shape{4096, 256, 5648, 800}}, 150, g, s, d_stamps, true))
            return 1;
    }
    // residency / dispatch census: the same total work in other launch shapes (each workgroup busy 8 us per 256 threads)
    if (run("census: 4096 x 256 threads, 8 us", {Shape{4096, 256, 5648, 800}}, 20, true, s, d_stamps, true)) return 1;
    if (run("census: 2048 x 256 threads, 8 us", {Shape{2048, 256, 5648, 800}}, 20, true, s, d_stamps, true)) return 1;
    if (run("census: 2048 x 512 threads, 8 us", {Shape{2048, 512, 11296, 800}}, 20, true, s, d_stamps, true)) return 1;
    if (run("census: 1024 x 1024 threads, 8 us", {Shape{1024, 1024, 22592, 800}}, 20, true, s, d_stamps, true)) return 1;
    if (run("census: 4096 x 128 threads, 8 us", {Shape{4096, 128, 5648, 800}}, 20, true, s, d_stamps, true)) return 1;
    if (run("census: 4096 x 256 threads, 2 us", {Shape{4096, 256, 5648, 200}}, 20, true, s, d_stamps, true)) return 1;
    if (run("census: 4096 x 256 threads, empty", {Shape{4096, 256, 5648, 0}}, 20, true, s, d_stamps, true)) return 1;
    return 0;
}
