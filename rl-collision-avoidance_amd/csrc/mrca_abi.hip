// mrca_abi.hip -- host side of include/mrca_env.h: config validation, the SoA device arena,
// table / map upload, kernel sequencing and optional HIP-event timing.  No torch, no Python.
#include <hip/hip_runtime.h>
#include <string>
#include <chrono>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/mrca_env.h"
#include "mrca_host.h"
#include "mrca_hostutil.h"
#include "mrca_kernels.h"
#include "mrca_rollout_store.h"

namespace {
thread_local char g_err[512] = "";
}

namespace mrca {
int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace mrca

namespace {

using mrca::DeviceGuard;
#define fail(...) mrca::set_error(__VA_ARGS__)

// log2 of the beams a marching thread of the ray cast owns (EnvView::ray_shift): 2 per thread, 4 in worlds of more than 64
// robots -- as long as that leaves the workgroup two whole wavefronts or more (the measurements: mrca_create)
// (the marching threads of a workgroup are whole wavefronts -- beams >> shift is a multiple of 64: a wave's ballot is one word of
// MRCA_F_HIT_BITS)
inline int32_t product_ray_shift(int32_t beams, int32_t big) {
    if (big && (beams >> 2) >= 128 && (beams >> 2) % 64 == 0) return 2;
    return (beams >= 256 && (beams >> 1) % 64 == 0) ? 1 : 0;
}

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) return fail(MRCA_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

constexpr size_t kAlign = 256;
size_t align_up(size_t v) { return (v + kAlign - 1) / kAlign * kAlign; }

struct Layout {
    size_t field_off[MRCA_F_COUNT];
    size_t field_bytes[MRCA_F_COUNT];
    size_t off_reset_mode, off_goal_mode, off_group_id, off_init_table, off_goal_table;
    size_t off_beam_cos, off_beam_sin, off_map, off_free_rect, off_cellfield, off_head, off_outline;
    // big worlds (robots_per_world > 64) only
    size_t off_bw_ticket, off_bw_prov, off_bw_state, off_bw_chead, off_bw_cnext, off_bw_lstart, off_bw_lcount, off_bw_lsorted, off_bw_lblock, off_bw_lcursor;
    size_t off_status;
    int32_t bw_cmask, bw_lmask;
    size_t total;
};

int validate(const mrca_config* c) {
    if (!c) return fail(MRCA_ERR_INVALID, "config is NULL");
    if (c->abi_version != MRCA_ABI_VERSION)
        return fail(MRCA_ERR_INVALID, "abi_version %d != %d", c->abi_version, MRCA_ABI_VERSION);
    if (c->num_worlds < 1) return fail(MRCA_ERR_INVALID, "num_worlds must be >= 1");
    if (c->robots_per_world < 1) return fail(MRCA_ERR_INVALID, "robots_per_world must be >= 1");
    // more than 64 robots per world: the per-robot-thread path with a per-tick broad phase (bw_* kernels).  Its
    // episodes restart per robot or not at all; group-synchronous episodes (Stage-2) need the one-wave-per-world path
    if (c->robots_per_world > 64 && c->auto_reset == MRCA_AUTO_GROUP)
        return fail(MRCA_ERR_UNSUPPORTED, "robots_per_world %d > 64 with group-synchronous episodes (auto_reset 2)",
                    c->robots_per_world);
    if (!(c->collision_raster >= 0.0f)) return fail(MRCA_ERR_INVALID, "collision_raster must be >= 0");
    if (c->collision_raster > 0.0f && (c->collision_raster < 0.1f || mrca::outline_span(1.0f / c->collision_raster) > mrca::kOutlineWin))
        return fail(MRCA_ERR_UNSUPPORTED, "collision_raster %.3f m: an outline is kept as an 8 x 8 bitmap of raster cells, which "
                    "holds cells of >= 0.1 m", (double)c->collision_raster);
    if (c->collision_raster > 0.0f && c->robots_per_world > 64)
        return fail(MRCA_ERR_UNSUPPORTED, "collision_raster with robots_per_world > 64");
    if ((int64_t)c->num_worlds * c->robots_per_world > (1 << 24))
        return fail(MRCA_ERR_UNSUPPORTED, "more than 2^24 robots in one environment");
    if (c->beams < 64 || c->beams > 1024 || c->beams % 64)
        return fail(MRCA_ERR_INVALID, "beams %d must be a multiple of 64 in [64,1024]", c->beams);
    if (c->frames < 1 || c->frames > 8) return fail(MRCA_ERR_INVALID, "frames %d out of [1,8]", c->frames);
    if (c->map_width < 1 || c->map_height < 1 || c->map_words_per_row < (c->map_width + 31) / 32)
        return fail(MRCA_ERR_INVALID, "bad map geometry %dx%d wpr %d", c->map_width, c->map_height,
                    c->map_words_per_row);
    // the grid walks keep cell coordinates in fp32 (exact integers, sub-cell estimates good to << 1 cell up to
    // 2^16) and index the per-cell fields with 24-bit multiplies
    if (c->map_width > 16384 || c->map_height > 16384)
        return fail(MRCA_ERR_UNSUPPORTED, "map %dx%d cells: at most 16384 per side", c->map_width, c->map_height);
    if (!(c->map_cell > 0.0f)) return fail(MRCA_ERR_INVALID, "map_cell must be > 0");
    if (!c->map_bits) return fail(MRCA_ERR_INVALID, "map_bits is NULL");
    if (c->auto_reset < 0 || c->auto_reset > 2) return fail(MRCA_ERR_INVALID, "auto_reset %d", c->auto_reset);
    const int R = c->robots_per_world;
    for (int i = 0; i < R; ++i) {
        if (c->reset_mode && (c->reset_mode[i] < 0 || c->reset_mode[i] > 2))
            return fail(MRCA_ERR_INVALID, "reset_mode[%d] = %d", i, c->reset_mode[i]);
        if (c->goal_mode && (c->goal_mode[i] < 0 || c->goal_mode[i] > 2))
            return fail(MRCA_ERR_INVALID, "goal_mode[%d] = %d", i, c->goal_mode[i]);
        if (c->group_id && (c->group_id[i] < 0 || c->group_id[i] > 15))
            return fail(MRCA_ERR_INVALID, "group_id[%d] = %d out of [0,15]", i, c->group_id[i]);
        if (c->reset_mode && c->reset_mode[i] == MRCA_RESET_TABLE && !c->init_table)
            return fail(MRCA_ERR_INVALID, "reset_mode[%d] is TABLE but init_table is NULL", i);
        if (c->goal_mode && c->goal_mode[i] == MRCA_RESET_TABLE && !c->goal_table)
            return fail(MRCA_ERR_INVALID, "goal_mode[%d] is TABLE but goal_table is NULL", i);
    }
    return MRCA_OK;
}

void make_layout(const mrca_config* c, Layout* L) {
    const size_t N = (size_t)c->num_worlds * c->robots_per_world;
    const size_t B = c->beams, F = c->frames, R = c->robots_per_world;
    size_t sz[MRCA_F_COUNT];
    sz[MRCA_F_POSE] = N * 3 * 4;
    sz[MRCA_F_SPEED] = N * 2 * 4;
    sz[MRCA_F_SPEED_GT] = N * 2 * 4;
    sz[MRCA_F_GOAL] = N * 2 * 4;
    sz[MRCA_F_INIT_POSE] = N * 3 * 4;
    sz[MRCA_F_SCAN] = N * B * 4;
    sz[MRCA_F_OBS] = N * F * B * 4;
    sz[MRCA_F_LOCAL_GOAL] = N * 2 * 4;
    sz[MRCA_F_REWARD] = N * 4;
    sz[MRCA_F_DONE] = N;
    sz[MRCA_F_RESULT] = N;
    sz[MRCA_F_FIRST_RESULT] = N;
    sz[MRCA_F_CRASHED] = N;
    sz[MRCA_F_LIVE] = N;
    sz[MRCA_F_FRESH] = N;
    sz[MRCA_F_T] = N * 4;
    sz[MRCA_F_EPISODE] = N * 4;
    sz[MRCA_F_PREV_DIST] = N * 4;
    sz[MRCA_F_SCAN_RING] = N * F * B * 4;
    sz[MRCA_F_RING_HEAD] = N;
    sz[MRCA_F_HIT_BITS] = N * F * (B / 64) * 8;
    size_t off = 0;
    for (int f = 0; f < MRCA_F_COUNT; ++f) {
        L->field_off[f] = off;
        L->field_bytes[f] = sz[f];
        off += align_up(sz[f]);
    }
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += align_up(bytes);
        return o;
    };
    L->off_reset_mode = take(R * 4);
    L->off_goal_mode = take(R * 4);
    L->off_group_id = take(R * 4);
    L->off_init_table = take(R * 3 * 4);
    L->off_goal_table = take(R * 2 * 4);
    L->off_beam_cos = take(B * 4);
    L->off_beam_sin = take(B * 4);
    L->off_map = take((size_t)c->map_height * c->map_words_per_row * 4);
    L->off_free_rect = take((size_t)(c->map_width + 2 * mrca::kFieldPadX) * (c->map_height + 2 * mrca::kFieldPadY) *
                            4 * sizeof(uint16_t));
    L->off_cellfield = take((size_t)c->map_width * c->map_height);
    L->off_head = take(N * sizeof(float4));
    L->off_outline = take(c->collision_raster > 0.0f ? N * sizeof(mrca::OutlineBits) : 0);   // fidelity mode only
    L->bw_cmask = L->bw_lmask = 0;
    if (c->robots_per_world > 64) {
        size_t mc = 1, ml = 1;
        while (mc < 4 * N) mc <<= 1;      // 2N entries (pose at tick start + provisional pose): load factor <= 0.5
        while (ml < 2 * N) ml <<= 1;
        L->bw_cmask = (int32_t)(mc - 1);
        L->bw_lmask = (int32_t)(ml - 1);
        L->off_bw_ticket = take(4);
        L->off_bw_prov = take(N * 2 * sizeof(float4));
        L->off_bw_state = take(N * 4);
        L->off_bw_chead = take(mc * 4);
        L->off_bw_cnext = take(2 * N * 4);
        L->off_bw_lstart = take((ml + 2) * 4);
        L->off_bw_lcount = take((ml + 1) * 4);
        L->off_bw_lsorted = take(N * 4);
        L->off_bw_lblock = take((ml / 1024 + 2) * 4);
        L->off_bw_lcursor = take((ml + 1) * 4);
    }
    L->off_status = take(4);
    L->total = off;
}

constexpr int kTimingRing = 1024;

// The free-rectangle field depends on the map only and takes ~1 s of host time for an 800 x 800 map:
// environments created on the same map in one process (one per rank thread, per test, per bench leg)
// share one host copy.  Keyed by the bitmap itself.
struct HostField {
    std::vector<uint32_t> bits;
    int32_t width, height, wpr;
    std::vector<uint16_t> entries;      // 4 quadrant entries per cell
    int pitch;
};
std::shared_ptr<const HostField> host_field(const mrca_config* c) {
    static std::mutex mu;
    static std::vector<std::shared_ptr<const HostField>> cache;
    const size_t words = (size_t)c->map_height * c->map_words_per_row;
    std::lock_guard<std::mutex> lock(mu);
    for (const auto& f : cache)
        if (f->width == c->map_width && f->height == c->map_height && f->wpr == c->map_words_per_row &&
            std::memcmp(f->bits.data(), c->map_bits, words * 4) == 0)
            return f;
    auto f = std::make_shared<HostField>();
    f->bits.assign(c->map_bits, c->map_bits + words);
    f->width = c->map_width;
    f->height = c->map_height;
    f->wpr = c->map_words_per_row;
    mrca::build_free_rect_field(c->map_bits, c->map_width, c->map_height, c->map_words_per_row, &f->entries, &f->pitch);
    if (cache.size() >= 4) cache.erase(cache.begin());
    cache.push_back(f);
    return f;
}

}  // namespace

struct mrca_env {
    mrca_config cfg;
    Layout layout;
    char* arena = nullptr;
    bool owns_arena = false;
    mrca::EnvView view;
    mrca::EnvView* view_dev = nullptr;    // `view` in device memory (outside the arena): what move_kernel / raycast_kernel read
    size_t lds_bytes = 0;
    // timing
    int timing = 0;      // 0 off, n > 0: record events on every n-th step
    int step_count = 0;
    std::vector<hipEvent_t> ev;  // 4 per recorded step: begin / end of the move launch, begin / end of the ray cast
    int ev_used = 0;
    int last_ray_count = 0;   // workgroups of the last ray-cast launch (profiling build: whose stamps are current)
    // mrca_step_many with chains > 1: the streams of world ranges 1 .. P-1 and the events that order them against the caller's
    std::vector<hipStream_t> chain_stream;
    std::vector<hipEvent_t> chain_moved, chain_done;
    hipEvent_t chain_fork = nullptr;
    // mrca_step_many's run-ahead schedule: the move launches of a call's ticks run on a stream of their own,
    // AHEAD of the ray casts, each tick writing the five things a ray cast reads of a move launch -- pose, head record, goal,
    // fresh flag, outline -- into a slot of its own, so that no move launch ever waits for a ray cast.  Slot 0 is the env's
    // own fields (where a call starts and where its last tick ends); slots 1 .. ahead_slots live in `ahead_mem`.
    char* ahead_mem = nullptr;
    size_t ahead_bytes = 0;                 // per slot
    size_t ahead_off[5] = {0, 0, 0, 0, 0};  // pose, head, goal, fresh, outline inside a slot
    int ahead_slots = 0;                    // 0: the ring could not be allocated -> the chained schedule
    hipStream_t move_stream = nullptr;
    std::vector<hipEvent_t> moved;          // [kAheadTicks] "tick k's move launch is through"
    // stream choice (choose_streams below): the caller's stream the env's streams were last checked against, how many
    // ranges that check covered, the streams found to share a hardware queue with another one (parked until mrca_destroy)
    std::vector<hipStream_t> checked_against;    // caller's streams the env's CURRENT streams have been checked against
    int checked_ranges = 0;
    std::vector<hipStream_t> parked;
    unsigned long long* probe_stamps = nullptr;   // [4] device: start / end of the two probe kernels
};

constexpr int kAheadTicks = 256;            // most ticks one run-ahead pass covers (a pass ends with every stream joined: ~90 us)
constexpr size_t kAheadMaxBytes = 256u << 20;
// World range 1 gets its stream in mrca_create (chains <= 2, bench.py's default, may be captured at once); further ranges get
// theirs at their first use -- an env does not park streams it may never use (the runtime maps a process's streams onto a
// few hardware queues, DESIGN.md 5.10 "what the schedule depends on").
constexpr int kChainStreamsAtCreate = 1;

// the env's view with slot b's buffers in place of the five fields (b = 0: the env's own)
static mrca::EnvView slot_view(const mrca_env* env, int b) {
    mrca::EnvView v = env->view;
    if (b > 0) {
        char* base = env->ahead_mem + (size_t)(b - 1) * env->ahead_bytes;
        v.pose = reinterpret_cast<float*>(base + env->ahead_off[0]);
        v.head = reinterpret_cast<float4*>(base + env->ahead_off[1]);
        v.goal = reinterpret_cast<float*>(base + env->ahead_off[2]);
        v.fresh = reinterpret_cast<uint8_t*>(base + env->ahead_off[3]);
        if (env->view.outline) v.outline = reinterpret_cast<mrca::OutlineBits*>(base + env->ahead_off[4]);
    }
    return v;
}

// env->view -> its device copy (mrca_create; the profiling build's debug switches).  Synchronous: no launch that reads the
// copy is in flight at either place.
static hipError_t upload_view(mrca_env* env) {
    hipError_t e = hipSuccess;
    if (!env->view_dev) e = hipMalloc(reinterpret_cast<void**>(&env->view_dev), sizeof(mrca::EnvView));
    if (e != hipSuccess) return e;
    env->view.dev = env->view_dev;
    if ((e = hipDeviceSynchronize()) != hipSuccess) return e;
    return hipMemcpy(env->view_dev, &env->view, sizeof(mrca::EnvView), hipMemcpyHostToDevice);
}

// the streams, events and the run-ahead ring an env owns beside its arena (mrca_destroy, and mrca_create when it gives up)
static void release_side_objects(mrca_env* env) {
    for (hipEvent_t e : env->ev) (void)hipEventDestroy(e);
    for (hipStream_t s : env->chain_stream) {
        (void)hipStreamSynchronize(s);
        (void)hipStreamDestroy(s);
    }
    if (env->move_stream) {
        (void)hipStreamSynchronize(env->move_stream);
        (void)hipStreamDestroy(env->move_stream);
    }
    for (hipEvent_t e : env->moved)
        if (e) (void)hipEventDestroy(e);
    if (env->ahead_mem) (void)hipFree(env->ahead_mem);
    if (env->view_dev) (void)hipFree(env->view_dev);
    env->view_dev = nullptr;
    env->view.dev = nullptr;
    if (env->probe_stamps) (void)hipFree(env->probe_stamps);
    env->probe_stamps = nullptr;
    for (hipStream_t s : env->parked) (void)hipStreamDestroy(s);
    env->parked.clear();
    for (hipEvent_t e : env->chain_moved) (void)hipEventDestroy(e);
    for (hipEvent_t e : env->chain_done) (void)hipEventDestroy(e);
    if (env->chain_fork) (void)hipEventDestroy(env->chain_fork);
    env->ev.clear();
    env->chain_stream.clear();
    env->moved.clear();
    env->chain_moved.clear();
    env->chain_done.clear();
    env->move_stream = nullptr;
    env->ahead_mem = nullptr;
    env->chain_fork = nullptr;
}

extern "C" {

int mrca_abi_version(void) { return MRCA_ABI_VERSION; }

const char* mrca_last_error(void) { return g_err; }

int mrca_arena_bytes(const mrca_config* cfg, size_t* bytes_out) {
    if (int rc = validate(cfg)) return rc;
    if (!bytes_out) return fail(MRCA_ERR_INVALID, "bytes_out is NULL");
    Layout L;
    make_layout(cfg, &L);
    *bytes_out = L.total;
    return MRCA_OK;
}

int mrca_create(const mrca_config* cfg, void* arena_dev, size_t arena_bytes, mrca_env** env_out) {
    if (int rc = validate(cfg)) return rc;
    if (!env_out) return fail(MRCA_ERR_INVALID, "env_out is NULL");
    *env_out = nullptr;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(MRCA_ERR_INVALID, "device %d not in [0,%d)", cfg->device, ndev);
    HIP_TRY(hipSetDevice(cfg->device));

    mrca_env* env = new (std::nothrow) mrca_env();
    if (!env) return fail(MRCA_ERR_NOMEM, "host allocation failed");
    env->cfg = *cfg;
    make_layout(cfg, &env->layout);
    const Layout& L = env->layout;
    if (arena_dev) {
        if (arena_bytes < L.total) {
            delete env;
            return fail(MRCA_ERR_NOMEM, "arena of %zu bytes < required %zu", arena_bytes, L.total);
        }
        if (reinterpret_cast<uintptr_t>(arena_dev) % kAlign) {
            delete env;
            return fail(MRCA_ERR_INVALID, "arena must be %zu-byte aligned", kAlign);
        }
        env->arena = static_cast<char*>(arena_dev);
    } else {
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&env->arena), L.total);
        if (e != hipSuccess) {
            delete env;
            return fail(MRCA_ERR_NOMEM, "hipMalloc(%zu) failed: %s", L.total, hipGetErrorString(e));
        }
        env->owns_arena = true;
    }
    auto bail = [&](int rc) {
        release_side_objects(env);
        if (env->owns_arena) (void)hipFree(env->arena);
        delete env;
        return rc;
    };
#define HIP_TRY_BAIL(expr)                                                                        \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return bail(fail(MRCA_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)));       \
    } while (0)

    HIP_TRY_BAIL(hipMemset(env->arena, 0, L.total));

    const int R = cfg->robots_per_world, B = cfg->beams;
    const size_t N = (size_t)cfg->num_worlds * R;
    // scenario tables
    std::vector<int32_t> reset_mode(R, MRCA_RESET_DISC), goal_mode(R, MRCA_RESET_DISC), group(R, 0);
    std::vector<float> init_tab(R * 3, 0.0f), goal_tab(R * 2, 0.0f);
    if (cfg->reset_mode) memcpy(reset_mode.data(), cfg->reset_mode, R * 4);
    if (cfg->goal_mode) memcpy(goal_mode.data(), cfg->goal_mode, R * 4);
    else if (cfg->reset_mode) goal_mode = reset_mode;
    if (cfg->group_id) memcpy(group.data(), cfg->group_id, R * 4);
    if (cfg->init_table) memcpy(init_tab.data(), cfg->init_table, R * 3 * 4);
    if (cfg->goal_table) memcpy(goal_tab.data(), cfg->goal_table, R * 2 * 4);
    int num_groups = 0;
    for (int i = 0; i < R; ++i) num_groups = group[i] + 1 > num_groups ? group[i] + 1 : num_groups;
    // beam directions: bearing_i = -pi/2 + i*pi/(B-1) (stageros.cpp:495-497), evaluated in
    // double, rounded once to fp32
    std::vector<float> bcos(B), bsin(B);
    for (int i = 0; i < B; ++i) {
        const double b = -M_PI / 2.0 + (double)i * (M_PI / (double)(B - 1));
        bcos[i] = (float)std::cos(b);
        bsin[i] = (float)std::sin(b);
    }
    HIP_TRY_BAIL(hipMemcpy(env->arena + L.off_reset_mode, reset_mode.data(), R * 4, hipMemcpyHostToDevice));
    HIP_TRY_BAIL(hipMemcpy(env->arena + L.off_goal_mode, goal_mode.data(), R * 4, hipMemcpyHostToDevice));
    HIP_TRY_BAIL(hipMemcpy(env->arena + L.off_group_id, group.data(), R * 4, hipMemcpyHostToDevice));
    HIP_TRY_BAIL(hipMemcpy(env->arena + L.off_init_table, init_tab.data(), R * 3 * 4, hipMemcpyHostToDevice));
    HIP_TRY_BAIL(hipMemcpy(env->arena + L.off_goal_table, goal_tab.data(), R * 2 * 4, hipMemcpyHostToDevice));
    {
        // before the first reset the robots stand at their table poses (= the agent lines of the world file): a start
        // sampled in Stage-2's region keeps 7 m away from the robot's CURRENT position, its first one included
        // (stage_world2.py:250-268)
        std::vector<float> pose0((size_t)N * 3);
        for (size_t n = 0; n < N; ++n)
            for (int k = 0; k < 3; ++k) pose0[n * 3 + k] = init_tab[(n % (size_t)R) * 3 + k];
        HIP_TRY_BAIL(hipMemcpy(env->arena + L.field_off[MRCA_F_POSE], pose0.data(), pose0.size() * 4, hipMemcpyHostToDevice));
    }
    HIP_TRY_BAIL(hipMemcpy(env->arena + L.off_beam_cos, bcos.data(), B * 4, hipMemcpyHostToDevice));
    HIP_TRY_BAIL(hipMemcpy(env->arena + L.off_beam_sin, bsin.data(), B * 4, hipMemcpyHostToDevice));
    HIP_TRY_BAIL(hipMemcpy(env->arena + L.off_map, cfg->map_bits,
                           (size_t)cfg->map_height * cfg->map_words_per_row * 4, hipMemcpyHostToDevice));
    int free_rect_pitch = 0;
    {
        const std::shared_ptr<const HostField> hf = host_field(cfg);
        free_rect_pitch = hf->pitch;
        HIP_TRY_BAIL(hipMemcpy(env->arena + L.off_free_rect, hf->entries.data(), hf->entries.size() * sizeof(uint16_t),
                               hipMemcpyHostToDevice));
    }
    {
        std::vector<uint8_t> cf;
        mrca::build_cell_field(cfg->map_bits, cfg->map_width, cfg->map_height, cfg->map_words_per_row, &cf);
        HIP_TRY_BAIL(hipMemcpy(env->arena + L.off_cellfield, cf.data(), cf.size(), hipMemcpyHostToDevice));
    }
    if (R > 64)    // the collision hash starts empty; every tick leaves it empty again (bw_finish_kernel)
        HIP_TRY_BAIL(hipMemset(env->arena + L.off_bw_chead, 0xFF, ((size_t)L.bw_cmask + 1) * 4));
    // live = 1, t = 1 at construction (a robot exists and is idle before the first reset)
    HIP_TRY_BAIL(hipMemset(env->arena + L.field_off[MRCA_F_LIVE], 1, N));
    {
        std::vector<int32_t> ones(N, 1);
        HIP_TRY_BAIL(hipMemcpy(env->arena + L.field_off[MRCA_F_T], ones.data(), N * 4, hipMemcpyHostToDevice));
    }
    env->cfg.map_bits = nullptr;  // host pointers are not retained
    env->cfg.reset_mode = env->cfg.goal_mode = env->cfg.group_id = nullptr;
    env->cfg.init_table = env->cfg.goal_table = nullptr;

    mrca::EnvView& v = env->view;
    char* a = env->arena;
    v.N = (int32_t)N;
    v.R = R;
    v.W = cfg->num_worlds;
    v.B = B;
    v.F = cfg->frames;
    v.pose = reinterpret_cast<float*>(a + L.field_off[MRCA_F_POSE]);
    v.speed = reinterpret_cast<float*>(a + L.field_off[MRCA_F_SPEED]);
    v.speed_gt = reinterpret_cast<float*>(a + L.field_off[MRCA_F_SPEED_GT]);
    v.goal = reinterpret_cast<float*>(a + L.field_off[MRCA_F_GOAL]);
    v.init_pose = reinterpret_cast<float*>(a + L.field_off[MRCA_F_INIT_POSE]);
    v.scan = reinterpret_cast<float*>(a + L.field_off[MRCA_F_SCAN]);
    v.obs = reinterpret_cast<float*>(a + L.field_off[MRCA_F_OBS]);
    v.scan_ring = reinterpret_cast<float*>(a + L.field_off[MRCA_F_SCAN_RING]);
    v.ring_head = reinterpret_cast<uint8_t*>(a + L.field_off[MRCA_F_RING_HEAD]);
    v.hit_bits = reinterpret_cast<unsigned long long*>(a + L.field_off[MRCA_F_HIT_BITS]);
    v.local_goal = reinterpret_cast<float*>(a + L.field_off[MRCA_F_LOCAL_GOAL]);
    v.reward = reinterpret_cast<float*>(a + L.field_off[MRCA_F_REWARD]);
    v.prev_dist = reinterpret_cast<float*>(a + L.field_off[MRCA_F_PREV_DIST]);
    v.done = reinterpret_cast<uint8_t*>(a + L.field_off[MRCA_F_DONE]);
    v.result = reinterpret_cast<uint8_t*>(a + L.field_off[MRCA_F_RESULT]);
    v.first_result = reinterpret_cast<uint8_t*>(a + L.field_off[MRCA_F_FIRST_RESULT]);
    v.crashed = reinterpret_cast<uint8_t*>(a + L.field_off[MRCA_F_CRASHED]);
    v.live = reinterpret_cast<uint8_t*>(a + L.field_off[MRCA_F_LIVE]);
    v.fresh = reinterpret_cast<uint8_t*>(a + L.field_off[MRCA_F_FRESH]);
    v.t = reinterpret_cast<int32_t*>(a + L.field_off[MRCA_F_T]);
    v.episode = reinterpret_cast<int32_t*>(a + L.field_off[MRCA_F_EPISODE]);
    v.reset_mode = reinterpret_cast<const int32_t*>(a + L.off_reset_mode);
    v.goal_mode = reinterpret_cast<const int32_t*>(a + L.off_goal_mode);
    v.group_id = reinterpret_cast<const int32_t*>(a + L.off_group_id);
    v.init_table = reinterpret_cast<const float*>(a + L.off_init_table);
    v.goal_table = reinterpret_cast<const float*>(a + L.off_goal_table);
    v.beam_cos = reinterpret_cast<const float*>(a + L.off_beam_cos);
    v.beam_sin = reinterpret_cast<const float*>(a + L.off_beam_sin);
    v.beam_step = mrca::kPi / (float)(B - 1);
    v.beam_inv_step = (float)(B - 1) / mrca::kPi;
    v.map_bits = reinterpret_cast<const uint32_t*>(a + L.off_map);
    v.free_rect = reinterpret_cast<const uint16_t*>(a + L.off_free_rect);
    v.free_rect_pitch = free_rect_pitch;
    v.cellfield = reinterpret_cast<const uint8_t*>(a + L.off_cellfield);
    v.head = reinterpret_cast<float4*>(a + L.off_head);
    v.outline = cfg->collision_raster > 0.0f ? reinterpret_cast<mrca::OutlineBits*>(a + L.off_outline) : nullptr;
    v.big = R > 64 ? 1 : 0;
    if (v.big) {
        v.bw_ticket = reinterpret_cast<uint32_t*>(a + L.off_bw_ticket);
        v.bw_prov = reinterpret_cast<float4*>(a + L.off_bw_prov);
        v.bw_state = reinterpret_cast<int32_t*>(a + L.off_bw_state);
        v.bw_chead = reinterpret_cast<int32_t*>(a + L.off_bw_chead);
        v.bw_cnext = reinterpret_cast<int32_t*>(a + L.off_bw_cnext);
        v.bw_lstart = reinterpret_cast<int32_t*>(a + L.off_bw_lstart);
        v.bw_lcount = reinterpret_cast<int32_t*>(a + L.off_bw_lcount);
        v.bw_lsorted = reinterpret_cast<int32_t*>(a + L.off_bw_lsorted);
        v.bw_lblock = reinterpret_cast<int32_t*>(a + L.off_bw_lblock);
        v.bw_lcursor = reinterpret_cast<int32_t*>(a + L.off_bw_lcursor);
        v.bw_cmask = L.bw_cmask;
        v.bw_lmask = L.bw_lmask;
    }
    v.status = reinterpret_cast<uint32_t*>(a + L.off_status);
    v.ray_first = 0;
    v.ray_count = (int32_t)N;
    v.world_first = 0;
    v.world_count = cfg->num_worlds;
    v.g.x0 = cfg->map_x0;
    v.g.y0 = cfg->map_y0;
    v.g.cell = cfg->map_cell;
    v.g.inv_cell = 1.0f / cfg->map_cell;
    v.g.width = cfg->map_width;
    v.g.height = cfg->map_height;
    v.g.wpr = cfg->map_words_per_row;
    v.timeout = cfg->timeout;
    v.w_thresh = cfg->w_thresh;
    v.pre_dist_zero = cfg->pre_dist_zero;
    v.hold_velocity = cfg->hold_velocity ? 1 : 0;
    v.auto_reset = cfg->auto_reset;
    v.num_groups = num_groups;
    v.key0 = (uint32_t)(cfg->seed & 0xFFFFFFFFull);
    v.key1 = (uint32_t)(cfg->seed >> 32);
    v.foot_hc = (int32_t)std::ceil(0.2907 * (double)v.g.inv_cell) + 1;
    v.edge_slots = mrca::edge_event_slots(v.g.inv_cell);
    v.raster_inv = cfg->collision_raster > 0.0f ? 1.0f / cfg->collision_raster : 0.0f;
    v.raster_res = cfg->collision_raster;
    // the ray cast tests 4 x 4 cells of a neighbour's outline window where that covers every outline (Stage's 0.2 m), else 8 x 8
    v.raster_kw = (cfg->collision_raster > 0.0f && mrca::outline_span(v.raster_inv) <= 4) ? 4 : 8;
    v.lidar_radius = 0.2917f;
    v.lidar_near = 0.30f;
    v.lidar_reach2 = mrca::kLidarReach2;
    if (cfg->collision_raster > 0.0f) {      // outline CELLS reach one cell diagonal beyond the rectangle
        v.lidar_radius = 0.2917f + 1.4143f * cfg->collision_raster;
        v.lidar_near = v.lidar_radius + 0.01f;
        const float reach = 6.0f + v.lidar_radius + 0.01f;
        v.lidar_reach2 = reach * reach;
    }
    {   // broad phase: rectangles further apart than 2 x circumradius cannot overlap; outlines further apart than that
        // plus one raster-cell diagonal cannot share a cell
        const float reach = 2.0f * 0.2907f + 0.001f + (cfg->collision_raster > 0.0f ? 1.4143f * cfg->collision_raster : 0.0f);
        v.collide_reach2 = cfg->collision_raster > 0.0f ? reach * reach : mrca::kCollideReach2;
    }
    v.r_magic = R <= 64 ? (uint32_t)(((1ull << 32) + (uint64_t)R - 1) / (uint64_t)R) : 0u;
    v.debug_flags = 0;
    v.dev = nullptr;                      // (upload_view, at the end of mrca_create)
    v.eager_views = 0;                    // (set per launch by the step paths of an env with lazy_obs = 0)
#if defined(MRCA_PROFILING)
    v.launch_stamps = nullptr;
    v.launch_slot = 0;
#endif
    // Launch shape of the ray cast, measured (profiles/r02/r02_c_ablation_launch_shapes.txt, 4096 / 8228 robots, HIP events):
    //   2 beams per thread one after the other, first wave prepares the neighbours   28.1 / 33.7 us   <- product
    //   1 beam per thread (512 threads), first wave prepares                           31.6 / 41.1 us
    //   2 beams per thread in lock step (two lookups in flight), first wave prepares   34.7 / 37.9 us
    //   the same three with a dedicated fifth preparation wave                          31.5-37.8 / 39.9-48.5 us
    // i.e. neither more lookups in flight per thread nor taking the preparation off the marching waves pays: the
    // lock-step loop costs 62 instead of 54 VALU instructions per jump and keeps finished rays idling, the extra
    // wave costs a resident workgroup per CU.
    // Worlds of more than 64 robots (the chunked neighbour lists of the big-world path), one circle of 50 000, PROFILING build
    // (profiles/r04_m_slice_probe*.txt; full launch / one rank's slice of 6 250 robots):
    //   4 beams per thread one after the other (2 waves per workgroup, 4096 workgroups resident)   445 /  77 us   <- product
    //   2 beams per thread one after the other                                                      536 /  88 us
    //   2 / 4 beams per thread in lock step (rounds 2-3)                                      548, 578 / 88, 94 us
    //   1 beam per thread                                                                           844 / 133 us
    v.ray_shift = product_ray_shift(cfg->beams, v.big);
    v.ray_prep_wave = 0;
    v.ray_sequential = 1;
    env->lds_bytes = mrca::ray_lds_bytes(v);
    if (!v.big && mrca::move_lds_bytes(v) > 64 * 1024)
        return bail(fail(MRCA_ERR_UNSUPPORTED, "map_cell %.4f m is too fine for the LDS patches: use >= 0.01 m",
                         (double)cfg->map_cell));
    if (env->lds_bytes > 160 * 1024)
        return bail(fail(MRCA_ERR_UNSUPPORTED, "the ray cast needs %zu B of LDS per robot (> 160 KiB): too many beams",
                         env->lds_bytes));
    if (!v.big) {
        // the run-ahead ring of mrca_step_many (outside the arena: a caller-provided arena keeps its documented size).  As many
        // slots as fit the budget, at most one per tick of a pass; none (allocation failed) = the chained schedule, no error
        size_t off = 0;
        const size_t part[5] = {N * 3 * 4, N * sizeof(float4), N * 2 * 4, N, v.outline ? N * sizeof(mrca::OutlineBits) : 0};
        for (int i = 0; i < 5; ++i) {
            env->ahead_off[i] = off;
            off += align_up(part[i]);
        }
        env->ahead_bytes = off;
        size_t slots = kAheadMaxBytes / off;
        if (slots > (size_t)kAheadTicks - 1) slots = kAheadTicks - 1;
        if (slots >= 1 && hipMalloc(reinterpret_cast<void**>(&env->ahead_mem), slots * off) == hipSuccess) {
            env->ahead_slots = (int)slots;
        } else {
            (void)hipGetLastError();
            env->ahead_mem = nullptr;
        }
        bool ok = hipStreamCreateWithFlags(&env->move_stream, hipStreamNonBlocking) == hipSuccess;
        env->moved.assign(kAheadTicks, nullptr);
        for (auto& e : env->moved) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        for (int c = 0; c < kChainStreamsAtCreate && ok; ++c) {
            hipStream_t st = nullptr;
            hipEvent_t ea = nullptr, eb = nullptr;
            ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
                 hipEventCreateWithFlags(&ea, hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&eb, hipEventDisableTiming) == hipSuccess;
            if (ok) {
                env->chain_stream.push_back(st);
                env->chain_moved.push_back(ea);
                env->chain_done.push_back(eb);
            }
        }
        if (!ok) {
            (void)hipGetLastError();
            env->ahead_slots = 0;       // (whatever was created is released by mrca_destroy)
        }
    }
    HIP_TRY_BAIL(upload_view(env));       // (the view is final from here on)
    mrca::launch_head_init(v, nullptr);   // head records of the construction-time poses (all at the origin)
    HIP_TRY_BAIL(hipGetLastError());
    HIP_TRY_BAIL(hipDeviceSynchronize());
    *env_out = env;
    return MRCA_OK;
#undef HIP_TRY_BAIL
}

int mrca_destroy(mrca_env* env) {
    if (!env) return MRCA_OK;
    release_side_objects(env);
    if (env->owns_arena) HIP_TRY(hipFree(env->arena));
    delete env;
    return MRCA_OK;
}

int mrca_get_field(mrca_env* env, int field, void** ptr_dev_out, size_t* offset_out, size_t* bytes_out) {
    if (!env) return fail(MRCA_ERR_INVALID, "env is NULL");
    if (field < 0 || field >= MRCA_F_COUNT) return fail(MRCA_ERR_INVALID, "field %d out of range", field);
    if (ptr_dev_out) *ptr_dev_out = env->arena + env->layout.field_off[field];
    if (offset_out) *offset_out = env->layout.field_off[field];
    if (bytes_out) *bytes_out = env->layout.field_bytes[field];
    return MRCA_OK;
}

int mrca_reset(mrca_env* env, const uint8_t* mask_dev, const float* poses_dev, const float* goals_dev, void* stream) {
    if (!env) return fail(MRCA_ERR_INVALID, "env is NULL");
    DeviceGuard guard(env->cfg.device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    mrca::launch_reset(env->view, mask_dev, poses_dev, goals_dev, s);
    mrca::launch_lidar_grid(env->view, /*counted=*/0, s);
    mrca::launch_raycast(env->view, /*only_fresh=*/1, s);
    if (!env->cfg.lazy_obs) mrca::launch_materialize(env->view, MRCA_VIEW_SCAN | MRCA_VIEW_OBS, s);
    HIP_TRY(hipGetLastError());
    return MRCA_OK;
}

int mrca_materialize(mrca_env* env, int32_t what, void* stream) {
    if (!env) return fail(MRCA_ERR_INVALID, "env is NULL");
    if (what & ~(MRCA_VIEW_SCAN | MRCA_VIEW_OBS)) return fail(MRCA_ERR_INVALID, "mrca_materialize: unknown view bits 0x%x", what);
    DeviceGuard guard(env->cfg.device);
    mrca::launch_materialize(env->view, what, static_cast<hipStream_t>(stream));
    HIP_TRY(hipGetLastError());
    return MRCA_OK;
}

int mrca_newest_obs(mrca_env* env, float* out_dev, void* stream) {
    if (!env) return fail(MRCA_ERR_INVALID, "env is NULL");
    if (!out_dev) return fail(MRCA_ERR_INVALID, "out_dev is NULL");
    if (reinterpret_cast<uintptr_t>(out_dev) % 16) return fail(MRCA_ERR_INVALID, "out_dev must be 16-byte aligned");
    DeviceGuard guard(env->cfg.device);
    mrca::launch_newest_obs(env->view, out_dev, static_cast<hipStream_t>(stream));
    HIP_TRY(hipGetLastError());
    return MRCA_OK;
}

enum { kPhaseMove = 1, kPhaseObserve = 2 };
static int step_impl(mrca_env* env, const float* actions_dev, int32_t first, int32_t count, void* stream,
                     int32_t world_first = 0, int32_t world_count = -1, int phases = kPhaseMove | kPhaseObserve) {
    if (!env) return fail(MRCA_ERR_INVALID, "env is NULL");
    if ((phases & kPhaseMove) && !actions_dev) return fail(MRCA_ERR_INVALID, "actions_dev is NULL");
    if (first < 0 || count < 0 || first + count > env->view.N)
        return fail(MRCA_ERR_INVALID, "ray-cast slice [%d, %d) outside [0, %d)", first, first + count, env->view.N);
    if (world_count < 0) world_count = env->view.W;
    DeviceGuard guard(env->cfg.device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool rec = phases == (kPhaseMove | kPhaseObserve) && env->timing > 0 && (env->step_count++ % env->timing) == 0 &&
                     env->ev_used + 4 <= (int)env->ev.size();
    mrca::EnvView v = env->view;
    v.ray_first = first;      // the robots whose lidar outputs (scan, frame stack, local goal) this call produces
    v.ray_count = count;
    v.world_first = world_first;   // the worlds the move launch advances (mrca_step_worlds; otherwise all of them)
    v.world_count = world_count;
    env->last_ray_count = count;
    // timing: the launches' own begin / end stamps (hipExtLaunchKernel), not event records around them
    hipEvent_t* ev = rec ? &env->ev[env->ev_used] : nullptr;
    if (phases & kPhaseMove) mrca::launch_move(v, actions_dev, s, rec ? ev[0] : nullptr, rec ? ev[1] : nullptr);
    if (phases & kPhaseObserve) {
        mrca::launch_lidar_grid(v, /*counted=*/1, s);
        // (lazy_obs = 0: the ray cast forms MRCA_F_SCAN / MRCA_F_OBS of its robots itself -- no materialize launch behind it)
        v.eager_views = env->cfg.lazy_obs ? 0 : (MRCA_VIEW_SCAN | MRCA_VIEW_OBS);
        mrca::launch_raycast(v, /*only_fresh=*/0, s, rec ? ev[2] : nullptr, rec ? ev[3] : nullptr);
    }
    if (rec) env->ev_used += 4;
    HIP_TRY(hipGetLastError());
    return MRCA_OK;
}

int mrca_step(mrca_env* env, const float* actions_dev, void* stream) {
    return step_impl(env, actions_dev, 0, env ? env->view.N : 0, stream);
}

int mrca_step_slice(mrca_env* env, const float* actions_dev, int32_t first_robot, int32_t num_robots, void* stream) {
    return step_impl(env, actions_dev, first_robot, num_robots, stream);
}

static int worlds_impl(mrca_env* env, const float* actions_dev, int32_t first_world, int32_t num_worlds, void* stream, int phases) {
    if (!env) return fail(MRCA_ERR_INVALID, "env is NULL");
    if (first_world < 0 || num_worlds < 0 || first_world + num_worlds > env->view.W)
        return fail(MRCA_ERR_INVALID, "world range [%d, %d) outside [0, %d)", first_world, first_world + num_worlds, env->view.W);
    if (env->view.big && (first_world != 0 || num_worlds != env->view.W || phases != (kPhaseMove | kPhaseObserve)))
        return fail(MRCA_ERR_UNSUPPORTED, "a part of the worlds / of the tick with robots_per_world > 64");
    const int32_t R = env->view.R;
    return step_impl(env, actions_dev, first_world * R, num_worlds * R, stream, first_world, num_worlds, phases);
}

int mrca_step_worlds(mrca_env* env, const float* actions_dev, int32_t first_world, int32_t num_worlds, void* stream) {
    return worlds_impl(env, actions_dev, first_world, num_worlds, stream, kPhaseMove | kPhaseObserve);
}

int mrca_move_worlds(mrca_env* env, const float* actions_dev, int32_t first_world, int32_t num_worlds, void* stream) {
    return worlds_impl(env, actions_dev, first_world, num_worlds, stream, kPhaseMove);
}

int mrca_observe_worlds(mrca_env* env, int32_t first_world, int32_t num_worlds, void* stream) {
    return worlds_impl(env, nullptr, first_world, num_worlds, stream, kPhaseObserve);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Which streams?  The HIP runtime maps a process's streams onto a few hardware queues (four by default, handed out by use
// count), and two streams that share a queue run their kernels ONE AFTER THE OTHER whatever the program says: the schedules
// below then lose what they are built for, silently (tools/queue_alias_probe.hip: of twelve streams every one shares its queue
// with two others; tools/stream_pressure_probe.py: the same mrca_step_many call at 305 M and at 150 M agent-steps/s on the
// Stage-2 map depending on what else the process had created -- round 5's chained schedule halves in the same states).
// Which streams share is the runtime's business, so the env MEASURES it: two 40 us probe kernels, one on each stream of a
// pair (both warmed first: a stream's first launch creates its queue), stamped with the constant clock; if the second started
// only after the first had ended, the pair shares a queue: the env parks that stream (a parked stream keeps its use count on
// its queue, so the next candidate lands elsewhere) and tries another.  At the first mrca_step_many call on a given caller's
// stream (again if that stream changes or more ranges are asked for), never inside a capture: ~1 ms, and the one place where a
// call of this library synchronises.
__global__ void stream_probe_kernel(unsigned long long ticks, unsigned long long* stamps, int slot) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0 && stamps) {
        stamps[slot * 2 + 0] = t0;
        stamps[slot * 2 + 1] = wall_clock64();
    }
}

// 1: the two streams ran the probes side by side; 0: one after the other; -1: a HIP call failed
static int streams_overlap(mrca_env* env, hipStream_t a, hipStream_t b) {
    int serial = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(stream_probe_kernel, dim3(1), dim3(64), 0, a, 4000ull, env->probe_stamps, 0);      // 40 us
        hipLaunchKernelGGL(stream_probe_kernel, dim3(1), dim3(64), 0, b, 4000ull, env->probe_stamps, 1);
        if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return -1;
        unsigned long long h[4];
        if (hipMemcpy(h, env->probe_stamps, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        if (h[2] >= h[1] || h[0] >= h[3]) ++serial;
    }
    return serial >= 2 ? 0 : 1;
}

static int warm_stream(hipStream_t st) {
    hipLaunchKernelGGL(stream_probe_kernel, dim3(1), dim3(64), 0, st, 0ull, (unsigned long long*)nullptr, 0);
    HIP_TRY(hipStreamSynchronize(st));
    return MRCA_OK;
}

// make `*slot` a stream that overlaps with every stream of `others`; the ones that do not are parked
static int choose_stream(mrca_env* env, hipStream_t* slot, const std::vector<hipStream_t>& others) {
    for (int attempt = 0; attempt < 6; ++attempt) {
        if (!*slot) HIP_TRY(hipStreamCreateWithFlags(slot, hipStreamNonBlocking));
        if (int rc = warm_stream(*slot)) return rc;
        bool good = true;
        for (hipStream_t o : others) {
            const int ov = streams_overlap(env, o, *slot);
            if (ov < 0) return fail(MRCA_ERR_HIP, "mrca_step_many: stream probe failed: %s", hipGetErrorString(hipGetLastError()));
            if (ov == 0) {
                good = false;
                break;
            }
        }
        if (good) return MRCA_OK;
        env->parked.push_back(*slot);
        *slot = nullptr;
    }
    // six candidates in a row shared a queue with somebody: take one more as it comes (correct, only not concurrent)
    HIP_TRY(hipStreamCreateWithFlags(slot, hipStreamNonBlocking));
    return MRCA_OK;
}

static int choose_streams(mrca_env* env, hipStream_t s0, int P) {
    if (!env->probe_stamps) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&env->probe_stamps), 4 * sizeof(unsigned long long)));
    const size_t parked_before = env->parked.size();
    if (int rc = warm_stream(s0)) return rc;
    std::vector<hipStream_t> chosen{s0};
    if (env->ahead_slots > 0) {
        if (int rc = choose_stream(env, &env->move_stream, chosen)) return rc;
        chosen.push_back(env->move_stream);
    }
    for (int c = 1; c < P; ++c) {
        if (int rc = choose_stream(env, &env->chain_stream[c - 1], chosen)) return rc;
        chosen.push_back(env->chain_stream[c - 1]);
    }
    // (a caller that alternates between a few streams is checked once per stream, not once per call -- unless a check had to
    // replace one of the env's streams: then what was verified before no longer holds)
    if (env->parked.size() != parked_before || P > env->checked_ranges) env->checked_against.clear();
    if (env->checked_against.size() >= 8) env->checked_against.erase(env->checked_against.begin());
    env->checked_against.push_back(s0);
    if (P > env->checked_ranges) env->checked_ranges = P;
    if (std::getenv("MRCA_DEBUG_STREAMS"))
        std::fprintf(stderr, "[mrca] stream check against %p: %d range stream(s) + move stream chosen, %zu candidate(s) parked\n",
                     (void*)s0, env->checked_ranges - 1, env->parked.size());
    return MRCA_OK;
}

#if defined(MRCA_PROFILING)
// Profiling build, MRCA_LAUNCH_STAMPS=1: the timeline of every run-ahead pass without a profiler attached (rocprofv3's tracing
// triples the host's cost per launch, and a short region is host-paced) -- every launch of the pass stamps its first start and
// last end on the device's constant clock (MRCA_LAUNCH_BEGIN / _END), the host notes when it enqueued it; printed to stderr
// after a device synchronisation at the end of the pass.  tools/region_once.py, DESIGN.md 5.10.
struct LaunchLog {
    unsigned long long* dev = nullptr;
    std::vector<std::string> what;
    std::vector<double> host_at;
    double host_t0 = 0.0;
    bool on = false;
};
static LaunchLog g_log;
constexpr int kLogSlots = 2048;
static double host_now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static void log_begin() {
    g_log.on = getenv("MRCA_LAUNCH_STAMPS") != nullptr;
    if (!g_log.on) return;
    if (!g_log.dev && hipMalloc(reinterpret_cast<void**>(&g_log.dev), kLogSlots * 16) != hipSuccess) {
        g_log.on = false;
        return;
    }
    std::vector<unsigned long long> init((size_t)kLogSlots * 2);
    for (int i = 0; i < kLogSlots; ++i) {
        init[2 * i] = ~0ull;
        init[2 * i + 1] = 0ull;
    }
    (void)hipMemcpy(g_log.dev, init.data(), init.size() * 8, hipMemcpyHostToDevice);
    g_log.what.clear();
    g_log.host_at.clear();
    g_log.host_t0 = host_now_us();
}
static void log_note(const char* kind, int tick, int range) {      // a host-side call that is not a launch
    if (!g_log.on) return;
    char buf[64];
    snprintf(buf, sizeof buf, "%s t%d r%d", kind, tick, range);
    fprintf(stderr, "  host %7.1f us: %s\n", host_now_us() - g_log.host_t0, buf);
}
static void log_tag(mrca::EnvView& v, const char* kind, int tick, int range) {
    v.launch_stamps = nullptr;
    if (!g_log.on || (int)g_log.what.size() >= kLogSlots) return;
    char buf[64];
    snprintf(buf, sizeof buf, "%s t%d r%d", kind, tick, range);
    v.launch_stamps = g_log.dev;
    v.launch_slot = (int)g_log.what.size();
    g_log.what.push_back(buf);
    g_log.host_at.push_back(host_now_us() - g_log.host_t0);
}
static void log_end() {
    if (!g_log.on) return;
    const double host_done = host_now_us() - g_log.host_t0;
    (void)hipDeviceSynchronize();
    const double synced = host_now_us() - g_log.host_t0;
    std::vector<unsigned long long> h((size_t)kLogSlots * 2);
    (void)hipMemcpy(h.data(), g_log.dev, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull;
    for (size_t i = 0; i < g_log.what.size(); ++i) t0 = h[2 * i] < t0 ? h[2 * i] : t0;
    fprintf(stderr, "# pass of %zu launches: host enqueued for %.1f us, synchronised at %.1f us; device times from the first start\n",
            g_log.what.size(), host_done, synced);
    for (size_t i = 0; i < g_log.what.size(); ++i)
        fprintf(stderr, "%-14s enqueued %7.1f | starts %7.2f ends %7.2f (%6.2f us)\n", g_log.what[i].c_str(), g_log.host_at[i],
                (double)(h[2 * i] - t0) / 100.0, (double)(h[2 * i + 1] - t0) / 100.0, (double)(h[2 * i + 1] - h[2 * i]) / 100.0);
}
#define MRCA_LOG_BEGIN() log_begin()
#define MRCA_LOG_TAG(v, kind, tick, range) log_tag(v, kind, tick, range)
#define MRCA_LOG_END() log_end()
#else
#define MRCA_LOG_BEGIN() ((void)0)
#define MRCA_LOG_TAG(v, kind, tick, range) ((void)0)
#define MRCA_LOG_END() ((void)0)
#endif


// One run-ahead pass of mrca_step_many: K <= ahead_slots + 1 ticks.  Tick k's move launch covers ALL worlds and writes slot
// w(k) = K - 1 - k (slot 0 = the env's own fields: the pass starts from them and its last tick leaves them current), reading
// slot w(k - 1); tick 0 goes out on the caller's stream, ticks 1 .. K - 1 on the env's move stream, back to back -- a slot per
// tick, so a move launch waits for no ray cast.  The ray casts of world range c run on the range's stream (range 0: the
// caller's), tick after tick, each BLOCK of ticks behind the event "the block's last move launch is through" (the blocks and the
// host order of the enqueues: below).  Nothing else is ordered: the ray casts are what a tick costs (16.4 us for 4096 robots as
// two ranges -- two residency rounds of a workgroup's lifetime), the move launches (8.5 us) run beside them.
// Every dependency is a stream order or an event; nothing spins.
static int run_ahead_pass(mrca_env* env, const float* const* act, int K, int P, hipStream_t s0) {
    const int W = env->view.W, R = env->view.R;
    auto stream_of = [&](int c) { return c == 0 ? s0 : env->chain_stream[c - 1]; };
    auto first_world = [&](int c) { return (int)((int64_t)c * W / P); };
    hipError_t herr = hipSuccess;
    bool move_forked = false;
    MRCA_LOG_BEGIN();
    // Ticks are enqueued in BLOCKS -- [0], [1], [2], [3], then fours: a block's move launches, ONE event behind the last of them,
    // and every range's stream waits for that event once before it takes the block's ray casts.  hipStreamWaitEvent is the
    // dearest call here (4.6 us of host time against ~3 for a launch: a build with host timers, profiles/r06_ai_*): a wait per
    // tick and range made the host 15.8 us per tick against the device's 19.1 -- any hiccup starved the queues.
    // The HOST ORDER matters as much: the host needs ~13 us per tick, the device ~16.5, so the device is never far behind the
    // host and what is enqueued late starts late.  The move launches of block b + 3 are therefore enqueued BEFORE the ray casts
    // of block b: they have a queue of their own, under load they come ~13 us apart (not 8.5: the launch stamps of the
    // profiling build, MRCA_LAUNCH_STAMPS), and a ray cast waits 10 us beyond the end of the move launch it depends on.
    // Measured (own ticks on the caller's stream x blocks of lead, profiles/r06_ai_*): lead 1 (round 6's first form) 463 us
    // per 20-tick region, lead 2 - 4 with one or two own ticks 436 - 445; tick 0 alone on the caller's stream and lead 3 kept.
    int first_of[kAheadTicks + 2];
    int nb = 0;
    for (int a = 0; a < K; a += a < 4 ? 1 : 4) first_of[nb++] = a;
    first_of[nb] = K;
    constexpr int own = 1;                       // ticks below this one: move launches on the caller's stream
    auto moves_of = [&](int b) {
        const int a = first_of[b], e = first_of[b + 1];
        hipStream_t sm = a < own ? s0 : env->move_stream;
        if (a >= own && !move_forked) {       // the move stream starts behind the caller's last move launch (and so behind the caller's work)
            herr = hipStreamWaitEvent(env->move_stream, env->moved[a - 1], 0);
            if (herr != hipSuccess) return;
            move_forked = true;
        }
        for (int k = a; k < e; ++k) {
            mrca::EnvView mv = slot_view(env, K - 1 - k);
            const mrca::EnvView in = slot_view(env, k == 0 ? 0 : K - k);
            mv.world_first = 0;
            mv.world_count = W;
            MRCA_LOG_TAG(mv, "move", k, 0);
            mrca::launch_move(mv, act[k], sm, nullptr, nullptr, &in);
        }
        herr = hipEventRecord(env->moved[e - 1], sm);
    };
    auto rays_of = [&](int b) {
        const int a = first_of[b], e = first_of[b + 1];
        for (int c = 0; c < P && herr == hipSuccess; ++c) {
            hipStream_t sc = stream_of(c);
            if (c > 0 || a >= own) {        // (range 0's first ray casts follow their ticks' move launches on the caller's stream itself)
                herr = hipStreamWaitEvent(sc, env->moved[e - 1], 0);
                if (herr != hipSuccess) return;
            }
            const int w0 = first_world(c), wn = first_world(c + 1) - w0;
            for (int k = a; k < e; ++k) {
                mrca::EnvView rv = slot_view(env, K - 1 - k);
                rv.ray_first = w0 * R;
                rv.ray_count = wn * R;
                rv.world_first = w0;
                rv.world_count = wn;
                rv.eager_views = env->cfg.lazy_obs ? 0 : (MRCA_VIEW_SCAN | MRCA_VIEW_OBS);
                MRCA_LOG_TAG(rv, "ray", k, c);
                mrca::launch_raycast(rv, /*only_fresh=*/0, sc);
            }
        }
    };
    constexpr int lead = 3;                      // how many blocks the move launches are enqueued ahead of the ray casts (>= 1)
    for (int b = 0; b < nb && b < lead && herr == hipSuccess; ++b) moves_of(b);
    for (int b = 0; b < nb && herr == hipSuccess; ++b) {
        if (b + lead < nb) moves_of(b + lead);
        if (herr == hipSuccess) rays_of(b);
    }
    // join: the caller's stream continues when every range is through (the move stream is: range 0 waited for its last launch)
    hipError_t jerr = hipSuccess;
    if (K > 0) {
        for (int c = 1; c < P; ++c) {
            hipError_t e1 = hipEventRecord(env->chain_done[c - 1], stream_of(c));
            hipError_t e2 = hipStreamWaitEvent(s0, env->chain_done[c - 1], 0);
            if (jerr == hipSuccess) jerr = e1 != hipSuccess ? e1 : e2;
        }
        if (move_forked && herr != hipSuccess) {     // an early exit: the move stream may still be forked off the caller's
            hipError_t e1 = hipEventRecord(env->chain_fork, env->move_stream);
            hipError_t e2 = hipStreamWaitEvent(s0, env->chain_fork, 0);
            if (jerr == hipSuccess) jerr = e1 != hipSuccess ? e1 : e2;
        }
    }
    MRCA_LOG_END();
    if (herr != hipSuccess) return fail(MRCA_ERR_HIP, "mrca_step_many: %s", hipGetErrorString(herr));
    if (jerr != hipSuccess) return fail(MRCA_ERR_HIP, "mrca_step_many (join): %s", hipGetErrorString(jerr));
    HIP_TRY(hipGetLastError());
    return MRCA_OK;
}

int mrca_step_many(mrca_env* env, const float* const* actions_dev, int32_t num_actions, int32_t first_tick, int32_t num_ticks,
                   int32_t chains, void* stream) {
    if (!env) return fail(MRCA_ERR_INVALID, "env is NULL");
    if (!actions_dev || num_actions < 1) return fail(MRCA_ERR_INVALID, "mrca_step_many: no actions");
    if (first_tick < 0 || num_ticks < 0) return fail(MRCA_ERR_INVALID, "mrca_step_many: first_tick %d, num_ticks %d", first_tick, num_ticks);
    for (int i = 0; i < num_actions; ++i)
        if (!actions_dev[i]) return fail(MRCA_ERR_INVALID, "mrca_step_many: actions_dev[%d] is NULL", i);
    const int W = env->view.W;
    const bool chained = chains < 0;          // chains = -P: round 5's schedule (P chains `move, ray, move, ray ...` half a tick apart)
    int P = chains < 0 ? -chains : chains;
    if (P < 1) P = 1;
    if (P > W) P = W;
    if (env->view.big) P = 1;
    auto act = [&](int k) { return actions_dev[(size_t)((int64_t)first_tick + k) % (size_t)num_actions]; };
    DeviceGuard guard(env->cfg.device);
    hipStream_t s0 = static_cast<hipStream_t>(stream);
    // streams and events of ranges beyond the ones mrca_create made: never inside a capture (stream creation is not capturable)
    while ((int)env->chain_stream.size() < P - 1) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(s0, &cs);
        if (cs != hipStreamCaptureStatusNone)
            return fail(MRCA_ERR_INVALID, "mrca_step_many: chains %d needs streams the env has not created yet -- call it once "
                                          "outside the capture first (mrca_create prepares chains <= %d)", P, kChainStreamsAtCreate + 1);
        hipStream_t s;
        hipEvent_t a, b;
        HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&b, hipEventDisableTiming));
        env->chain_stream.push_back(s);
        env->chain_moved.push_back(a);
        env->chain_done.push_back(b);
    }
    if (!env->chain_fork) HIP_TRY(hipEventCreateWithFlags(&env->chain_fork, hipEventDisableTiming));
    if (num_ticks == 0) return MRCA_OK;
    if (!env->view.big && (P > 1 || (!chained && env->ahead_slots > 0)) &&
        (env->checked_ranges < P ||
         std::find(env->checked_against.begin(), env->checked_against.end(), s0) == env->checked_against.end())) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(s0, &cs);
        if (cs == hipStreamCaptureStatusNone)          // (inside a capture: the streams as they are)
            if (int rc = choose_streams(env, s0, P)) return rc;
    }
    if (!chained && !env->view.big && env->ahead_slots > 0) {
        // the run-ahead schedule, in passes of at most ahead_slots + 1 ticks (a pass ends with every stream joined)
        const int per = env->ahead_slots + 1 < kAheadTicks ? env->ahead_slots + 1 : kAheadTicks;
        std::vector<const float*> a((size_t)per);
        for (int k0 = 0; k0 < num_ticks; k0 += per) {
            const int K = num_ticks - k0 < per ? num_ticks - k0 : per;
            for (int k = 0; k < K; ++k) a[(size_t)k] = act(k0 + k);
            if (int rc = run_ahead_pass(env, a.data(), K, P, s0)) return rc;
        }
        return MRCA_OK;
    }
    if (P == 1) {
        for (int k = 0; k < num_ticks; ++k)
            if (int rc = step_impl(env, act(k), 0, env->view.N, stream)) return rc;
        return MRCA_OK;
    }
    auto stream_of = [&](int c) { return c == 0 ? s0 : env->chain_stream[c - 1]; };
    auto first_world = [&](int c) { return (int)((int64_t)c * W / P); };
    // (fork: range c's stream starts behind range c - 1's FIRST move launch -- which sits behind everything queued on the caller's
    // stream before this call, so that one wait is the fork AND the half tick between the ranges; the first launch of the call
    // goes out before any event is touched)
    int rc = MRCA_OK;
    hipError_t herr = hipSuccess;
    int forked = 1;          // ranges below this index have launches on their streams: they must be joined on every path
    for (int k = 0; k < num_ticks && rc == MRCA_OK && herr == hipSuccess; ++k) {
        const float* a = act(k);
        for (int c = 0; c < P; ++c) {
            hipStream_t sc = stream_of(c);
            const int w0 = first_world(c), wn = first_world(c + 1) - w0;
            // half a tick behind the previous range, once: its first move launch has finished, its first ray cast is starting
            if (k == 0 && c > 0) {
                herr = hipStreamWaitEvent(sc, env->chain_moved[c - 1], 0);
                if (herr != hipSuccess) break;
                forked = c + 1;
            }
            if ((rc = worlds_impl(env, a, w0, wn, sc, kPhaseMove))) break;
            if (k == 0 && c + 1 < P && (herr = hipEventRecord(env->chain_moved[c], sc)) != hipSuccess) break;
            if ((rc = worlds_impl(env, nullptr, w0, wn, sc, kPhaseObserve))) break;
        }
    }
    // join: the caller's stream continues when every range is through -- on the error paths too (a forked stream left
    // unjoined would dangle from a capture, and its launches would race the caller's next call)
    for (int c = 1; c < forked; ++c) {
        hipError_t e1 = hipEventRecord(env->chain_done[c - 1], stream_of(c));
        hipError_t e2 = hipStreamWaitEvent(s0, env->chain_done[c - 1], 0);
        if (herr == hipSuccess) herr = e1 != hipSuccess ? e1 : e2;
    }
    if (rc != MRCA_OK) return rc;
    if (herr != hipSuccess) return fail(MRCA_ERR_HIP, "mrca_step_many: %s", hipGetErrorString(herr));
    return MRCA_OK;
}

int mrca_check(mrca_env* env, void* stream) {
    if (!env) return fail(MRCA_ERR_INVALID, "env is NULL");
    DeviceGuard guard(env->cfg.device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    uint32_t bits = 0;
    HIP_TRY(hipMemcpyAsync(&bits, env->view.status, sizeof(bits), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (bits == 0) return MRCA_OK;
    HIP_TRY(hipMemsetAsync(env->view.status, 0, sizeof(bits), s));
    if (bits & mrca::kStatusCollideUndecided)
        return fail(MRCA_ERR_HIP, "collision pass of a world with more than 64 robots gave up waiting for a lower-indexed "
                                  "robot (its bounded wait ran out): at least one robot was left undecided since the last "
                                  "check; the env's state is not to be trusted");
    if (bits & mrca::kStatusOutlineWindow)
        return fail(MRCA_ERR_HIP, "fidelity mode: a cell of a robot's outline fell outside the 8 x 8 window of its bitmap since the "
                                  "last check (coordinates beyond the supported range?); collisions and lidar returns of that "
                                  "robot are not to be trusted");
    if (bits & mrca::kStatusBadRolloutRow)
        return fail(MRCA_ERR_INVALID, "mrca_rollout_store_*: the device-side tick counter was outside [0, horizon) since the last "
                                      "check (that launch stored nothing)");
    if (bits & mrca::kStatusBadBeamIndex)
        return fail(MRCA_ERR_INVALID, "mrca_sparse_obs: the beam table held an index outside [0, beams) since the last check "
                                      "(clamped to the nearest beam)");
    return fail(MRCA_ERR_HIP, "device status word 0x%x", bits);
}

static int check_rows(const mrca_rollout_rows* r) {
    if (!r) return fail(MRCA_ERR_INVALID, "rows is NULL");
    if (!r->frames || !r->fidx || !r->cur || !r->goal || !r->speed || !r->action || !r->logprob || !r->value || !r->reward ||
        !r->done)
        return fail(MRCA_ERR_INVALID, "mrca_rollout_rows: NULL pointer");
    if (r->horizon < 1) return fail(MRCA_ERR_INVALID, "mrca_rollout_rows: horizon %d", r->horizon);
    if ((reinterpret_cast<uintptr_t>(r->frames) % 16) || (reinterpret_cast<uintptr_t>(r->goal) % 8) ||
        (reinterpret_cast<uintptr_t>(r->speed) % 8) || (reinterpret_cast<uintptr_t>(r->action) % 8))
        return fail(MRCA_ERR_INVALID, "mrca_rollout_rows: frames must be 16-byte, goal / speed / action 8-byte aligned");
    return MRCA_OK;
}

int mrca_rollout_store_state(mrca_env* env, const mrca_rollout_rows* rows, const int64_t* tick_dev, const float* action_dev,
                             const float* logprob_dev, const float* value_dev, void* stream) {
    if (!env) return fail(MRCA_ERR_INVALID, "env is NULL");
    if (int rc = check_rows(rows)) return rc;
    if (!tick_dev || !action_dev || !logprob_dev || !value_dev)
        return fail(MRCA_ERR_INVALID, "mrca_rollout_store_state: NULL pointer");
    if (reinterpret_cast<uintptr_t>(action_dev) % 8)
        return fail(MRCA_ERR_INVALID, "mrca_rollout_store_state: action_dev must be 8-byte aligned");
    DeviceGuard guard(env->cfg.device);
    mrca::launch_rollout_store_state(env->view, *rows, tick_dev, action_dev, logprob_dev, value_dev, static_cast<hipStream_t>(stream));
    HIP_TRY(hipGetLastError());
    return MRCA_OK;
}

int mrca_rollout_store_outcome(mrca_env* env, const mrca_rollout_rows* rows, int64_t* tick_dev, uint32_t* ticket_dev,
                               void* stream) {
    if (!env) return fail(MRCA_ERR_INVALID, "env is NULL");
    if (int rc = check_rows(rows)) return rc;
    if (!tick_dev || !ticket_dev) return fail(MRCA_ERR_INVALID, "mrca_rollout_store_outcome: NULL pointer");
    DeviceGuard guard(env->cfg.device);
    mrca::launch_rollout_store_outcome(env->view, *rows, tick_dev, ticket_dev, static_cast<hipStream_t>(stream));
    HIP_TRY(hipGetLastError());
    return MRCA_OK;
}

int mrca_normalize_scans(const float* in_dev, float* out_dev, size_t count, void* stream) {
    if (!in_dev || !out_dev) return fail(MRCA_ERR_INVALID, "mrca_normalize_scans: NULL pointer");
    if (count % 4 || reinterpret_cast<uintptr_t>(in_dev) % 16 || reinterpret_cast<uintptr_t>(out_dev) % 16)
        return fail(MRCA_ERR_INVALID, "mrca_normalize_scans: count must be a multiple of 4 and the buffers 16-byte aligned");
    DeviceGuard guard(mrca::device_of(in_dev));
    mrca::launch_normalize(in_dev, out_dev, (long long)count, static_cast<hipStream_t>(stream));
    HIP_TRY(hipGetLastError());
    return MRCA_OK;
}

int mrca_sparse_obs(mrca_env* env, const int32_t* index_dev, int32_t beam_num, float* out_dev, void* stream) {
    if (!env) return fail(MRCA_ERR_INVALID, "env is NULL");
    if (!index_dev || !out_dev) return fail(MRCA_ERR_INVALID, "mrca_sparse_obs: NULL pointer");
    if (beam_num < 1 || beam_num > env->view.B) return fail(MRCA_ERR_INVALID, "mrca_sparse_obs: beam_num %d not in [1, %d]", beam_num, env->view.B);
    DeviceGuard guard(env->cfg.device);
    mrca::launch_sparse_obs(env->view, index_dev, beam_num, out_dev, static_cast<hipStream_t>(stream));
    HIP_TRY(hipGetLastError());
    return MRCA_OK;
}

int mrca_gae(const float* rewards_dev, const float* values_dev, const float* last_value_dev,
             const uint8_t* dones_dev, float gamma, float lam, int32_t T, int32_t N, float* targets_dev,
             float* advs_dev, void* stream) {
    if (!rewards_dev || !values_dev || !last_value_dev || !dones_dev || !targets_dev || !advs_dev)
        return fail(MRCA_ERR_INVALID, "mrca_gae: NULL pointer");
    if (T < 1 || N < 1) return fail(MRCA_ERR_INVALID, "mrca_gae: T=%d N=%d", T, N);
    DeviceGuard guard(mrca::device_of(rewards_dev));      // launch where the buffers live
    mrca::launch_gae(rewards_dev, values_dev, last_value_dev, dones_dev, gamma, lam, T, N, targets_dev, advs_dev,
                     static_cast<hipStream_t>(stream));
    HIP_TRY(hipGetLastError());
    return MRCA_OK;
}

int mrca_enable_timing(mrca_env* env, int32_t on) {
    if (!env) return fail(MRCA_ERR_INVALID, "env is NULL");
    if (on && env->ev.empty()) {
        env->ev.resize(4 * kTimingRing);
        for (auto& e : env->ev) HIP_TRY(hipEventCreate(&e));
    }
    env->timing = on > 0 ? on : 0;
    env->step_count = 0;
    env->ev_used = 0;
    return MRCA_OK;
}

// What an event pair reads with NOTHING between its two records: the processing time of the second marker, which every
// (event, kernel, event) measurement of mrca_read_timing contains once per kernel.  bench.py reports its kernel
// averages net of this figure so that they add up to no more than the tick they are part of.
int mrca_event_pair_overhead(void* stream, int32_t samples, float* us_out) {
    if (!us_out || samples < 1 || samples > 4096) return fail(MRCA_ERR_INVALID, "mrca_event_pair_overhead: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int dev = 0;
    if (s) {     // the events belong on the stream's device, whatever the caller's current one is
        HIP_TRY(hipStreamGetDevice(s, &dev));
    } else {
        HIP_TRY(hipGetDevice(&dev));
    }
    DeviceGuard guard(dev);
    hipEvent_t a = nullptr, b = nullptr;
    HIP_TRY(hipEventCreate(&a));
    hipError_t err = hipEventCreate(&b);
    double sum = 0.0;
    for (int i = 0; err == hipSuccess && i < samples; ++i) {
        float ms = 0.0f;
        if ((err = hipEventRecord(a, s)) != hipSuccess) break;
        if ((err = hipEventRecord(b, s)) != hipSuccess) break;
        if ((err = hipEventSynchronize(b)) != hipSuccess) break;
        if ((err = hipEventElapsedTime(&ms, a, b)) != hipSuccess) break;
        sum += ms;
    }
    (void)hipEventDestroy(a);          // on every path
    if (b) (void)hipEventDestroy(b);
    if (err != hipSuccess) return fail(MRCA_ERR_HIP, "mrca_event_pair_overhead: %s", hipGetErrorString(err));
    *us_out = (float)(sum / samples * 1e3);
    return MRCA_OK;
}

#if defined(MRCA_PROFILING)
// Profiling build only (libmrca_env_prof.so): ablation switches (results are WRONG while bits 0-5 are set) and
// launch-shape knobs (results unchanged): bits 8-10 = k > 0:
// 1 << (k-1) beams per marching thread; bit 11: a dedicated fifth preparation wave; bit 12: the beams of a thread marched
// in lock step instead of one after the other.
int mrca_set_debug_flags(mrca_env* env, int32_t flags) {
    if (!env) return fail(MRCA_ERR_INVALID, "env is NULL");
    env->view.debug_flags = flags & 0xFF;
    const int knob = (flags >> 8) & 7;
    if (knob) {
        const int shift = knob - 1;
        const int threads = env->cfg.beams >> shift;
        if (shift > 2 || threads < 64 || threads % 64 || threads < (env->cfg.beams >> 2))
            return fail(MRCA_ERR_INVALID, "beams-per-thread knob %d out of range", knob);
        env->view.ray_shift = shift;
    } else {
        env->view.ray_shift = product_ray_shift(env->cfg.beams, env->view.big);   // the product's launch shape
    }
    env->view.ray_prep_wave = (flags & 0x800) ? 1 : 0;
    env->view.ray_sequential = (flags & 0x1000) ? 0 : 1;
    DeviceGuard guard(env->cfg.device);
    HIP_TRY(upload_view(env));            // move_kernel / raycast_kernel read the switches from the device copy
    return MRCA_OK;
}
#endif

#if defined(MRCA_PROFILING)
// Profiling build only: average s_memtime ticks between the move kernel's phase stamps (0 -> 1 ... 7 -> 8) over the worlds
// of the LAST launch (synchronises the device).
int mrca_debug_move_stamps(mrca_env* env, double* avg_ticks_out /* [9]: 8 deltas + entry-to-end */) {
    if (!env || !avg_ticks_out) return fail(MRCA_ERR_INVALID, "NULL argument");
    DeviceGuard guard(env->cfg.device);
    HIP_TRY(hipDeviceSynchronize());
    const int W = env->view.W < 4096 ? env->view.W : 4096;
    std::vector<unsigned long long> h((size_t)10 * W);
    mrca::read_move_stamps(h.data(), W);
    for (int k = 0; k < 8; ++k) {
        double sum = 0.0;
        for (int w = 0; w < W; ++w) sum += (double)(h[(size_t)(k + 1) * W + w] - h[(size_t)k * W + w]);
        avg_ticks_out[k] = sum / W;
    }
    double sum = 0.0;
    for (int w = 0; w < W; ++w) sum += (double)(h[(size_t)8 * W + w] - h[w]);
    avg_ticks_out[8] = sum / W;
    return MRCA_OK;
}
#endif

#if defined(MRCA_PROFILING)
// Profiling build only: the raw stamps of the LAST move launch, out[k * worlds + w] = stamp k (0..8) of world w -- for the
// DISTRIBUTION over worlds (a launch lasts as long as its slowest world): tools/move_tail.py
int mrca_debug_move_stamps_raw(mrca_env* env, unsigned long long* out /* [10 * worlds] */, int32_t* worlds_out) {
    if (!env || !out || !worlds_out) return fail(MRCA_ERR_INVALID, "NULL argument");
    DeviceGuard guard(env->cfg.device);
    HIP_TRY(hipDeviceSynchronize());
    const int W = env->view.W < 4096 ? env->view.W : 4096;
    mrca::read_move_stamps(out, W);
    *worlds_out = W;
    return MRCA_OK;
}
#endif

#if defined(MRCA_PROFILING)
// Profiling build only: s_memtime stamps of the LAST ray-cast launch (synchronises the device).  out[w * 7 + k], w = 0, 1
// (wave 0 prepares the neighbour list, wave 1 only marches), k = 0..6: mean over workgroups of stamp k minus the
// workgroup's entry stamp; out[14..16] = 0 (reserved).
int mrca_debug_ray_stamps(mrca_env* env, double* out /* [17] */) {
    if (!env || !out) return fail(MRCA_ERR_INVALID, "NULL argument");
    DeviceGuard guard(env->cfg.device);
    HIP_TRY(hipDeviceSynchronize());
    const int nb = env->last_ray_count < 8192 ? env->last_ray_count : 8192;    // the workgroups whose stamps are this launch's
    if (nb < 1) return fail(MRCA_ERR_INVALID, "no ray-cast launch to read stamps of");
    std::vector<unsigned long long> h((size_t)14 * nb);
    mrca::read_ray_stamps(h.data(), nb);
    auto at = [&](int w, int k, int b) { return h[((size_t)w * 7 + k) * nb + b]; };
    for (int w = 0; w < 2; ++w)
        for (int k = 0; k < 7; ++k) {
            double sum = 0.0;
            for (int b = 0; b < nb; ++b) sum += (double)(long long)(at(w, k, b) - at(0, 0, b));
            out[w * 7 + k] = sum / nb;
        }
    // [14..16] reserved: stamps of different workgroups cannot be compared (s_memtime counters have unrelated origins from
    // die to die and within one), only differences inside a workgroup mean something
    out[14] = out[15] = out[16] = 0.0;
    return MRCA_OK;
}
#endif

int mrca_read_timing(mrca_env* env, float* move_ms_total, float* ray_ms_total, int32_t* launches) {
    if (!env) return fail(MRCA_ERR_INVALID, "env is NULL");
    float mv = 0.0f, ry = 0.0f;
    const int n = env->ev_used / 4;
    if (n > 0) HIP_TRY(hipEventSynchronize(env->ev[env->ev_used - 1]));
    for (int i = 0; i < n; ++i) {
        float a = 0.0f, b = 0.0f;
        HIP_TRY(hipEventElapsedTime(&a, env->ev[4 * i], env->ev[4 * i + 1]));
        HIP_TRY(hipEventElapsedTime(&b, env->ev[4 * i + 2], env->ev[4 * i + 3]));
        mv += a;
        ry += b;
    }
    if (move_ms_total) *move_ms_total = mv;
    if (ray_ms_total) *ray_ms_total = ry;
    if (launches) *launches = n;
    env->ev_used = 0;
    return MRCA_OK;
}

}  // extern "C"
