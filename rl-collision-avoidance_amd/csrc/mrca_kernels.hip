// mrca_kernels.hip -- gfx950 kernels of the batched Stage tick.
//
//   move_kernel    one 64-lane wavefront per world, lane = robot.  Latch action, integrate,
//                  outline-vs-grid test (skipped where the per-cell distance field proves the
//                  footprint free; otherwise patches pulled into LDS by the whole wave and walked
//                  four lanes per robot), then the collision pass in robot order (Stage moves its
//                  models one after another): a broad phase picks the robots that can touch
//                  anybody, each of them takes a turn -- its provisional pose is broadcast, every
//                  lane runs the rectangle SAT against the pose it has at that point of the
//                  order, a wavefront ballot decides revert+stall.  Reward / terminal / episode
//                  bookkeeping (wave-parallel Philox resets, group ballots) follow.
//   raycast_kernel one workgroup per robot, beams/K threads, K beams per thread (product: K = 2, marched one
//                  after the other; the lock-step form and a dedicated preparation wave are measured variants).
//                  The first wave compacts the other robots of the world within lidar reach into LDS (ballot +
//                  popcount) together with a per-beam bitmask of who can touch which beam, then marches like
//                  the others: the exact skipping march over the per-cell free-rectangle field (~2 dependent
//                  L2 lookups per ray).  The robot's own sin/cos and the field entry of its cell come from the
//                  16-byte `head` record the move kernel published (scalar loads), so no wave recomputes them.
//                  Each thread slab-tests its own beams against the flagged neighbours and stores their range itself,
//                  into the robot's ring of raw scans (one slot per tick, no shift, no second copy).
//   materialize_kernel  the newest scan (MRCA_F_SCAN) / the normalised stack in deque order (MRCA_F_OBS) out of the ring of
//                  raw scans, on demand; newest_obs_kernel: the rollout buffer's row.
//   bw_*           the move kernel's tick for worlds with more than 64 robots (per-robot threads, spatial
//                  hashes, ordered collision pass as dependency rounds); raycast_kernel<K, true> is its ray cast.
//   reset_kernel   explicit reset_pose / control_pose / generate_goal_point.
//   gae_kernel     reverse GAE scan, thread per robot, coalesced over N.
//
// No dense contraction anywhere in the environment: MFMA is deliberately unused here (BASELINE.json north_star); the
// policy's conv front end (mrca_policy.hip) is where it is used.
#include "mrca_kernels.h"

#include <hip/hip_ext.h>

namespace mrca {

namespace {

constexpr int kWave = 64;
constexpr int kPatchLoads = 6;   // independent patch-word loads a thread of move_kernel keeps in flight
constexpr int kMoveWaves = 4;    // wavefronts per world in move_kernel
constexpr int kEventOutlineTasks = 3 * kWave * kMoveWaves;   // outline test by events up to three passes of the block's threads, else by walks

__device__ __forceinline__ void begin_episode(const EnvView& e, int n, int local, float curx, float cury, float* px,
                                              float* py, float* pth, float* gx, float* gy, float* pdist,
                                              const float* pose_override, const float* goal_override) {
    const uint32_t ep = (uint32_t)e.episode[n];
    float x, y, th;
    if (pose_override) {
        x = pose_override[0];
        y = pose_override[1];
        th = pose_override[2];
    } else {
        const int mode = e.reset_mode[local];
        if (mode == 0) {
            x = e.init_table[local * 3 + 0];
            y = e.init_table[local * 3 + 1];
            th = wrap_angle(e.init_table[local * 3 + 2]);
        } else {
            sample_pose(mode, (uint32_t)n, ep, e.key0, e.key1, curx, cury, &x, &y, &th);
        }
    }
    float qx, qy;
    if (goal_override) {
        qx = goal_override[0];
        qy = goal_override[1];
    } else {
        const int gmode = e.goal_mode[local];
        if (gmode == 0) {
            qx = e.goal_table[local * 2 + 0];
            qy = e.goal_table[local * 2 + 1];
        } else {
            sample_goal(gmode, (uint32_t)n, ep, e.key0, e.key1, x, y, &qx, &qy);
        }
    }
    const float ddx = qx - x, ddy = qy - y;
    const float d = sqrtf(ddx * ddx + ddy * ddy);
    *px = x;
    *py = y;
    *pth = th;
    *gx = qx;
    *gy = qy;
    *pdist = e.pre_dist_zero ? 0.0f : d;
    e.init_pose[n * 3 + 0] = x;
    e.init_pose[n * 3 + 1] = y;
    e.init_pose[n * 3 + 2] = th;
}

// lane must be wave-uniform: v_readlane_b32
__device__ __forceinline__ int ibcast(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float fbcast(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// Wave-parallel rejection sampling for FOUR robots at once: the wave as four groups of 16 lanes, group g sampling for "its"
// robot (all arguments uniform within a group; `need` false = the group has nothing to sample), 16 consecutive attempts k per
// step, the lowest acceptable k wins -- the same draw the one-lane loop (sample_pose / sample_goal) returns, without a
// wavefront waiting on one unlucky robot's long tail.  (Rounds 2-4: the whole wave sampled 64 attempts for ONE robot at a time;
// a world's restarts took turns, 2 700 ticks each,
// and a launch lasts as long as its slowest world: with 0.5 restarts per world and tick on the Stage-1 worlds some world of
// every launch has three (profiles/r05_k_*: mean workgroup 13 100 ticks, launch 22 300), and a Stage-2 group of ten
// region-sampled robots restarts together.)
__device__ __forceinline__ void group_sample_pose(int g, int sub, bool need, int mode, uint32_t gid, uint32_t ep, uint32_t k0,
                                                  uint32_t k1, float curx, float cury, float* px, float* py, float* pth) {
    bool done = !need;
    for (int base = 0; base < kMaxTriesPose; base += 16) {
        float x, y, th;
        const bool ok = pose_try(mode, gid, ep, (uint32_t)(base + sub), k0, k1, curx, cury, &x, &y, &th);
        const bool last = base + 16 >= kMaxTriesPose;
        const unsigned long long m = __ballot(!done && (ok || (last && sub == kMaxTriesPose - 1 - base)));
        const uint32_t m16 = (uint32_t)(m >> (16 * g)) & 0xFFFFu;
        const int w = 16 * g + (m16 ? __ffs((int)m16) - 1 : 0);
        const float wx = __shfl(x, w, kWave), wy = __shfl(y, w, kWave), wth = __shfl(th, w, kWave);
        if (!done && m16) {
            *px = wx;
            *py = wy;
            *pth = wth;
            done = true;
        }
        if (__ballot(!done) == 0ull) break;
    }
}

__device__ __forceinline__ void group_sample_goal(int g, int sub, bool need, int mode, uint32_t gid, uint32_t ep, uint32_t k0,
                                                  uint32_t k1, float curx, float cury, float* gx, float* gy) {
    bool done = !need;
    for (int base = 0; base < kMaxTriesGoal; base += 16) {
        float x, y;
        const bool ok = goal_try(mode, gid, ep, (uint32_t)(base + sub), k0, k1, curx, cury, &x, &y);
        const bool last = base + 16 >= kMaxTriesGoal;
        const unsigned long long m = __ballot(!done && (ok || (last && sub == kMaxTriesGoal - 1 - base)));
        const uint32_t m16 = (uint32_t)(m >> (16 * g)) & 0xFFFFu;
        const int w = 16 * g + (m16 ? __ffs((int)m16) - 1 : 0);
        const float wx = __shfl(x, w, kWave), wy = __shfl(y, w, kWave);
        if (!done && m16) {
            *gx = wx;
            *gy = wy;
            done = true;
        }
        if (__ballot(!done) == 0ull) break;
    }
}

struct MiniGrid {  // the move kernel's per-robot occupancy patch in LDS
    const uint32_t* t;
    int y0, w0, stride;
    __device__ __forceinline__ bool operator()(int ix, int iy) const {
        return (t[(iy - y0) * stride + ((ix >> 5) - w0)] >> (ix & 31)) & 1u;
    }
};

// The observation stack (ppo_stage1.py:59-60,87-89: a deque of the last F normalised scans) is stored as a RING of RAW
// scans per robot: scan_ring[n][slot][beam] with ring_head[n] = the slot of the NEWEST scan; logical frame f (0 = oldest)
// sits in slot (head + 1 + f) mod F.  A tick writes ONE row per robot (the ray cast's epilogue) and bumps the head --
// rounds 1-2 shifted the whole stack down every tick (33.6 of the tick's 56.6 MB of HBM traffic at 4096 robots), round 3
// still stored every beam twice (the scan and its affine image x / 6 - 0.5: 16.8 MB of the tick's 22.9).  The affine map
// (stage_world1.py:140) is applied by whoever READS: the policy's front end while it stages a scan (mrca_policy.hip),
// newest_obs_kernel for the rollout buffer's row, and this kernel, which makes the two views a reference-shaped caller
// wants -- MRCA_F_SCAN (the newest scan, contiguous) and MRCA_F_OBS (the normalised stack in deque order, what
// CNNPolicy.forward eats) -- on demand (mrca_materialize) or after every call when the env was created with
// lazy_obs = 0.  A thread owns float4 columns of robots [ray_first, ray_first + ray_count).
// (ABI 4-5 kept what a beam hit in the ring's sign bit; the readers' |x| -- a source modifier, free -- stays as a belt)
__device__ __forceinline__ float4 norm_obs4(float4 v) {
    return make_float4(norm_obs(fabsf(v.x)), norm_obs(fabsf(v.y)), norm_obs(fabsf(v.z)), norm_obs(fabsf(v.w)));
}
__device__ __forceinline__ float4 fabs4(float4 v) { return make_float4(fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w)); }

__global__ void materialize_kernel(EnvView e, int what) {
    const int fstride = e.B >> 2;
    const long long total = (long long)e.ray_count * fstride;
    const float4* ring = reinterpret_cast<const float4*>(e.scan_ring) + (size_t)e.ray_first * e.F * fstride;
    float4* out = reinterpret_cast<float4*>(e.obs) + (size_t)e.ray_first * e.F * fstride;
    float4* scan = reinterpret_cast<float4*>(e.scan) + (size_t)e.ray_first * fstride;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += stride) {
        const long long r = k / fstride;
        const int col = (int)(k - r * fstride);
        const int hd = e.ring_head[e.ray_first + r];
        const float4* src = ring + r * e.F * fstride + col;
        const float4 newest = src[hd * fstride];
        if (what & 1) scan[r * fstride + col] = fabs4(newest);
        if (what & 2) {
            float4* dst = out + r * e.F * fstride + col;
            if (e.F == 3) {
                const int s0 = hd == 2 ? 0 : hd + 1, s1 = s0 == 2 ? 0 : s0 + 1;
                const float4 a = src[s0 * fstride], b = src[s1 * fstride];
                dst[0] = norm_obs4(a);
                dst[fstride] = norm_obs4(b);
                dst[2 * fstride] = norm_obs4(newest);
            } else {
                for (int f = 0; f < e.F; ++f) dst[f * fstride] = norm_obs4(src[((hd + 1 + f) % e.F) * fstride]);
            }
        }
    }
}

// out[n][:] = x / 6 - 0.5 of robot n's newest scan: the one row per tick a single-frame rollout buffer keeps
__global__ void newest_obs_kernel(EnvView e, float* __restrict__ out) {
    const int fstride = e.B >> 2;
    const long long total = (long long)e.N * fstride;
    const float4* ring = reinterpret_cast<const float4*>(e.scan_ring);
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += stride) {
        const long long r = k / fstride;
        const int col = (int)(k - r * fstride);
        const int hd = e.ring_head[r];
        reinterpret_cast<float4*>(out)[k] = norm_obs4(ring[(r * e.F + hd) * fstride + col]);
    }
}

__global__ void normalize_kernel(const float4* __restrict__ in, float4* __restrict__ out, long long count4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < count4; k += stride) out[k] = norm_obs4(in[k]);
}

// get_laser_observation for beam_num != the raw sample count (stage_world1.py:126-139: a left half picked ascending and a
// right half picked descending from the raw scan): out[n][f][k] = x / 6 - 0.5 of beam index[k] of robot n's logical frame f.
// The index table is the caller's (the reference builds it by repeated float64 addition; mrca/vec_env.py restates that).
__global__ void sparse_obs_kernel(EnvView e, const int32_t* __restrict__ index, int nb, float* __restrict__ out) {
    const long long total = (long long)e.N * e.F * nb;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += stride) {
        const long long rf = k / nb;
        const int j = (int)(k - rf * nb);
        const long long r = rf / e.F;
        const int f = (int)(rf - r * e.F);
        const int hd = e.ring_head[r];
        int slot = hd + 1 + f;
        slot -= slot >= e.F ? e.F : 0;
        // (the table is the caller's: an entry outside [0, B) must not become an out-of-bounds read -- clamped, and flagged)
        int bi = index[j];
        if ((unsigned)bi >= (unsigned)e.B) {
            atomicOr(e.status, kStatusBadBeamIndex);
            bi = bi < 0 ? 0 : e.B - 1;
        }
        out[k] = norm_obs(fabsf(e.scan_ring[((size_t)r * e.F + slot) * e.B + bi]));
    }
}

#if defined(MRCA_PROFILING)
// profiling build only: s_memtime stamps of the move kernel's phases (every memory operation drained first, so that a
// phase owns its latency), one row per world; read through mrca_debug_move_stamps (tools/ablate.py)
constexpr int kMoveStamps = 10, kMoveStampWorlds = 4096;
__device__ unsigned long long g_move_stamps[kMoveStamps][kMoveStampWorlds];
#define MRCA_STAMP(k)                                                                      \
    do {                                                                                   \
        __builtin_amdgcn_s_waitcnt(0);                                                     \
        if (threadIdx.x == 0 && blockIdx.x < kMoveStampWorlds)                             \
            g_move_stamps[k][blockIdx.x] = __builtin_amdgcn_s_memtime();                   \
    } while (0)
// the same for the ray cast: lane 0 of waves 0 and 1 of every workgroup; memory is drained at a stamp only with debug
// flag 64 (without it the stamps leave the kernel's overlap as it is)
constexpr int kRayStamps = 7, kRayStampBlocks = 8192;
__device__ unsigned long long g_ray_stamps[2][kRayStamps][kRayStampBlocks];
#define MRCA_RSTAMP(k)                                                                                  \
    do {                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        if (MRCA_DBG(e, 64)) __builtin_amdgcn_s_waitcnt(0);                                             \
        if ((threadIdx.x & 63) == 0 && threadIdx.x < 128 && blockIdx.x < kRayStampBlocks)               \
            g_ray_stamps[threadIdx.x >> 6][k][blockIdx.x] = __builtin_amdgcn_s_memtime();               \
        __builtin_amdgcn_sched_barrier(0);                                                              \
    } while (0)
#else
#define MRCA_STAMP(k) do { } while (0)
#define MRCA_RSTAMP(k) do { } while (0)
#endif

// kMoveWaves wavefronts per world.  Every wave loads and integrates the whole world (lane = robot: ~100 B per robot, and
// each wave can then shuffle any robot's values without a trip through LDS); the three phases that are loops over
// robots -- the broad phase of the collision pass, the patch loads, the outline walks -- are split across the waves;
// wave 0 alone carries on with the ordered pass, rewards, restarts and the stores.  (One wave per world, rounds 1-2:
// those three phases were 73 % of the kernel's chain on the Stage-2 map, profiles/r03/r03_g_ablate.txt.)
// (Leading scalar arguments: the first 14 dwords of a kernel's arguments are PRELOADED into SGPRs by the command processor
// (csrc/build.sh: -amdgpu-kernarg-preload-count), so the loads the tick's dependent chain starts with go out at once instead
// of one memory round trip later, behind the s_load of a 600-byte EnvView.  A/B on one box, three alternating runs each:
// move 11.12 -> 10.85 us, ray cast 21.35 -> 20.51 us, profiles/r04_y_ab_kernarg_preload.txt.)
// (pose_p / head_p / goal_p / outline_p are what the tick READS -- the poses, head records, goals and outlines as the tick
// before left them; e.pose / e.head / e.goal / e.fresh / e.outline are where it WRITES them.  The same buffers for an ordinary
// call; mrca_step_many alternates two sets, so that tick k + 1's move launch may run while tick k's ray cast still reads
// tick k's poses: DESIGN.md 5.10.  With the two scalars they fill the 14 preloaded dwords exactly.)
__global__ __launch_bounds__(kWave * kMoveWaves) void move_kernel(int R_, int world_first, const float* pose_p,
                                                                  const float4* head_p, const float* __restrict__ actions,
                                                                  const uint8_t* live_p, const float* goal_p,
                                                                  const OutlineBits* outline_p, const EnvView* __restrict__ view_p,
                                                                  MoveOut out) {
    extern __shared__ __attribute__((aligned(16))) uint32_t mini[];
    // the env's view from device memory (uniform loads off a read-only pointer: scalar, like the kernel-argument loads they
    // replace), with this launch's output slot in place of the five fields a slot holds
    EnvView e = *view_p;
    e.pose = out.pose;
    e.head = out.head;
    e.goal = out.goal;
    e.fresh = out.fresh;
    e.outline = out.outline;
#if defined(MRCA_PROFILING)
    e.launch_stamps = out.launch_stamps;
    e.launch_slot = out.launch_slot;
#endif
    MRCA_LAUNCH_BEGIN(e);
    MRCA_STAMP(0);
    const int world = world_first + blockIdx.x;     // (mrca_step_worlds: a launch may cover a range of worlds)
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid >> 6;
    const bool valid = lane < R_;
    const int n = world * R_ + (valid ? lane : 0);

    // every per-robot input is requested up front so that all of it arrives in ONE memory round trip
    float x = pose_p[n * 3 + 0], y = pose_p[n * 3 + 1], th = pose_p[n * 3 + 2];
    const bool live = live_p[n] != 0;
    const float act_v = actions[n * 2 + 0], act_w = actions[n * 2 + 1];
    float gx = goal_p[n * 2 + 0], gy = goal_p[n * 2 + 1];
    float pdist = e.prev_dist[n];
    int t = e.t[n];
    float reward = e.reward[n];
    uint8_t done = e.done[n];
    uint8_t res = e.result[n];
    uint8_t first = e.first_result[n];
    uint8_t crashed = e.crashed[n];
    int ep = e.episode[n];
    const int rmode = valid ? e.reset_mode[lane] : 0;
    const int gmode = valid ? e.goal_mode[lane] : 0;
    const int gid = valid ? e.group_id[lane] : -1;
    // this lane's rows of the pose / goal tables: a restart below takes them by broadcast instead of a load per robot
    const float tix = valid ? e.init_table[lane * 3 + 0] : 0.0f, tiy = valid ? e.init_table[lane * 3 + 1] : 0.0f;
    const float tith = valid ? e.init_table[lane * 3 + 2] : 0.0f;
    const float tgx = valid ? e.goal_table[lane * 2 + 0] : 0.0f, tgy = valid ? e.goal_table[lane * 2 + 1] : 0.0f;
    const float4 hd = head_p[n];   // sin / cos of th and the field entry of the robot's cell, kept by whoever moved it
    // fidelity mode (collision_raster): robots collide when their OUTLINES SHARE A RASTER CELL (Stage's rule) instead of when
    // their rectangles overlap.  An outline is an anchored 8 x 8 bitmap (mrca_device.h OutlineBits); the one of the pose at
    // tick start comes with the robot's record, kept by whoever moved it
    const bool raster = e.raster_inv > 0.0f;
    OutlineBits ob_old{0, 0, 0u, 0u};
    if (raster) ob_old = outline_p[n];
    // a robot whose script no longer sends cmd_vel (dead, ppo_stage2.py:72-74): idles, or -- hold_velocity, what Stage
    // does with the last SetSpeed -- keeps driving at the command it was given last
    float held_v = 0.0f, held_w = 0.0f;
    if (e.hold_velocity) {
        held_v = e.speed[n * 2 + 0];
        held_w = e.speed[n * 2 + 1];
    }
    const float v = live ? sane_cmd(act_v) : held_v;
    const float w = live ? sane_cmd(act_w) : held_w;

    // integrate: explicit Euler with the heading at tick start
    float s = hd.x, c = hd.y;
    uint32_t cellv = __float_as_uint(hd.z), cellw = __float_as_uint(hd.w);
    const float d = v * kDt;
    const float nx = x + d * c;
    const float ny = y + d * s;
    const float nth = wrap_angle(th + w * kDt);
    float ns, nc;
    sincos_det(nth, &ns, &nc);
    const bool moving = valid && ((v != 0.0f) || (w != 0.0f));
    MRCA_STAMP(1);      // state loaded, integrated

    // --- outline-vs-grid test.  Skipped (same answer: free) when the coarse free-distance field says
    //     every block within the footprint's circumradius of the provisional centre is empty.  For the
    //     others the (2*hc+1)-row patch under the footprint is pulled into LDS by the WHOLE wave, every robot's words
    //     in flight together, and each robot then walks its outline in LDS.
    const int hc = e.foot_hc;
    const int prow = 2 * hc + 1;
    const int pwords = (prow + 31) / 32 + 1;
    const int psize = prow * pwords;
    const int pix = (int)floorf((nx - e.g.x0) * e.g.inv_cell);
    const int piy = (int)floorf((ny - e.g.y0) * e.g.inv_cell);
    const int py0 = piy - hc;
    const int pw0 = (pix - hc) >> 5;
    // nothing occupied within hc cells (Chebyshev) of the centre cell => every cell the outline walk
    // could visit is free => no hit, without touching the bitmap.  The byte is requested here and used
    // after the broad phase below, which needs no memory and so runs inside this load's latency.
    const bool inside = pix >= 0 && piy >= 0 && pix < e.g.width && piy < e.g.height;
    const bool check_map = valid && !MRCA_DBG(e, 8);
    const uint8_t clearance = (check_map && inside) ? e.cellfield[(size_t)piy * e.g.width + pix] : 0;
    // the field entry of the provisional cell rides in the same round trip: it becomes the `head` entry if the
    // move is committed
    const FreeRectField rect_field{e.free_rect, e.g.width, e.g.height, e.free_rect_pitch};
    uint32_t cellv_new, cellw_new;
    rect_field.cell(pix, piy, &cellv_new, &cellw_new);

    // --- broad phase of the robot-robot collision pass (the pass itself follows the outline test): robot i
    //     can only touch robot j if its provisional centre comes within 2 x circumradius of j's old or
    //     new centre, so only the (few) robots with such a neighbour take a turn in the ordered pass;
    //     everybody else commits straight away -- their outcome does not depend on the order.
    int* need_list = reinterpret_cast<int*>(mini + kWave * psize);   // [64] lanes that need the walk
    int* hit_flag = need_list + kWave;                               // [64] result per robot lane
    int* inv_part = hit_flag + kWave;                                // [kMoveWaves][64] broad-phase partial results
    bool involved = false;
    if (!MRCA_DBG(e, 16)) {
        // every robot's old and provisional centre goes through LDS once (the patch area is free until the outline test
        // below) and is read back as ONE broadcast 16-byte load per candidate; wave q looks at candidates q, q + 4, ...
        float4* centres = reinterpret_cast<float4*>(mini);
        if (wave == 0) centres[lane] = make_float4(x, y, nx, ny);
        __syncthreads();
        bool part = false;
#pragma unroll 4
        for (int j = wave; j < e.R; j += kMoveWaves) {
            const float4 q = centres[j];
            const float ax = nx - q.x, ay = ny - q.y;
            const float bx2 = nx - q.z, by2 = ny - q.w;
            const float d_old = ax * ax + ay * ay, d_new = bx2 * bx2 + by2 * by2;
            if (j != lane && (d_old <= e.collide_reach2 || d_new <= e.collide_reach2)) part = true;  // (2*0.2907 + 0.001)^2
        }
        inv_part[wave * kWave + lane] = (part && valid) ? 1 : 0;
        __syncthreads();   // ... which also ends the reads of `centres`: the patches below reuse this LDS
        int any = 0;
#pragma unroll
        for (int q = 0; q < kMoveWaves; ++q) any |= inv_part[q * kWave + lane];
        involved = any != 0;
    }
    MRCA_STAMP(2);      // clearance byte + field entry loaded, broad phase done
    const bool need = check_map && !(inside && clearance > hc);
    // the robots that need the walk, compacted; then ALL their patches in one or two memory round trips: (robot, word)
    // pairs are dealt round-robin to the block's threads, kPatchLoads independent loads per thread in flight (a Stage-2
    // world keeps ~30 of its 44 robots near walls, 30 words each -- round 2 fetched four robots per round trip)
    const unsigned long long need_mask = __ballot(need);             // identical in every wave
    const int n_need = __popcll(need_mask);
    if (wave == 0) {
        if (need) need_list[__popcll(need_mask & ((1ull << lane) - 1ull))] = lane;
        hit_flag[lane] = 0;
    }
    __syncthreads();
    {
        const int total = n_need * psize;
        const float inv_psize = 1.0f / (float)psize, inv_pwords = 1.0f / (float)pwords;
        for (int k0 = 0; k0 < total; k0 += kWave * kMoveWaves * kPatchLoads) {
            uint32_t val[kPatchLoads];
            int dst[kPatchLoads];
#pragma unroll
            for (int u = 0; u < kPatchLoads; ++u) {
                const int idx = k0 + u * kWave * kMoveWaves + tid;
                const bool on = idx < total;
                // idx / psize and k / pwords by float reciprocal + one correction step (all operands < 2^24)
                int q = on ? (int)((float)idx * inv_psize) : 0;
                q -= (q * psize > idx) ? 1 : 0;
                q += ((q + 1) * psize <= idx && on) ? 1 : 0;
                const int k = idx - q * psize;
                int r = (int)((float)k * inv_pwords);
                r -= (r * pwords > k) ? 1 : 0;
                r += ((r + 1) * pwords <= k) ? 1 : 0;
                const int wi = k - r * pwords;
                const int src = need_list[on ? q : 0];
                const int gy = __shfl(py0, src, kWave) + r;
                const int gw = __shfl(pw0, src, kWave) + wi;
                val[u] = 0u;
                dst[u] = on ? src * psize + k : -1;
                if (on && gy >= 0 && gy < e.g.height && gw >= 0 && gw < e.g.wpr) val[u] = e.map_bits[gy * e.g.wpr + gw];
            }
#pragma unroll
            for (int u = 0; u < kPatchLoads; ++u)
                if (dst[u] >= 0) mini[dst[u]] = val[u];
        }
    }
    __syncthreads();
    MRCA_STAMP(3);      // patches in LDS
    // the outline test, one lane per crossing EVENT of an edge's walk (the start cell, every x crossing, every y crossing: which
    // cell an event enters is a closed form, mrca_device.h walk_event_hits) -- rounds 1-4 walked: four lanes per robot, ~13
    // dependent steps of ~25 instructions each on a wave that issues one instruction per 4.2 cycles: 6 000 ticks whenever
    // any robot of the world was near a wall.  Tasks (robot in need, edge, event) are dealt to the block's threads in order.
    // (An edge has 2Q + 1 event slots -- 21 at 0.05 m cells -- and uses ~14: beyond ~20 robots near walls, a Stage-2 world, the
    // events are more work than the walks are long; measured 14.4 vs 12.7 us per Stage-2 launch, profiles/r05_s_*.  Such a world
    // walks, four lanes per robot, as before.)
    if (n_need * 4 * (2 * e.edge_slots + 1) <= kEventOutlineTasks) {
        const int Q = e.edge_slots, per_edge = 2 * Q + 1, per_robot = 4 * per_edge;
        const int total = n_need * per_robot;
        const float inv_per_robot = 1.0f / (float)per_robot, inv_per_edge = 1.0f / (float)per_edge;
        for (int base = 0; base < total; base += kWave * kMoveWaves) {
            const int task = base + tid;
            const bool act = task < total;
            // task / per_robot and rem / per_edge by float reciprocal + one correction step (all operands < 2^24)
            int rq = act ? (int)((float)task * inv_per_robot) : 0;
            rq -= (rq * per_robot > task) ? 1 : 0;
            rq += ((rq + 1) * per_robot <= task && act) ? 1 : 0;
            const int rem = act ? task - rq * per_robot : 0;
            int k = (int)((float)rem * inv_per_edge);
            k -= (k * per_edge > rem) ? 1 : 0;
            k += ((k + 1) * per_edge <= rem) ? 1 : 0;
            const int ev = rem - k * per_edge;
            const int src = need_list[rq];
            const float sx_ = __shfl(nx, src, kWave), sy_ = __shfl(ny, src, kWave);
            const float ss_ = __shfl(ns, src, kWave), sc_ = __shfl(nc, src, kWave);
            const int sy0 = __shfl(py0, src, kWave), sw0 = __shfl(pw0, src, kWave);
            if (act) {
                const MiniGrid mg{mini + src * psize, sy0, sw0, pwords};
                if (static_edge_event_hits(mg, e.g, sx_, sy_, ss_, sc_, k, ev, Q)) hit_flag[src] = 1;
            }
        }
    } else {
        for (int base = 0; base < n_need; base += kWave * kMoveWaves / 4) {
            const int q = base + (tid >> 2);
            const bool act = q < n_need;
            const int src = act ? need_list[q] : 0;
            const float sx_ = __shfl(nx, src, kWave), sy_ = __shfl(ny, src, kWave);
            const float ss_ = __shfl(ns, src, kWave), sc_ = __shfl(nc, src, kWave);
            const int sy0 = __shfl(py0, src, kWave), sw0 = __shfl(pw0, src, kWave);
            if (act) {
                const MiniGrid mg{mini + src * psize, sy0, sw0, pwords};
                if (static_edge_hit(mg, e.g, sx_, sy_, ss_, sc_, tid & 3)) hit_flag[src] = 1;
            }
        }
    }
    // fidelity mode: the outline of every PROVISIONAL pose, four lanes per robot (one per edge: a walk of ~5 raster cells),
    // all 64 robots of the world in one pass over the block -- next to the walks above, behind the same barrier.  (Rounds
    // 3-4: ONE lane walked the four edges of the robot taking its turn INSIDE the ordered pass, into a list in LDS, between
    // two workgroup barriers per turn: 181 us per launch on the Stage-1 worlds, 316 us on Stage-2.)
    uint2* bm_new = reinterpret_cast<uint2*>(inv_part + kMoveWaves * kWave);   // [64] bitmap of the provisional outline
    if (raster) {
        const int q = tid >> 2, k = tid & 3;
        const float qx = __shfl(nx, q, kWave), qy = __shfl(ny, q, kWave);
        const float qs = __shfl(ns, q, kWave), qc = __shfl(nc, q, kWave);
        const bool qmoving = __shfl(moving ? 1 : 0, q, kWave) != 0;
        uint32_t lo = 0u, hi = 0u;
        bool ok = true;
        if (qmoving)
            ok = outline_edge_bits(e.raster_inv, qx, qy, qs, qc, k, outline_anchor(qx, e.raster_inv),
                                   outline_anchor(qy, e.raster_inv), &lo, &hi);
        lo |= __shfl_xor(lo, 1, kWave);
        hi |= __shfl_xor(hi, 1, kWave);
        lo |= __shfl_xor(lo, 2, kWave);
        hi |= __shfl_xor(hi, 2, kWave);
        if (k == 0) bm_new[q] = make_uint2(lo, hi);
        if (!ok) atomicOr(e.status, kStatusOutlineWindow);     // never, for res >= 0.1 m (mrca_check reports it)
    }
    __syncthreads();
    if (wave != 0) return;      // the helpers are done; wave 0 carries the rest of the tick
    const bool shit = need && hit_flag[lane] != 0;
    MRCA_STAMP(4);      // outline walks done
    OutlineBits ob_new = ob_old;              // a robot that stands still keeps its outline
    if (raster && moving) {
        const uint2 b = bm_new[lane];
        ob_new = OutlineBits{outline_anchor(nx, e.raster_inv), outline_anchor(ny, e.raster_inv), b.x, b.y};
    }
    OutlineBits ob_cur = ob_old;              // follows the commits, like (x, y, s, c)

    // --- collision pass in robot order (Stage's sequential model loop)
    // committed pose of a robot that is not involved: moves unless the map stops it
    const float ox_ = x, oy_ = y, os_ = s, oc_ = c;  // pose at tick start
    bool moved = moving && !involved && !shit;
    if (moving && !involved) crashed = shit ? 1 : 0;
    if (moved) {
        x = nx;
        y = ny;
        th = nth;
        s = ns;
        c = nc;
        cellv = cellv_new;
        cellw = cellw_new;
        ob_cur = ob_new;
    }
    {
        unsigned long long turn = __ballot(involved);
        while (turn) {
            const int i = __ffsll((long long)turn) - 1;
            turn &= turn - 1;
            const float xi = fbcast(nx, i), yi = fbcast(ny, i), si = fbcast(ns, i), ci = fbcast(nc, i);
            // robots after i in the order have not moved yet when i is tested
            const bool later = lane > i;
            const float cx_ = later ? ox_ : x, cy_ = later ? oy_ : y, cs_ = later ? os_ : s, cc_ = later ? oc_ : c;
            bool ov;
            if (raster) {
                // robot i's provisional outline is broadcast like its pose; every lane intersects it, in registers, with the
                // outline of the pose it has at this point of the order
                const OutlineBits oi{ibcast(ob_new.ax, i), ibcast(ob_new.ay, i), (uint32_t)ibcast((int)ob_new.lo, i),
                                     (uint32_t)ibcast((int)ob_new.hi, i)};
                ov = valid && (lane != i) && outline_bits_meet(oi, later ? ob_old : ob_cur);
            } else {
                ov = valid && (lane != i) && obb_overlap(xi, yi, si, ci, cx_, cy_, cs_, cc_);
            }
            const unsigned long long m = __ballot(ov);
            if (lane == i && moving) {
                const bool hit = shit || (m != 0ull);
                if (!hit) {
                    x = nx;
                    y = ny;
                    th = nth;
                    s = ns;
                    c = nc;
                    cellv = cellv_new;
                    cellw = cellw_new;
                    moved = true;
                    ob_cur = ob_new;
                }
                crashed = hit ? 1 : 0;
            }
        }
    }

    MRCA_STAMP(5);      // ordered collision pass done
    // GT velocity = finite difference of the pose (stageros.cpp:585-590)
    const float vgt = moved ? fabsf(v) : 0.0f;
    const float wgt = moved ? w : 0.0f;

    // reward / terminal (stage_world1.py:180-211)
    const float ddx = gx - x, ddy = gy - y;
    const float dist = sqrtf(ddx * ddx + ddy * ddy);
    float rg = (pdist - dist) * kKProgress;
    const bool reach = dist < kGoalRadius;
    rg = reach ? kRArrive : rg;
    const bool crash = crashed == 1;
    const float rc = crash ? kRCrash : 0.0f;
    const float aw = fabsf(wgt);
    const float rw = (aw > e.w_thresh) ? kKOmega * aw : 0.0f;
    const bool tout = t > e.timeout;
    uint8_t result = reach ? 1 : 0;
    result = crash ? 2 : result;
    result = tout ? 3 : result;
    const bool done_now = reach || crash || tout;
    uint8_t lv = live ? 1 : 0;
    if (live) {
        reward = (rg + rc) + rw;
        done = done_now ? 1 : 0;
        res = result;
        pdist = dist;
        t = t + 1;
        if (done_now && first == 0) first = result;
    }

    // episode bookkeeping
    bool fresh = false;
    if (e.auto_reset == 1) {
        fresh = valid && live && done_now;
    } else if (e.auto_reset == 2) {
        if (live && done_now) lv = 0;
        for (int g = 0; g < e.num_groups; ++g) {
            const bool in = valid && (gid == g);
            const unsigned long long members = __ballot(in);
            const unsigned long long finished = __ballot(in && (done != 0));
            if (in && members == finished) fresh = true;
        }
    }
    float spv = v, spw = w, ovgt = vgt, owgt = wgt;
    if (MRCA_DBG(e, 32)) fresh = false;
    MRCA_STAMP(6);      // reward / terminal / group ballots done
    // new episodes, FOUR robots per round: group g = lane / 16 samples for the g-th lowest restarting robot of the round
    unsigned long long pending = __ballot(fresh);
    while (pending) {
        int sel[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sel[q] = pending ? __ffsll((long long)pending) - 1 : -1;
            pending &= pending - 1;        // (0 & anything = 0: stays empty once it is)
        }
        const int g = lane >> 4, sub = lane & 15;
        const int src = g == 0 ? sel[0] : g == 1 ? sel[1] : g == 2 ? sel[2] : sel[3];      // uniform within a group
        const bool act = src >= 0;
        const int srcl = act ? src : 0;
        // the robot's values come out of its lane (every lane of the wave takes part in the shuffles)
        const uint32_t nsrc = (uint32_t)(world * e.R + srcl);
        const uint32_t eps = (uint32_t)(__shfl(ep, srcl, kWave) + 1);
        const int rm = __shfl(rmode, srcl, kWave), gm = __shfl(gmode, srcl, kWave);
        const float cx_ = __shfl(x, srcl, kWave), cy_ = __shfl(y, srcl, kWave);
        const float six = __shfl(tix, srcl, kWave), siy = __shfl(tiy, srcl, kWave), sith = __shfl(tith, srcl, kWave);
        const float sgx = __shfl(tgx, srcl, kWave), sgy = __shfl(tgy, srcl, kWave);
        float px = six, py = siy, pth = wrap_angle(sith);      // table rows (mode 0) unless sampled
        group_sample_pose(g, sub, act && rm != 0, rm, nsrc, eps, e.key0, e.key1, cx_, cy_, &px, &py, &pth);
        float qx = sgx, qy = sgy;
        group_sample_goal(g, sub, act && gm != 0, gm, nsrc, eps, e.key0, e.key1, px, py, &qx, &qy);
        // ... and go back to it: the robot's own lane reads its group's results from the group's first lane
        const int gi = lane == sel[0] ? 0 : lane == sel[1] ? 1 : lane == sel[2] ? 2 : lane == sel[3] ? 3 : -1;
        const int from = 16 * (gi < 0 ? 0 : gi);
        const float rpx = __shfl(px, from, kWave), rpy = __shfl(py, from, kWave), rpth = __shfl(pth, from, kWave);
        const float rqx = __shfl(qx, from, kWave), rqy = __shfl(qy, from, kWave);
        if (gi >= 0) {
            ep = ep + 1;
            x = rpx;
            y = rpy;
            th = rpth;
            gx = rqx;
            gy = rqy;
            const float ex = rqx - rpx, ey = rqy - rpy;
            const float d0 = sqrtf(ex * ex + ey * ey);
            pdist = e.pre_dist_zero ? 0.0f : d0;
            e.init_pose[n * 3 + 0] = rpx;
            e.init_pose[n * 3 + 1] = rpy;
            e.init_pose[n * 3 + 2] = rpth;
            t = 1;
            crashed = 0;
            lv = 1;
            ovgt = owgt = 0.0f;
            if (!e.hold_velocity) spv = spw = 0.0f;   // hold_velocity: the odom twist survives the teleport
        }
    }
    // head records of the new poses, for every restarted robot AT ONCE: inside the loop above the field entry of each new
    // cell was a dependent global load per restart -- a world with three restarts in a tick waited three round trips,
    // and the launch lasts as long as its slowest world
    if (fresh) {
        sincos_det(th, &s, &c);
        rect_field.cell((int)floorf((x - e.g.x0) * e.g.inv_cell), (int)floorf((y - e.g.y0) * e.g.inv_cell), &cellv, &cellw);
    }
    if (raster) {   // ... and their outlines: lanes 0..3 walk the four edges of one restarted robot's new pose
        unsigned long long fm = __ballot(fresh);
        bool ok = true;
        while (fm) {
            const int src = __ffsll((long long)fm) - 1;
            fm &= fm - 1;
            const float fx_ = fbcast(x, src), fy_ = fbcast(y, src), fs_ = fbcast(s, src), fc_ = fbcast(c, src);
            const int ax = outline_anchor(fx_, e.raster_inv), ay = outline_anchor(fy_, e.raster_inv);
            uint32_t lo = 0u, hi = 0u;
            if (lane < 4) ok = outline_edge_bits(e.raster_inv, fx_, fy_, fs_, fc_, lane, ax, ay, &lo, &hi) && ok;
            lo |= __shfl_xor(lo, 1, kWave);
            hi |= __shfl_xor(hi, 1, kWave);
            lo |= __shfl_xor(lo, 2, kWave);
            hi |= __shfl_xor(hi, 2, kWave);
            const uint32_t lo0 = (uint32_t)ibcast((int)lo, 0), hi0 = (uint32_t)ibcast((int)hi, 0);
            if (lane == src) ob_cur = OutlineBits{ax, ay, lo0, hi0};
        }
        if (!ok) atomicOr(e.status, kStatusOutlineWindow);
    }

    MRCA_STAMP(7);      // restarts done
    if (valid) {
        e.pose[n * 3 + 0] = x;
        e.pose[n * 3 + 1] = y;
        e.pose[n * 3 + 2] = th;
        e.speed[n * 2 + 0] = spv;
        e.speed[n * 2 + 1] = spw;
        e.speed_gt[n * 2 + 0] = ovgt;
        e.speed_gt[n * 2 + 1] = owgt;
        e.goal[n * 2 + 0] = gx;
        e.goal[n * 2 + 1] = gy;
        e.prev_dist[n] = pdist;
        e.t[n] = t;
        e.reward[n] = reward;
        e.done[n] = done;
        e.result[n] = res;
        e.first_result[n] = first;
        e.crashed[n] = crashed;
        e.live[n] = lv;
        e.episode[n] = ep;
        e.fresh[n] = fresh ? 1 : 0;
        e.head[n] = make_float4(s, c, __uint_as_float(cellv), __uint_as_float(cellw));
        if (raster) e.outline[n] = ob_cur;
    }
    MRCA_STAMP(8);      // stores drained
    MRCA_LAUNCH_END(e);
}

__device__ __forceinline__ void write_head(const EnvView& e, int n, float x, float y, float th) {
    float s, c;
    sincos_det(th, &s, &c);
    const FreeRectField rect_field{e.free_rect, e.g.width, e.g.height, e.free_rect_pitch};
    uint32_t v0, v1;
    rect_field.cell((int)floorf((x - e.g.x0) * e.g.inv_cell), (int)floorf((y - e.g.y0) * e.g.inv_cell), &v0, &v1);
    e.head[n] = make_float4(s, c, __uint_as_float(v0), __uint_as_float(v1));
    if (e.raster_inv > 0.0f) {   // fidelity mode: the outline of the new pose
        OutlineBits o;
        if (!outline_bits(e.raster_inv, x, y, s, c, &o)) atomicOr(e.status, kStatusOutlineWindow);
        e.outline[n] = o;
    }
}

// head records from the poses as they are (mrca_create: before the first reset every robot sits at the origin)
__global__ void head_init_kernel(EnvView e) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < e.N) write_head(e, n, e.pose[n * 3 + 0], e.pose[n * 3 + 1], e.pose[n * 3 + 2]);
}

__global__ void reset_kernel(EnvView e, const uint8_t* __restrict__ mask, const float* __restrict__ poses,
                             const float* __restrict__ goals) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= e.N) return;
    const bool sel = mask ? (mask[n] != 0) : true;
    e.fresh[n] = sel ? 1 : 0;
    if (!sel) return;
    const int local = n % e.R;
    e.episode[n] += 1;
    float x, y, th, gx, gy, pd;
    begin_episode(e, n, local, e.pose[n * 3 + 0], e.pose[n * 3 + 1], &x, &y, &th, &gx, &gy, &pd,
                  poses ? poses + n * 3 : nullptr, goals ? goals + n * 2 : nullptr);
    e.pose[n * 3 + 0] = x;
    e.pose[n * 3 + 1] = y;
    e.pose[n * 3 + 2] = th;
    write_head(e, n, x, y, th);
    e.goal[n * 2 + 0] = gx;
    e.goal[n * 2 + 1] = gy;
    e.prev_dist[n] = pd;
    e.t[n] = 1;
    e.crashed[n] = 0;
    e.live[n] = 1;
    if (!e.hold_velocity) {
        e.speed[n * 2 + 0] = 0.0f;
        e.speed[n * 2 + 1] = 0.0f;
    }
    e.speed_gt[n * 2 + 0] = 0.0f;
    e.speed_gt[n * 2 + 1] = 0.0f;
    e.done[n] = 0;
    e.result[n] = 0;
    e.reward[n] = 0.0f;
    e.first_result[n] = 0;
}

// LDS |= without a return value (ds_or_b64): nothing to wait for
__device__ __forceinline__ void mask_or(unsigned long long* p, unsigned long long v) {
    (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// blockIdx -> robot: consecutive robots (one world's robots) share an XCD's L2 (block b runs on
// XCD b % 8, guide T1); a pure permutation, so correctness never depends on it.
// (unsigned arithmetic: b and N are never negative, and a signed % 8 and / 8 are eleven scalar instructions where three do)
__device__ __forceinline__ int block_to_robot(int b, int N) {
    const uint32_t ub = (uint32_t)b, un = (uint32_t)N;
    if (un & 7u) return b;
    return (int)((ub & 7u) * (un >> 3) + (ub >> 3));
}

// The march reads the free-rectangle field straight from its L1/L2-resident global copy: ~2 dependent
// lookups per ray.  (Staging a tile of it in LDS per robot was measured slower at every granularity tried,
// the tile costs more to fill than the few lookups it serves.  So were persistent workgroups
// walking several robots each -- 39 vs 37 us, profiles/r01/r01_ad_ablation.txt -- and nontemporal stores made
// no difference.)  Marching the K beams of a thread in LOCK STEP (grid_march_skip_n: K lookups in flight per wait) is
// implemented and measured too: slower than one after the other (34.7 vs 28.1 us, profiles/r02/r02_c_*), see mrca_abi.hip.
// RKW > 0: fidelity mode's lidar (the other robots seen through the collision raster; RKW = cells per side of an outline's
// window, 4 or 8) -- a kernel of its own so that the default one does not carry the code: with the raster path behind a
// run-time branch the default launch was 0.85 us slower (A/B on one box, profiles/r04_h_ab_raster_path_in_default_kernel.txt:
// twice the instructions for the same instruction cache).  Since round 5 a beam's return from another robot's outline is a
// closed form over that robot's 16-byte outline record (ray_outline_entry) instead of a walk through a window of LDS bits.
template <int K, bool BIG, bool SEQ, int RKW, bool VIEWS>
__device__ __forceinline__ void raycast_body(int only_fresh, int ray_first, int ray_count, int R_, const float* __restrict__ pose_p,
                                             const float4* __restrict__ head_p, const float* __restrict__ bcos_p,
                                             const float* __restrict__ bsin_p, uint8_t* ring_head_p, const EnvView& e, int views);

// VIEWS: the epilogue also forms MRCA_F_SCAN / MRCA_F_OBS (lazy_obs = 0) -- an instantiation of its own, so that the default
// kernel carries nothing of it (as a run-time branch it put 22 scalar instructions per wave into every launch)
template <int K, bool BIG, bool SEQ, int RKW = 0, bool VIEWS = false>
__global__ __launch_bounds__(1024, (RKW == 4 ? 8 : 1)) void raycast_kernel(int only_fresh, int ray_first, int ray_count, int R_,
                                                       const float* __restrict__ pose_p, const float4* __restrict__ head_p,
                                                       const float* __restrict__ bcos_p, const float* __restrict__ bsin_p,
                                                       uint8_t* ring_head_p, const EnvView* __restrict__ view_p, RayIn in) {
    EnvView e = *view_p;        // (the env's view from device memory, see move_kernel; this launch's slot: what it reads of it)
    e.pose = const_cast<float*>(pose_p);
    e.head = const_cast<float4*>(head_p);
    e.goal = const_cast<float*>(in.goal);
    e.fresh = const_cast<uint8_t*>(in.fresh);
    e.outline = const_cast<OutlineBits*>(in.outline);
#if defined(MRCA_PROFILING)
    e.launch_stamps = in.launch_stamps;
    e.launch_slot = in.launch_slot;
#endif
    MRCA_LAUNCH_BEGIN(e);
    raycast_body<K, BIG, SEQ, RKW, VIEWS>(only_fresh, ray_first, ray_count, R_, pose_p, head_p, bcos_p, bsin_p, ring_head_p, e, in.views);
    MRCA_LAUNCH_END(e);
#if defined(MRCA_PROFILING)
    // debug flag 128: every workgroup casts its robot's beams a SECOND time inside the same launch -- everything it touches
    // is in its L1 / L2 by then, there is no launch boundary in between, no kernel argument to wait for: what a tick's ray
    // cast would cost inside a persistent kernel that never leaves the chip.  (The second pass writes another ring slot: the
    // results of a launch under this flag are for the clock only.)  tools/hot_pass_probe.py, DESIGN.md 10.2.
    if (MRCA_DBG(e, 128)) {
        __syncthreads();
        raycast_body<K, BIG, SEQ, RKW, VIEWS>(only_fresh, ray_first, ray_count, R_, pose_p, head_p, bcos_p, bsin_p, ring_head_p, e, in.views);
    }
#endif
}

template <int K, bool BIG, bool SEQ, int RKW, bool VIEWS>
__device__ __forceinline__ void raycast_body(int only_fresh, int ray_first, int ray_count, int R_, const float* __restrict__ pose_p,
                                             const float4* __restrict__ head_p, const float* __restrict__ bcos_p,
                                             const float* __restrict__ bsin_p, uint8_t* ring_head_p, const EnvView& e, int views) {
    // (the leading arguments repeat e.ray_first, e.ray_count, e.R, e.pose, e.head, e.beam_cos, e.beam_sin, e.ring_head: 14
    // dwords preloaded into SGPRs, see move_kernel)
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    MRCA_RSTAMP(0);
    const int n = ray_first + block_to_robot(blockIdx.x, ray_count);
    const int tid = threadIdx.x;
    // (A variant of this kernel without the early exit -- so that nothing is waited for before every request of the
    // workgroup is out -- was measured and changed nothing: 27.96 us either way, profiles/r03/r03_l_bench_env.json.  The
    // launch uses ~45 % of the VALU issue slots: what it waits for is a workgroup's chain of dependent round trips at full
    // occupancy, and one early exit less does not shorten that chain -- DESIGN.md 5.2.)
    if (only_fresh && e.fresh[n] == 0) return;  // block-uniform

    float4* nb = reinterpret_cast<float4*>(lds);
    int2* nbi = reinterpret_cast<int2*>(nb + kWave);
    int* nb_count = reinterpret_cast<int*>(nbi + kWave);
    unsigned long long* nbmask = reinterpret_cast<unsigned long long*>(nb_count + 4);   // [B] neighbours per beam
    int* nb_more = nb_count + 1;                                      // big worlds: another chunk of neighbours follows
    // fidelity mode (never in big worlds): the neighbours' outline records
    constexpr bool raster = RKW > 0 && !BIG;
    int4* nbo = reinterpret_cast<int4*>(nbmask + e.B);   // [64] OutlineBits as (ax, ay, lo, hi)

    const int T = e.B / K;                    // marching threads
    const bool extra = (int)blockDim.x > T;   // a dedicated preparation wave sits behind the marching ones
    const int prep_base = extra ? T : 0;
    const bool is_prep = tid >= prep_base && tid < prep_base + kWave;   // wave-uniform
    const bool marches = tid < T;                                       // wave-uniform
    // n / R without the ~25 scalar instructions of a 32-bit division: small worlds (R <= 64, n < 2^24) take the exact multiply-high
    // with the host's ceil(2^32 / R) (error n x (m R - 2^32) < 2^24 x 64 < 2^32).  (A scalar instruction costs the launch twice
    // what a vector one costs -- one scalar unit per CU for 32 waves: profiles/r06_ac_*.)
    const int world = BIG ? n / R_ : (R_ == 1 ? n : (int)__umulhi((uint32_t)n, e.r_magic));   // (2^32 / 1 does not fit the magic)
    const int local = n - world * R_;
    // the robot's own record: pose, sin / cos and the field entry of its cell.  Block-uniform -- but fetched with VECTOR
    // loads (the index goes through an opaque zero): as scalar loads they shared the out-of-order scalar counter with
    // the kernel arguments, and the neighbour candidate below could not be requested before they were back.
    // (Not in big worlds: there the neighbour enumeration hashes the robot's cell per chunk, wave-uniform work that
    // belongs on the scalar unit -- measured: 513 vs 492 us per 50 000-robot launch, profiles/r03/r03_s_bigworld_shards8.jsonl.)
    int lane_zero = 0;
    if constexpr (!BIG) asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
    const int nv = n + lane_zero;
    const float x = pose_p[nv * 3 + 0], y = pose_p[nv * 3 + 1];
    const float4 hd = head_p[nv];
    const float s = hd.x, c = hd.y;
    // the preparation wave requests "its" neighbour candidate in the same memory round trip
    const int pl = tid - prep_base;
    const bool cand = !BIG && is_prep && (pl < R_) && (pl != local);
    const int jn = world * R_ + (cand ? pl : local);
    float xj = 0.0f, yj = 0.0f;
    float4 hj = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    int4 oj = make_int4(0, 0, 0, 0);
    if (!BIG && is_prep) {
        xj = pose_p[jn * 3 + 0];
        yj = pose_p[jn * 3 + 1];
        if constexpr (raster) oj = reinterpret_cast<const int4*>(e.outline)[jn];
        else hj = head_p[jn];
    }
    // beam directions in the robot frame for this thread's K beams (tid + k*T)
    float bc[K], bs[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int b = (marches ? tid : 0) + k * T;
        bc[k] = bcos_p[b];
        bs[k] = bsin_p[b];
    }
    // slot of the newest frame so far (read by every thread BEFORE the first barrier, advanced by thread 0 after it) and
    // the fresh flag: requested last, used last
    const uint8_t fresh_byte = e.fresh[n];
    const int ring_slot = ring_head_p[n];
    // (The frame stack -- ppo_stage1.py:87-89: popleft / append -- is a ring of raw scans: only the newest one is written,
    // into the slot behind the previous newest one; see materialize_kernel.)
    // big worlds: the candidates come from the lidar hash (3 x 3 cells of 6.5 m around the robot's cell) and may
    // exceed the 64 a chunk holds: the preparation wave walks the nine bucket ranges 64 entries at a time and hands
    // the marching threads one chunk of <= 64 neighbours per barrier pair (see the chunk loop below).
    // The nine ranges are ONE list to it: lanes 0..8 fetch "their" cell's range in the same memory round trip, a prefix sum
    // over those lanes numbers the entries, and a batch of 64 takes entries off.. off + 63 of that list whichever cells they
    // belong to.  (Rounds 2-3 walked the cells one after the other -- range, entry, pose, head: four dependent round trips
    // per cell, 36 per workgroup, and a workgroup lived 25 us whatever else the chip was doing:
    // profiles/r04_o_slice_sweep.txt.  Now it is five.)
    int big_off = 0;                   // enumeration state of the preparation wave (wave-uniform): entries consumed so far
    auto big_chunk = [&]() {
        const int icx = hash_cell_coord(x, kLidarCell), icy = hash_cell_coord(y, kLidarCell);
        for (int b = pl; b < e.B; b += kWave) nbmask[b] = 0ull;
        int cell_start = 0, cell_count = 0;
        if (pl < 9) {
            const uint32_t h = hash_cell(icx + pl % 3 - 1, icy + pl / 3 - 1, world) & (uint32_t)e.bw_lmask;
            cell_start = e.bw_lstart[h];
            cell_count = e.bw_lstart[h + 1] - cell_start;
        }
        int cell_end = cell_count;         // inclusive prefix sum over lanes 0..8 (lanes >= 9 hold zeros)
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            const int up = __shfl_up(cell_end, d, kWave);
            if (pl >= d) cell_end += up;
        }
        const int total = __shfl(cell_end, 8, kWave);
        int cnt = 0;
        while (big_off < total) {
            const int entry = big_off + pl;
            int q = 0;                     // the cell entry falls into: the number of cells ending at or before it
#pragma unroll
            for (int t = 0; t < 8; ++t) q += __shfl(cell_end, t, kWave) <= entry ? 1 : 0;
            const bool valid = entry < total;          // then q <= 8
            q = valid ? q : 0;
            const int qx = icx + q % 3 - 1, qy = icy + q / 3 - 1;
            const int q_end = __shfl(cell_end, q, kWave), q_count = __shfl(cell_count, q, kWave);
            const int idx = __shfl(cell_start, q, kWave) + (entry - (q_end - q_count));
            const int j = valid ? e.bw_lsorted[idx] : -1;
            bool keep = false;
            float cxj = 0.0f, cyj = 0.0f;
            float4 chj = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            int lo = 0, hi = -1;
            if (j >= 0 && j != n && j / R_ == world) {
                cxj = pose_p[j * 3 + 0];
                cyj = pose_p[j * 3 + 1];
                // a bucket may hold other cells too (hash collisions) and the same bucket may serve two of the nine
                // cells: a robot counts only for the cell it really is in, so nobody is listed twice
                if (hash_cell_coord(cxj, kLidarCell) == qx && hash_cell_coord(cyj, kLidarCell) == qy) {
                    const float ddx = cxj - x, ddy = cyj - y;
                    if (ddx * ddx + ddy * ddy <= kLidarReach2) {
                        beam_interval(ddx * c + ddy * s, ddy * c - ddx * s, e.B, e.beam_step, e.beam_inv_step, e.lidar_radius,
                                      e.lidar_near, &lo, &hi);
                        keep = lo <= hi;
                        if (keep) chj = head_p[j];
                    }
                }
            }
            const unsigned long long m = __ballot(keep);
            const int add = __popcll(m);
            if (cnt + add > kWave) break;          // this batch opens the next chunk
            if (keep) {
                const int idx2 = cnt + __popcll(m & ((1ull << pl) - 1ull));
                float olx, oly;         // the lidar's origin in the neighbour's frame: once per neighbour, not per beam
                ray_box_origin(x, y, cxj, cyj, chj.x, chj.y, &olx, &oly);
                nb[idx2] = make_float4(olx, oly, chj.x, chj.y);
                nbi[idx2] = make_int2(lo, hi);
            }
            cnt += add;
            big_off += kWave;
        }
        if (pl == 0) {
            *nb_count = MRCA_DBG(e, 1) ? 0 : cnt;
            *nb_more = big_off < total ? 1 : 0;
        }
        for (int k = 0; k < cnt; ++k) {
            const int2 iv = nbi[k];
            for (int b = iv.x + pl; b <= iv.y; b += kWave) mask_or(&nbmask[b], 1ull << k);
        }
    };
    MRCA_RSTAMP(1);     // robot record, beam table, neighbour candidate requested (debug flag 64: arrived)
    if (is_prep) {
      if constexpr (BIG) {
        big_chunk();
      } else {
        for (int b = pl; b < e.B; b += kWave) nbmask[b] = 0ull;
        const float ddx = xj - x, ddy = yj - y;
        // conservative cull: a hit below 6 m needs the centre within 6 + circumradius(0.2907) m (fidelity mode: + one
        // raster-cell diagonal -- what is tested there are the robot's outline CELLS)
        bool keep = cand && (ddx * ddx + ddy * ddy <= e.lidar_reach2);
        int lo = 0, hi = -1;
        if (keep) {
            beam_interval(ddx * c + ddy * s, ddy * c - ddx * s, e.B, e.beam_step, e.beam_inv_step, e.lidar_radius, e.lidar_near,
                          &lo, &hi);
            keep = lo <= hi;
        }
        const unsigned long long m = __ballot(keep);
        if (keep) {
            const int idx = __popcll(m & ((1ull << pl) - 1ull));
            // the slab tests want the lidar's origin in the neighbour's frame (once per neighbour, not per beam); the
            // fidelity mode's closed form wants the neighbour's outline record
            if constexpr (raster) {
                nbo[idx] = oj;
            } else {
                float olx, oly;
                ray_box_origin(x, y, xj, yj, hj.x, hj.y, &olx, &oly);
                nb[idx] = make_float4(olx, oly, hj.x, hj.y);
            }
            nbi[idx] = make_int2(lo, hi);
        }
        const int cnt0 = MRCA_DBG(e, 1) ? 0 : __popcll(m);
        if (pl == 0) *nb_count = cnt0;
        // scatter: bit k of nbmask[b] = "neighbour k can touch beam b".  ds_or_b64 without a return value: the wave
        // fires one per neighbour and moves on (as a read-modify-write every neighbour cost an LDS round trip, ~2 000
        // of the 5 600 ticks wave 0 spent preparing, profiles/r03/r03_k_ablate_raycast_phase_stamps.txt).
        for (int k = 0; k < cnt0; ++k) {
            const int2 iv = nbi[k];
            for (int b = iv.x + pl; b <= iv.y; b += kWave) mask_or(&nbmask[b], 1ull << k);
        }
      }
    }
    MRCA_RSTAMP(2);     // wave 0: neighbour list and per-beam masks built
    // --- the march: K beams per thread in lock step
    float dx[K], dy[K], rng[K];
    bool from_robot[K];        // the range is a return from another robot (ranger_return 0.5: LaserScan intensity 0)
#pragma unroll
    for (int k = 0; k < K; ++k) {
        dx[k] = c * bc[k] - s * bs[k];
        dy[k] = s * bc[k] + c * bs[k];
        rng[k] = kRangeMax;
        from_robot[k] = false;
    }
    if (marches && !MRCA_DBG(e, 2)) {
        const FreeRectField field{e.free_rect, e.g.width, e.g.height, e.free_rect_pitch};
        MarchOrigin org;                       // once per robot: shared by all its beams
        org.fx = (x - e.g.x0) * e.g.inv_cell;
        org.fy = (y - e.g.y0) * e.g.inv_cell;
        org.ix0 = (int)floorf(org.fx);
        org.iy0 = (int)floorf(org.fy);
        org.v_lo = __float_as_uint(hd.z);
        org.v_hi = __float_as_uint(hd.w);
        // Big worlds are open worlds (scenario.circle_big: the map is a token patch at the origin, the robots stand
        // kilometres from it), and outside the map the field knows nothing: a ray crawls from cell to cell, one dependent
        // lookup of the zero border each -- 40 000 of the 54 000 ticks a workgroup of the 50 000-robot circle lived
        // (profiles/r04_t_bigworld_raycast_phases.txt), to return kRangeMax.  A robot whose 6 m cannot reach the map's
        // bounding box (two cells of slack for the roundings of fx / fy) has nothing to march through: every beam is
        // kRangeMax exactly as the march would return it.
        bool map_in_reach = true;
        if constexpr (BIG) {
            const float reach = kRangeMax * e.g.inv_cell + 2.0f;
            map_in_reach = org.fx + reach >= 0.0f && org.fx - reach <= (float)e.g.width && org.fy + reach >= 0.0f &&
                           org.fy - reach <= (float)e.g.height;
        }
        if (map_in_reach) {
            if constexpr (K == 1 || SEQ) {   // one ray at a time: the hand-tuned single-ray loop (54 VALU per jump)
#pragma unroll
                for (int k = 0; k < K; ++k) rng[k] = grid_march_skip(field, e.g, org, dx[k], dy[k], kRangeMax);
            } else {                         // K rays in lock step: K lookups in flight per wait
                grid_march_skip_n<K>(field, e.g, org, dx, dy, kRangeMax, rng);
            }
        }
    }
    MRCA_RSTAMP(3);     // this wave's beams marched
    __syncthreads();  // neighbour list ready (the preparation wave built it while the others marched)
    MRCA_RSTAMP(4);     // through the barrier / past the flag
    if constexpr (!BIG) {
        if (!marches) return;  // the dedicated preparation wave is done (whole wave: the barrier below counts live waves)
    }
    for (;;) {
        const int cnt = *nb_count;
        const int more = BIG ? *nb_more : 0;
        if (marches) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int b = tid + k * T;
                float r = rng[k];
                unsigned long long m = cnt > 0 ? nbmask[b] : 0ull;
                if constexpr (raster) {
                    // fidelity mode: the entry time of the first raster cell of each flagged neighbour's outline the beam's
                    // walk visits -- in closed form, the same times grid_march's walk over the raster would compare
                    if (m) {
                        const float fxr = x * e.raster_inv, fyr = y * e.raster_inv;
                        const int ixr = (int)floorf(fxr), iyr = (int)floorf(fyr);
                        const float tmax_c = kRangeMax * e.raster_inv;
                        const float inv_dx = dx[k] != 0.0f ? rcp_exact(dx[k]) : kInf;
                        const float inv_dy = dy[k] != 0.0f ? rcp_exact(dy[k]) : kInf;
                        // (the 4 x 4 form: an axis the ray never steps along gets origin -inf, see ray_outline_entry4)
                        const float fxe = dx[k] != 0.0f ? fxr : -kInf, fye = dy[k] != 0.0f ? fyr : -kInf;
                        const bool xpos = dx[k] > 0.0f, ypos = dy[k] > 0.0f;
                        do {
                            const int q = __ffsll((long long)m) - 1;
                            m &= m - 1;
                            const int4 oq = nbo[q];
                            const OutlineBits ob{oq.x, oq.y, (uint32_t)oq.z, (uint32_t)oq.w};
                            const float tc = RKW == 4 ? ray_outline_entry4(fxe, fye, ixr, iyr, xpos, ypos, inv_dx, inv_dy, ob)
                                                      : ray_outline_entry<RKW>(fxr, fyr, ixr, iyr, dx[k], dy[k], inv_dx, inv_dy, ob);
                            const float t = tc < tmax_c ? tc * e.raster_res : kInf;
                            from_robot[k] = from_robot[k] || t < r;
                            r = t < r ? t : r;
                        } while (m);
                    }
                } else {
                    while (m) {
                        const int q = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        const float4 nbq = nb[q];
                        const float t = ray_box_local(nbq.x, nbq.y, dx[k], dy[k], nbq.z, nbq.w);
                        from_robot[k] = from_robot[k] || t < r;
                        r = t < r ? t : r;
                    }
                }
                rng[k] = r;
            }
        }
        if (!more) break;
        __syncthreads();          // everybody is through with this chunk ...
        if (is_prep) big_chunk();
        __syncthreads();          // ... and the next one is ready
    }
    if (!marches) return;
    MRCA_RSTAMP(5);     // neighbour slab tests done
    // --- the scan (stageros.cpp:479-516) goes into the ring slot behind the newest one -- ONE store stream: the
    //     observation x / 6 - 0.5 (stage_world1.py:140) and the deque order (ppo_stage1.py:59-60,87-89) are the readers'
    //     business (materialize_kernel).  Every thread stores its own beams -- lane l of a wave holds beam base + l, so
    //     each store instruction of a wave covers 256 contiguous bytes.
    {
        // (row = n x F fits 32 bits -- n < 2^24, F <= 8 --: one 32 x 32 -> 64 multiply per address instead of a 64 x 32 chain of ten)
        const uint32_t row = (uint32_t)n * (uint32_t)e.F;
        float* ring_row = e.scan_ring + (size_t)row * (uint32_t)e.B;
        const int words = e.B >> 6;
        unsigned long long* hit_row = e.hit_bits + (size_t)row * (uint32_t)words;
        const int new_slot = ring_slot + 1 == e.F ? 0 : ring_slot + 1;
        const bool fresh = __builtin_amdgcn_readfirstlane((int)fresh_byte) != 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int b = tid + k * T;
            // (|x|: fidelity mode's closed form may return -0.0 for a beam that starts inside a marked cell)
            const float r = fabsf(rng[k] < kRangeMax ? rng[k] : kRangeMax);
            // what the beam hit: one bit per beam beside the ring (MRCA_F_HIT_BITS), set = another robot.  stageros casts
            // Stage's return value to uint8 for LaserScan.intensities (stageros.cpp:506): 1 floorplan, 0 robot or miss.  A wave
            // holds 64 consecutive beams (T is a multiple of 64: product_ray_shift), so its ballot IS the row's word b >> 6.
            // (ABI 4-5 kept the flag in the range's sign bit: a reader that forgot |x| got negative ranges.)
            const unsigned long long hm = __ballot(from_robot[k] && rng[k] < kRangeMax);
            const bool word_lane = (tid & (kWave - 1)) == 0;
            // (nontemporal stores: the launch does not read its rows again.  Measured A/B on one box, round 4: 21.7 us with
            // them, 22.9 us with plain stores (profiles/r04_g_ab_nontemporal_row_stores.txt) -- although FETCH_SIZE does not
            // move, 2502 vs 2504 KiB: what they relieve is the write path, not the free-rectangle field's residency)
            if (fresh) {     // deque([obs] * F), ppo_stage1.py:59-60: every slot, the head stays where it is
                for (int f = 0; f < e.F; ++f) {
                    __builtin_nontemporal_store(r, &ring_row[f * e.B + b]);
                    if (word_lane) hit_row[f * words + (b >> 6)] = hm;
                }
            } else {
                __builtin_nontemporal_store(r, &ring_row[new_slot * e.B + b]);
                if (word_lane) hit_row[new_slot * words + (b >> 6)] = hm;
            }
            // lazy_obs = 0: the two reference-shaped views of this robot, formed here instead of by a materialize_kernel launch
            // behind every ray cast (get_laser_observation, stage_world1.py:127-141: the newest scan; the stack in deque order,
            // x / 6 - 0.5) -- the same numbers: norm_obs of the ring's rows, the older ones as earlier launches stored them.  The
            // launch uses 5 % of HBM: the 8 kB per robot ride along (131 -> ... M agent-steps/s for a reference-shaped caller).
            if (VIEWS && views) {
                const float nr = norm_obs(r);
                // (nontemporal like the ring row: the launch does not read them again)
                if (views & 1) __builtin_nontemporal_store(r, &e.scan[(size_t)n * (uint32_t)e.B + b]);
                if (views & 2) {
                    float* dst = e.obs + (size_t)row * (uint32_t)e.B + b;
                    if (fresh) {
                        for (int f = 0; f < e.F; ++f) __builtin_nontemporal_store(nr, &dst[f * e.B]);
                    } else {
                        int slot = new_slot;
                        for (int f = 0; f < e.F - 1; ++f) {        // oldest first: the slot behind the newest, and on round the ring
                            slot = slot + 1 == e.F ? 0 : slot + 1;
                            __builtin_nontemporal_store(norm_obs(fabsf(ring_row[slot * e.B + b])), &dst[f * e.B]);
                        }
                        __builtin_nontemporal_store(nr, &dst[(e.F - 1) * e.B]);
                    }
                }
            }
        }
        if (tid == 0 && !fresh) ring_head_p[n] = (uint8_t)new_slot;
    }
    if (tid == 0) {  // get_local_goal (stage_world1.py:155-160)
        const float gx = e.goal[n * 2 + 0] - x, gy = e.goal[n * 2 + 1] - y;
        e.local_goal[n * 2 + 0] = gx * c + gy * s;
        e.local_goal[n * 2 + 1] = gy * c - gx * s;
    }
    MRCA_RSTAMP(6);     // stores issued (debug flag 64: acknowledged)
}


// ================================================================================================================
// Worlds with more than 64 robots (a single 500 / 50 000-robot circle): the tick of move_kernel
// spread over per-robot threads.  Same rules, same arithmetic, same ORDER: robots are tested one after another in
// index order against the poses the others have at that point (Stage's sequential model loop).
//
//   bw_integrate  thread per robot: latch, integrate, outline-vs-grid test, provisional pose -> bw_prov; both the
//                 pose at tick start and the provisional pose go into the collision hash (0.7 m cells, chained).
//   bw_collide    thread per robot, the ordered pass as DEPENDENCY ROUNDS: robot i looks at the 3 x 3 cells around
//                 its provisional centre; anybody listed there within 2 x circumradius can matter.  i is decided once
//                 every such lower-indexed moving robot is (their outcome tells which of their two poses counts);
//                 then it runs the SAT tests and publishes its own outcome (an agent-scope atomic on bw_state).  The
//                 lowest undecided robot never waits, and robot blocks are handed to workgroups by TICKET in the order
//                 they start, so every lower-indexed robot is held by a workgroup that is already running or done: the
//                 loop terminates whatever order the hardware dispatches workgroups in (a bounded wait + status word
//                 stay as a belt).  Robots with nobody in reach decide in their first round.
//   bw_finish     thread per robot: commit, GT velocity, reward / terminal, episode bookkeeping (per-robot resets),
//                 head record.
//   bw_lidar_*    counting sort of the FINAL poses into the lidar hash (6.5 m cells) the ray cast enumerates.
// ================================================================================================================
constexpr int kFlagMoving = 1, kFlagStaticHit = 2, kFlagLive = 4;

__global__ void bw_integrate_kernel(EnvView e, const float* __restrict__ actions) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n == 0) *e.bw_ticket = 0;          // the collision pass that follows hands out its robot blocks by ticket
    if (n >= e.N) return;
    const float x = e.pose[n * 3 + 0], y = e.pose[n * 3 + 1], th = e.pose[n * 3 + 2];
    const float4 hd = e.head[n];
    const bool live = e.live[n] != 0;
    const float v = live ? sane_cmd(actions[n * 2 + 0]) : (e.hold_velocity ? e.speed[n * 2 + 0] : 0.0f);
    const float w = live ? sane_cmd(actions[n * 2 + 1]) : (e.hold_velocity ? e.speed[n * 2 + 1] : 0.0f);
    const float s = hd.x, c = hd.y;
    const float d = v * kDt;
    const float nx = x + d * c;
    const float ny = y + d * s;
    const float nth = wrap_angle(th + w * kDt);
    float ns, nc;
    sincos_det(nth, &ns, &nc);
    const bool moving = (v != 0.0f) || (w != 0.0f);
    // outline-vs-grid test: free for sure when the footprint's patch lies outside the map (cells outside are
    // free) or the distance field clears it; otherwise the four outline edges are walked in the bitmap
    const int hc = e.foot_hc;
    const int pix = (int)floorf((nx - e.g.x0) * e.g.inv_cell);
    const int piy = (int)floorf((ny - e.g.y0) * e.g.inv_cell);
    const bool touches = pix + hc >= 0 && piy + hc >= 0 && pix - hc < e.g.width && piy - hc < e.g.height;
    const bool inside = pix >= 0 && piy >= 0 && pix < e.g.width && piy < e.g.height;
    bool shit = false;
    if (touches && !MRCA_DBG(e, 8)) {
        const bool clear = inside && e.cellfield[(size_t)piy * e.g.width + pix] > hc;
        if (!clear) shit = static_hit(GlobalGrid{e.map_bits, e.g.width, e.g.height, e.g.wpr}, e.g, nx, ny, ns, nc);
    }
    const int flags = (moving ? kFlagMoving : 0) | (shit ? kFlagStaticHit : 0) | (live ? kFlagLive : 0);
    e.bw_prov[2 * n + 0] = make_float4(nx, ny, nth, __int_as_float(flags));
    e.bw_prov[2 * n + 1] = make_float4(ns, nc, v, w);
    e.bw_state[n] = moving ? 0 : 1;
    const int world = n / e.R;
    {
        const uint32_t h = hash_cell(hash_cell_coord(x, kCollideCell), hash_cell_coord(y, kCollideCell), world) &
                           (uint32_t)e.bw_cmask;
        e.bw_cnext[2 * n] = atomicExch(&e.bw_chead[h], 2 * n);
    }
    if (moving) {
        const uint32_t h = hash_cell(hash_cell_coord(nx, kCollideCell), hash_cell_coord(ny, kCollideCell), world) &
                           (uint32_t)e.bw_cmask;
        e.bw_cnext[2 * n + 1] = atomicExch(&e.bw_chead[h], 2 * n + 1);
    }
}

__global__ void bw_collide_kernel(EnvView e) {
    // A workgroup takes the next block of robots when it STARTS (a ticket), not by its blockIdx: the robots below any
    // robot of this workgroup then belong to workgroups that have already started -- they are resident or finished,
    // never waiting to be dispatched -- so a robot only ever waits for waves that are running.  (Rounds 2-3 mapped
    // blockIdx -> robots and relied on gfx950 dispatching workgroups in index order: observed, not guaranteed.)
    __shared__ int block_ticket;
    if (threadIdx.x == 0) block_ticket = (int)atomicAdd(e.bw_ticket, 1u);
    __syncthreads();
    const int n = block_ticket * blockDim.x + threadIdx.x;
    if (n >= e.N) return;
    const float4 p0 = e.bw_prov[2 * n], p1 = e.bw_prov[2 * n + 1];
    const int flags = __float_as_int(p0.w);
    if (!(flags & kFlagMoving)) return;               // stays where it is; its stall flag is left alone
    const float nx = p0.x, ny = p0.y, ns = p1.x, nc = p1.y;
    const int world = n / e.R;
    const int icx = hash_cell_coord(nx, kCollideCell), icy = hash_cell_coord(ny, kCollideCell);
    // the nine bucket heads are independent loads: all in flight at once (one dependent round trip instead of nine)
    int head9[9];
#pragma unroll
    for (int q = 0; q < 9; ++q)
        head9[q] = e.bw_chead[hash_cell(icx + q % 3 - 1, icy + q / 3 - 1, world) & (uint32_t)e.bw_cmask];
    // Dependency rounds.  A robot only ever waits for LOWER-indexed robots, and the lowest undecided robot of the
    // launch never waits, so somebody can always make progress: every lower-indexed robot sits in a workgroup that took
    // its ticket earlier, i.e. one that is resident (its waves are scheduled: a waiting wave sleeps, it does not starve
    // the others) or already done.  The guard below stays as a belt: should a wait ever run out, the launch ends
    // instead of hanging the GPU, the robot stays undecided (treated as not moved) and bit 0 of the env's status word
    // is raised -- mrca_check() turns it into MRCA_ERR_HIP, never a silent wrong state.  The atomic store of the
    // outcome sits inside the loop by construction (done-flag form), not by grace of the optimiser.
    bool done = false;
    for (int guard = 0; !done && guard < (1 << 22); ++guard) {
        bool ready = true;
        bool hit = (flags & kFlagStaticHit) != 0;
        if (!MRCA_DBG(e, 16)) {
            // The nine chains are walked in LOCK STEP, one LEVEL per pass: the link and the centre of the entry each chain
            // stands at are 27 independent loads, in flight together; what follows an entry within reach (rare outside a
            // jam) is the slow path below.  (Walked one chain after the other, entry by entry, a robot with nobody near it
            // still paid a dependent round trip for each of the ~9 entries hash collisions and its own two poses put into
            // its buckets: 31 us per 50 000-robot launch, profiles/r04_r_bigworld_kernel_stats.csv.)
            int cur[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) cur[q] = head9[q];
            for (;;) {
                int nxt[9];
                float cx[9], cy[9];
                bool any = false;
#pragma unroll
                for (int q = 0; q < 9; ++q) {
                    const int en = cur[q];
                    nxt[q] = -1;
                    cx[q] = cy[q] = 0.0f;
                    if (en >= 0) {
                        any = true;
                        const int j = en >> 1;
                        // the centre this entry stands for: j's provisional pose (odd entries) or its pose at tick start
                        const float* ctr = (en & 1) ? reinterpret_cast<const float*>(e.bw_prov + 2 * j) : e.pose + 3 * j;
                        nxt[q] = e.bw_cnext[en];
                        cx[q] = ctr[0];
                        cy[q] = ctr[1];
                    }
                }
                if (!any) break;
#pragma unroll
                for (int q = 0; q < 9; ++q) {
                    const int en = cur[q];
                    cur[q] = nxt[q];
                    if (en < 0) continue;
                    const int j = en >> 1;
                    if (j == n || j / e.R != world) continue;
                    // within reach of my provisional centre?
                    const float ax = nx - cx[q], ay = ny - cy[q];
                    if (!(ax * ax + ay * ay <= e.collide_reach2)) continue;
                    const float4 q0 = e.bw_prov[2 * j], q1 = e.bw_prov[2 * j + 1];
                    const float ox = e.pose[j * 3 + 0], oy = e.pose[j * 3 + 1];
                    // the pose j has when it is my turn: robots after me have not moved yet; a robot before me is
                    // at its provisional pose iff its own test came out free
                    int sj = 1;
                    if (j < n && (__float_as_int(q0.w) & kFlagMoving)) {
                        // (relaxed: the outcome IS the message -- everything else this thread reads about j was written
                        // by the launch before.  As an acquire / release pair every decided robot's wave wrote the L2
                        // back (buffer_wbl2 sc1) and every look invalidated it (buffer_inv sc1): the jam's collision
                        // pass took 777 us instead of 341, profiles/r04_{q,r}_bigworld_kernel_stats.csv)
                        sj = __hip_atomic_load(&e.bw_state[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (sj == 0) {
                            ready = false;
                            break;
                        }
                    }
                    const float4 hj = e.head[j];
                    const bool at_new = sj == 2;
                    hit = obb_overlap(nx, ny, ns, nc, at_new ? q0.x : ox, at_new ? q0.y : oy, at_new ? q1.x : hj.x,
                                      at_new ? q1.y : hj.y) || hit;
                }
                if (!ready) break;
            }
        }
        if (ready) {
            e.crashed[n] = hit ? 1 : 0;
            __hip_atomic_store(&e.bw_state[n], hit ? 1 : 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            done = true;
        } else {
            __builtin_amdgcn_s_sleep(2);
        }
    }
    if (!done) atomicOr(e.status, kStatusCollideUndecided);
}

__global__ void bw_finish_kernel(EnvView e) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= e.N) return;
    const int local = n % e.R;
    const float4 p0 = e.bw_prov[2 * n], p1 = e.bw_prov[2 * n + 1];
    const int flags = __float_as_int(p0.w);
    const bool live = (flags & kFlagLive) != 0;
    const bool moved = e.bw_state[n] == 2;
    const float4 hd = e.head[n];
    const float ox0 = e.pose[n * 3 + 0], oy0 = e.pose[n * 3 + 1];    // the pose at tick start (hashed by bw_integrate)
    float x = moved ? p0.x : ox0;
    float y = moved ? p0.y : oy0;
    float th = moved ? p0.z : e.pose[n * 3 + 2];
    float s = moved ? p1.x : hd.x, c = moved ? p1.y : hd.y;
    const float v = p1.z, w = p1.w;
    float gx = e.goal[n * 2 + 0], gy = e.goal[n * 2 + 1];
    float pdist = e.prev_dist[n];
    int t = e.t[n];
    float reward = e.reward[n];
    uint8_t done = e.done[n], res = e.result[n], first = e.first_result[n], crashed = e.crashed[n];
    int ep = e.episode[n];

    // GT velocity = finite difference of the pose (stageros.cpp:585-590)
    const float vgt = moved ? fabsf(v) : 0.0f;
    const float wgt = moved ? w : 0.0f;
    // reward / terminal (stage_world1.py:180-211)
    const float ddx = gx - x, ddy = gy - y;
    const float dist = sqrtf(ddx * ddx + ddy * ddy);
    float rg = (pdist - dist) * kKProgress;
    const bool reach = dist < kGoalRadius;
    rg = reach ? kRArrive : rg;
    const bool crash = crashed == 1;
    const float rc = crash ? kRCrash : 0.0f;
    const float aw = fabsf(wgt);
    const float rw = (aw > e.w_thresh) ? kKOmega * aw : 0.0f;
    const bool tout = t > e.timeout;
    uint8_t result = reach ? 1 : 0;
    result = crash ? 2 : result;
    result = tout ? 3 : result;
    const bool done_now = reach || crash || tout;
    uint8_t lv = live ? 1 : 0;
    if (live) {
        reward = (rg + rc) + rw;
        done = done_now ? 1 : 0;
        res = result;
        pdist = dist;
        t = t + 1;
        if (done_now && first == 0) first = result;
    }
    float spv = v, spw = w, ovgt = vgt, owgt = wgt;
    const bool fresh = e.auto_reset == 1 && live && done_now && !MRCA_DBG(e, 32);
    if (fresh) {   // new episode (ppo_stage1.py:51-58): the one-lane form of the sampling loops
        ep = ep + 1;
        const int rm = e.reset_mode[local], gm = e.goal_mode[local];
        float px, py, pth, qx, qy;
        if (rm == 0) {
            px = e.init_table[local * 3 + 0];
            py = e.init_table[local * 3 + 1];
            pth = wrap_angle(e.init_table[local * 3 + 2]);
        } else {
            sample_pose(rm, (uint32_t)n, (uint32_t)ep, e.key0, e.key1, x, y, &px, &py, &pth);
        }
        if (gm == 0) {
            qx = e.goal_table[local * 2 + 0];
            qy = e.goal_table[local * 2 + 1];
        } else {
            sample_goal(gm, (uint32_t)n, (uint32_t)ep, e.key0, e.key1, px, py, &qx, &qy);
        }
        x = px;
        y = py;
        th = pth;
        sincos_det(pth, &s, &c);
        gx = qx;
        gy = qy;
        const float ex = qx - px, ey = qy - py;
        pdist = e.pre_dist_zero ? 0.0f : sqrtf(ex * ex + ey * ey);
        e.init_pose[n * 3 + 0] = px;
        e.init_pose[n * 3 + 1] = py;
        e.init_pose[n * 3 + 2] = pth;
        t = 1;
        crashed = 0;
        lv = 1;
        ovgt = owgt = 0.0f;
        if (!e.hold_velocity) spv = spw = 0.0f;
    }
    const FreeRectField rect_field{e.free_rect, e.g.width, e.g.height, e.free_rect_pitch};
    uint32_t cellv, cellw;
    rect_field.cell((int)floorf((x - e.g.x0) * e.g.inv_cell), (int)floorf((y - e.g.y0) * e.g.inv_cell), &cellv, &cellw);
    e.pose[n * 3 + 0] = x;
    e.pose[n * 3 + 1] = y;
    e.pose[n * 3 + 2] = th;
    e.speed[n * 2 + 0] = spv;
    e.speed[n * 2 + 1] = spw;
    e.speed_gt[n * 2 + 0] = ovgt;
    e.speed_gt[n * 2 + 1] = owgt;
    e.goal[n * 2 + 0] = gx;
    e.goal[n * 2 + 1] = gy;
    e.prev_dist[n] = pdist;
    e.t[n] = t;
    e.reward[n] = reward;
    e.done[n] = done;
    e.result[n] = res;
    e.first_result[n] = first;
    e.crashed[n] = crashed;
    e.live[n] = lv;
    e.episode[n] = ep;
    e.fresh[n] = fresh ? 1 : 0;
    e.head[n] = make_float4(s, c, __uint_as_float(cellv), __uint_as_float(cellw));
    // the collision hash is per tick: every robot empties the (at most two) buckets it filled, so the next tick needs
    // no 4N-entry memset; and the lidar hash's population count of the FINAL pose rides here as well
    const int world = n / e.R;
    e.bw_chead[hash_cell(hash_cell_coord(ox0, kCollideCell), hash_cell_coord(oy0, kCollideCell), world) &
               (uint32_t)e.bw_cmask] = -1;
    if (flags & kFlagMoving)
        e.bw_chead[hash_cell(hash_cell_coord(p0.x, kCollideCell), hash_cell_coord(p0.y, kCollideCell), world) &
                   (uint32_t)e.bw_cmask] = -1;
    atomicAdd(&e.bw_lcount[hash_cell(hash_cell_coord(x, kLidarCell), hash_cell_coord(y, kLidarCell), world) &
                           (uint32_t)e.bw_lmask], 1);
}

// lidar hash = counting sort of the robots by the bucket of their (final) cell.  (Stand-alone count: after explicit
// resets; a tick counts inside bw_finish_kernel.)
__global__ void bw_lidar_count_kernel(EnvView e) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= e.N) return;
    const uint32_t h = hash_cell(hash_cell_coord(e.pose[n * 3 + 0], kLidarCell), hash_cell_coord(e.pose[n * 3 + 1], kLidarCell),
                                 n / e.R) & (uint32_t)e.bw_lmask;
    atomicAdd(&e.bw_lcount[h], 1);
}

// Exclusive scan of the bucket populations in three small launches (a single workgroup walking all 2N buckets took
// 296 us at 50 000 robots -- 70 % of the move phase, profiles/r03/r03_d_bigworld_50000_kernel_stats.csv):
//   scan_local   one workgroup per 1024 buckets: coalesced load, block scan, local prefix -> bw_lstart, total -> bw_lblock;
//                zeroes the counts (for the next tick) and the fill cursors
//   scan_apply   adds its block's offset (the totals of the blocks before it, summed by the block itself)
__device__ __forceinline__ int block_scan_1024(int v, int* part) {     // inclusive scan over the workgroup's 1024 threads
    const int tid = threadIdx.x;
    part[tid] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int add = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    return part[tid];
}

__global__ __launch_bounds__(1024) void bw_lidar_scan_local_kernel(EnvView e) {
    __shared__ int part[1024];
    const int M = e.bw_lmask + 1;
    const int b = blockIdx.x * 1024 + threadIdx.x;
    const int cnt = b < M ? e.bw_lcount[b] : 0;
    const int incl = block_scan_1024(cnt, part);
    if (b < M) {
        e.bw_lstart[b] = incl - cnt;
        e.bw_lcount[b] = 0;                 // ready for the next count (no memset per tick)
        e.bw_lcursor[b] = 0;                // the fill cursor
    }
    if (threadIdx.x == 1023) e.bw_lblock[blockIdx.x] = incl;
}

// adds the offset of its 1024-bucket block = the sum of the totals of the blocks before it (every block adds those up
// itself: at most 1024 values per pass, cheaper than a launch of its own)
__global__ __launch_bounds__(1024) void bw_lidar_scan_apply_kernel(EnvView e) {
    __shared__ int part[1024];
    const int M = e.bw_lmask + 1;
    int offset = 0;
    for (int base = 0; base < (int)blockIdx.x; base += 1024) {
        const int k = base + threadIdx.x;
        const int v = k < (int)blockIdx.x ? e.bw_lblock[k] : 0;
        block_scan_1024(v, part);
        offset += part[1023];
        __syncthreads();
    }
    const int b = blockIdx.x * 1024 + threadIdx.x;
    if (b < M) e.bw_lstart[b] += offset;
    if (b == M - 1) e.bw_lstart[M] = e.N;   // every robot is in exactly one bucket
}

__global__ void bw_lidar_fill_kernel(EnvView e) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= e.N) return;
    const uint32_t h = hash_cell(hash_cell_coord(e.pose[n * 3 + 0], kLidarCell), hash_cell_coord(e.pose[n * 3 + 1], kLidarCell),
                                 n / e.R) & (uint32_t)e.bw_lmask;
    e.bw_lsorted[e.bw_lstart[h] + atomicAdd(&e.bw_lcursor[h], 1)] = n;
}

// generate_train_data (model/ppo.py:122-139)
__global__ void gae_kernel(const float* __restrict__ rewards, const float* __restrict__ values,
                           const float* __restrict__ last_value, const uint8_t* __restrict__ dones, float gamma,
                           float lam, int T, int N, float* __restrict__ targets, float* __restrict__ advs) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float gl = gamma * lam;
    float vnext = last_value[n];
    float g = 0.0f;
    for (int t = T - 1; t >= 0; --t) {
        const size_t k = (size_t)t * N + n;
        const float nd = 1.0f - (float)dones[k];
        const float v = values[k];
        const float delta = (rewards[k] + gamma * vnext * nd) - v;
        g = delta + gl * nd * g;
        const float tg = g + v;
        targets[k] = tg;
        advs[k] = tg - v;
        vnext = v;
    }
}

}  // namespace

#if defined(MRCA_PROFILING)
void read_ray_stamps(unsigned long long* host, int blocks) {      // [2][kRayStamps][blocks]
    for (int k = 0; k < 2 * kRayStamps; ++k)
        (void)hipMemcpyFromSymbol(host + (size_t)k * blocks, HIP_SYMBOL(g_ray_stamps), sizeof(unsigned long long) * blocks,
                                  sizeof(unsigned long long) * ((size_t)k * kRayStampBlocks), hipMemcpyDeviceToHost);
}
void read_move_stamps(unsigned long long* host, int worlds) {      // [kMoveStamps][worlds]
    for (int k = 0; k < kMoveStamps; ++k)
        (void)hipMemcpyFromSymbol(host + (size_t)k * worlds, HIP_SYMBOL(g_move_stamps), sizeof(unsigned long long) * worlds,
                                  sizeof(unsigned long long) * ((size_t)k * kMoveStampWorlds), hipMemcpyDeviceToHost);
}
#endif


size_t ray_lds_bytes(const EnvView& e) {
    size_t b = kWave * (sizeof(float4) + sizeof(int2)) + 16 + (size_t)e.B * 8;
    if (e.raster_inv > 0.0f && !e.big) b += kWave * sizeof(OutlineBits);   // fidelity mode: the neighbours' outline records
    return b;
}

size_t move_lds_bytes(const EnvView& e) {
    const int rows = 2 * e.foot_hc + 1;
    const int words = (rows + 31) / 32 + 1;
    size_t b = (size_t)kWave * rows * words * 4 + (2 + kMoveWaves) * kWave * sizeof(int);
    if (e.raster_inv > 0.0f) b += (size_t)kWave * sizeof(uint2);   // fidelity mode: the provisional outlines' bitmaps
    return b;
}

void launch_move(const EnvView& e, const float* actions, hipStream_t s, hipEvent_t start, hipEvent_t stop, const EnvView* in,
                 unsigned flags) {
    if (!e.big) {
        if (e.world_count <= 0) return;
        const EnvView& r = in ? *in : e;      // what the tick reads (the tick before's poses, heads, goals, outlines)
        MoveOut out{e.pose, e.head, e.goal, e.fresh, e.outline};
#if defined(MRCA_PROFILING)
        out.launch_stamps = e.launch_stamps;
        out.launch_slot = e.launch_slot;
#endif
        if (start || stop || flags)
            hipExtLaunchKernelGGL(move_kernel, dim3(e.world_count), dim3(kWave * kMoveWaves), (uint32_t)move_lds_bytes(e), s, start,
                                  stop, flags, e.R, e.world_first, r.pose, r.head, actions, e.live, r.goal, r.outline, e.dev, out);
        else
            hipLaunchKernelGGL(move_kernel, dim3(e.world_count), dim3(kWave * kMoveWaves), move_lds_bytes(e), s, e.R, e.world_first,
                               r.pose, r.head, actions, e.live, r.goal, r.outline, e.dev, out);
        return;
    }
    // (the collision hash's heads and the lidar hash's counts are left clean by the tick before: bw_finish_kernel /
    // bw_lidar_scan_local_kernel; mrca_create clears them once)
    const int bs = 256, nb = (e.N + bs - 1) / bs;
    if (start || stop) {     // the move phase's span: begin of its first kernel .. end of its last
        hipExtLaunchKernelGGL(bw_integrate_kernel, dim3(nb), dim3(bs), 0, s, start, nullptr, 0, e, actions);
        hipLaunchKernelGGL(bw_collide_kernel, dim3(nb), dim3(bs), 0, s, e);
        hipExtLaunchKernelGGL(bw_finish_kernel, dim3(nb), dim3(bs), 0, s, nullptr, stop, 0, e);
        return;
    }
    hipLaunchKernelGGL(bw_integrate_kernel, dim3(nb), dim3(bs), 0, s, e, actions);
    hipLaunchKernelGGL(bw_collide_kernel, dim3(nb), dim3(bs), 0, s, e);
    hipLaunchKernelGGL(bw_finish_kernel, dim3(nb), dim3(bs), 0, s, e);      // + the lidar hash's counts
}

void launch_materialize(const EnvView& e, int what, hipStream_t s) {
    if (e.ray_count <= 0 || !(what & 3)) return;
    const long long cols = (long long)e.ray_count * (e.B >> 2);
    long long nb = (cols + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(materialize_kernel, dim3((int)nb), dim3(256), 0, s, e, what);
}

void launch_normalize(const float* in, float* out, long long count, hipStream_t s) {
    const long long c4 = count / 4;
    long long blocks = (c4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) return;
    hipLaunchKernelGGL(normalize_kernel, dim3((int)blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(in),
                       reinterpret_cast<float4*>(out), c4);
}

void launch_sparse_obs(const EnvView& e, const int32_t* index, int nb, float* out, hipStream_t s) {
    const long long total = (long long)e.N * e.F * nb;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(sparse_obs_kernel, dim3((int)blocks), dim3(256), 0, s, e, index, nb, out);
}

void launch_newest_obs(const EnvView& e, float* out, hipStream_t s) {
    const long long cols = (long long)e.N * (e.B >> 2);
    long long nb = (cols + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(newest_obs_kernel, dim3((int)nb), dim3(256), 0, s, e, out);
}

// the lidar hash of the current poses.  counted = 1: the populations are in bw_lcount already (a tick: bw_finish_kernel)
void launch_lidar_grid(const EnvView& e, int counted, hipStream_t s) {
    if (!e.big) return;
    const int bs = 256, nb = (e.N + bs - 1) / bs;
    if (!counted) hipLaunchKernelGGL(bw_lidar_count_kernel, dim3(nb), dim3(bs), 0, s, e);
    const int sb = (e.bw_lmask + 1 + 1023) / 1024;
    hipLaunchKernelGGL(bw_lidar_scan_local_kernel, dim3(sb), dim3(1024), 0, s, e);
    hipLaunchKernelGGL(bw_lidar_scan_apply_kernel, dim3(sb), dim3(1024), 0, s, e);
    hipLaunchKernelGGL(bw_lidar_fill_kernel, dim3(nb), dim3(bs), 0, s, e);
}

void launch_reset(const EnvView& e, const uint8_t* mask, const float* poses, const float* goals, hipStream_t s) {
    const int bs = 256;
    hipLaunchKernelGGL(reset_kernel, dim3((e.N + bs - 1) / bs), dim3(bs), 0, s, e, mask, poses, goals);
}

void launch_head_init(const EnvView& e, hipStream_t s) {
    const int bs = 256;
    hipLaunchKernelGGL(head_init_kernel, dim3((e.N + bs - 1) / bs), dim3(bs), 0, s, e);
}

void launch_raycast(const EnvView& e, int only_fresh, hipStream_t s, hipEvent_t start, hipEvent_t stop) {
    const bool raster_mode = !e.big && e.raster_inv > 0.0f;
    // fidelity mode launches the product's shapes only (2 beams per thread one after the other, or 1; the first wave prepares)
    const int threads = raster_mode ? (e.B >> (e.ray_shift == 0 ? 0 : 1)) : (e.B >> e.ray_shift) + (e.ray_prep_wave ? kWave : 0);
    const size_t lds = ray_lds_bytes(e);
    const dim3 grid(e.ray_count);
    if (e.ray_count <= 0) return;
    const bool seq = e.ray_sequential != 0;
    RayIn rin{e.goal, e.fresh, e.outline, e.eager_views};
#if defined(MRCA_PROFILING)
    rin.launch_stamps = e.launch_stamps;
    rin.launch_slot = e.launch_slot;
#endif
#define MRCA_RAY(K, BIG, SEQ) MRCA_RAY4(K, BIG, SEQ, 0)
#define MRCA_RAY4(K, BIG, SEQ, RKWV)                                                                                          \
    do {                                                                                                               \
        if (rin.views)                                                                                                 \
            hipExtLaunchKernelGGL((raycast_kernel<K, BIG, SEQ, RKWV, true>), grid, dim3(threads), (uint32_t)lds, s, start, stop, 0, \
                                  only_fresh, e.ray_first, e.ray_count, e.R, e.pose, e.head, e.beam_cos, e.beam_sin,         \
                                  e.ring_head, e.dev, rin);                                                              \
        else if (start || stop)                                                                                        \
            hipExtLaunchKernelGGL((raycast_kernel<K, BIG, SEQ, RKWV, false>), grid, dim3(threads), (uint32_t)lds, s, start, stop, 0, \
                                  only_fresh, e.ray_first, e.ray_count, e.R, e.pose, e.head, e.beam_cos, e.beam_sin,         \
                                  e.ring_head, e.dev, rin);                                                              \
        else                                                                                                           \
            hipLaunchKernelGGL((raycast_kernel<K, BIG, SEQ, RKWV, false>), grid, dim3(threads), lds, s, only_fresh, e.ray_first,    \
                               e.ray_count, e.R, e.pose, e.head, e.beam_cos, e.beam_sin, e.ring_head, e.dev, rin);     \
    } while (0)
    if (raster_mode) {
        if (e.raster_kw <= 4) {
            if (e.ray_shift == 0) MRCA_RAY4(1, false, false, 4);
            else MRCA_RAY4(2, false, true, 4);
        } else {
            if (e.ray_shift == 0) MRCA_RAY4(1, false, false, 8);
            else MRCA_RAY4(2, false, true, 8);
        }
        return;
    }
    if (e.big) {
        switch (e.ray_shift) {
            case 0: MRCA_RAY(1, true, false); break;
            case 1:
                if (seq) MRCA_RAY(2, true, true);
                else MRCA_RAY(2, true, false);
                break;
            default:
                if (seq) MRCA_RAY(4, true, true);      // the product's shape for big worlds (mrca_abi.hip)
                else MRCA_RAY(4, true, false);
                break;
        }
        return;
    }
    switch (e.ray_shift) {
        case 0: MRCA_RAY(1, false, false); break;
        case 1:
            if (seq) MRCA_RAY(2, false, true);
            else MRCA_RAY(2, false, false);
            break;
        default:
            if (seq) MRCA_RAY(4, false, true);      // two waves per workgroup: all 4096 robots resident at once
            else MRCA_RAY(4, false, false);
            break;
    }
#undef MRCA_RAY
#undef MRCA_RAY4
}

void launch_gae(const float* rewards, const float* values, const float* last_value, const uint8_t* dones, float gamma,
                float lam, int T, int N, float* targets, float* advs, hipStream_t s) {
    const int bs = 256;
    hipLaunchKernelGGL(gae_kernel, dim3((N + bs - 1) / bs), dim3(bs), 0, s, rewards, values, last_value, dones, gamma,
                       lam, T, N, targets, advs);
}

}  // namespace mrca
