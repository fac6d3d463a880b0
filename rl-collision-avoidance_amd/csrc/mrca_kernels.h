// mrca_kernels.h -- launch interface between the C ABI (mrca_abi.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mrca_device.h"

namespace mrca {

// Everything a kernel needs.  All pointers are device memory inside the env's arena.  Most kernels take it by value (kernarg
// segment); the two of the tick -- move_kernel, raycast_kernel -- read the env's DEVICE copy of it (`dev`) and take what
// differs from launch to launch as small arguments of their own (MoveOut / RayIn): a launch costs the host 2.6 us with up
// to 104 bytes of kernel arguments and 3.2 - 3.5 us from 128 bytes on (tools/launch_cost_probe.hip), and a short
// mrca_step_many region is paced by the host (profiles/r06_ai_*).
struct EnvView {
    int32_t N, R, W, B, F;
    // per-robot state (SoA arena, 256-B aligned fields)
    float* pose;        // [N,3]
    float* speed;       // [N,2]
    float* speed_gt;    // [N,2]
    float* goal;        // [N,2]
    float* init_pose;   // [N,3]
    float* scan;        // [N,B]   the newest scan -- a COPY out of scan_ring, made by materialize_kernel
    float* obs;         // [N,F,B] the normalised frame stack in deque order -- made from scan_ring by materialize_kernel
    float* scan_ring;   // [N,F,B] the last F scans (RAW ranges) as a ring: slot ring_head[n] holds robot n's newest one
    uint8_t* ring_head; // [N]
    unsigned long long* hit_bits;   // [N,F,B/64] what each beam of the ring hit: bit set = another robot (MRCA_F_HIT_BITS)
    float* local_goal;  // [N,2]
    float* reward;      // [N]
    float* prev_dist;   // [N]
    uint8_t* done;
    uint8_t* result;
    uint8_t* first_result;
    uint8_t* crashed;
    uint8_t* live;
    uint8_t* fresh;
    int32_t* t;
    int32_t* episode;
    // internal, one 16-byte record per robot, rewritten by whoever changes a pose (move / reset kernels): sin and
    // cos of the heading (deterministic sincos_det) and the four quadrant entries of the free-rectangle field for the
    // cell the robot stands in (as float bits).  The ray cast starts from it instead of recomputing all three per wave.
    float4* head;       // [N] (sin, cos, bits(quadrants 0 | 1 << 16), bits(quadrants 2 | 3 << 16))
    // fidelity mode only (raster_inv > 0): the robot's outline as an anchored 8 x 8 bitmap of raster cells (mrca_device.h
    // OutlineBits), kept like `head` by whoever changes a pose: the collision pass intersects two of them in registers, the
    // ray cast tests a beam against one in closed form
    OutlineBits* outline;   // [N]
    // worlds with more than 64 robots ("big" worlds: one wavefront no longer holds a world): scratch of the
    // per-tick broad phase, see the bw_* kernels.  All NULL / 0 otherwise.
    int32_t big;            // 1: robots_per_world > 64
    float4* bw_prov;        // [N][2] provisional pose of the tick: (nx, ny, nth, bits(flags)), (ns, nc, v, w)
    int32_t* bw_state;      // [N] ordered collision pass: 0 undecided, 1 stays, 2 moved
    int32_t* bw_chead;      // [bw_cmask+1] collision hash (0.7 m cells): bucket -> first entry, -1 = empty
    int32_t* bw_cnext;      // [2N] entry e = 2*robot + (0: pose at tick start | 1: provisional pose) -> next entry
    int32_t bw_cmask;
    uint32_t* bw_ticket;    // [1] the collision pass hands its robot blocks to workgroups in the order they START
    int32_t* bw_lstart;     // [bw_lmask+2] lidar hash (6.5 m cells) over the FINAL poses: bucket -> first slot
    int32_t* bw_lcount;     // [bw_lmask+1] bucket population (counted by bw_finish_kernel, zeroed again by the scan)
    int32_t* bw_lcursor;    // [bw_lmask+1] fill cursor of the counting sort
    int32_t* bw_lsorted;    // [N] robots ordered by bucket
    int32_t* bw_lblock;     // [(bw_lmask+1)/1024 + 1] totals / offsets of the scan's 1024-bucket blocks
    int32_t bw_lmask;
    int32_t ray_first, ray_count;   // the ray cast covers robots [ray_first, ray_first + ray_count) (mrca_step_slice)
    int32_t world_first, world_count;   // the move launch covers worlds [world_first, world_first + world_count) (mrca_step_worlds)
    // scenario tables, per local index
    const int32_t* reset_mode;
    const int32_t* goal_mode;
    const int32_t* group_id;
    const float* init_table;  // [R,3]
    const float* goal_table;  // [R,2]
    // lidar beam directions in the robot frame (stageros.cpp:495-497), fp64-computed, fp32-rounded
    const float* beam_cos;
    const float* beam_sin;
    float beam_step, beam_inv_step;   // pi / (B - 1) and (B - 1) / pi as fp32 quotients (mrca_device.h:beam_interval)
    // occupancy grid (move kernel) + per-cell free-rectangle field (grid_march_skip, ray-cast kernel)
    const uint32_t* map_bits;
    const uint16_t* free_rect;   // [map_h + 2*kFieldPadY][free_rect_pitch][4 quadrants], see FreeRectField
    int32_t free_rect_pitch;     // padded row length in CELLS
    const uint8_t* cellfield;  // per-cell Chebyshev distance to the nearest occupied cell [map_h][map_w]
    GridGeom g;
    // rules
    int32_t timeout;
    float w_thresh;
    int32_t pre_dist_zero;
    int32_t auto_reset;
    int32_t num_groups;
    int32_t hold_velocity;  // 1: Stage's SetSpeed persistence (dead robots keep driving, speed survives a reset)
    uint32_t key0, key1;
    float raster_inv;     // fidelity mode: 1 / collision_raster (0 = exact rectangles), see mrca_device.h outline_cells
    float raster_res;     // collision_raster itself (fidelity mode: the lidar sees the other robots through this raster too)
    int32_t raster_kw;    // cells per side of an outline's window the ray cast tests: 4 (res >= 0.195 m) or 8
    // the ray cast's neighbour culls: what another robot lies inside of seen from its centre (circumradius + 1 mm; in
    // fidelity mode + one raster-cell diagonal), the centre distance below which every beam is kept, and the squared centre
    // distance beyond which it cannot return a range below 6 m
    float lidar_radius, lidar_near, lidar_reach2;
    float collide_reach2; // squared centre distance beyond which two robots cannot collide (broad phase)
    int32_t foot_hc;      // half extent (cells) of the move kernel's per-robot mini tile
    int32_t edge_slots;   // crossings per axis a footprint edge can have on this map (mrca_device.h edge_event_slots)
    int32_t ray_shift;    // raycast_kernel marches 1 << ray_shift beams per thread in lock step
    int32_t ray_sequential; // 1 (with ray_shift 1): the two beams of a thread are marched one after the other
    int32_t ray_prep_wave;  // 1: a dedicated wave prepares the neighbour list (blockDim = beams >> ray_shift + 64)
    uint32_t r_magic;     // ceil(2^32 / R) for R <= 64: n / R == umulhi(n, r_magic) for every robot index n < 2^24
    int32_t debug_flags;  // profiling ablations only (mrca_set_debug_flags, -DMRCA_PROFILING builds)
    uint32_t* status;     // [1] sticky device-side error bits (kStatus*), read and cleared by mrca_check()
    int32_t eager_views;  // HOST side only (travels to the kernel in RayIn): the views the next ray cast forms itself, see RayIn
    const EnvView* dev;   // the env's copy of this struct in device memory (mrca_abi.hip: uploaded by mrca_create, again by the
                          // profiling build's debug switches); a host-side copy with other slots / ranges keeps pointing at it
#if defined(MRCA_PROFILING)
    // profiling build only (MRCA_LAUNCH_STAMPS, mrca_abi.hip): [2 * slots] first start / last end of a launch on the 100 MHz
    // constant clock, by thread 0 of every 64th workgroup and the last (every workgroup: 4096 atomics on one word made a launch five times longer) -- the timeline of a mrca_step_many pass without a profiler attached
    unsigned long long* launch_stamps;
    int32_t launch_slot;
#endif
};

// bits of EnvView::status
constexpr uint32_t kStatusCollideUndecided = 1u;   // bw_collide_kernel gave up waiting for a lower-indexed robot
constexpr uint32_t kStatusBadBeamIndex = 4u;       // mrca_sparse_obs: an entry of the caller's beam table was outside [0, beams)
constexpr uint32_t kStatusOutlineWindow = 2u;      // fidelity mode: an outline cell fell outside the 8 x 8 window of its bitmap

// Ablation switches exist only in the profiling build of the library (csrc/build.sh --profiling ->
// libmrca_env_prof.so, used by tools/ablate.py); in the product they fold to `false` at compile time.
#if defined(MRCA_PROFILING)
#define MRCA_DBG(e, bit) (((e).debug_flags & (bit)) != 0)
#define MRCA_LAUNCH_BEGIN(e)                                                                                      \
    do {                                                                                                          \
        if ((e).launch_stamps && threadIdx.x == 0 && ((blockIdx.x & 63) == 0 || blockIdx.x == gridDim.x - 1)) atomicMin((e).launch_stamps + 2 * (e).launch_slot, wall_clock64()); \
    } while (0)
#define MRCA_LAUNCH_END(e)                                                                                            \
    do {                                                                                                              \
        if ((e).launch_stamps && threadIdx.x == 0 && ((blockIdx.x & 63) == 0 || blockIdx.x == gridDim.x - 1)) atomicMax((e).launch_stamps + 2 * (e).launch_slot + 1, wall_clock64()); \
    } while (0)
#else
#define MRCA_DBG(e, bit) false
#define MRCA_LAUNCH_BEGIN(e) ((void)0)
#define MRCA_LAUNCH_END(e) ((void)0)
#endif

// What a move launch WRITES that mrca_step_many gives a slot per tick (the rest of its outputs it finds in *dev), and what a ray
// cast reads of such a slot beside its leading arguments; in the profiling build also the launch's stamp slot.
struct MoveOut {
    float* pose;
    float4* head;
    float* goal;
    uint8_t* fresh;
    OutlineBits* outline;
#if defined(MRCA_PROFILING)
    unsigned long long* launch_stamps;
    int32_t launch_slot;
#endif
};
struct RayIn {
    const float* goal;
    const uint8_t* fresh;
    const OutlineBits* outline;
    int32_t views;          // MRCA_VIEW_SCAN | MRCA_VIEW_OBS: the ray cast also forms these views of its robots (lazy_obs = 0)
#if defined(MRCA_PROFILING)
    unsigned long long* launch_stamps;
    int32_t launch_slot;
#endif
};

size_t ray_lds_bytes(const EnvView& e);
size_t move_lds_bytes(const EnvView& e);

// `start` / `stop` (both or neither): events stamped with the BEGIN of the first and the END of the last kernel of the launch
// (hipExtLaunchKernel: the dispatch's own timestamps -- what rocprofv3 reports -- instead of event records around it, which
// read 2.5 us longer per kernel; bench.py)
// in: the view whose pose / head / goal / outline the tick READS (nullptr: e's own -- an in-place tick); flags: hipExtLaunchKernel's
void launch_move(const EnvView& e, const float* actions, hipStream_t s, hipEvent_t start = nullptr, hipEvent_t stop = nullptr,
                 const EnvView* in = nullptr, unsigned flags = 0);
void launch_reset(const EnvView& e, const uint8_t* mask, const float* poses, const float* goals, hipStream_t s);
void launch_head_init(const EnvView& e, hipStream_t s);
void launch_lidar_grid(const EnvView& e, int counted, hipStream_t s);   // big worlds: hash of the current poses for the ray cast
void launch_raycast(const EnvView& e, int only_fresh, hipStream_t s, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);
#if defined(MRCA_PROFILING)
void read_ray_stamps(unsigned long long* host, int blocks);     // profiling build: [2 waves][7 stamps][blocks] of raycast_kernel
void read_move_stamps(unsigned long long* host, int worlds);   // profiling build: s_memtime stamps of move_kernel's phases
#endif
// scan_ring -> scan (what & 1) and / or obs (what & 2: normalised, deque order) for robots [ray_first, ray_first + ray_count)
void launch_materialize(const EnvView& e, int what, hipStream_t s);
// the newest frame of every robot, normalised, into out[N,B] (the row a one-frame rollout buffer stores per tick)
void launch_newest_obs(const EnvView& e, float* out, hipStream_t s);
// out[N,F,nb] = x / 6 - 0.5 of beams index[0..nb) of every frame in deque order (get_laser_observation with beam_num != raw)
// out[i] = in[i] / 6 - 0.5 exactly as the env's own views form it (norm_obs), for `count` floats (count % 4 == 0)
void launch_normalize(const float* in, float* out, long long count, hipStream_t s);
void launch_sparse_obs(const EnvView& e, const int32_t* index, int nb, float* out, hipStream_t s);
void launch_gae(const float* rewards, const float* values, const float* last_value, const uint8_t* dones, float gamma,
                float lam, int T, int N, float* targets, float* advs, hipStream_t s);

}  // namespace mrca
