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
//   raycast_kernel one workgroup per robot: beams/K marching threads, each marching K beams in LOCK STEP
//                  (K independent field lookups in flight per thread: the march is a chain of dependent
//                  L2 lookups, ~2 per ray), plus one preparation wave that compacts the other robots of
//                  the world within lidar reach into LDS (ballot + popcount) together with a per-beam
//                  bitmask of who can touch which beam.  The robot's own sin/cos and the field entry of
//                  its cell come from the 16-byte `head` record the move kernel published (scalar
//                  loads), so no wave recomputes them.  Each thread then slab-tests its own beams against
//                  the flagged neighbours; scan, normalised observation and the frame-stack shift leave
//                  through LDS as 16-byte stores.
//   reset_kernel   explicit reset_pose / control_pose / generate_goal_point.
//   gae_kernel     reverse GAE scan, thread per robot, coalesced over N.
//
// No dense contraction anywhere: MFMA is deliberately unused (BASELINE.json north_star).
#include "mrca_kernels.h"

namespace mrca {

namespace {

constexpr int kWave = 64;
constexpr int kPatchBatch = 4;   // footprint patches fetched per memory round trip in move_kernel

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

// Wave-parallel rejection sampling: the 64 lanes evaluate 64 consecutive attempts k at once and the
// lowest acceptable k wins -- the same draw the one-lane loop (sample_pose / sample_goal) returns,
// without a wavefront waiting on one unlucky robot's long tail.  All arguments are wave-uniform.
// lane must be wave-uniform: v_readlane_b32
__device__ __forceinline__ int ibcast(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float fbcast(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

__device__ __forceinline__ void wave_sample_pose(int lane, int mode, uint32_t gid, uint32_t ep, uint32_t k0,
                                                 uint32_t k1, float curx, float cury, float* px, float* py,
                                                 float* pth) {
    for (int base = 0; base < kMaxTriesPose; base += kWave) {
        float x, y, th;
        const bool ok = pose_try(mode, gid, ep, (uint32_t)(base + lane), k0, k1, curx, cury, &x, &y, &th);
        const bool last = base + kWave >= kMaxTriesPose;
        const unsigned long long m = __ballot(ok || (last && lane == kMaxTriesPose - 1 - base));
        if (m) {
            const int w = __ffsll((long long)m) - 1;
            *px = fbcast(x, w);
            *py = fbcast(y, w);
            *pth = fbcast(th, w);
            return;
        }
    }
}

__device__ __forceinline__ void wave_sample_goal(int lane, int mode, uint32_t gid, uint32_t ep, uint32_t k0,
                                                 uint32_t k1, float curx, float cury, float* gx, float* gy) {
    for (int base = 0; base < kMaxTriesGoal; base += kWave) {
        float x, y;
        const bool ok = goal_try(mode, gid, ep, (uint32_t)(base + lane), k0, k1, curx, cury, &x, &y);
        const bool last = base + kWave >= kMaxTriesGoal;
        const unsigned long long m = __ballot(ok || (last && lane == kMaxTriesGoal - 1 - base));
        if (m) {
            const int w = __ffsll((long long)m) - 1;
            *gx = fbcast(x, w);
            *gy = fbcast(y, w);
            return;
        }
    }
}

struct MiniGrid {  // the move kernel's per-robot occupancy patch in LDS
    const uint32_t* t;
    int y0, w0, stride;
    __device__ __forceinline__ bool operator()(int ix, int iy) const {
        return (t[(iy - y0) * stride + ((ix >> 5) - w0)] >> (ix & 31)) & 1u;
    }
};

__global__ __launch_bounds__(kWave) void move_kernel(EnvView e, const float* __restrict__ actions) {
    extern __shared__ __attribute__((aligned(16))) uint32_t mini[];
    const int world = blockIdx.x;
    const int lane = threadIdx.x;
    const bool valid = lane < e.R;
    const int n = world * e.R + (valid ? lane : 0);

    // every per-robot input is requested up front so that all of it arrives in ONE memory round trip
    float x = e.pose[n * 3 + 0], y = e.pose[n * 3 + 1], th = e.pose[n * 3 + 2];
    const bool live = e.live[n] != 0;
    const float act_v = actions[n * 2 + 0], act_w = actions[n * 2 + 1];
    float gx = e.goal[n * 2 + 0], gy = e.goal[n * 2 + 1];
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
    const float4 hd = e.head[n];   // sin / cos of th and the field entry of the robot's cell, kept by whoever moved it
    const float v = live ? sane_cmd(act_v) : 0.0f;
    const float w = live ? sane_cmd(act_w) : 0.0f;

    // integrate: explicit Euler with the heading at tick start
    float s = hd.x, c = hd.y;
    uint32_t cellv = __float_as_uint(hd.z);
    const float d = v * kDt;
    const float nx = x + d * c;
    const float ny = y + d * s;
    const float nth = wrap_angle(th + w * kDt);
    float ns, nc;
    sincos_det(nth, &ns, &nc);
    const bool moving = valid && ((v != 0.0f) || (w != 0.0f));

    // --- outline-vs-grid test.  Skipped (same answer: free) when the coarse free-distance field says
    //     every block within the footprint's circumradius of the provisional centre is empty.  For the
    //     others the (2*hc+1)-row patch under the footprint is pulled into LDS by the WHOLE wave, kPatchBatch
    //     robots' loads in flight at a time, and each robot then walks its outline in LDS.
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
    const uint32_t cellv_new = rect_field(pix, piy);

    // --- broad phase of the robot-robot collision pass (the pass itself follows the outline test): robot i
    //     can only touch robot j if its provisional centre comes within 2 x circumradius of j's old or
    //     new centre, so only the (few) robots with such a neighbour take a turn in the ordered pass;
    //     everybody else commits straight away -- their outcome does not depend on the order.
    bool involved = false;
    if (!MRCA_DBG(e, 16)) {
        for (int j = 0; j < e.R; ++j) {
            const float ax = nx - fbcast(x, j), ay = ny - fbcast(y, j);
            const float bx2 = nx - fbcast(nx, j), by2 = ny - fbcast(ny, j);
            const float d_old = ax * ax + ay * ay, d_new = bx2 * bx2 + by2 * by2;
            if (j != lane && (d_old <= 0.3392f || d_new <= 0.3392f)) involved = true;  // (2*0.2907 + 0.001)^2
        }
        involved = involved && valid;
    }
    const bool need = check_map && !(inside && clearance > hc);
    {
        unsigned long long todo = __ballot(need);
        while (todo) {
            int src[kPatchBatch];
#pragma unroll
            for (int q = 0; q < kPatchBatch; ++q) {
                src[q] = todo ? (__ffsll((long long)todo) - 1) : -1;
                if (todo) todo &= todo - 1;
            }
            for (int k0 = 0; k0 < psize; k0 += kWave) {
                const int k = k0 + lane;
                const int r = k / pwords, wi = k - r * pwords;
                uint32_t val[kPatchBatch];
#pragma unroll
                for (int q = 0; q < kPatchBatch; ++q) {
                    val[q] = 0u;
                    if (src[q] >= 0 && k < psize) {
                        const int gy = ibcast(py0, src[q]) + r;
                        const int gw = ibcast(pw0, src[q]) + wi;
                        if (gy >= 0 && gy < e.g.height && gw >= 0 && gw < e.g.wpr) val[q] = e.map_bits[gy * e.g.wpr + gw];
                    }
                }
#pragma unroll
                for (int q = 0; q < kPatchBatch; ++q)
                    if (src[q] >= 0 && k < psize) mini[src[q] * psize + k] = val[q];
            }
        }
    }
    // outline walks, four lanes per robot (one per edge, 16 robots per pass): a walk is a chain of
    // dependent LDS reads, so spreading the edges over lanes cuts the chain by four
    int* need_list = reinterpret_cast<int*>(mini + kWave * psize);   // [64] lanes that need the walk
    int* hit_flag = need_list + kWave;                               // [64] result per robot lane
    const unsigned long long need_mask = __ballot(need);
    if (need) need_list[__popcll(need_mask & ((1ull << lane) - 1ull))] = lane;
    hit_flag[lane] = 0;
    __syncthreads();  // one wave per block: orders the cooperative LDS writes before the reads below
    const int n_need = __popcll(need_mask);
    for (int base = 0; base < n_need; base += 16) {
        const int q = base + (lane >> 2);
        const bool act = q < n_need;
        const int src = act ? need_list[q] : 0;
        const float sx_ = __shfl(nx, src, kWave), sy_ = __shfl(ny, src, kWave);
        const float ss_ = __shfl(ns, src, kWave), sc_ = __shfl(nc, src, kWave);
        const int sy0 = __shfl(py0, src, kWave), sw0 = __shfl(pw0, src, kWave);
        if (act) {
            const MiniGrid mg{mini + src * psize, sy0, sw0, pwords};
            if (static_edge_hit(mg, e.g, sx_, sy_, ss_, sc_, lane & 3)) hit_flag[src] = 1;
        }
    }
    __syncthreads();
    const bool shit = need && hit_flag[lane] != 0;

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
            const bool ov = valid && (lane != i) && obb_overlap(xi, yi, si, ci, cx_, cy_, cs_, cc_);
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
                    moved = true;
                }
                crashed = hit ? 1 : 0;
            }
        }
    }

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
    // new episodes, one robot at a time with the whole wave sampling for it
    unsigned long long pending = __ballot(fresh);
    while (pending) {
        const int src = __ffsll((long long)pending) - 1;
        pending &= pending - 1;
        const int nsrc = world * e.R + src;
        const uint32_t eps = (uint32_t)(ibcast(ep, src) + 1);
        const int rm = ibcast(rmode, src), gm = ibcast(gmode, src);
        float px, py, pth, qx, qy;
        if (rm == 0) {
            px = e.init_table[src * 3 + 0];
            py = e.init_table[src * 3 + 1];
            pth = wrap_angle(e.init_table[src * 3 + 2]);
        } else {
            wave_sample_pose(lane, rm, (uint32_t)nsrc, eps, e.key0, e.key1, fbcast(x, src), fbcast(y, src), &px, &py,
                             &pth);
        }
        if (gm == 0) {
            qx = e.goal_table[src * 2 + 0];
            qy = e.goal_table[src * 2 + 1];
        } else {
            wave_sample_goal(lane, gm, (uint32_t)nsrc, eps, e.key0, e.key1, px, py, &qx, &qy);
        }
        // head record of the new pose (wave-uniform values; only lane src keeps them)
        float rs_, rc_;
        sincos_det(pth, &rs_, &rc_);
        const uint32_t rv_ = rect_field((int)floorf((px - e.g.x0) * e.g.inv_cell),
                                        (int)floorf((py - e.g.y0) * e.g.inv_cell));
        if (lane == src) {
            ep = (int)eps;
            x = px;
            y = py;
            th = pth;
            s = rs_;
            c = rc_;
            cellv = rv_;
            gx = qx;
            gy = qy;
            const float ex = qx - px, ey = qy - py;
            const float d0 = sqrtf(ex * ex + ey * ey);
            pdist = e.pre_dist_zero ? 0.0f : d0;
            e.init_pose[n * 3 + 0] = px;
            e.init_pose[n * 3 + 1] = py;
            e.init_pose[n * 3 + 2] = pth;
            t = 1;
            crashed = 0;
            lv = 1;
            spv = spw = ovgt = owgt = 0.0f;
        }
    }

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
        e.head[n] = make_float4(s, c, __uint_as_float(cellv), 0.0f);
    }
}

__device__ __forceinline__ void write_head(const EnvView& e, int n, float x, float y, float th) {
    float s, c;
    sincos_det(th, &s, &c);
    const FreeRectField rect_field{e.free_rect, e.g.width, e.g.height, e.free_rect_pitch};
    const uint32_t v0 = rect_field((int)floorf((x - e.g.x0) * e.g.inv_cell), (int)floorf((y - e.g.y0) * e.g.inv_cell));
    e.head[n] = make_float4(s, c, __uint_as_float(v0), 0.0f);
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
    e.speed[n * 2 + 0] = 0.0f;
    e.speed[n * 2 + 1] = 0.0f;
    e.speed_gt[n * 2 + 0] = 0.0f;
    e.speed_gt[n * 2 + 1] = 0.0f;
    e.done[n] = 0;
    e.result[n] = 0;
    e.reward[n] = 0.0f;
    e.first_result[n] = 0;
}

// blockIdx -> robot: consecutive robots (one world's robots) share an XCD's L2 (block b runs on
// XCD b % 8, guide T1); a pure permutation, so correctness never depends on it.
__device__ __forceinline__ int block_to_robot(int b, int N) {
    if (N % 8) return b;
    const int per = N / 8;
    return (b % 8) * per + b / 8;
}

// The march reads the free-rectangle field straight from its L1/L2-resident global copy: ~2 dependent
// lookups per ray.  (Staging a tile of it in LDS per robot was measured slower at every granularity tried,
// the tile costs more to fill than the few lookups it serves.  So were persistent workgroups
// walking several robots each -- 39 vs 37 us, profiles/r01_ad_ablation.txt -- and nontemporal stores made
// no difference.)  What a chain of dependent lookups wants is more of them in flight: each thread marches its
// K beams in lock step (grid_march_skip_n), so one wait covers K lookups.
template <int K>
__global__ __launch_bounds__(1024) void raycast_kernel(EnvView e, int only_fresh) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int n = block_to_robot(blockIdx.x, e.N);
    const int tid = threadIdx.x;
    // the fresh flag comes through the scalar cache (n is block-uniform; the aligned word holding the byte),
    // so nothing below queues behind it in the vector-memory counter
    const uint32_t fresh_word = reinterpret_cast<const uint32_t*>(e.fresh)[n >> 2];

    const bool fresh = ((fresh_word >> ((n & 3) * 8)) & 0xFFu) != 0;
    if (only_fresh && !fresh) return;  // block-uniform

    float4* nb = reinterpret_cast<float4*>(lds);
    int2* nbi = reinterpret_cast<int2*>(nb + kWave);
    int* nb_count = reinterpret_cast<int*>(nbi + kWave);
    float* rbuf = reinterpret_cast<float*>(nb_count + 4);             // [B] ranges for the wide epilogue
    float* obuf = rbuf + e.B;                                         // [B] normalised ranges
    unsigned long long* nbmask = reinterpret_cast<unsigned long long*>(obuf + e.B);   // [B] neighbours per beam

    const int T = e.B / K;                    // marching threads
    const bool extra = (int)blockDim.x > T;   // a dedicated preparation wave sits behind the marching ones
    const int prep_base = extra ? T : 0;
    const bool is_prep = tid >= prep_base && tid < prep_base + kWave;   // wave-uniform
    const bool marches = tid < T;                                       // wave-uniform
    const int world = n / e.R;
    const int local = n - world * e.R;
    // the robot's own record is block-uniform: pose, sin / cos and the field entry of its cell
    const float x = e.pose[n * 3 + 0], y = e.pose[n * 3 + 1];
    const float4 hd = e.head[n];
    const float s = hd.x, c = hd.y;
    // the preparation wave requests "its" neighbour candidate in the same memory round trip
    const int pl = tid - prep_base;
    const bool cand = is_prep && (pl < e.R) && (pl != local);
    const int jn = world * e.R + (cand ? pl : local);
    float xj = 0.0f, yj = 0.0f;
    float4 hj = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (is_prep) {
        xj = e.pose[jn * 3 + 0];
        yj = e.pose[jn * 3 + 1];
        hj = e.head[jn];
    }
    // beam directions in the robot frame for this thread's K beams (tid + k*T)
    float bc[K], bs[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int b = (marches ? tid : 0) + k * T;
        bc[k] = e.beam_cos[b];
        bs[k] = e.beam_sin[b];
    }
    const bool wide = tid < (e.B >> 2);
    const int fstride = e.B >> 2;
    float4* ob4 = reinterpret_cast<float4*>(e.obs + (size_t)n * e.F * e.B);
    // The frame-stack shift (ppo_stage1.py:87-89: popleft / append) does not depend on this tick's ranges at all:
    // the preparation wave moves frames 1.. down to 0.. on its own -- requested here, in the same round trip as its
    // neighbour candidate, stored as soon as they arrive -- while the other waves march.  The marching waves never
    // wait on an HBM load and carry no frame registers; they only append the newest frame.  (A robot that started
    // an episode gets all its frames from the epilogue instead.)
    const bool shifter = is_prep && !fresh && e.F == 3;
    const int chunks = (fstride + kWave - 1) / kWave;          // float4 chunks per lane and frame (2 at 512 beams)
    const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const bool sh0 = shifter && chunks <= 2 && pl < fstride;            // this lane's first / second column
    const bool sh1 = shifter && chunks <= 2 && pl + kWave < fstride;
    const int j0 = sh0 ? pl : 0, j1 = sh1 ? pl + kWave : 0;
    float4 keep1a = zero4, keep2a = zero4, keep1b = zero4, keep2b = zero4;
    if (sh0) {
        keep1a = ob4[fstride + j0];
        keep2a = ob4[2 * fstride + j0];
    }
    if (sh1) {
        keep1b = ob4[fstride + j1];
        keep2b = ob4[2 * fstride + j1];
    }
    // --- preparation wave: compact the world's other robots within lidar reach into LDS, each with the
    //     (conservative) interval of beams that can touch it.  It alone touches the masks before the barrier.
    if (is_prep) {
        for (int b = pl; b < e.B; b += kWave) nbmask[b] = 0ull;
        const float ddx = xj - x, ddy = yj - y;
        // conservative cull: a hit below 6 m needs the centre within 6 + circumradius(0.2907) m
        bool keep = cand && (ddx * ddx + ddy * ddy <= 39.69f);
        int lo = 0, hi = -1;
        if (keep) {
            beam_interval(ddx * c + ddy * s, ddy * c - ddx * s, e.B, &lo, &hi);
            keep = lo <= hi;
        }
        const unsigned long long m = __ballot(keep);
        if (keep) {
            const int idx = __popcll(m & ((1ull << pl) - 1ull));
            nb[idx] = make_float4(xj, yj, hj.x, hj.y);
            nbi[idx] = make_int2(lo, hi);
        }
        const int cnt0 = MRCA_DBG(e, 1) ? 0 : __popcll(m);
        if (pl == 0) *nb_count = cnt0;
        // scatter: bit k of nbmask[b] = "neighbour k can touch beam b".  One wave, program order: the
        // read-modify-writes of successive k never race, and within one k the lanes hit distinct beams.
        for (int k = 0; k < cnt0; ++k) {
            const int2 iv = nbi[k];
            for (int b = iv.x + pl; b <= iv.y; b += kWave) nbmask[b] |= 1ull << k;
        }
        // frame-stack shift: a lane only ever touches "its" float4 columns, so reads and writes of different
        // lanes never meet, and within a lane every store waits for the loads it depends on
        if (shifter && chunks <= 2) {
            if (sh0) {
                ob4[j0] = keep1a;
                ob4[fstride + j0] = keep2a;
            }
            if (sh1) {
                ob4[j1] = keep1b;
                ob4[fstride + j1] = keep2b;
            }
        } else if (is_prep && !fresh) {   // any other stack depth / beam count: frame by frame, column by column
            for (int j = pl; j < fstride; j += kWave)
                for (int f = 0; f + 1 < e.F; ++f) ob4[f * fstride + j] = ob4[(f + 1) * fstride + j];
        }
    }
    // --- the march: K beams per thread in lock step
    float dx[K], dy[K], rng[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        dx[k] = c * bc[k] - s * bs[k];
        dy[k] = s * bc[k] + c * bs[k];
        rng[k] = kRangeMax;
    }
    if (marches && !MRCA_DBG(e, 2)) {
        const FreeRectField field{e.free_rect, e.g.width, e.g.height, e.free_rect_pitch};
        MarchOrigin org;                       // once per robot: shared by all its beams
        org.fx = (x - e.g.x0) * e.g.inv_cell;
        org.fy = (y - e.g.y0) * e.g.inv_cell;
        org.ix0 = (int)floorf(org.fx);
        org.iy0 = (int)floorf(org.fy);
        org.v0 = __float_as_uint(hd.z);
        grid_march_skip_n<K>(field, e.g, org, dx, dy, kRangeMax, rng);
    }
    __syncthreads();  // neighbour list ready (the preparation wave built it while the others marched)
    if (!marches) return;  // the dedicated preparation wave is done (whole wave: the barrier below counts live waves)
    const int cnt = *nb_count;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int b = tid + k * T;
        float r = rng[k];
        unsigned long long m = cnt > 0 ? nbmask[b] : 0ull;
        while (m) {
            const int q = __ffsll((long long)m) - 1;
            m &= m - 1;
            const float4 nbq = nb[q];
            const float t = ray_box(x, y, dx[k], dy[k], nbq.x, nbq.y, nbq.z, nbq.w);
            r = t < r ? t : r;
        }
        r = r < kRangeMax ? r : kRangeMax;
        rbuf[b] = r;
        obuf[b] = norm_obs(r);     // stage_world1.py:140, once per value, spread over all marching threads
    }

    // --- scan, normalised observation, frame stack (ppo_stage1.py:59-60,87-89): ranges went through LDS so
    //     that a quarter of the threads can move 16 bytes each
    __syncthreads();
    if (wide) {
        const float4 r4 = reinterpret_cast<const float4*>(rbuf)[tid];
        const float4 o4 = reinterpret_cast<const float4*>(obuf)[tid];
        reinterpret_cast<float4*>(e.scan + (size_t)n * e.B)[tid] = r4;
        if (fresh) {
            for (int f = 0; f < e.F; ++f) ob4[f * fstride + tid] = o4;
        } else {
            ob4[(e.F - 1) * fstride + tid] = o4;      // frames below were shifted by the preparation wave
        }
    }
    if (tid == 0) {  // get_local_goal (stage_world1.py:155-160)
        const float gx = e.goal[n * 2 + 0] - x, gy = e.goal[n * 2 + 1] - y;
        e.local_goal[n * 2 + 0] = gx * c + gy * s;
        e.local_goal[n * 2 + 1] = gy * c - gx * s;
    }
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

size_t ray_lds_bytes(const EnvView& e) {
    return kWave * (sizeof(float4) + sizeof(int2)) + 16 + (size_t)e.B * 16;
}

size_t move_lds_bytes(const EnvView& e) {
    const int rows = 2 * e.foot_hc + 1;
    const int words = (rows + 31) / 32 + 1;
    return (size_t)kWave * rows * words * 4 + 2 * kWave * sizeof(int);
}

void launch_move(const EnvView& e, const float* actions, hipStream_t s) {
    hipLaunchKernelGGL(move_kernel, dim3(e.W), dim3(kWave), move_lds_bytes(e), s, e, actions);
}

void launch_reset(const EnvView& e, const uint8_t* mask, const float* poses, const float* goals, hipStream_t s) {
    const int bs = 256;
    hipLaunchKernelGGL(reset_kernel, dim3((e.N + bs - 1) / bs), dim3(bs), 0, s, e, mask, poses, goals);
}

void launch_head_init(const EnvView& e, hipStream_t s) {
    const int bs = 256;
    hipLaunchKernelGGL(head_init_kernel, dim3((e.N + bs - 1) / bs), dim3(bs), 0, s, e);
}

void launch_raycast(const EnvView& e, int only_fresh, hipStream_t s) {
    const int threads = (e.B >> e.ray_shift) + (e.ray_prep_wave ? kWave : 0);
    const size_t lds = ray_lds_bytes(e);
    switch (e.ray_shift) {
        case 0: hipLaunchKernelGGL(raycast_kernel<1>, dim3(e.N), dim3(threads), lds, s, e, only_fresh); break;
        case 1: hipLaunchKernelGGL(raycast_kernel<2>, dim3(e.N), dim3(threads), lds, s, e, only_fresh); break;
        default: hipLaunchKernelGGL(raycast_kernel<4>, dim3(e.N), dim3(threads), lds, s, e, only_fresh); break;
    }
}

void launch_gae(const float* rewards, const float* values, const float* last_value, const uint8_t* dones, float gamma,
                float lam, int T, int N, float* targets, float* advs, hipStream_t s) {
    const int bs = 256;
    hipLaunchKernelGGL(gae_kernel, dim3((N + bs - 1) / bs), dim3(bs), 0, s, rewards, values, last_value, dones, gamma,
                       lam, T, N, targets, advs);
}

}  // namespace mrca
