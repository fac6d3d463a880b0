// mrca_device.h -- per-lane arithmetic of the Stage tick for gfx950 (and a plain host build
// used only by the CPU test harness under tests/host_emul/).
//
// Every function is built from separately rounded IEEE fp32 +,-,*,/,sqrt and comparisons, in a
// fixed order (compile with -ffp-contract=off, correctly rounded div/sqrt), so a launch is
// reproducible bit-for-bit and can be checked against the fp32 mode of the oracle.
//
// Reference behaviour restated here (paths relative to the reference checkout):
//   kinematic tick .......... libstage ModelPosition (un-vendored) driven by stageros.cpp:445-449
//   lidar geometry .......... stageros.cpp:479-516, worlds/stage1.world:9-15
//   reward / terminal ....... stage_world1.py:180-211 (+ stage_world2.py:175-208, circle_world.py:171-203)
//   local goal .............. stage_world1.py:155-160
//   reset distributions ..... stage_world1.py:251-274, stage_world2.py:250-287
#pragma once
#include <stdint.h>
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#define MRCA_HD __host__ __device__ __forceinline__
#else
#define MRCA_HD inline
#endif

namespace mrca {

constexpr float kDt = 0.1f;           // Stage default interval_sim (worlds/*.world set none)
constexpr float kRangeMax = 6.0f;     // stage1.world:13
constexpr float kHalfLen = 0.22f;     // stage1.world:83  size [0.44 0.38 0.22]
constexpr float kHalfWid = 0.19f;
constexpr float kGoalRadius = 0.5f;   // stage_world1.py:34
constexpr float kRArrive = 15.0f;     // stage_world1.py:195
constexpr float kRCrash = -15.0f;     // stage_world1.py:200
constexpr float kKProgress = 2.5f;    // stage_world1.py:187
constexpr float kKOmega = -0.1f;      // stage_world1.py:204
constexpr float kPi = 3.14159265358979323846f;
constexpr float kTwoPi = 6.28318530717958647692f;
constexpr int kMaxTriesPose = 64;
constexpr int kMaxTriesGoal = 256;
constexpr uint32_t kStreamPose = 0u, kStreamGoal = 1u;
constexpr float kInf = __builtin_huge_valf();
constexpr int kMaxMarchSteps = 1 << 14;   // termination guard of the grid walks (never reached on finite in-map input)

// ------------------------------------------------------------------------------------------
// sincos: Cody-Waite reduction by pi/2 + Cephes single-precision minimax polynomials.
MRCA_HD void sincos_det(float th, float* sn, float* cs) {
    const float k = rintf(th * 0.6366197723675814f);
    const float r = ((th - k * 1.5703125f) - k * 4.837512969970703125e-4f) - k * 7.54978995489188e-8f;
    const float z = r * r;
    const float s = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z + -1.6666654611e-1f) * z * r + r;
    const float c = ((2.443315711809948e-5f * z + -1.388731625493765e-3f) * z + 4.166664568298827e-2f) * (z * z) +
                    (1.0f - 0.5f * z);
    const int q = ((int)k) & 3;
    *sn = (q == 0) ? s : (q == 1) ? c : (q == 2) ? -s : -c;
    *cs = (q == 0) ? c : (q == 1) ? -s : (q == 2) ? -c : s;
}

// 1 / d, correctly rounded.  On the device: v_rcp_f32 (1 ulp) + ONE Newton step in FMA arithmetic -- bit-identical to the
// IEEE quotient for every float with 2^-100 <= |d| <= 2^100 (exhaustive sweep over all 2^32 bit patterns on the MI355X,
// tools/check_rcp.hip, profiles/r02/r02_e_check_rcp.txt: 0 mismatches; the mismatches outside are results that underflow) --
// in 3 instructions instead of the ~11 of the compiler's division expansion; anything outside that range takes the IEEE
// division.  The test for "outside" is ONE wave-uniform branch (a ballot): written as a per-lane `if` it cost every caller
// eleven scalar instructions and three branches of exec-mask bookkeeping around a path no beam direction ever takes
// (profiles/r06_ac_*: a scalar instruction costs the ray cast twice a vector one).  PRECONDITION d != 0 -- a zero yields NaN,
// not an infinity: every caller tests for it anyway (`d != 0 ? r : inf` with r computed unconditionally, so that the select
// stays a select), and an axis-parallel beam must not send its whole wave down the slow path.
// The host build (test harness) is the plain quotient, which is the specification.
MRCA_HD float rcp_exact(float d) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float a = fabsf(d);
    const float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    const float fast = __builtin_fmaf(e, r, r);
    const bool outside = !(a >= 7.8886090522101181e-31f && a <= 1.2676506002282294e30f) && a != 0.0f;
    if (__builtin_amdgcn_ballot_w64(outside) != 0ull) return outside ? 1.0f / d : fast;
    return fast;
#else
    return 1.0f / d;
#endif
}

// A non-finite command (a diverged policy) is treated as 0: the robot idles instead of poisoning the state.
MRCA_HD float sane_cmd(float a) { return (a - a == 0.0f) ? a : 0.0f; }

// (-pi, pi]: the GT yaw after the quaternion round trip (stageros.cpp:575-583, stage_world1.py:88-91)
MRCA_HD float wrap_angle(float th) {
    th = (th > kPi) ? th - kTwoPi : th;
    th = (th <= -kPi) ? th + kTwoPi : th;
    return th;
}

// ------------------------------------------------------------------------------------------
// Philox4x32-10 (Random123).  Counter = (robot id, episode, try, stream), key = seed.
struct U4 {
    uint32_t x, y, z, w;
};

MRCA_HD U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        if (r) {
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    }
    return U4{c0, c1, c2, c3};
}

MRCA_HD float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

// ------------------------------------------------------------------------------------------
// Occupancy grid views.  Cells outside [0,W)x[0,H) are free.
struct GridGeom {
    float x0, y0, cell, inv_cell;
    int32_t width, height, wpr;
};

struct GlobalGrid {  // bit lookups straight from the (L2-resident) bitmap
    const uint32_t* bits;
    int32_t width, height, wpr;
    MRCA_HD bool operator()(int ix, int iy) const {
        if (ix < 0 || iy < 0 || ix >= width || iy >= height) return false;
        return in_map(ix, iy);
    }
    // for cells known to lie inside the map (a cell of a non-empty block always does)
    MRCA_HD bool in_map(int ix, int iy) const {
        return (bits[(uint32_t)(iy * wpr + (ix >> 5))] >> (ix & 31)) & 1u;
    }
};

// First occupied cell along o + t*d, 0 <= t < tmax (metres, |d| = 1).  The visited cell sequence
// is driven by closed-form boundary times t(b) = (float(b) - f) * (1/d) in cell units, so it is
// independent of how the walk is organised.  Returns the entry distance of the first occupied
// cell (0 when the start cell is occupied) or tmax.  Ties step in y.
template <class Occ>
MRCA_HD float grid_march(const Occ& occ, const GridGeom& g, float ox, float oy, float dx, float dy, float tmax) {
    const float fx = (ox - g.x0) * g.inv_cell;
    const float fy = (oy - g.y0) * g.inv_cell;
    int ix = (int)floorf(fx);
    int iy = (int)floorf(fy);
    const float tmax_c = tmax * g.inv_cell;
    if (occ(ix, iy)) return 0.0f;
    if (!(tmax_c > 0.0f)) return tmax;
    const bool xnz = dx != 0.0f, ynz = dy != 0.0f;
    const float inv_dx = xnz ? rcp_exact(dx) : kInf;
    const float inv_dy = ynz ? rcp_exact(dy) : kInf;
    const int sx = dx > 0.0f ? 1 : -1;
    const int sy = dy > 0.0f ? 1 : -1;
    int bx = dx > 0.0f ? ix + 1 : ix;
    int by = dy > 0.0f ? iy + 1 : iy;
    float tx = xnz ? ((float)bx - fx) * inv_dx : kInf;
    float ty = ynz ? ((float)by - fy) * inv_dy : kInf;
    // a walk of tmax metres enters fewer than 2*tmax/cell + 2 cells; the cap only ever triggers on
    // non-finite / absurd inputs and guarantees termination on the GPU
    for (int guard = 0; guard < kMaxMarchSteps; ++guard) {
        float t;
        if (tx < ty) {
            t = tx;
            ix += sx;
            bx += sx;
            tx = ((float)bx - fx) * inv_dx;
        } else {
            t = ty;
            iy += sy;
            by += sy;
            ty = ynz ? ((float)by - fy) * inv_dy : kInf;
        }
        if (t >= tmax_c) return tmax;
        if (occ(ix, iy)) return t * g.cell;
    }
    return tmax;
}

// ------------------------------------------------------------------------------------------
// Same result as grid_march, far fewer steps: empty-space skipping over a per-cell field of FREE
// RECTANGLES.  A ray only ever needs free space AHEAD of it, so the field stores, for every empty cell, FOUR
// rectangles of empty cells -- one per quadrant of the direction of travel, each with the cell in the corner the ray
// enters through: quadrant q = (dx > 0) | (dy > 0) << 1 has the 16-bit entry ex | ey << 8, meaning the cells
// [ix, ix + sx*ex] x [iy, iy + sy*ey] are all empty -- and 0xFFFF in all four for an occupied cell.  A ray jumps
// straight to the face where it leaves the rectangle of the cell it is in and looks again: ~1.2 lookups per lidar ray
// on the Stage-1 rink, and the cell bitmap is never read.  (Rounds 1-3 stored ONE rectangle per cell, grown on all four
// sides -- the two extents a ray does not use narrowed the two it does: 1.7 lookups per ray, and 3.0 instead of 2.6 loop
// iterations per 64-beam wavefront; tools/field_probe.cpp)
//
// Exactness: grid_march is a 2-way merge of the x-crossing events tx(b) and the y-crossing events
// ty(b) (both monotone in b), ties -> y first.  Leaving the rectangle through its x face Bx happens
// iff tx(Bx) < ty(By); at that moment exactly the y events with ty(b) <= tx(Bx) have been consumed
// (symmetrically: x events with tx(b) < ty(By)).  The next pending boundary is found from an
// arithmetic estimate that is always within one boundary of the truth and is settled with the SAME
// closed-form times, so the walk resumes in precisely the cell, and with precisely the pending
// boundaries, the cell-by-cell walk would have -- every later comparison, and the returned entry
// time, are bit-identical.  ANY field of valid empty rectangles gives the same numbers; the field only decides how many
// jumps a ray takes.
//
// Granularity (measured, 4096 / 8228 robots, profiles/r01/r01_u..z_ablation.txt): 4x4-cell blocks with
// 4-bit extents 48 / 130 us per ray-cast launch on stage-1 / stage-2; 2x2-cell blocks with 8-bit
// extents 38 / 77 us; per cell 34 / 43 us.  The per-cell field is 8 bytes per cell: 1.3 / 5.1 MB for the 20 m / 40 m
// maps at 0.05 m.
constexpr int kFieldMaxExtent = 254;             // cells per side (8-bit; 255 | 255 << 8 marks an occupied cell)
constexpr uint32_t kCellOccupied = 0xFFFFu;      // a quadrant entry of an occupied cell

// The field is stored with a border of empty cells (kFieldPadX columns left/right, kFieldPadY rows
// below/above, value 0 = "empty, no extent"): clamping the cell coordinates into the border replaces
// every bounds check, and cells outside the map read as what they are.
constexpr int kFieldPadX = 2, kFieldPadY = 1;
MRCA_HD int imin(int a, int b) { return a < b ? a : b; }
MRCA_HD int imax(int a, int b) { return a > b ? a : b; }

// median of three: clamp(x, lo, hi) for lo <= hi in ONE v_med3_i32 on the device (the compiler cannot prove
// lo <= hi for run-time bounds and would emit max + min).  LO is a small compile-time constant (an inline
// operand), hi may live in a scalar register.
MRCA_HD int med3_i32(int a, int b, int c) {
#if defined(__HIP_DEVICE_COMPILE__)
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    const int lo = imin(a, imin(b, c)), hi = imax(a, imax(b, c));
    return a + b + c - lo - hi;   // no overflow for the operands used here (two of them < 2^20, one < 2^30)
#endif
}
// a * b + c for |a|, |b| < 2^23 in ONE full-rate v_mad_i32_i24 (left to itself the compiler turns the multiply of an
// operand it knows to be small into v_mad_u64_u32: a quarter-rate 64-bit instruction in the march's inner loop)
MRCA_HD int mad24(int a, int b, int c) {
#if defined(__HIP_DEVICE_COMPILE__)
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    return a * b + c;
#endif
}
template <int LO>
MRCA_HD int clamp_from(int x, int hi) {   // clamp(x, LO, hi), LO <= hi
#if defined(__HIP_DEVICE_COMPILE__)
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "n"(LO), "s"(hi));
    return r;
#else
    return imin(imax(x, LO), hi);
#endif
}

struct FreeRectField {   // quadrant free-rectangle field straight from global memory (L1/L2-resident)
    const uint16_t* d;   // base of the padded array: 4 entries (quadrants 0..3) per cell
    int32_t w, h, pitch; // map size in cells, CELLS per padded row
    // byte offset of cell (ix, iy)'s record from d (cells outside the map clamp into the zero border)
    MRCA_HD uint32_t cell_offset(int ix, int iy) const {
        const int x = clamp_from<-kFieldPadX>(ix, w + kFieldPadX - 1);
        const int y = clamp_from<-kFieldPadY>(iy, h + kFieldPadY - 1);
        // |y|, pitch < 2^23: 24-bit multiply-add
#if defined(__HIP_DEVICE_COMPILE__)
        const int idx = __mul24(y, pitch) + x;
#else
        const int idx = y * pitch + x;
#endif
        return (uint32_t)(idx + (kFieldPadY * pitch + kFieldPadX)) << 3;
    }
    // the entry of quadrant q (qbytes = 2 * q) of a cell
    MRCA_HD uint32_t operator()(int ix, int iy, uint32_t qbytes) const {
        return *reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(d) + (cell_offset(ix, iy) + qbytes));
    }
    // all four entries of a cell: .x = quadrants 0 | 1 << 16, .y = quadrants 2 | 3 << 16 (what a `head` record carries)
    MRCA_HD void cell(int ix, int iy, uint32_t* lo, uint32_t* hi) const {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(d) + cell_offset(ix, iy));
        *lo = p[0];
        *hi = p[1];
    }
};

// What all rays leaving one point share: the origin in cell units, its cell and that cell's field entry
// (a lidar computes it once per robot, not once per beam).
struct MarchOrigin {
    float fx, fy;
    int32_t ix0, iy0;
    uint32_t v_lo, v_hi;     // the four quadrant entries of the origin's cell (FreeRectField::cell)
};
template <class Field>
MRCA_HD MarchOrigin march_origin(const Field& field, const GridGeom& g, float ox, float oy) {
    MarchOrigin o;
    o.fx = (ox - g.x0) * g.inv_cell;
    o.fy = (oy - g.y0) * g.inv_cell;
    o.ix0 = (int)floorf(o.fx);
    o.iy0 = (int)floorf(o.fy);
    field.cell(o.ix0, o.iy0, &o.v_lo, &o.v_hi);
    return o;
}

// Flat, branch-poor form (one loop, one jump per iteration, x/y handled by selects) so the 64 rays of
// a wavefront stay in lock step.  No (tx, ty) state is carried -- boundary times are always re-derived
// from the closed form, which is what makes every path through here produce the same numbers as
// grid_march.
// x exit of a jump: a crossing is consumed when its time is <= t; y exit: when it is < t.  Both as ONE strict comparison against
// this bound: t itself, or the next float above it.  t >= 0 (a time along the ray; a -0 counts as +0), finite.
MRCA_HD float at_or_before(float t, bool inclusive) {
    uint32_t u;
    __builtin_memcpy(&u, &t, 4);
    u = (u & 0x7FFFFFFFu) + (inclusive ? 1u : 0u);
    float r;
    __builtin_memcpy(&r, &u, 4);
    return r;
}

// a loop counter kept PER LANE in a vector register (the compiler would otherwise hold a uniform counter in an SGPR and turn
// its test into a lane mask: s_add, s_cmp, s_cselect, s_or, s_mov per trip)
MRCA_HD int lane_counter(int x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(x));
#endif
    return x;
}

template <class Field>
MRCA_HD float grid_march_skip(const Field& field, const GridGeom& g, const MarchOrigin& org, float dx, float dy,
                              float tmax) {
    const float fx = org.fx, fy = org.fy;
    const int ix0 = org.ix0, iy0 = org.iy0;
    const float tmax_c = tmax * g.inv_cell;
    const bool xnz = dx != 0.0f, ynz = dy != 0.0f;
    const bool xpos = dx > 0.0f, ypos = dy > 0.0f;
    // the ray's quadrant picks which of the four rectangles of a cell it reads -- from the origin's cell onwards
    const uint32_t qbytes = (xpos ? 2u : 0u) + (ypos ? 4u : 0u);
    const uint32_t vq = ypos ? org.v_hi : org.v_lo;
    uint32_t v = xpos ? vq >> 16 : vq & 0xFFFFu;  // carried: one field lookup per jump
    if (v == kCellOccupied) return 0.0f;
    if (!(tmax_c > 0.0f)) return tmax;
    const float rdx = rcp_exact(dx), rdy = rcp_exact(dy);   // (unconditionally: a select, not a branch around the ballot inside)
    const float inv_dx = xnz ? rdx : kInf;
    const float inv_dy = ynz ? rdy : kInf;
    const int sx = xpos ? 1 : -1, sy = ypos ? 1 : -1;
    const int ux = xpos ? 1 : 0, uy = ypos ? 1 : 0;      // current cell = pending boundary - u, on each axis
    // a pending boundary only ever moves in the direction of travel: clamp(b, from b0) = med3(b, b0, lim)
    const int limx = xpos ? 0x3FFFFFFF : -0x3FFFFFFF, limy = ypos ? 0x3FFFFFFF : -0x3FFFFFFF;
    // An axis-parallel ray never crosses a boundary of its zero axis -- which then is always the secondary
    // axis below (its exit time is +inf), so the whole estimate is skipped for such a ray.
    const bool bothnz = xnz & ynz;
    // The state is just the next pending boundary on each axis (the cell follows from it).
    int bx = ix0 + ux;
    int by = iy0 + uy;
    // The loop's bookkeeping the vector way (a scalar instruction costs the launch twice a vector one, profiles/r06_ac_*):
    //   * the result is selected inside the loop -- a ray hits once, then leaves it;
    //   * ONE exit: `left`, a per-lane trip counter that a hit, the end of the beam and the trip guard all set to 0 (as lane
    //     MASKS the three exits cost eleven scalar instructions per trip);
    //   * the secondary axis is settled for every lane and selected away for an axis-parallel ray (its numbers are NaNs then,
    //     never used): a branch around it cost the exec-mask bookkeeping of a region per trip;
    //   * "consumed at or before t" is "before the next float above t" (t >= 0 by construction; -0 counts as +0; denormals are
    //     honoured, .amdhsa_float_denorm_mode_32 3): one bit trick instead of an equality compare and two mask operations per
    //     candidate.
    float out = tmax;
    int left = lane_counter(kMaxMarchSteps);
    do {
        // faces of the rectangle known to be free: e cells beyond the current cell's own far face
        const int ex = (int)(v & 255u), ey = (int)((v >> 8) & 255u);   // (both as 8-bit fields: one v_and / v_bfe + a 24-bit mad each)
        const int Bx = mad24(ex, sx, bx);
        const int By = mad24(ey, sy, by);
        const float rawx = ((float)Bx - fx) * inv_dx;
        const float rawy = ((float)By - fy) * inv_dy;
        const float tBx = xnz ? rawx : kInf;
        const float tBy = ynz ? rawy : kInf;
        const bool xe = tBx < tBy;  // leaves through the x face (ties: y first)
        const float t = xe ? tBx : tBy;
        if (t >= tmax_c) {
            left = 0;
        } else {
            // the other ("secondary") axis: which of its crossings were consumed before time t?
            // x exit: y crossings with ty(b) <= t;  y exit: x crossings with tx(b) < t.
            const float fS = xe ? fy : fx;
            const float invS = xe ? inv_dy : inv_dx;
            const int sS = xe ? sy : sx;
            const int bS0 = xe ? by : bx;
            // position on the secondary axis at time t.  The consumed crossings are exactly those on
            // the near side of p* = fS + t*dS*(1 +- 2.4e-7) (rounding of 1/dS and of the closed form),
            // and pT differs from p* by far less than one cell (|t*dS| <= 170, |fS| <= 2^16: < 0.01),
            // so the first pending boundary is floor(pT) (+1 going up) or one of its two neighbours.
            // Which one is decided with the closed-form times themselves, branch-free (a data-dependent
            // "only when close to a boundary" test would make the whole wavefront take the slow path
            // whenever one of its 64 rays is close).
            const float pT = fS + (xe ? dy : dx) * t;
            const int b = med3_i32((int)floorf(pT) + (xe ? uy : ux), bS0, xe ? limy : limx);
            const int bprev = b - sS;
            const float tp = ((float)bprev - fS) * invS;      // the crossing before b ...
            const float tc = ((float)b - fS) * invS;          // ... and b itself: consumed before t?
            const float tle = at_or_before(t, xe);            // x exit: "<= t", y exit: "< t"
            const bool cons_p = tp < tle;
            const bool cons_c = tc < tle;
            int bS = cons_c ? b + sS : b;                     // cons_c implies cons_p (times are monotone)
            bS = ((b != bS0) & !cons_p) ? bprev : bS;
            // An axis-parallel ray never crosses a boundary of its zero axis -- which then is always the secondary
            // axis (its exit time is +inf): it keeps its pending boundary
            bS = bothnz ? bS : bS0;
            // pending boundaries after the jump: the exit-axis face is crossed, the secondary axis resumes at bS
            bx = xe ? Bx + sx : bS;
            by = xe ? bS : By + sy;
            v = field(bx - ux, by - uy, qbytes);  // cells outside the map read as empty (the zero border)
            const bool hit = v == kCellOccupied;
            out = hit ? t * g.cell : out;
            left = hit ? 0 : left - 1;
        }
        left = lane_counter(left);
    } while (left > 0);
    return out;
}

template <class Field>
MRCA_HD float grid_march_skip(const Field& field, const GridGeom& g, float ox, float oy, float dx, float dy,
                              float tmax) {
    return grid_march_skip(field, g, march_origin(field, g, ox, oy), dx, dy, tmax);
}

// K rays from the same origin marched in LOCK STEP by one thread: the same per-ray arithmetic as
// grid_march_skip (every number a ray produces is bit-identical), but the K field lookups of an iteration
// are independent and issued back to back, so one wait covers K dependent-load latencies instead of one.
// The loop runs while any ray is still marching; a finished ray keeps re-reading its last cell (an L1 hit)
// and commits nothing.  Per wavefront the iteration count is max over lanes and rays instead of the sum over
// rays of the max over lanes -- the VALU work is the same, the exposed latency is 1/K.
template <int K, class Field>
MRCA_HD void grid_march_skip_n(const Field& field, const GridGeom& g, const MarchOrigin& org, const float (&dx)[K],
                               const float (&dy)[K], float tmax, float (&out)[K]) {
    const float fx = org.fx, fy = org.fy;
    const float tmax_c = tmax * g.inv_cell;
    if ((org.v_lo & 0xFFFFu) == kCellOccupied) {      // an occupied cell says so in all four quadrants
#pragma unroll
        for (int k = 0; k < K; ++k) out[k] = 0.0f;
        return;
    }
    if (!(tmax_c > 0.0f)) {
#pragma unroll
        for (int k = 0; k < K; ++k) out[k] = tmax;
        return;
    }
    float inv_dx[K], inv_dy[K], t[K];
    int bx[K], by[K];
    uint32_t v[K], qb[K];
    bool act[K], hit[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        inv_dx[k] = dx[k] != 0.0f ? rcp_exact(dx[k]) : kInf;
        inv_dy[k] = dy[k] != 0.0f ? rcp_exact(dy[k]) : kInf;
        bx[k] = org.ix0 + (dx[k] > 0.0f ? 1 : 0);
        by[k] = org.iy0 + (dy[k] > 0.0f ? 1 : 0);
        qb[k] = (dx[k] > 0.0f ? 2u : 0u) + (dy[k] > 0.0f ? 4u : 0u);
        const uint32_t vq = dy[k] > 0.0f ? org.v_hi : org.v_lo;
        v[k] = dx[k] > 0.0f ? vq >> 16 : vq & 0xFFFFu;
        t[k] = 0.0f;
        act[k] = true;
        hit[k] = false;
    }
    int guard = kMaxMarchSteps;
    bool any = true;
    while (any) {
        int cx[K], cy[K], nbx[K], nby[K];
        float tn[K];
        bool go[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const bool xnz = dx[k] != 0.0f, ynz = dy[k] != 0.0f;
            const bool xpos = dx[k] > 0.0f, ypos = dy[k] > 0.0f;
            const int sx = xpos ? 1 : -1, sy = ypos ? 1 : -1;
            const int ux = xpos ? 1 : 0, uy = ypos ? 1 : 0;
            const int limx = xpos ? 0x3FFFFFFF : -0x3FFFFFFF, limy = ypos ? 0x3FFFFFFF : -0x3FFFFFFF;
            const int ex = (int)(v[k] & 255u), ey = (int)((v[k] >> 8) & 255u);
            const int Bx = mad24(ex, sx, bx[k]);
            const int By = mad24(ey, sy, by[k]);
            const float rawx = ((float)Bx - fx) * inv_dx[k];
            const float rawy = ((float)By - fy) * inv_dy[k];
            const float tBx = xnz ? rawx : kInf;
            const float tBy = ynz ? rawy : kInf;
            const bool xe = tBx < tBy;
            const float tt = xe ? tBx : tBy;
            const float fS = xe ? fy : fx;
            const float invS = xe ? inv_dy[k] : inv_dx[k];
            const int sS = xe ? sy : sx;
            const int bS0 = xe ? by[k] : bx[k];
            int bS = bS0;
            if (xnz & ynz) {
                const float pT = fS + (xe ? dy[k] : dx[k]) * tt;
                const int b = med3_i32((int)floorf(pT) + (xe ? uy : ux), bS0, xe ? limy : limx);
                const int bprev = b - sS;
                const float tp = ((float)bprev - fS) * invS;
                const float tc = ((float)b - fS) * invS;
                const bool cons_p = (tp < tt) | ((tp == tt) & xe);
                const bool cons_c = (tc < tt) | ((tc == tt) & xe);
                bS = cons_c ? b + sS : b;
                bS = ((b != bS0) & !cons_p) ? bprev : bS;
            }
            go[k] = act[k] && !(tt >= tmax_c);
            tn[k] = tt;
            nbx[k] = go[k] ? (xe ? Bx + sx : bS) : bx[k];
            nby[k] = go[k] ? (xe ? bS : By + sy) : by[k];
            cx[k] = nbx[k] - ux;
            cy[k] = nby[k] - uy;
        }
        uint32_t nv[K];
#pragma unroll
        for (int k = 0; k < K; ++k) nv[k] = field(cx[k], cy[k], qb[k]);   // K independent lookups in flight
        --guard;
        any = false;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            bx[k] = nbx[k];
            by[k] = nby[k];
            t[k] = go[k] ? tn[k] : t[k];
            v[k] = go[k] ? nv[k] : v[k];
            hit[k] = go[k] ? (nv[k] == kCellOccupied) : hit[k];
            act[k] = go[k] && !hit[k] && guard > 0;
            any = any || act[k];
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) out[k] = hit[k] ? t[k] * g.cell : tmax;
}

// scan / 6 - 0.5 (stage_world1.py:140).  The quotient is formed as q = x * RN(1/6), r = fma(-q, 6, x),
// q' = fma(r, RN(1/6), q) -- Markstein's correction, 3 instructions instead of the ~12 of an IEEE division --
// and q' - 0.5 equals RN(RN(x / 6) - 0.5) for EVERY float x in [0, 6] (checked exhaustively over all 1 086 324 737
// of them: tests/test_geometry_properties.py::test_norm_obs_equals_ieee_division samples them, tools/check_div6.c
// is the full sweep; q' itself differs from RN(x / 6) only where x / 6 is subnormal, which the subtraction
// absorbs).
MRCA_HD float norm_obs(float x) {
    const float inv6 = 1.0f / 6.0f;
    const float q = x * inv6;
    const float r = __builtin_fmaf(-q, 6.0f, x);
    return __builtin_fmaf(r, inv6, q) - 0.5f;
}

// ------------------------------------------------------------------------------------------
// Robot outline (0.44 x 0.38 rectangle) against the grid: march the four edges.
// Edge k of the footprint (corner k -> corner k+1, corners (+,+),(-,+),(-,-),(+,-)): does the walk
// along it meet an occupied cell?
template <class Occ>
MRCA_HD bool static_edge_hit(const Occ& occ, const GridGeom& g, float x, float y, float s, float c, int k) {
    const float hx = (k == 0 || k == 3) ? kHalfLen : -kHalfLen;
    const float hy = (k < 2) ? kHalfWid : -kHalfWid;
    const float ex = (k == 0) ? -c : (k == 1) ? s : (k == 2) ? c : -s;
    const float ey = (k == 0) ? -s : (k == 1) ? -c : (k == 2) ? s : c;
    const float el = (k & 1) ? 2.0f * kHalfWid : 2.0f * kHalfLen;
    const float cx = x + (hx * c - hy * s);
    const float cy = y + (hx * s + hy * c);
    return grid_march(occ, g, cx, cy, ex, ey, el) < el;
}

// ------------------------------------------------------------------------------------------
// The same edge test WITHOUT walking (round 5).  grid_march's walk is a merge of the x crossings and the y crossings of
// the edge; every crossing that happens within the walk (its time t < tmax in cell units -- crossings are taken in time
// order, so those are exactly the ones taken) enters one cell, and WHICH cell follows from a count: after the i-th x
// crossing the ray is in column ix0 + sx * i and in the row it has reached by then -- the y crossings with ty <= tx_i are
// behind it (ties step in y first) -- and after the j-th y crossing in row iy0 + sy * j and the column given by the x
// crossings with tx < ty_j.  The count is grid_march_skip's settlement: an arithmetic estimate of the pending boundary,
// within one of the truth, settled with the same closed-form times.  So one lane can take ONE event -- q = 0 the start
// cell, 1..Q the x crossings, Q + 1..2Q the y crossings -- and an edge is 2Q + 1 independent lanes instead of a chain of
// ~13 dependent steps of one (move_kernel: 6 000 of a workgroup's 13 000 - 22 000 ticks whenever a robot of the world is
// near a wall, profiles/r05_r_move_tail.txt).  The edge hits iff some event enters an occupied cell at t * cell < tmax:
// entry times never decrease along the walk, so if the FIRST occupied cell fails that test every later one does, and if it
// passes the OR is true -- grid_march(...) < tmax exactly.  Q = edge_event_slots(): crossings per axis an edge can have.
MRCA_HD int edge_event_slots(float inv_cell) { return (int)ceilf(2.0f * kHalfLen * inv_cell) + 1; }

template <class Occ>
MRCA_HD bool walk_event_hits(const Occ& occ, const GridGeom& g, float ox, float oy, float dx, float dy, float tmax, int q,
                             int Q) {
    const float fx = (ox - g.x0) * g.inv_cell;
    const float fy = (oy - g.y0) * g.inv_cell;
    const int ix0 = (int)floorf(fx);
    const int iy0 = (int)floorf(fy);
    if (q == 0) return occ(ix0, iy0);                  // grid_march: an occupied start cell is a hit at distance 0
    const float tmax_c = tmax * g.inv_cell;
    if (!(tmax_c > 0.0f)) return false;
    const bool isx = q <= Q;                           // an x crossing (the primary axis P = x), else a y crossing (P = y)
    const int i = isx ? q : q - Q;                     // the i-th crossing of the primary axis, 1-based
    const float fP = isx ? fx : fy, dP = isx ? dx : dy;
    const float fS = isx ? fy : fx, dS = isx ? dy : dx;
    const int iP0 = isx ? ix0 : iy0, iS0 = isx ? iy0 : ix0;
    if (!(dP != 0.0f)) return false;                   // an axis the edge does not move along is never crossed
    const float invP = rcp_exact(dP);
    const int sP = dP > 0.0f ? 1 : -1, uP = dP > 0.0f ? 1 : 0;
    const float t = ((float)(iP0 + uP + sP * (i - 1)) - fP) * invP;        // grid_march's tx / ty of that boundary
    if (!(t < tmax_c)) return false;                   // the walk ends at the first crossing with t >= tmax
    // the other axis: its first pending boundary once every crossing that the merge takes BEFORE this event is consumed
    // (before an x crossing: ty <= t; before a y crossing: tx < t) -- grid_march_skip's settlement
    const int sS = dS > 0.0f ? 1 : -1, uS = dS > 0.0f ? 1 : 0;
    const int bS0 = iS0 + uS;
    int bS = bS0;
    if (dS != 0.0f) {
        const float invS = rcp_exact(dS);
        const float pT = fS + dS * t;
        const int b = med3_i32((int)floorf(pT) + uS, bS0, dS > 0.0f ? 0x3FFFFFFF : -0x3FFFFFFF);
        const int bprev = b - sS;
        const float tp = ((float)bprev - fS) * invS;
        const float tc = ((float)b - fS) * invS;
        const bool cons_p = (tp < t) | ((tp == t) & isx);
        const bool cons_c = (tc < t) | ((tc == t) & isx);
        bS = cons_c ? b + sS : b;
        bS = ((b != bS0) & !cons_p) ? bprev : bS;
    }
    const int cP = iP0 + sP * i, cS = bS - uS;
    return occ(isx ? cP : cS, isx ? cS : cP) && (t * g.cell < tmax);
}

// event q of edge k of the footprint at (x, y, sin, cos): the OR over q = 0 .. 2Q is static_edge_hit(..., k)
template <class Occ>
MRCA_HD bool static_edge_event_hits(const Occ& occ, const GridGeom& g, float x, float y, float s, float c, int k, int q, int Q) {
    const float hx = (k == 0 || k == 3) ? kHalfLen : -kHalfLen;
    const float hy = (k < 2) ? kHalfWid : -kHalfWid;
    const float ex = (k == 0) ? -c : (k == 1) ? s : (k == 2) ? c : -s;
    const float ey = (k == 0) ? -s : (k == 1) ? -c : (k == 2) ? s : c;
    const float el = (k & 1) ? 2.0f * kHalfWid : 2.0f * kHalfLen;
    const float cx = x + (hx * c - hy * s);
    const float cy = y + (hx * s + hy * c);
    return walk_event_hits(occ, g, cx, cy, ex, ey, el, q, Q);
}

template <class Occ>
MRCA_HD bool static_hit(const Occ& occ, const GridGeom& g, float x, float y, float s, float c) {
    bool hit = false;
    for (int k = 0; k < 4; ++k) hit = static_edge_hit(occ, g, x, y, s, c, k) || hit;
    return hit;
}

// ------------------------------------------------------------------------------------------
// Stage-like RASTER collision between robots (fidelity mode, mrca_config.collision_raster = res > 0): Stage maps a
// model's outline into its world raster (cells of `resolution` metres, 0.2 m in stage1/2.world:3) and reports a
// collision when a cell of the outline also holds another model [libstage, SURVEY Appendix B].  Restated: the outline
// cells of a pose are the cells the closed-form grid walk visits along its four edges (the walk of grid_march on a
// raster of `res` metres aligned at the world origin, start cell included, cells entered at t < edge length); two
// robots collide iff their outlines share a cell.  At most kMaxOutlineCells per robot (res >= 0.1 m).
constexpr int kMaxOutlineCells = 40;
MRCA_HD long long pack_cell(int ix, int iy) { return (long long)(((unsigned long long)(unsigned int)ix << 32) | (unsigned int)iy); }

template <class Emit>
MRCA_HD void walk_cells(float inv_res, float ox, float oy, float dx, float dy, float tmax, Emit&& emit) {
    const float fx = ox * inv_res;
    const float fy = oy * inv_res;
    int ix = (int)floorf(fx);
    int iy = (int)floorf(fy);
    const float tmax_c = tmax * inv_res;
    emit(ix, iy);
    if (!(tmax_c > 0.0f)) return;
    const bool xnz = dx != 0.0f, ynz = dy != 0.0f;
    const float inv_dx = xnz ? rcp_exact(dx) : kInf;
    const float inv_dy = ynz ? rcp_exact(dy) : kInf;
    const int sx = dx > 0.0f ? 1 : -1;
    const int sy = dy > 0.0f ? 1 : -1;
    int bx = dx > 0.0f ? ix + 1 : ix;
    int by = dy > 0.0f ? iy + 1 : iy;
    float tx = xnz ? ((float)bx - fx) * inv_dx : kInf;
    float ty = ynz ? ((float)by - fy) * inv_dy : kInf;
    for (int guard = 0; guard < 4 * kMaxOutlineCells; ++guard) {
        float t;
        if (tx < ty) {
            t = tx;
            ix += sx;
            bx += sx;
            tx = ((float)bx - fx) * inv_dx;
        } else {
            t = ty;
            iy += sy;
            by += sy;
            ty = ynz ? ((float)by - fy) * inv_dy : kInf;
        }
        if (t >= tmax_c) return;
        emit(ix, iy);
    }
}

// outline cells of the 0.44 x 0.38 footprint at (x, y, sin, cos): edges as in static_edge_hit; returns the count
MRCA_HD int outline_cells(float inv_res, float x, float y, float s, float c, long long* out) {
    int n = 0;
    for (int k = 0; k < 4; ++k) {
        const float hx = (k == 0 || k == 3) ? kHalfLen : -kHalfLen;
        const float hy = (k < 2) ? kHalfWid : -kHalfWid;
        const float ex = (k == 0) ? -c : (k == 1) ? s : (k == 2) ? c : -s;
        const float ey = (k == 0) ? -s : (k == 1) ? -c : (k == 2) ? s : c;
        const float el = (k & 1) ? 2.0f * kHalfWid : 2.0f * kHalfLen;
        const float cx = x + (hx * c - hy * s);
        const float cy = y + (hx * s + hy * c);
        walk_cells(inv_res, cx, cy, ex, ey, el, [&](int ix, int iy) {
            if (n < kMaxOutlineCells) out[n++] = pack_cell(ix, iy);
        });
    }
    return n;
}

MRCA_HD bool cells_intersect(const long long* a, int na, const long long* b, int nb) {
    bool hit = false;
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) hit = hit || (a[i] == b[j]);
    return hit;
}

// The same outline as an ANCHORED BITMAP (what the kernels keep: round 5).  Every point of the outline lies within the
// footprint's circumradius (0.29069 m) of the centre, so its raster cells fall into a window of
// floor(2 * 0.2927 / res) + 2 <= 7 columns / rows (res >= 0.1 m) that starts at the cell of (centre - 0.2927 m): the SET of
// outline cells -- all the collision rule and the raster lidar ever ask about -- is 64 bits, bit jy * 8 + jx for the cell
// (ax + jx, ay + jy), plus the anchor.  A robot's record is 16 bytes, lives in registers, and is published per robot next
// to its `head` record by whoever moves it (EnvView::outline); two outlines meet iff one bitmap, shifted by the anchors'
// difference, ANDs with the other -- a dozen integer instructions where rounds 3-4 compared two 40-entry cell lists in LDS
// (1600 64-bit compares per pair, between two workgroup barriers per turn of the ordered pass).
constexpr float kOutlineReach = 0.2927f;   // circumradius + 2 mm: far beyond any rounding of the corner positions
constexpr int kOutlineWin = 8;             // window side in cells
struct OutlineBits {
    int32_t ax, ay;       // raster cell of the window's (0, 0) corner
    uint32_t lo, hi;      // rows 0..3 | rows 4..7, bit (jy & 3) * 8 + jx
};
MRCA_HD int outline_anchor(float v, float inv_res) { return (int)floorf((v - kOutlineReach) * inv_res); }
// columns / rows of the window an outline can occupy at this resolution (host side: validation, kernel variant)
MRCA_HD int outline_span(float inv_res) { return (int)floorf((0.2907f + kOutlineReach + 0.002f) * inv_res) + 2; }

// the cells of outline edge k of the pose (x, y, sin, cos) -- walk_cells, i.e. exactly outline_cells' cells -- ORed into
// (*lo, *hi) relative to the anchor; false if a cell fell outside the window (never, for res >= 0.1 m: the callers raise
// the env's status word instead of dropping it silently)
MRCA_HD bool outline_edge_bits(float inv_res, float x, float y, float s, float c, int k, int ax, int ay, uint32_t* lo,
                               uint32_t* hi) {
    const float hx = (k == 0 || k == 3) ? kHalfLen : -kHalfLen;
    const float hy = (k < 2) ? kHalfWid : -kHalfWid;
    const float ex = (k == 0) ? -c : (k == 1) ? s : (k == 2) ? c : -s;
    const float ey = (k == 0) ? -s : (k == 1) ? -c : (k == 2) ? s : c;
    const float el = (k & 1) ? 2.0f * kHalfWid : 2.0f * kHalfLen;
    const float cx = x + (hx * c - hy * s);
    const float cy = y + (hx * s + hy * c);
    bool ok = true;
    uint32_t l = *lo, h = *hi;
    walk_cells(inv_res, cx, cy, ex, ey, el, [&](int ix, int iy) {
        const int jx = ix - ax, jy = iy - ay;
        const bool in = (unsigned)jx < (unsigned)kOutlineWin && (unsigned)jy < (unsigned)kOutlineWin;
        ok = ok && in;
        const uint32_t m = in ? 1u << (((jy & 3) << 3) + jx) : 0u;
        l |= jy < 4 ? m : 0u;
        h |= jy < 4 ? 0u : m;
    });
    *lo = l;
    *hi = h;
    return ok;
}
MRCA_HD bool outline_bits(float inv_res, float x, float y, float s, float c, OutlineBits* out) {
    OutlineBits o{outline_anchor(x, inv_res), outline_anchor(y, inv_res), 0u, 0u};
    bool ok = true;
    for (int k = 0; k < 4; ++k) ok = outline_edge_bits(inv_res, x, y, s, c, k, o.ax, o.ay, &o.lo, &o.hi) && ok;
    *out = o;
    return ok;
}

// do two outlines share a raster cell?  q's bitmap is moved into p's window (columns by the anchors' x difference, rows by
// the y difference; what leaves the window cannot be a cell of p) and ANDed.
MRCA_HD bool outline_bits_meet(const OutlineBits& p, const OutlineBits& q) {
    const int dx = q.ax - p.ax, dy = q.ay - p.ay;
    if (dx <= -kOutlineWin || dx >= kOutlineWin || dy <= -kOutlineWin || dy >= kOutlineWin) return false;
    const uint64_t P = (uint64_t)p.lo | ((uint64_t)p.hi << 32);
    uint64_t Q = (uint64_t)q.lo | ((uint64_t)q.hi << 32);
    const int adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
    const uint64_t keep = 0x0101010101010101ull * (uint64_t)(0xFFu >> adx);      // the columns that stay inside the window
    Q = dx >= 0 ? (Q & keep) << adx : (Q >> adx) & keep;
    Q = dy >= 0 ? Q << (8 * ady) : Q >> (8 * ady);
    return (P & Q) != 0ull;
}

// Fidelity mode's lidar return of ONE other robot, in closed form.  grid_march's walk is a two-way merge of the x
// crossings tx(b) = (float(b) - fx) * (1 / dx) and the y crossings ty(b) (both monotone in b; tx < ty steps in x, ties and
// ty < tx in y), so WHICH cells a ray visits and WHEN it enters them follows from four crossing times per cell, without
// walking: the ray is in column c from the crossing of c's near face (time -inf for the origin's column, +inf for a column
// behind the origin or for any other column of a ray with dx = 0) until the crossing of its far face, likewise for rows;
// cell (c, r) is visited iff   enterX(c) < leaveY(r)   (the merge takes x_i before y_{j+1} iff tx_i < ty_{j+1})   and
// not leaveX(c) < enterY(r)   (it takes y_j before x_{i+1} iff not tx_{i+1} < ty_j),   and is entered at
// max(enterX(c), enterY(r)) -- the later of the two crossings, the value grid_march returns.  Entry times never decrease
// along the walk, so the first marked cell the walk meets is the marked visited cell with the smallest entry time, and the
// minimum over several robots' outlines is the walk through the union of their marks.  KW x KW cells of the neighbour's
// window are tested (KW = 4 covers res >= 0.195 m -- Stage's 0.2 m -- KW = 8 everything down to 0.1 m).  Returns the entry time
// in CELLS (0 when the origin's cell is marked), +inf when the ray meets no marked cell; the caller applies t < tmax and
// * res exactly as grid_march does.  Rounds 3-4 walked a window of LDS bits cell by cell (up to 60 dependent LDS reads per
// beam); tests/test_fidelity_closed_form.py holds this against that walk on adversarial rays.
template <int KW>
MRCA_HD float ray_outline_entry(float fx, float fy, int ix0, int iy0, float dx, float dy, float inv_dx, float inv_dy,
                                const OutlineBits& o) {
    const bool xnz = dx != 0.0f, ynz = dy != 0.0f;
    const bool xpos = dx > 0.0f, ypos = dy > 0.0f;
    float eX[KW], lX[KW], eY[KW], lY[KW];
    {
        float T[KW + 1];
#pragma unroll
        for (int q = 0; q <= KW; ++q) T[q] = ((float)(o.ax + q) - fx) * inv_dx;
#pragma unroll
        for (int j = 0; j < KW; ++j) {
            const int rel = o.ax + j - ix0;
            const bool behind = xpos ? rel < 0 : rel > 0;
            const float en = xpos ? T[j] : T[j + 1], lv = xpos ? T[j + 1] : T[j];
            eX[j] = rel == 0 ? -kInf : ((behind || !xnz) ? kInf : en);
            lX[j] = xnz ? lv : kInf;
        }
#pragma unroll
        for (int q = 0; q <= KW; ++q) T[q] = ((float)(o.ay + q) - fy) * inv_dy;
#pragma unroll
        for (int j = 0; j < KW; ++j) {
            const int rel = o.ay + j - iy0;
            const bool behind = ypos ? rel < 0 : rel > 0;
            const float en = ypos ? T[j] : T[j + 1], lv = ypos ? T[j + 1] : T[j];
            eY[j] = rel == 0 ? -kInf : ((behind || !ynz) ? kInf : en);
            lY[j] = ynz ? lv : kInf;
        }
    }
    float best = kInf;
#pragma unroll
    for (int jy = 0; jy < KW; ++jy) {
        const uint32_t row = ((jy < 4 ? o.lo : o.hi) >> ((jy & 3) << 3)) & 0xFFu;
#pragma unroll
        for (int jx = 0; jx < KW; ++jx) {
            const bool visited = (eX[jx] < lY[jy]) & !(lX[jx] < eY[jy]) & (((row >> jx) & 1u) != 0u);
            const float t = eX[jx] < eY[jy] ? eY[jy] : eX[jx];
            best = (visited & (t < best)) ? t : best;
        }
    }
    return best > 0.0f ? best : 0.0f;      // the origin's own cell is entered at "-inf": range 0
}

// Separating-axis test of two robot rectangles; touching counts as overlap.
MRCA_HD bool obb_overlap(float xi, float yi, float si, float ci, float xj, float yj, float sj, float cj) {
    const float tx = xj - xi;
    const float ty = yj - yi;
    const float a0 = fabsf(ci * cj + si * sj);
    const float a1 = fabsf(si * cj - ci * sj);
    const float ex = kHalfLen + (kHalfLen * a0 + kHalfWid * a1);
    const float ey = kHalfWid + (kHalfLen * a1 + kHalfWid * a0);
    const bool sep = (fabsf(tx * ci + ty * si) > ex) || (fabsf(ty * ci - tx * si) > ey) ||
                     (fabsf(tx * cj + ty * sj) > ex) || (fabsf(ty * cj - tx * sj) > ey);
    return !sep;
}

MRCA_HD void slab(float lo, float ld, float h, float* t0, float* t1) {
    const bool par = fabsf(ld) < 1e-12f;
    const float inv = rcp_exact(par ? 1.0f : ld);
    const float ta = (-h - lo) * inv;
    const float tb = (h - lo) * inv;
    const bool out = fabsf(lo) > h;
    *t0 = par ? (out ? kInf : -kInf) : (ta < tb ? ta : tb);
    *t1 = par ? (out ? -kInf : kInf) : (ta < tb ? tb : ta);
}

// Entry distance of the ray into robot j's rectangle; +inf on a miss.  Robots are visible to
// each other's lidar: ranger_return 0.5 (stage1.world:95).
// The ray's ORIGIN in robot j's frame is the same for all beams of a lidar: ray_box_origin once per neighbour (the preparing
// wave), ray_box_local per tested beam -- the same expressions in the same order as ray_box, so the same bits.
MRCA_HD void ray_box_origin(float ox, float oy, float xj, float yj, float sj, float cj, float* lx, float* ly) {
    const float rx = ox - xj;
    const float ry = oy - yj;
    *lx = rx * cj + ry * sj;
    *ly = ry * cj - rx * sj;
}
MRCA_HD float ray_box_local(float lx, float ly, float dx, float dy, float sj, float cj);
MRCA_HD float ray_box(float ox, float oy, float dx, float dy, float xj, float yj, float sj, float cj) {
    float lx, ly;
    ray_box_origin(ox, oy, xj, yj, sj, cj, &lx, &ly);
    return ray_box_local(lx, ly, dx, dy, sj, cj);
}
MRCA_HD float ray_box_local(float lx, float ly, float dx, float dy, float sj, float cj) {
    const float ldx = dx * cj + dy * sj;
    const float ldy = dy * cj - dx * sj;
    float t0x, t1x, t0y, t1y;
    slab(lx, ldx, kHalfLen, &t0x, &t1x);
    slab(ly, ldy, kHalfWid, &t0y, &t1y);
    const float tin = t0x > t0y ? t0x : t0y;
    const float tout = t1x < t1y ? t1x : t1y;
    const bool hit = (tin <= tout) && (tout >= 0.0f);
    return hit ? (tin > 0.0f ? tin : 0.0f) : kInf;
}

// Conservative set of beams that can touch another robot: the rectangle lies inside its
// circumscribed circle (radius 0.2907 m), so only beams within asin(r/dist) of the bearing of its
// centre can hit it.  Only the cull uses these numbers, never a reported value, and culled beams provably miss,
// so the minimum over the kept tests equals the minimum over all of them -- which is why the two angles may be
// cheap UPPER / approximate bounds instead of libm calls (each ~100 instructions on the path of the preparing
// wave): asin(x) <= x + (pi/2 - 1) x^3 on [0, 1] (the gap asin(x) - x over x^3 grows monotonically from 1/6 to
// pi/2 - 1), and a degree-7 odd polynomial for atan (|error| < 1.2e-4 rad).  Two beam widths + 1 mm of radius +
// 3e-3 rad of slack absorb the approximation and every rounding in here.
MRCA_HD float atan2_approx(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    const float a = mx > 0.0f ? mn / mx : 0.0f;
    const float z = a * a;
    float r = ((-0.0464964749f * z + 0.15931422f) * z - 0.327622764f) * z * a + a;
    r = ay > ax ? 1.57079637f - r : r;
    r = x < 0.0f ? 3.14159274f - r : r;
    return y < 0.0f ? -r : r;
}

// step = kPi / (beams - 1) and inv_step = (beams - 1) / kPi as fp32 quotients: the same for every robot and beam, so the
// kernel takes them from the host (two IEEE divisions, ~24 VALU instructions, per neighbour candidate otherwise)
// `radius`: what the other robot lies inside of, seen from its centre -- the circumscribed circle of the rectangle + 1 mm
// (0.2917 m), or, in fidelity mode, that plus one raster-cell diagonal (its outline CELLS); `near`: centre distances up
// to which every beam is kept (0.30 m / radius + 0.01 m).
MRCA_HD void beam_interval(float lx, float ly, int beams, float step, float inv_step, float radius, float near, int* lo,
                           int* hi) {
    const float dist = sqrtf(lx * lx + ly * ly);
    if (dist <= near) {
        *lo = 0;
        *hi = beams - 1;
        return;
    }
    float ratio = radius / dist;
    ratio = ratio < 1.0f ? ratio : 1.0f;
    const float alpha = (ratio + 0.5708f * (ratio * ratio * ratio)) + 2.0f * step + 3e-3f;
    const float phi = atan2_approx(ly, lx);
    const float flo = (phi - alpha + 0.5f * kPi) * inv_step;
    const float fhi = (phi + alpha + 0.5f * kPi) * inv_step;
    int l = (int)floorf(flo), h = (int)ceilf(fhi);
    *lo = l < 0 ? 0 : l;
    *hi = h > beams - 1 ? beams - 1 : h;
}
MRCA_HD void beam_interval(float lx, float ly, int beams, int* lo, int* hi) {
    beam_interval(lx, ly, beams, kPi / (float)(beams - 1), (float)(beams - 1) / kPi, 0.2917f, 0.30f, lo, hi);
}

// ray_outline_entry<4> again, rearranged for the instruction count (what the fidelity ray cast runs at Stage's 0.2 m: 286 -> ~185
// vector instructions per tested (beam, neighbour) pair).  Same times, same comparisons, same result:
//   * the window's columns and rows are taken in WALK order (column w = the w-th the ray can reach: ax + w going right, ax + 3 - w
//     going left): the boundary in front of column w is base + s * w whichever way the ray goes, so enter(w) = T[w] and
//     leave(w) = T[w + 1] with no select per column, and the bitmap is mirrored once per neighbour instead (v_bfrev / v_perm);
//   * a ray with dx = 0 gets fxe = -inf and 1 / dx = +inf from the caller: every crossing time then evaluates to +inf by itself
//     ((b - -inf) * inf), which is what grid_march gives an axis it never steps along;
//   * columns / rows behind the origin leave the bitmap through two shifted masks instead of a select per column;
//   * an unmarked cell turns its enter time into a NaN (OR with a sign-extended bit: one v_bfe_i32 + one v_or), so that the
//     "entered before the row is left" compare fails by itself, and the minimum is a v_max / v_cndmask / v_min chain.
MRCA_HD uint32_t bitrev32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bitreverse32(v);
#else
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    return __builtin_bswap32(v);
#endif
}
MRCA_HD float ray_outline_entry4(float fxe, float fye, int ix0, int iy0, bool xpos, bool ypos, float inv_dx, float inv_dy,
                                 const OutlineBits& o) {
    const int sx = xpos ? 1 : -1, sy = ypos ? 1 : -1;
    const int bx0 = o.ax + (xpos ? 0 : 4), by0 = o.ay + (ypos ? 0 : 4);
    float TX[5], TY[5];
#pragma unroll
    for (int q = 0; q <= 4; ++q) {
        TX[q] = ((float)(bx0 + sx * q) - fxe) * inv_dx;
        TY[q] = ((float)(by0 + sy * q) - fye) * inv_dy;
    }
    const int w0 = xpos ? ix0 - o.ax : o.ax + 3 - ix0;      // the origin's column / row in walk order (may lie outside 0..3)
    const int v0 = ypos ? iy0 - o.ay : o.ay + 3 - iy0;
    float eX[4], eY[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        eX[w] = w == w0 ? -kInf : TX[w];
        eY[w] = w == v0 ? -kInf : TY[w];
    }
    // the bitmap in walk order (rows 0..3 live in o.lo, one byte each, columns in the low nibble) ...
    const uint32_t both = bitrev32(o.lo) >> 4;                                  // columns AND rows mirrored
    const uint32_t cols = __builtin_bswap32(both), rows = __builtin_bswap32(o.lo);
    uint32_t bm = xpos ? (ypos ? o.lo : rows) : (ypos ? cols : both);
    // ... without the columns and rows behind the origin (walk index below w0 / v0)
    const int cw = w0 < 0 ? 0 : (w0 > 4 ? 4 : w0), cv = v0 < 0 ? 0 : (v0 > 4 ? 4 : v0);
    bm &= ((0xFu << cw) & 0xFu) * 0x01010101u;
    bm &= (uint32_t)(0xFFFFFFFFull << (8 * cv));
    const uint32_t unmarked = ~bm;
    float best = kInf;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            // 0 for a marked cell, all ones (a NaN once ORed into any float) for an unmarked one
            const int pos = 8 * v + w;
            const uint32_t nanmask = (uint32_t)((int32_t)(unmarked << (31 - pos)) >> 31);
            uint32_t exb;
            float ex;
            memcpy(&exb, &eX[w], 4);
            exb |= nanmask;
            memcpy(&ex, &exb, 4);
            const bool visited = (ex < TY[v + 1]) & !(TX[w + 1] < eY[v]);
            const float cand = visited ? fmaxf(eX[w], eY[v]) : kInf;     // (no NaN among these: v_max_f32 / v_min_f32 as they are)
            best = fminf(best, cand);
        }
    }
    return best > 0.0f ? best : 0.0f;
}

// ------------------------------------------------------------------------------------------
// Fidelity mode, lidar: the other robots as Stage's ranger sees them -- through the world raster they are mapped into
// (worlds/stage1.world:3 `resolution 0.2`, :94-95 `ranger_return`) [libstage; restated].  A window of
// raster cells around the robot's own cell -- every cell a 6 m beam can enter -- holds one bit per cell that carries a
// piece of another robot's outline (outline_cells' walk); a beam walks the raster with grid_march's closed-form boundary
// times (raster aligned at the world origin: GridGeom{0, 0, res, 1 / res}) and its range is the entry distance of the first
// marked cell, 0 when it starts in one.
struct RasterWindow {
    const uint32_t* bits;      // [side][wpr]
    int32_t ix0, iy0, side, wpr;
    MRCA_HD bool operator()(int ix, int iy) const {
        const int jx = ix - ix0, jy = iy - iy0;
        if ((unsigned)jx >= (unsigned)side || (unsigned)jy >= (unsigned)side) return false;
        return (bits[jy * wpr + (jx >> 5)] >> (jx & 31)) & 1u;
    }
};
MRCA_HD int raster_window_reach(float inv_res) { return (int)ceilf(kRangeMax * inv_res) + 2; }   // cells either side

// ------------------------------------------------------------------------------------------
// Spatial hashes of the big-world broad phase (worlds with more than 64 robots).  A point goes to the cell
// (floor(x / cs), floor(y / cs)); a query looks at the 3 x 3 cells around its own, which contain every point
// within `reach` of it as long as cs exceeds reach by the rounding slack of the fp32 quotient (cs = 0.7 m for the
// 0.5824 m collision reach, 6.5 m for the 6.3 m lidar reach: 20 % / 3 % of margin against ~1e-7 relative error).
constexpr float kCollideCell = 0.7f, kLidarCell = 6.5f;
constexpr float kCollideReach2 = 0.3392f;   // (2 * 0.2907 + 0.001)^2, the move kernel's broad-phase radius
constexpr float kLidarReach2 = 39.69f;      // (6 + 0.3)^2, the ray cast's neighbour cull
MRCA_HD int hash_cell_coord(float x, float cs) { return (int)floorf(x * (1.0f / cs)); }
MRCA_HD uint32_t hash_cell(int ix, int iy, int world) {
    return ((uint32_t)ix * 73856093u) ^ ((uint32_t)iy * 19349663u) ^ ((uint32_t)world * 83492791u);
}

// ------------------------------------------------------------------------------------------
// reset_pose / generate_goal_point
MRCA_HD void region_xy(float ua, float ub, float* x, float* y) {  // stage_world2.py:252-257
    *x = 9.0f + 10.0f * ua;
    *y = (ub <= 0.4f) ? -(ub * 10.0f + 1.0f) : -(ub * 10.0f + 9.0f);
}

// One rejection-sampling attempt k of reset_pose.  mode: 1 disc (stage_world1.py:251-260),
// 2 region (stage_world2.py:250-268).  Returns whether the draw is acceptable; the sampled pose is
// the FIRST acceptable k, or k = kMaxTriesPose-1 if none is (bounded loop).
MRCA_HD bool pose_try(int mode, uint32_t gid, uint32_t episode, uint32_t k, uint32_t k0, uint32_t k1, float curx,
                      float cury, float* px, float* py, float* pth) {
    const U4 r = philox4x32_10(gid, episode, k, kStreamPose, k0, k1);
    const float ua = u01(r.x), ub = u01(r.y), uc = u01(r.z);
    float x, y;
    bool ok;
    if (mode == 1) {
        x = -9.0f + 18.0f * ua;
        y = -9.0f + 18.0f * ub;
        ok = sqrtf(x * x + y * y) <= 9.0f;
    } else {
        region_xy(ua, ub, &x, &y);
        const float ddx = x - curx, ddy = y - cury;
        ok = !(sqrtf(ddx * ddx + ddy * ddy) < 7.0f);
    }
    *px = x;
    *py = y;
    *pth = wrap_angle(kTwoPi * uc);
    return ok;
}

// One attempt k of generate_goal_point.  mode: 1 disc with 8..10 m from the robot
// (stage_world1.py:262-274), 2 region (stage_world2.py:270-287).
MRCA_HD bool goal_try(int mode, uint32_t gid, uint32_t episode, uint32_t k, uint32_t k0, uint32_t k1, float curx,
                      float cury, float* gx, float* gy) {
    const U4 r = philox4x32_10(gid, episode, k, kStreamGoal, k0, k1);
    const float ua = u01(r.x), ub = u01(r.y);
    float x, y;
    bool ok;
    if (mode == 1) {
        x = -9.0f + 18.0f * ua;
        y = -9.0f + 18.0f * ub;
        const float d_o = sqrtf(x * x + y * y);
        const float ex = x - curx, ey = y - cury;
        const float d_g = sqrtf(ex * ex + ey * ey);
        ok = !((d_o > 9.0f) || (d_g > 10.0f) || (d_g < 8.0f));
    } else {
        region_xy(ua, ub, &x, &y);
        const float ex = x - curx, ey = y - cury;
        ok = !(sqrtf(ex * ex + ey * ey) < 7.0f);
    }
    *gx = x;
    *gy = y;
    return ok;
}

MRCA_HD void sample_pose(int mode, uint32_t gid, uint32_t episode, uint32_t k0, uint32_t k1, float curx, float cury,
                         float* px, float* py, float* pth) {
    for (int k = 0; k < kMaxTriesPose; ++k)
        if (pose_try(mode, gid, episode, (uint32_t)k, k0, k1, curx, cury, px, py, pth) || k == kMaxTriesPose - 1) return;
}

MRCA_HD void sample_goal(int mode, uint32_t gid, uint32_t episode, uint32_t k0, uint32_t k1, float curx, float cury,
                         float* gx, float* gy) {
    for (int k = 0; k < kMaxTriesGoal; ++k)
        if (goal_try(mode, gid, episode, (uint32_t)k, k0, k1, curx, cury, gx, gy) || k == kMaxTriesGoal - 1) return;
}

}  // namespace mrca
