// tests/host_emul/emul.cpp -- TEST HARNESS ONLY (never shipped, never loaded by the product).
//
// Drives the product's per-lane arithmetic (rl-collision-avoidance_amd/csrc/mrca_device.h, the
// exact functions the gfx950 kernels inline) with plain host loops, so the arithmetic can be
// checked bit-for-bit against the oracle's fp32 mode in the CPU-only container before any GPU
// time is spent.  The loop structure mirrors move_kernel / raycast_kernel / reset_kernel in
// mrca_kernels.hip (wavefront ballot -> a plain OR over robots).
//
// Build: g++ -O2 -ffp-contract=off -shared -fPIC emul.cpp -o libmrca_emul.so
#include <stdint.h>
#include <string.h>

#include <map>
#include <vector>

#include "../../rl-collision-avoidance_amd/csrc/mrca_device.h"
#include "../../rl-collision-avoidance_amd/csrc/mrca_host.h"
#include "../../rl-collision-avoidance_amd/csrc/mrca_policy_layout.h"

using namespace mrca;

extern "C" {

struct EmulEnv {
    int32_t N, R, W, B, F;
    float *pose, *speed, *speed_gt, *goal, *init_pose, *scan, *obs, *local_goal, *reward, *prev_dist;
    uint8_t *done, *result, *first_result, *crashed, *live, *fresh;
    int32_t *t, *episode;
    const int32_t *reset_mode, *goal_mode, *group_id;
    const float *init_table, *goal_table;
    const float *beam_cos, *beam_sin;
    const uint32_t* map_bits;
    float x0, y0, cell;
    int32_t width, height, wpr;
    int32_t timeout;
    float w_thresh;
    int32_t pre_dist_zero, auto_reset, num_groups;
    uint32_t key0, key1;
    int32_t hold_velocity;
};

// the free-rectangle field of a map, built once per distinct map (keyed by a hash of the bitmap)
struct CachedField {
    std::vector<uint16_t> f;
    int pitch = 0;
};
static const CachedField& field_of(const EmulEnv* e) {
    static std::map<uint64_t, CachedField> cache;
    uint64_t h = 1469598103934665603ull ^ ((uint64_t)e->width << 32) ^ (uint64_t)e->height;
    for (size_t i = 0; i < (size_t)e->height * e->wpr; ++i) h = (h ^ e->map_bits[i]) * 1099511628211ull;
    CachedField& c = cache[h];
    if (c.f.empty()) build_free_rect_field(e->map_bits, e->width, e->height, e->wpr, &c.f, &c.pitch);
    return c;
}

static GridGeom geom(const EmulEnv* e) {
    GridGeom g;
    g.x0 = e->x0;
    g.y0 = e->y0;
    g.cell = e->cell;
    g.inv_cell = 1.0f / e->cell;
    g.width = e->width;
    g.height = e->height;
    g.wpr = e->wpr;
    return g;
}

static void begin_episode(const EmulEnv* e, int n, int local, float curx, float cury, const float* po,
                          const float* go) {
    const uint32_t ep = (uint32_t)e->episode[n];
    float x, y, th;
    if (po) {
        x = po[0]; y = po[1]; th = po[2];
    } else if (e->reset_mode[local] == 0) {
        x = e->init_table[local * 3];
        y = e->init_table[local * 3 + 1];
        th = wrap_angle(e->init_table[local * 3 + 2]);
    } else {
        sample_pose(e->reset_mode[local], (uint32_t)n, ep, e->key0, e->key1, curx, cury, &x, &y, &th);
    }
    float gx, gy;
    if (go) {
        gx = go[0]; gy = go[1];
    } else if (e->goal_mode[local] == 0) {
        gx = e->goal_table[local * 2];
        gy = e->goal_table[local * 2 + 1];
    } else {
        sample_goal(e->goal_mode[local], (uint32_t)n, ep, e->key0, e->key1, x, y, &gx, &gy);
    }
    const float ddx = gx - x, ddy = gy - y;
    const float d = sqrtf(ddx * ddx + ddy * ddy);
    e->pose[n * 3] = x; e->pose[n * 3 + 1] = y; e->pose[n * 3 + 2] = th;
    e->init_pose[n * 3] = x; e->init_pose[n * 3 + 1] = y; e->init_pose[n * 3 + 2] = th;
    e->goal[n * 2] = gx; e->goal[n * 2 + 1] = gy;
    e->prev_dist[n] = e->pre_dist_zero ? 0.0f : d;
    e->t[n] = 1;
    e->crashed[n] = 0;
    e->live[n] = 1;
    if (!e->hold_velocity) e->speed[n * 2] = e->speed[n * 2 + 1] = 0.0f;
    e->speed_gt[n * 2] = e->speed_gt[n * 2 + 1] = 0.0f;
}

void emul_raycast(const EmulEnv* e, int only_fresh) {
    const GridGeom g = geom(e);
    const CachedField& cf = field_of(e);
    const FreeRectField dist{cf.f.data(), e->width, e->height, cf.pitch};
    for (int n = 0; n < e->N; ++n) {
        const bool fresh = e->fresh[n] != 0;
        if (only_fresh && !fresh) continue;
        const int world = n / e->R, local = n % e->R;
        const float x = e->pose[n * 3], y = e->pose[n * 3 + 1], th = e->pose[n * 3 + 2];
        float s, c;
        sincos_det(th, &s, &c);
        std::vector<float> nb;
        std::vector<int> nbi;
        for (int j = 0; j < e->R; ++j) {
            if (j == local) continue;
            const int m = world * e->R + j;
            const float xj = e->pose[m * 3], yj = e->pose[m * 3 + 1];
            const float ddx = xj - x, ddy = yj - y;
            if (!(ddx * ddx + ddy * ddy <= 39.69f)) continue;
            int lo, hi;
            beam_interval(ddx * c + ddy * s, ddy * c - ddx * s, e->B, &lo, &hi);
            if (lo > hi) continue;
            float sj, cj;
            sincos_det(e->pose[m * 3 + 2], &sj, &cj);
            nb.insert(nb.end(), {xj, yj, sj, cj});
            nbi.insert(nbi.end(), {lo, hi});
        }
        // the kernel's thread t marches beams t and t + B/2 in lock step (grid_march_skip_n<2>)
        std::vector<float> marched(e->B);
        const MarchOrigin org = march_origin(dist, g, x, y);
        for (int b = 0; b < e->B / 2; ++b) {
            float ddx[2], ddy[2], out[2];
            for (int k = 0; k < 2; ++k) {
                const float bc = e->beam_cos[b + k * (e->B / 2)], bs = e->beam_sin[b + k * (e->B / 2)];
                ddx[k] = c * bc - s * bs;
                ddy[k] = s * bc + c * bs;
            }
            grid_march_skip_n<2>(dist, g, org, ddx, ddy, kRangeMax, out);
            marched[b] = out[0];
            marched[b + e->B / 2] = out[1];
        }
        for (int b = 0; b < e->B; ++b) {
            const float bc = e->beam_cos[b], bs = e->beam_sin[b];
            const float dx = c * bc - s * bs;
            const float dy = s * bc + c * bs;
            float rng = marched[b];
            for (size_t k = 0; k < nb.size(); k += 4) {
                if (b < nbi[k / 2] || b > nbi[k / 2 + 1]) continue;
                const float t = ray_box(x, y, dx, dy, nb[k], nb[k + 1], nb[k + 2], nb[k + 3]);
                rng = t < rng ? t : rng;
            }
            rng = rng < kRangeMax ? rng : kRangeMax;
            e->scan[(size_t)n * e->B + b] = rng;
            const float o = norm_obs(rng);
            float* ob = e->obs + (size_t)n * e->F * e->B + b;
            if (fresh) {
                for (int f = 0; f < e->F; ++f) ob[f * e->B] = o;
            } else {
                for (int f = 0; f + 1 < e->F; ++f) ob[f * e->B] = ob[(f + 1) * e->B];
                ob[(e->F - 1) * e->B] = o;
            }
        }
        const float gx = e->goal[n * 2] - x, gy = e->goal[n * 2 + 1] - y;
        e->local_goal[n * 2] = gx * c + gy * s;
        e->local_goal[n * 2 + 1] = gy * c - gx * s;
    }
}

void emul_reset(const EmulEnv* e, const uint8_t* mask, const float* poses, const float* goals) {
    for (int n = 0; n < e->N; ++n) {
        const bool sel = mask ? mask[n] != 0 : true;
        e->fresh[n] = sel;
        if (!sel) continue;
        e->episode[n] += 1;
        begin_episode(e, n, n % e->R, e->pose[n * 3], e->pose[n * 3 + 1], poses ? poses + n * 3 : nullptr,
                      goals ? goals + n * 2 : nullptr);
        e->done[n] = 0;
        e->result[n] = 0;
        e->reward[n] = 0.0f;
        e->first_result[n] = 0;
    }
    emul_raycast(e, 1);
}

void emul_step(const EmulEnv* e, const float* actions) {
    const GridGeom g = geom(e);
    const GlobalGrid occ{e->map_bits, e->width, e->height, e->wpr};
    const int R = e->R;
    std::vector<float> x(R), y(R), th(R), s(R), c(R), nx(R), ny(R), nth(R), ns(R), nc(R), v(R), w(R);
    std::vector<char> moving(R), shit(R), moved(R), livev(R), done_now(R);
    for (int world = 0; world < e->W; ++world) {
        for (int l = 0; l < R; ++l) {
            const int n = world * R + l;
            x[l] = e->pose[n * 3]; y[l] = e->pose[n * 3 + 1]; th[l] = e->pose[n * 3 + 2];
            livev[l] = e->live[n] != 0;
            v[l] = livev[l] ? sane_cmd(actions[n * 2]) : (e->hold_velocity ? e->speed[n * 2] : 0.0f);
            w[l] = livev[l] ? sane_cmd(actions[n * 2 + 1]) : (e->hold_velocity ? e->speed[n * 2 + 1] : 0.0f);
            sincos_det(th[l], &s[l], &c[l]);
            const float d = v[l] * kDt;
            nx[l] = x[l] + d * c[l];
            ny[l] = y[l] + d * s[l];
            nth[l] = wrap_angle(th[l] + w[l] * kDt);
            sincos_det(nth[l], &ns[l], &nc[l]);
            moving[l] = (v[l] != 0.0f) || (w[l] != 0.0f);
            shit[l] = static_hit(occ, g, nx[l], ny[l], ns[l], nc[l]);
            moved[l] = 0;
        }
        for (int i = 0; i < R; ++i) {
            bool any = false;
            for (int j = 0; j < R; ++j)
                if (j != i) any = any || obb_overlap(nx[i], ny[i], ns[i], nc[i], x[j], y[j], s[j], c[j]);
            if (moving[i]) {
                const bool hit = shit[i] || any;
                if (!hit) {
                    x[i] = nx[i]; y[i] = ny[i]; th[i] = nth[i]; s[i] = ns[i]; c[i] = nc[i];
                    moved[i] = 1;
                }
                e->crashed[world * R + i] = hit ? 1 : 0;
            }
        }
        for (int l = 0; l < R; ++l) {
            const int n = world * R + l;
            e->pose[n * 3] = x[l]; e->pose[n * 3 + 1] = y[l]; e->pose[n * 3 + 2] = th[l];
            e->speed[n * 2] = v[l]; e->speed[n * 2 + 1] = w[l];
            const float vgt = moved[l] ? fabsf(v[l]) : 0.0f;
            const float wgt = moved[l] ? w[l] : 0.0f;
            e->speed_gt[n * 2] = vgt; e->speed_gt[n * 2 + 1] = wgt;
            const float ddx = e->goal[n * 2] - x[l], ddy = e->goal[n * 2 + 1] - y[l];
            const float dist = sqrtf(ddx * ddx + ddy * ddy);
            float rg = (e->prev_dist[n] - dist) * kKProgress;
            const bool reach = dist < kGoalRadius;
            rg = reach ? kRArrive : rg;
            const bool crash = e->crashed[n] == 1;
            const float rc = crash ? kRCrash : 0.0f;
            const float aw = fabsf(wgt);
            const float rw = (aw > e->w_thresh) ? kKOmega * aw : 0.0f;
            const bool tout = e->t[n] > e->timeout;
            uint8_t result = reach ? 1 : 0;
            result = crash ? 2 : result;
            result = tout ? 3 : result;
            done_now[l] = reach || crash || tout;
            if (livev[l]) {
                e->reward[n] = (rg + rc) + rw;
                e->done[n] = done_now[l];
                e->result[n] = result;
                e->prev_dist[n] = dist;
                e->t[n] += 1;
                if (done_now[l] && e->first_result[n] == 0) e->first_result[n] = result;
            }
            e->fresh[n] = 0;
        }
        if (e->auto_reset == 1) {
            for (int l = 0; l < R; ++l)
                if (livev[l] && done_now[l]) e->fresh[world * R + l] = 1;
        } else if (e->auto_reset == 2) {
            for (int l = 0; l < R; ++l)
                if (livev[l] && done_now[l]) e->live[world * R + l] = 0;
            for (int gi = 0; gi < e->num_groups; ++gi) {
                bool all = true, any = false;
                for (int l = 0; l < R; ++l)
                    if (e->group_id[l] == gi) {
                        any = true;
                        all = all && e->done[world * R + l];
                    }
                if (any && all)
                    for (int l = 0; l < R; ++l)
                        if (e->group_id[l] == gi) e->fresh[world * R + l] = 1;
            }
        }
        for (int l = 0; l < R; ++l) {
            const int n = world * R + l;
            if (e->fresh[n]) {
                e->episode[n] += 1;
                begin_episode(e, n, l, e->pose[n * 3], e->pose[n * 3 + 1], nullptr, nullptr);
            }
        }
    }
    emul_raycast(e, 0);
}

// the product's quadrant free-rectangle field, un-padded [height][width][4] uint16, for the soundness test
int emul_free_rect_field(const EmulEnv* e, uint16_t* out, int cap) {
    const CachedField& cf = field_of(e);
    if (e->width * e->height * 4 > cap) return -1;
    for (int y = 0; y < e->height; ++y)
        memcpy(out + (size_t)y * e->width * 4, cf.f.data() + ((size_t)(y + kFieldPadY) * cf.pitch + kFieldPadX) * 4,
               (size_t)e->width * 8);
    return 0;
}

void emul_march(const EmulEnv* e, int n, const float* ox, const float* oy, const float* dx, const float* dy,
                const float* tmax, float* out_plain, float* out_skip) {
    const GridGeom g = geom(e);
    const GlobalGrid occ{e->map_bits, e->width, e->height, e->wpr};
    const CachedField& cf = field_of(e);
    const FreeRectField dist{cf.f.data(), e->width, e->height, cf.pitch};
    for (int i = 0; i < n; ++i) {
        out_plain[i] = grid_march(occ, g, ox[i], oy[i], dx[i], dy[i], tmax[i]);
        out_skip[i] = grid_march_skip(dist, g, ox[i], oy[i], dx[i], dy[i], tmax[i]);
    }
}

// the lock-step K-ray march: ray i is marched together with K-1 other rays from the SAME origin and range whose
// directions are taken from rays i+1.. (so partners finish at different iterations); out[i] = result of ray i
void emul_march_lockstep(const EmulEnv* e, int n, int K, const float* ox, const float* oy, const float* dx,
                         const float* dy, const float* tmax, float* out) {
    const GridGeom g = geom(e);
    const CachedField& cf = field_of(e);
    const FreeRectField dist{cf.f.data(), e->width, e->height, cf.pitch};
    for (int i = 0; i < n; ++i) {
        const MarchOrigin org = march_origin(dist, g, ox[i], oy[i]);
        if (K == 2) {
            const float ddx[2] = {dx[i], dx[(i + 1) % n]}, ddy[2] = {dy[i], dy[(i + 1) % n]};
            float o[2];
            grid_march_skip_n<2>(dist, g, org, ddx, ddy, tmax[i], o);
            out[i] = o[0];
        } else {
            const float ddx[4] = {dx[(i + 3) % n], dx[(i + 1) % n], dx[i], dx[(i + 2) % n]};
            const float ddy[4] = {dy[(i + 3) % n], dy[(i + 1) % n], dy[i], dy[(i + 2) % n]};
            float o[4];
            grid_march_skip_n<4>(dist, g, org, ddx, ddy, tmax[i], o);
            out[i] = o[2];
        }
    }
}

// beam-interval culling: for n neighbour centres (robot frame), the interval the product would test
void emul_beam_interval(int n, const float* lx, const float* ly, int beams, int* lo, int* hi) {
    for (int i = 0; i < n; ++i) beam_interval(lx[i], ly[i], beams, &lo[i], &hi[i]);
}

void emul_ray_box(int n, const float* ox, const float* oy, const float* dx, const float* dy, const float* xj,
                  const float* yj, const float* sj, const float* cj, float* t) {
    for (int i = 0; i < n; ++i) t[i] = ray_box(ox[i], oy[i], dx[i], dy[i], xj[i], yj[i], sj[i], cj[i]);
}

// the move kernel's outline test in both forms: grid_march with its early exit (the specification) and the OR over the walk's
// independent crossing events (walk_event_hits: one lane per event on the device) -- out[i] = 4 bits per pose, one per edge
int emul_outline_hits(const EmulEnv* e, int n, const float* x, const float* y, const float* th, int* out_march, int* out_events) {
    const GridGeom g = geom(e);
    const GlobalGrid occ{e->map_bits, e->width, e->height, e->wpr};
    const int Q = edge_event_slots(g.inv_cell);
    for (int i = 0; i < n; ++i) {
        float s, c;
        sincos_det(th[i], &s, &c);
        int a = 0, b = 0;
        for (int k = 0; k < 4; ++k) {
            a |= (static_edge_hit(occ, g, x[i], y[i], s, c, k) ? 1 : 0) << k;
            bool hit = false;
            for (int q = 0; q <= 2 * Q; ++q) hit = static_edge_event_hits(occ, g, x[i], y[i], s, c, k, q, Q) || hit;
            b |= (hit ? 1 : 0) << k;
        }
        out_march[i] = a;
        out_events[i] = b;
    }
    return Q;
}

int emul_obb(float xi, float yi, float si, float ci, float xj, float yj, float sj, float cj) {
    return obb_overlap(xi, yi, si, ci, xj, yj, sj, cj) ? 1 : 0;
}

void emul_norm_obs(const float* x, int n, float* out) {
    for (int i = 0; i < n; ++i) out[i] = norm_obs(x[i]);
}

void emul_sincos(const float* th, int n, float* s, float* c) {
    for (int i = 0; i < n; ++i) sincos_det(th[i], &s[i], &c[i]);
}

void emul_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
    const U4 r = philox4x32_10(c0, c1, c2, c3, k0, k1);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

// --- fidelity mode (round 5): the anchored outline bitmaps and the closed-form raster lidar of the kernels against the
//     cell lists / the cell-by-cell walk they replace (tests/test_fidelity_closed_form.py)
int emul_outline_bits(float res, float x, float y, float th, int32_t* out /* ax, ay, lo, hi */) {
    float s, c;
    sincos_det(th, &s, &c);
    OutlineBits o;
    const bool ok = outline_bits(1.0f / res, x, y, s, c, &o);
    out[0] = o.ax; out[1] = o.ay; out[2] = (int32_t)o.lo; out[3] = (int32_t)o.hi;
    return ok ? 1 : 0;
}
int emul_outline_span(float res) { return outline_span(1.0f / res); }
int emul_outline_cells(float res, float x, float y, float th, long long* out /* [kMaxOutlineCells] */) {
    float s, c;
    sincos_det(th, &s, &c);
    return outline_cells(1.0f / res, x, y, s, c, out);
}
static OutlineBits ob_of(const int32_t* p) { return OutlineBits{p[0], p[1], (uint32_t)p[2], (uint32_t)p[3]}; }
int emul_outline_meet(const int32_t* p, const int32_t* q) { return outline_bits_meet(ob_of(p), ob_of(q)) ? 1 : 0; }

// n rays (ox, oy, dx, dy) against ONE outline each (ob[4 * i ..]): the closed form of the kernel (window side kw = 4 | 8) and
// grid_march's walk through a window of marked cells around the origin (what rounds 3-4 ran on the device)
void emul_ray_outline(int n, int kw, float res, const float* ox, const float* oy, const float* dx, const float* dy,
                      const int32_t* ob, float tmax, float* out_closed, float* out_walk) {
    const float inv = 1.0f / res;
    const int reach = raster_window_reach(inv);
    const int side = 2 * reach + 1, wpr = (side + 31) >> 5;
    std::vector<uint32_t> bits((size_t)side * wpr);
    const GridGeom gr{0.0f, 0.0f, res, inv, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        const OutlineBits o = ob_of(ob + 4 * i);
        const float fx = ox[i] * inv, fy = oy[i] * inv;
        const int ix0 = (int)floorf(fx), iy0 = (int)floorf(fy);
        const float inv_dx = dx[i] != 0.0f ? rcp_exact(dx[i]) : kInf, inv_dy = dy[i] != 0.0f ? rcp_exact(dy[i]) : kInf;
        // kw = 4: the rearranged form the fidelity ray cast runs (mrca_device.h ray_outline_entry4), kw = -4: the plain 4 x 4 form
        const float fxe = dx[i] != 0.0f ? fx : -kInf, fye = dy[i] != 0.0f ? fy : -kInf;
        const float t = kw == 4 ? ray_outline_entry4(fxe, fye, ix0, iy0, dx[i] > 0.0f, dy[i] > 0.0f, inv_dx, inv_dy, o)
                        : kw == -4 ? ray_outline_entry<4>(fx, fy, ix0, iy0, dx[i], dy[i], inv_dx, inv_dy, o)
                                   : ray_outline_entry<8>(fx, fy, ix0, iy0, dx[i], dy[i], inv_dx, inv_dy, o);
        out_closed[i] = t < tmax * inv ? t * res : tmax;
        std::fill(bits.begin(), bits.end(), 0u);
        const RasterWindow win{bits.data(), ix0 - reach, iy0 - reach, side, wpr};
        for (int jy = 0; jy < kOutlineWin; ++jy)
            for (int jx = 0; jx < kOutlineWin; ++jx) {
                const uint32_t half = jy < 4 ? o.lo : o.hi;
                if (!((half >> (((jy & 3) << 3) + jx)) & 1u)) continue;
                const int wx = o.ax + jx - win.ix0, wy = o.ay + jy - win.iy0;
                if (wx >= 0 && wy >= 0 && wx < side && wy < side) bits[(size_t)wy * wpr + (wx >> 5)] |= 1u << (wx & 31);
            }
        out_walk[i] = grid_march(win, gr, ox[i], oy[i], dx[i], dy[i], tmax);
    }
}

// --- the LDS image / operand address formulas of the policy's backward kernel (mrca_policy_layout.h), for
//     tests/test_policy_bwd_layout.py
int pl_rowmap(int reg, int hl) { return mrca_pbwd::rowmap(reg, hl); }
int pl_x_operand_base(int kk) { return mrca_pbwd::x_operand_base(kk); }
int pl_conv1_pstart(int h) { return mrca_pbwd::conv1_pstart(h); }
int pl_h1_store_off(int p, int h) { return mrca_pbwd::h1_store_off(p, h); }
void pl_constants(int* out) {
    using namespace mrca_pbwd;
    const int v[] = {kXPitch, kGPitch, kHPitch, kXE, kXO, kG2, kH1E, kH1O, kWaveFloats, kWavesPerBlock,
                     kPartDw2, kPartDw1, kPartDb1, kPartDb2, kPartFloats, kHalf, kBlockFloats};
    for (unsigned i = 0; i < sizeof(v) / sizeof(v[0]); ++i) out[i] = v[i];
}

// --- the same for the forward kernel (namespace mrca_pfwd), for tests/test_policy_conv_layout.py
int pf_conv1_kk(int s, int hl) { return mrca_pfwd::conv1_kk(s, hl); }
int pf_conv1_family(int s) { return mrca_pfwd::conv1_family(s); }
int pf_conv1_step_off(int s) { return mrca_pfwd::conv1_step_off(s); }
int pf_conv2_ci(int s, int hl) { return mrca_pfwd::conv2_ci(s, hl); }
int pf_conv2_tap(int s, int hl) { return mrca_pfwd::conv2_tap(s, hl); }
int pf_conv2_step_off(int s) { return mrca_pfwd::conv2_step_off(s); }
int pf_h1_store_off(int p) { return mrca_pfwd::h1_store_off(p); }
void pf_constants(int* out) {
    using namespace mrca_pfwd;
    const int v[] = {kXPitch, kHPitch, kXE, kXO, kH1E, kH1O, kWaveFloats, kWavesPerBlock};
    for (unsigned i = 0; i < sizeof(v) / sizeof(v[0]); ++i) out[i] = v[i];
}

}  // extern "C"
