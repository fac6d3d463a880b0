// How many field lookups ("jumps") does a lidar beam need?  Host-only experiment behind the choice of the free-space
// field of the ray cast (csrc/mrca_host.h): the march is exact for ANY field of valid empty rectangles, so the field
// only decides the jump count -- which the kernel's VALU-bound half is proportional to.
//
//   (a) the round-1..3 field: ONE rectangle per cell, grown greedily on all four sides (a ray only ever uses the two
//       extents that face its direction of travel; the growth of the other two sides narrows them);
//   (b) QUADRANT fields: four rectangles per cell, each with the cell in the corner the ray ENTERS through (a ray of
//       quadrant (+,+) looks up the rectangle that extends right and up only), grown greedily on its two sides.
//
//   g++ -O2 -std=c++17 tools/field_probe.cpp -o tools/_build/field_probe && tools/_build/field_probe map.bin [robots]
// map.bin = int32 width, height, wpr; float cell, x0, y0; uint32 bits[height*wpr]   (tools/field_probe.py writes it)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../rl-collision-avoidance_amd/csrc/mrca_host.h"

using namespace mrca;

struct Map {
    int32_t w, h, wpr;
    float cell, x0, y0;
    std::vector<uint32_t> bits;
    bool occ(int x, int y) const {
        if (x < 0 || y < 0 || x >= w || y >= h) return false;
        return (bits[(size_t)y * wpr + (x >> 5)] >> (x & 31)) & 1u;
    }
};

// quadrant q = (sx > 0) | (sy > 0) << 1; entry = ex | ey << 8 (cells beyond the cell itself in the travel direction)
static void build_quadrant(const Map& m, int q, std::vector<uint16_t>* out, int policy) {
    const int sx = (q & 1) ? 1 : -1, sy = (q & 2) ? 1 : -1;
    const size_t sw = (size_t)m.w + 1;
    std::vector<int32_t> sat(sw * ((size_t)m.h + 1), 0);
    for (int y = 0; y < m.h; ++y)
        for (int x = 0; x < m.w; ++x)
            sat[(size_t)(y + 1) * sw + x + 1] =
                (int)m.occ(x, y) + sat[(size_t)y * sw + x + 1] + sat[(size_t)(y + 1) * sw + x] - sat[(size_t)y * sw + x];
    auto count = [&](int x0, int y0, int x1, int y1) -> int {
        if (x0 > x1) std::swap(x0, x1);
        if (y0 > y1) std::swap(y0, y1);
        x0 = std::max(x0, 0); y0 = std::max(y0, 0); x1 = std::min(x1, m.w - 1); y1 = std::min(y1, m.h - 1);
        if (x0 > x1 || y0 > y1) return 0;
        return sat[(size_t)(y1 + 1) * sw + x1 + 1] - sat[(size_t)y0 * sw + x1 + 1] - sat[(size_t)(y1 + 1) * sw + x0] +
               sat[(size_t)y0 * sw + x0];
    };
    out->assign((size_t)m.w * m.h, 0xFFFF);
    const int M = 254;
    for (int y = 0; y < m.h; ++y)
        for (int x = 0; x < m.w; ++x) {
            if (m.occ(x, y)) continue;
            int ex = 0, ey = 0;
            if (policy == 0) {          // alternate
                for (bool grew = true; grew;) {
                    grew = false;
                    if (ex < M && count(x + sx * (ex + 1), y, x + sx * (ex + 1), y + sy * ey) == 0) { ++ex; grew = true; }
                    if (ey < M && count(x, y + sy * (ey + 1), x + sx * ex, y + sy * (ey + 1)) == 0) { ++ey; grew = true; }
                }
            } else if (policy == 3) {   // the corner rectangle of the largest AREA (staircase enumeration)
                long long best = -1;
                int w = M;
                for (int h = 0; h <= M; ++h) {
                    // widest free run in row y + sy*h, limited by the rows below it
                    int run = 0;
                    if (m.occ(x, y + sy * h)) break;
                    while (run < w && !m.occ(x + sx * (run + 1), y + sy * h)) ++run;
                    w = std::min(w, run);
                    const long long area = (long long)(w + 1) * (h + 1);
                    if (area > best) { best = area; ex = w; ey = h; }
                }
            } else if (policy == 2) {   // the largest SQUARE only (one byte per quadrant)
                while (ex < M && count(x, y, x + sx * (ex + 1), y + sy * (ex + 1)) == 0) ++ex;
                ey = ex;
            } else {                    // the largest SQUARE first, then alternate
                while (ex < M && count(x, y, x + sx * (ex + 1), y + sy * (ex + 1)) == 0) ++ex;
                ey = ex;
                for (bool grew = true; grew;) {
                    grew = false;
                    if (ex < M && count(x + sx * (ex + 1), y, x + sx * (ex + 1), y + sy * ey) == 0) { ++ex; grew = true; }
                    if (ey < M && count(x, y + sy * (ey + 1), x + sx * ex, y + sy * (ey + 1)) == 0) { ++ey; grew = true; }
                }
            }
            (*out)[(size_t)y * m.w + x] = (uint16_t)(ex | (ey << 8));
        }
}

// returns the number of field lookups AFTER the origin's own entry (which every beam of a robot shares)
template <class Ext>
static int march_count(const Map& m, float ox, float oy, float dx, float dy, float tmax, Ext ext, float* range) {
    const double fx = (ox - m.x0) / m.cell, fy = (oy - m.y0) / m.cell, tmc = tmax / m.cell;
    int ix = (int)std::floor(fx), iy = (int)std::floor(fy);
    const int sx = dx > 0 ? 1 : -1, sy = dy > 0 ? 1 : -1;
    int lookups = 0;
    *range = tmax;
    if (m.occ(ix, iy)) { *range = 0; return 0; }
    for (int guard = 0; guard < 4096; ++guard) {
        int ex, ey;
        ext(ix, iy, sx, sy, &ex, &ey);
        const int Bx = sx > 0 ? ix + ex + 1 : ix - ex, By = sy > 0 ? iy + ey + 1 : iy - ey;
        const double tBx = dx != 0 ? (Bx - fx) / dx : 1e30, tBy = dy != 0 ? (By - fy) / dy : 1e30;
        const bool xe = tBx < tBy;
        const double t = xe ? tBx : tBy;
        if (t >= tmc) return lookups;
        const double px = fx + dx * t, py = fy + dy * t;
        if (xe) { ix = sx > 0 ? Bx : Bx - 1; iy = (int)std::floor(py); }
        else    { iy = sy > 0 ? By : By - 1; ix = (int)std::floor(px); }
        ++lookups;
        if (m.occ(ix, iy)) { *range = (float)(t * m.cell); return lookups; }
    }
    return lookups;
}

int main(int argc, char** argv) {
    if (argc < 2) return 1;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    Map m;
    if (fread(&m.w, 4, 1, f) != 1 || fread(&m.h, 4, 1, f) != 1 || fread(&m.wpr, 4, 1, f) != 1 || fread(&m.cell, 4, 1, f) != 1 ||
        fread(&m.x0, 4, 1, f) != 1 || fread(&m.y0, 4, 1, f) != 1)
        return 1;
    m.bits.resize((size_t)m.h * m.wpr);
    if (fread(m.bits.data(), 4, m.bits.size(), f) != m.bits.size()) return 1;
    fclose(f);
    const int robots = argc > 2 ? atoi(argv[2]) : 2000;
    const float rmax = argc > 3 ? (float)atof(argv[3]) : 9.0f;        // robots uniform in a disc of this radius (stage 1)
    std::vector<uint32_t> field;
    int pitch = 0;
    build_free_rect_field(m.bits.data(), m.w, m.h, m.wpr, &field, &pitch);
    std::vector<uint16_t> quad[4][4];
    for (int pol = 0; pol < 4; ++pol)
        for (int q = 0; q < 4; ++q) build_quadrant(m, q, &quad[pol][q], pol);
    auto ext_a = [&](int ix, int iy, int sx, int sy, int* ex, int* ey) {
        const int x = std::min(std::max(ix, -kFieldPadX), m.w + kFieldPadX - 1), y = std::min(std::max(iy, -kFieldPadY), m.h + kFieldPadY - 1);
        const uint32_t v = field[(size_t)(y + kFieldPadY) * pitch + x + kFieldPadX];
        *ex = (v >> (sx > 0 ? 8 : 0)) & 255;
        *ey = (v >> (sy > 0 ? 24 : 16)) & 255;
    };
    auto make_q = [&](int pol) {
        return [&, pol](int ix, int iy, int sx, int sy, int* ex, int* ey) {
            if (ix < 0 || iy < 0 || ix >= m.w || iy >= m.h) { *ex = *ey = 0; return; }
            const uint16_t v = quad[pol][(sx > 0) | ((sy > 0) << 1)][(size_t)iy * m.w + ix];
            *ex = v & 255;
            *ey = v >> 8;
        };
    };
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> U(-1.0f, 1.0f);
    long long la = 0, lq[3] = {0, 0, 0}, beams = 0, mism = 0; long long wave_s = 0; int wmax_s = 0;
    long long hist_a[16] = {0}, hist_q[16] = {0};
    long long wave_a = 0, wave_q = 0, waves = 0;      // per 64-beam wavefront: the MAXIMUM over its lanes is what it executes
    for (int r = 0; r < robots; ++r) {
        float x, y;
        do {
            x = U(rng) * rmax;
            y = U(rng) * rmax;
        } while (x * x + y * y > rmax * rmax ||
                 m.occ((int)std::floor((x - m.x0) / m.cell), (int)std::floor((y - m.y0) / m.cell)));
        const float th = U(rng) * 3.14159265f;
        int wmax_a = 0, wmax_q = 0;
        for (int b = 0; b < 512; ++b) {
            const double a = th - 3.14159265358979 / 2 + b * 3.14159265358979 / 511;
            const float dx = (float)std::cos(a), dy = (float)std::sin(a);
            float ra, rq;
            const int ca = march_count(m, x, y, dx, dy, 6.0f, ext_a, &ra);
            const int cq0 = march_count(m, x, y, dx, dy, 6.0f, make_q(0), &rq);
            const int cq1 = march_count(m, x, y, dx, dy, 6.0f, make_q(3), &rq);
            float rs;
            const int cq2 = march_count(m, x, y, dx, dy, 6.0f, make_q(2), &rs);
            la += ca; lq[0] += cq0; lq[1] += cq1; lq[2] += cq2;
            wmax_s = std::max(wmax_s, cq2 + (rs >= 6.0f ? 1 : 0));
            if ((b & 63) == 63) { wave_s += wmax_s; wmax_s = 0; }
            ++beams;
            mism += std::fabs(ra - rq) > 1e-4f;
            ++hist_a[std::min(ca, 15)];
            ++hist_q[std::min(cq0, 15)];
            // loop iterations = lookups + 1 unless the beam ended in a hit (the last iteration then found the wall)
            const int ia = ca + (ra >= 6.0f ? 1 : 0), iq = cq0 + (rq >= 6.0f ? 1 : 0);
            wmax_a = std::max(wmax_a, ia);
            wmax_q = std::max(wmax_q, iq);
            if ((b & 63) == 63) {
                wave_a += wmax_a; wave_q += wmax_q; ++waves;
                wmax_a = wmax_q = 0;
            }
        }
    }
    printf("map %d x %d cells of %.3f m, %d robots x 512 beams\n", m.w, m.h, m.cell, robots);
    printf("lookups per beam after the origin's own entry (jumps = lookups + 1 when the beam ends beyond the last rectangle):\n");
    printf("  one 4-sided rectangle per cell (product)      %.3f\n", (double)la / beams);
    printf("  quadrant corner rectangles, alternate growth  %.3f\n", (double)lq[0] / beams);
    printf("  quadrant corner rectangles, LARGEST AREA      %.3f\n", (double)lq[1] / beams);
    printf("  quadrant corner SQUARES (one byte per quadrant) %.3f lookups, wavefront iterations %.3f\n", (double)lq[2] / beams,
           (double)wave_s / waves);
    printf("  loop iterations a 64-beam WAVEFRONT executes per beam pass (max over its lanes): product %.3f, quadrant %.3f\n",
           (double)wave_a / waves, (double)wave_q / waves);
    printf("  range mismatches between fields (float walk, diagnostics only): %lld of %lld\n", mism, beams);
    printf("  histogram of lookups   product: ");
    for (int k = 0; k < 10; ++k) printf("%d:%.3f ", k, (double)hist_a[k] / beams);
    printf("\n                        quadrant: ");
    for (int k = 0; k < 10; ++k) printf("%d:%.3f ", k, (double)hist_q[k] / beams);
    printf("\n");
    return 0;
}
