// mrca_host.h -- host-side helpers of the env library (plain C++, no HIP).
#pragma once
#include <stdint.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "mrca_device.h"

namespace mrca {

// Quadrant free-rectangle field for grid_march_skip: FOUR 16-bit entries per cell, one per quadrant of the direction of
// travel, q = (dx > 0) | (dy > 0) << 1 with sx = +-1, sy = +-1 accordingly.  For an EMPTY cell entry q packs the extents
// ex | ey << 8 of a rectangle of empty cells [x, x + sx*ex] x [y, y + sy*ey] that has the cell in the corner a ray of
// that quadrant enters through (cells outside the map are empty); an occupied cell gets kCellOccupied in all four.  Each
// rectangle is grown greedily, alternately in x and in y, while the strip added is entirely empty (up to
// kFieldMaxExtent cells per side) -- any empty rectangle is valid for the march, larger ones just save jumps; measured
// against the alternatives (one 4-sided rectangle per cell, largest square, largest area) in tools/field_probe.cpp.
// Storage: (height + 2*kFieldPadY) rows of `pitch` CELLS (4 x uint16 each), cell (0,0) at [kFieldPadY][kFieldPadX], the
// border filled with 0 (see FreeRectField).  Rows are independent and are built by a few host threads.
inline void build_free_rect_field(const uint32_t* bits, int width, int height, int wpr, std::vector<uint16_t>* out,
                                  int* pitch_out) {
    // summed-area table of occupied cells for O(1) strip tests
    const size_t sw = (size_t)width + 1;
    std::vector<int32_t> sat(sw * ((size_t)height + 1), 0);
    auto occupied = [&](int x, int y) -> int { return (bits[(size_t)y * wpr + (x >> 5)] >> (x & 31)) & 1u; };
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x)
            sat[(size_t)(y + 1) * sw + x + 1] =
                occupied(x, y) + sat[(size_t)y * sw + x + 1] + sat[(size_t)(y + 1) * sw + x] - sat[(size_t)y * sw + x];
    auto count = [&](int x0, int y0, int x1, int y1) -> int {  // inclusive cell rectangle (either order), clipped to the map
        if (x0 > x1) std::swap(x0, x1);
        if (y0 > y1) std::swap(y0, y1);
        x0 = std::max(x0, 0); y0 = std::max(y0, 0); x1 = std::min(x1, width - 1); y1 = std::min(y1, height - 1);
        if (x0 > x1 || y0 > y1) return 0;
        return sat[(size_t)(y1 + 1) * sw + x1 + 1] - sat[(size_t)y0 * sw + x1 + 1] - sat[(size_t)(y1 + 1) * sw + x0] +
               sat[(size_t)y0 * sw + x0];
    };
    const int pitch = width + 2 * kFieldPadX;
    out->assign((size_t)pitch * (height + 2 * kFieldPadY) * 4, 0);
    uint16_t* field = out->data();
    auto rows = [&](int y_begin, int y_end) {
        const int M = kFieldMaxExtent;   // < 255, so a packed entry can never equal kCellOccupied
        for (int y = y_begin; y < y_end; ++y)
            for (int x = 0; x < width; ++x) {
                uint16_t* e = field + ((size_t)(y + kFieldPadY) * pitch + x + kFieldPadX) * 4;
                if (occupied(x, y)) {
                    e[0] = e[1] = e[2] = e[3] = (uint16_t)kCellOccupied;
                    continue;
                }
                for (int q = 0; q < 4; ++q) {
                    const int sx = (q & 1) ? 1 : -1, sy = (q & 2) ? 1 : -1;
                    int ex = 0, ey = 0;
                    for (bool grew = true; grew;) {
                        grew = false;
                        if (ex < M && count(x + sx * (ex + 1), y, x + sx * (ex + 1), y + sy * ey) == 0) { ++ex; grew = true; }
                        if (ey < M && count(x, y + sy * (ey + 1), x + sx * ex, y + sy * (ey + 1)) == 0) { ++ey; grew = true; }
                    }
                    e[q] = (uint16_t)(ex | (ey << 8));
                }
            }
    };
    const int nthreads = std::max(1, std::min({(int)std::thread::hardware_concurrency(), 16, height / 64}));
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads; ++t)
        pool.emplace_back(rows, (int)((int64_t)height * t / nthreads), (int)((int64_t)height * (t + 1) / nthreads));
    rows(0, height / nthreads);
    for (std::thread& th : pool) th.join();
    *pitch_out = pitch;
}

// Per-cell Chebyshev distance (in cells, saturated at 255) to the nearest occupied cell; 0 = occupied.
// Lets the move kernel prove "nothing within the footprint's reach" with one byte load.
inline void build_cell_field(const uint32_t* bits, int width, int height, int wpr, std::vector<uint8_t>* out) {
    std::vector<uint8_t>& d = *out;
    d.assign((size_t)width * height, 255);
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x)
            if ((bits[(size_t)y * wpr + (x >> 5)] >> (x & 31)) & 1u) d[(size_t)y * width + x] = 0;
    auto at = [&](int x, int y) -> int { return (x < 0 || y < 0 || x >= width || y >= height) ? 255 : d[(size_t)y * width + x]; };
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            int m = d[(size_t)y * width + x];
            m = std::min(m, std::min(std::min(at(x - 1, y - 1), at(x, y - 1)), std::min(at(x + 1, y - 1), at(x - 1, y))) + 1);
            d[(size_t)y * width + x] = (uint8_t)std::min(m, 255);
        }
    for (int y = height - 1; y >= 0; --y)
        for (int x = width - 1; x >= 0; --x) {
            int m = d[(size_t)y * width + x];
            m = std::min(m, std::min(std::min(at(x + 1, y + 1), at(x, y + 1)), std::min(at(x - 1, y + 1), at(x + 1, y))) + 1);
            d[(size_t)y * width + x] = (uint8_t)std::min(m, 255);
        }
}

}  // namespace mrca
