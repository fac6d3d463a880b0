// mrca_host.h -- host-side helpers of the env library (plain C++, no HIP).
#pragma once
#include <stdint.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "mrca_device.h"

namespace mrca {

// Free-rectangle field for grid_march_skip, one entry per cell.  For an EMPTY cell the entry packs four
// 8-bit extents L | R<<8 | D<<16 | U<<24 of a rectangle of empty cells [x-L, x+R] x [y-D, y+U] around it
// (cells outside the map are empty); an occupied cell gets kCellOccupied.  The rectangle is grown
// greedily, one side at a time in the order left, right, down, up, while the strip added is entirely
// empty (up to kFieldMaxExtent cells per side) -- any empty rectangle containing the cell is valid for
// the march, larger ones just save jumps.
// Storage: (height + 2*kFieldPadY) rows of `pitch` entries, cell (0,0) at [kFieldPadY][kFieldPadX], the
// border filled with 0 (see FreeRectField).  Rows are independent and are built by a few host threads.
inline void build_free_rect_field(const uint32_t* bits, int width, int height, int wpr, std::vector<uint32_t>* out,
                                  int* pitch_out) {
    // summed-area table of occupied cells for O(1) strip tests
    const size_t sw = (size_t)width + 1;
    std::vector<int32_t> sat(sw * ((size_t)height + 1), 0);
    auto occupied = [&](int x, int y) -> int { return (bits[(size_t)y * wpr + (x >> 5)] >> (x & 31)) & 1u; };
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x)
            sat[(size_t)(y + 1) * sw + x + 1] =
                occupied(x, y) + sat[(size_t)y * sw + x + 1] + sat[(size_t)(y + 1) * sw + x] - sat[(size_t)y * sw + x];
    auto count = [&](int x0, int y0, int x1, int y1) -> int {  // inclusive cell rectangle, clipped to the map
        x0 = std::max(x0, 0); y0 = std::max(y0, 0); x1 = std::min(x1, width - 1); y1 = std::min(y1, height - 1);
        if (x0 > x1 || y0 > y1) return 0;
        return sat[(size_t)(y1 + 1) * sw + x1 + 1] - sat[(size_t)y0 * sw + x1 + 1] - sat[(size_t)(y1 + 1) * sw + x0] +
               sat[(size_t)y0 * sw + x0];
    };
    const int pitch = width + 2 * kFieldPadX;
    out->assign((size_t)pitch * (height + 2 * kFieldPadY), 0);
    uint32_t* field = out->data();
    auto rows = [&](int y_begin, int y_end) {
        const int M = kFieldMaxExtent;   // < 255, so a packed entry can never equal kCellOccupied
        for (int y = y_begin; y < y_end; ++y)
            for (int x = 0; x < width; ++x) {
                uint32_t& e = field[(size_t)(y + kFieldPadY) * pitch + x + kFieldPadX];
                if (occupied(x, y)) {
                    e = kCellOccupied;
                    continue;
                }
                int l = 0, r = 0, d = 0, u = 0;
                for (bool grew = true; grew;) {
                    grew = false;
                    if (l < M && count(x - l - 1, y - d, x - l - 1, y + u) == 0) { ++l; grew = true; }
                    if (r < M && count(x + r + 1, y - d, x + r + 1, y + u) == 0) { ++r; grew = true; }
                    if (d < M && count(x - l, y - d - 1, x + r, y - d - 1) == 0) { ++d; grew = true; }
                    if (u < M && count(x - l, y + u + 1, x + r, y + u + 1) == 0) { ++u; grew = true; }
                }
                e = (uint32_t)l | ((uint32_t)r << 8) | ((uint32_t)d << 16) | ((uint32_t)u << 24);
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
