// mrca_host.h -- host-side helpers of the env library (plain C++, no HIP).
#pragma once
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "mrca_device.h"

namespace mrca {

// Coarse free-distance field for grid_march_skip: blocks of kSkipK x kSkipK cells;
// out[cy*cw+cx] = Chebyshev distance in blocks to the nearest block holding an occupied cell
// (0 = this block is not empty), saturated at 255.  Two-pass chamfer, exact for L-infinity.
// Rows are padded to a multiple of 4 blocks with the value 1 ("that block is free") so that a
// kernel can fetch four blocks per 32-bit load; *pitch_out is the padded row length.
inline void build_skip_field(const uint32_t* bits, int width, int height, int wpr, std::vector<uint8_t>* out,
                             int* cw_out, int* ch_out, int* pitch_out = nullptr) {
    const int cw = (width + kSkipK - 1) / kSkipK, ch = (height + kSkipK - 1) / kSkipK;
    std::vector<int> d((size_t)cw * ch, 255);
    for (int y = 0; y < height; ++y)
        for (int w = 0; w < wpr; ++w) {
            uint32_t v = bits[(size_t)y * wpr + w];
            while (v) {
                const int b = __builtin_ctz(v);
                v &= v - 1;
                const int x = w * 32 + b;
                if (x < width) d[(size_t)(y >> kSkipShift) * cw + (x >> kSkipShift)] = 0;
            }
        }
    auto at = [&](int x, int y) -> int { return (x < 0 || y < 0 || x >= cw || y >= ch) ? 255 : d[(size_t)y * cw + x]; };
    for (int y = 0; y < ch; ++y)
        for (int x = 0; x < cw; ++x) {
            int m = d[(size_t)y * cw + x];
            m = std::min(m, std::min(std::min(at(x - 1, y - 1), at(x, y - 1)), std::min(at(x + 1, y - 1), at(x - 1, y))) + 1);
            d[(size_t)y * cw + x] = std::min(m, 255);
        }
    for (int y = ch - 1; y >= 0; --y)
        for (int x = cw - 1; x >= 0; --x) {
            int m = d[(size_t)y * cw + x];
            m = std::min(m, std::min(std::min(at(x + 1, y + 1), at(x, y + 1)), std::min(at(x - 1, y + 1), at(x + 1, y))) + 1);
            d[(size_t)y * cw + x] = std::min(m, 255);
        }
    const int pitch = pitch_out ? ((cw + 3) & ~3) : cw;
    out->assign((size_t)pitch * ch, 1);
    for (int y = 0; y < ch; ++y)
        for (int x = 0; x < cw; ++x) (*out)[(size_t)y * pitch + x] = (uint8_t)d[(size_t)y * cw + x];
    *cw_out = cw;
    *ch_out = ch;
    if (pitch_out) *pitch_out = pitch;
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
