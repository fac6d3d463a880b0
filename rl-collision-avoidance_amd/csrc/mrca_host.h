// mrca_host.h -- host-side helpers of the env library (plain C++, no HIP).
#pragma once
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "mrca_device.h"

namespace mrca {

// Free-rectangle field for grid_march_skip: blocks of kSkipK x kSkipK cells.  For an EMPTY block the
// entry packs four 8-bit extents L | R<<8 | D<<16 | U<<24 of a rectangle of empty blocks
// [cx-L, cx+R] x [cy-D, cy+U] around it (blocks outside the map are empty); a block holding an
// occupied cell gets kBlockFull.  The rectangle is grown greedily, one side at a time in the order
// left, right, down, up, while the strip added is entirely empty (up to kSkipMaxExtent blocks per side) -- any
// empty rectangle containing the block is valid for the march, larger ones just save steps.
// Storage: (ch + 2*kSkipPadY) rows of `pitch` entries, the map's block (0,0) at [kSkipPadY][kSkipPadX], the
// border filled with 0 (see GlobalDist).
inline void build_skip_field(const uint32_t* bits, int width, int height, int wpr, std::vector<uint32_t>* out,
                             int* cw_out, int* ch_out, int* pitch_out) {
    const int cw = (width + kSkipK - 1) / kSkipK, ch = (height + kSkipK - 1) / kSkipK;
    std::vector<uint8_t> full((size_t)cw * ch, 0);
    for (int y = 0; y < height; ++y)
        for (int w = 0; w < wpr; ++w) {
            uint32_t v = bits[(size_t)y * wpr + w];
            while (v) {
                const int b = __builtin_ctz(v);
                v &= v - 1;
                const int x = w * 32 + b;
                if (x < width)  // 2x2 occupancy of the block, bit (y&1)*2 + (x&1)
                    full[(size_t)(y >> kSkipShift) * cw + (x >> kSkipShift)] |= (uint8_t)(1u << (((y & 1) << 1) | (x & 1)));
            }
        }
    // summed-area table of non-empty blocks for O(1) strip tests (blocks outside the map count as empty)
    std::vector<int> sat((size_t)(cw + 1) * (ch + 1), 0);
    for (int y = 0; y < ch; ++y)
        for (int x = 0; x < cw; ++x)
            sat[(size_t)(y + 1) * (cw + 1) + x + 1] = (full[(size_t)y * cw + x] != 0) + sat[(size_t)y * (cw + 1) + x + 1] +
                                                      sat[(size_t)(y + 1) * (cw + 1) + x] - sat[(size_t)y * (cw + 1) + x];
    auto count = [&](int x0, int y0, int x1, int y1) -> int {  // inclusive block rectangle, clipped to the map
        x0 = std::max(x0, 0); y0 = std::max(y0, 0); x1 = std::min(x1, cw - 1); y1 = std::min(y1, ch - 1);
        if (x0 > x1 || y0 > y1) return 0;
        return sat[(size_t)(y1 + 1) * (cw + 1) + x1 + 1] - sat[(size_t)y0 * (cw + 1) + x1 + 1] -
               sat[(size_t)(y1 + 1) * (cw + 1) + x0] + sat[(size_t)y0 * (cw + 1) + x0];
    };
    const int pitch = (cw + 2 * kSkipPadX + 1) & ~1;
    out->assign((size_t)pitch * (ch + 2 * kSkipPadY), 0);
    for (int y = 0; y < ch; ++y)
        for (int x = 0; x < cw; ++x) {
            if (full[(size_t)y * cw + x]) {
                (*out)[(size_t)(y + kSkipPadY) * pitch + x + kSkipPadX] = kBlockFull | full[(size_t)y * cw + x];
                continue;
            }
            int l = 0, r = 0, d = 0, u = 0;
            for (bool grew = true; grew;) {
                grew = false;
                const int M = kSkipMaxExtent;   // < 255, so a packed entry can never equal kBlockFull
                if (l < M && count(x - l - 1, y - d, x - l - 1, y + u) == 0) { ++l; grew = true; }
                if (r < M && count(x + r + 1, y - d, x + r + 1, y + u) == 0) { ++r; grew = true; }
                if (d < M && count(x - l, y - d - 1, x + r, y - d - 1) == 0) { ++d; grew = true; }
                if (u < M && count(x - l, y + u + 1, x + r, y + u + 1) == 0) { ++u; grew = true; }
            }
            (*out)[(size_t)(y + kSkipPadY) * pitch + x + kSkipPadX] =
                (uint32_t)l | ((uint32_t)r << 8) | ((uint32_t)d << 16) | ((uint32_t)u << 24);
        }
    *cw_out = cw;
    *ch_out = ch;
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
