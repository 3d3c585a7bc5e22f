// expand_kernel.inl -- k_expand: sparse coefficient events -> the dense 8x8 LEVEL blocks k_recon reads.
//
// The host side of state.rs:295-381 (inverse_rle without its arithmetic) only ever places a handful of
// non-zero LEVELs per coded block; shipping each as one 32-bit event (level << 16 | raster position x + 8*y)
// instead of a 128-byte block cuts the PCIe traffic of a typical P picture four-fold.  This kernel rebuilds the
// dense pool in HBM: 8 lanes per block, lane r owns coefficient row r (16 bytes), scans the block's events and
// writes its row with one 16-byte store -- every byte of the pool is written, nothing needs pre-zeroing.
#pragma once

#include "dev_common.h"

namespace h263mi {

constexpr int EXPAND_THREADS = 256;        // 32 blocks per workgroup

H263_DEV void expand_lane(const uint32_t *block_first_event, const uint32_t *events, int16_t *coeffs, uint32_t n_blocks,
                          uint32_t block, int row)
{
    if (block >= n_blocks) return;
    const uint32_t first = block_first_event[block], last = block_first_event[block + 1];
    uint32_t w[4] = {0, 0, 0, 0};
    for (uint32_t e = first; e < last; e++) {
        const uint32_t ev = events[e];
        const uint32_t pos = ev & 63u;
        if ((int)(pos >> 3) != row) continue;
        const uint32_t col = pos & 7u, level = ev >> 16;
        // a later event on the same position replaces the earlier one (the dense writer of the parser does too)
        w[col >> 1] = (w[col >> 1] & ~(0xffffu << (16 * (col & 1)))) | (level << (16 * (col & 1)));
    }
    *reinterpret_cast<uint4 *>(coeffs + (size_t)block * 64 + (size_t)row * 8) = make_uint4(w[0], w[1], w[2], w[3]);
}

}  // namespace h263mi
