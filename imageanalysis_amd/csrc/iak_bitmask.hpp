// Rows of an image as bit masks: which pixels are above a threshold, and the nine mask bits
// around a pixel as a table index.  Shared by the kernels that run a 3x3 operation on a
// thresholded image by table look-up.
#pragma once
#include "iak_common.hpp"

namespace iak {

// 16 bytes of the image at flat offset voff (a multiple of 4, possibly negative or past the
// end): bytes outside [0, N) read 0.  A wholly negative offset is out of range as unsigned; the
// clamp keeps the compiler from splitting the offset into a negative register part and a
// positive immediate (the range check is done on the unwrapped sum).  Only a load that
// straddles offset 0 needs care: the range check would drop all of it, so its bytes are
// fetched one by one.
__device__ __forceinline__ u32x4 load16_flat(__amdgpu_buffer_rsrc_t rsrc, int voff, int N) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)min((uint32_t)voff, (uint32_t)N), 0, 0);
    if (voff < 0 && voff > -16) {
#pragma unroll
        for (int w = 0; w < 4; w++) {
            uint32_t x = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int o = voff + 4 * w + b;
                const uint32_t byte = __builtin_amdgcn_raw_buffer_load_b8(rsrc, (int)min((uint32_t)o, (uint32_t)N), 0, 0);
                x |= (o < 0 ? 0u : byte) << (8 * b);
            }
            v[w] = x;
        }
    }
    return v;
}

// byte b of the result = 1 if byte b of x is greater than t, with kw = (255 - t) in every byte:
// the carry out of the byte-wise sum x + kw
__device__ __forceinline__ uint32_t bytes_above(uint32_t x, uint32_t kw) {
    const uint32_t lo = (x & 0x7f7f7f7fu) + (kw & 0x7f7f7f7fu);
    const uint32_t carry = (x & kw) | ((x | kw) & lo);
    return (carry >> 7) & 0x01010101u;
}

// bit 4 * w + b of the result = byte b of x[w] is above the threshold (byte-wise dot products
// with 1, 2, 4, ... gather the flags)
__device__ __forceinline__ uint32_t above16(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t kw) {
    const uint32_t lo = __builtin_amdgcn_udot4(bytes_above(x1, kw), 0x80402010u,
                                               __builtin_amdgcn_udot4(bytes_above(x0, kw), 0x08040201u, 0u, false), false);
    const uint32_t hi = __builtin_amdgcn_udot4(bytes_above(x3, kw), 0x80402010u,
                                               __builtin_amdgcn_udot4(bytes_above(x2, kw), 0x08040201u, 0u, false), false);
    return lo | (hi << 8);
}

// One 16-byte row load (columns col0-4 .. col0+11 at flat offset voff + 4) as 10 mask bits:
// bit i = pixel col0 - 1 + i is above the threshold.
__device__ __forceinline__ uint32_t mask_row10(__amdgpu_buffer_rsrc_t rsrc, int voff, int N, uint32_t kw) {
    const u32x4 v = load16_flat(rsrc, voff, N);
    return (above16(v[0], v[1], v[2], v[3], kw) >> 3) & 0x3ffu;
}

// the 9-bit table index of pixel j of a row from the three 10-bit rows packed side by side
// (up | mid << 10 | down << 20): bits j..j+2 of each, gathered by one 24-bit multiply whose
// three partial products do not overlap
__device__ __forceinline__ uint32_t nine_bits(uint32_t packed_rows, int j) {
    return (__umul24((packed_rows >> j) & 0x00701c07u, 0x4081u) >> 14) & 0x1ffu;
}

}  // namespace iak
