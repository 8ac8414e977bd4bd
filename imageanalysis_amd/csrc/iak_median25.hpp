// 5x5 median, all weights 1, as a comparator network on packed 16-bit pairs in registers.
//
// A lane owns 8 pixels x 8 rows of the output, as in the 8-pixel 3x3 kernels, and walks down the
// 12 input rows (one 16-byte buffer load each: columns col0-4 .. col0+11, of which it uses
// col0-2 .. col0+9), fetched four rows ahead and unpacked once into six registers of two 16-bit
// pixels.  Per output row:
//   (1) the twelve image columns of the five rows are sorted, two columns per instruction
//       (v_pk_min_u16 / v_pk_max_u16): nine comparators for 5 values.  A sorted column serves
//       the five windows that hold it -- this is the part a per-pixel network would repeat;
//   (2) the sorted columns once more in the odd alignment (pairs of columns 1-2, 3-4, ...), so
//       that for an output PAIR (x, x+1) window column k of both is one whole register;
//   (3) per output pair the 61 comparators (98 min/max operations) of iak_median25_net.inc:
//       a sort of each rank across the five columns and a selection among the 13 positions
//       that can still hold the median.  The network is derived and checked on all 2^25
//       zero-one inputs by tools/make_median25_net.py (zero-one principle).
// About 68 instructions per pixel instead of the ~140 of the column-histogram kernel, which
// keeps every other window size.  The result is sorted[12] of the window (imageAnalysis.cu:641-680
// with a race-free private window); taps outside the image read 0 through the buffer
// descriptor (flat view: columns past a row's end are the next row's first ones); border pixels
// (row < 2, row > H-2, col < 2, col > W-2) are 0.  W % 8 == 0, images 4-byte and outputs 8-byte
// aligned.
#pragma once
#include "iak_bitmask.hpp"
#include "iak_common.hpp"
#include "iak_stencil_ops.hpp"

namespace iak {

typedef unsigned short m25_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t m25_min(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(m25_u16x2, a), __builtin_bit_cast(m25_u16x2, b)));
}
__device__ __forceinline__ uint32_t m25_max(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(m25_u16x2, a), __builtin_bit_cast(m25_u16x2, b)));
}

__global__ __launch_bounds__(256) void median25_rows_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int W,
                                                            int H, long long in_stride, long long out_stride) {
    constexpr int A = 2, S = 2, ROWS = 8 + 2 * A, AHEAD = 4;
    const uint8_t *src = in + (long long)blockIdx.y * in_stride;
    uint8_t *dst = out + (long long)blockIdx.y * out_stride;
    const int N = W * H;  // the launcher keeps W * (H + 16) below 2^31
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, N, 0x00020000);
    const int chunks = W >> 3;
    const int items = chunks * ((H + 7) >> 3);
    for (int it = blockIdx.x * blockDim.x + threadIdx.x; it < items; it += gridDim.x * blockDim.x) {
        const int strip = it / chunks;
        const int col0 = (it - strip * chunks) << 3;
        const int y0 = strip << 3;
        const int voff = (y0 - A) * W + col0 - 4;
        // the border rule by column: output x survives iff S - col0 <= x <= W - S - col0
        uint32_t keep = 0xffu;
        {
            const int lo = S - col0, hi = W - S - col0;
            if (lo > 0) keep &= lo >= 8 ? 0u : (0xffu << lo);
            if (hi < 7) keep &= hi < 0 ? 0u : ((1u << (hi + 1)) - 1u);
        }
        const uint32_t keep0 = spread4(keep & 0xfu), keep1 = spread4((keep >> 4) & 0xfu);
        u32x4 raw[AHEAD + 1];
#pragma unroll
        for (int r = 0; r < AHEAD; r++) raw[r] = load16_flat(rsrc, voff + r * W, N);
        uint32_t px[5][6];  // the last five rows: px[r % 5][m] = pixels col0-2+2m, col0-1+2m as 16-bit halves
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            if (r + AHEAD < ROWS) raw[(r + AHEAD) % (AHEAD + 1)] = load16_flat(rsrc, voff + (r + AHEAD) * W, N);
            {
                const u32x4 q = raw[r % (AHEAD + 1)];  // bytes 2 .. 13 are columns col0-2 .. col0+9
                uint32_t *e = px[r % 5];
                e[0] = __builtin_amdgcn_perm(0u, q[0], 0x0c030c02u);
                e[1] = __builtin_amdgcn_perm(0u, q[1], 0x0c010c00u);
                e[2] = __builtin_amdgcn_perm(0u, q[1], 0x0c030c02u);
                e[3] = __builtin_amdgcn_perm(0u, q[2], 0x0c010c00u);
                e[4] = __builtin_amdgcn_perm(0u, q[2], 0x0c030c02u);
                e[5] = __builtin_amdgcn_perm(0u, q[3], 0x0c010c00u);
            }
            if (r < 2 * A) continue;
            const int row = y0 + r - 2 * A;
            // (1) the columns of the five rows, sorted: c[rank][m]
            uint32_t c[5][6];
#pragma unroll
            for (int i = 0; i < 5; i++)
#pragma unroll
                for (int m = 0; m < 6; m++) c[i][m] = px[(r - 4 + i) % 5][m];
#define M25_COL(a, b)                                  \
    _Pragma("unroll") for (int m = 0; m < 6; m++) {    \
        const uint32_t lo = m25_min(c[a][m], c[b][m]); \
        c[b][m] = m25_max(c[a][m], c[b][m]);           \
        c[a][m] = lo;                                  \
    }
            M25_COL(0, 1) M25_COL(3, 4) M25_COL(2, 4) M25_COL(2, 3) M25_COL(1, 4) M25_COL(0, 3) M25_COL(0, 2) M25_COL(1, 3) M25_COL(1, 2)
#undef M25_COL
            // (2) the odd alignment: o[rank][m] = columns 2m+1, 2m+2
            uint32_t o[5][5];
#pragma unroll
            for (int i = 0; i < 5; i++)
#pragma unroll
                for (int m = 0; m < 5; m++) o[i][m] = __builtin_amdgcn_alignbit(c[i][m + 1], c[i][m], 16);
            // (3) per output pair (2j, 2j+1): window column k is image column 2j+k (and 2j+k+1)
            uint32_t med[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t w[25];
#pragma unroll
                for (int i = 0; i < 5; i++)
#pragma unroll
                    for (int k = 0; k < 5; k++) {
                        const int q = 2 * j + k;
                        w[5 * i + k] = (q & 1) ? o[i][(q - 1) >> 1] : c[i][q >> 1];
                    }
#define M25_CE(a, b)                               \
    {                                              \
        const uint32_t lo = m25_min(w[a], w[b]);   \
        w[b] = m25_max(w[a], w[b]);                \
        w[a] = lo;                                 \
    }
#define M25_MIN(a, b) w[a] = m25_min(w[a], w[b]);
#define M25_MAX(a, b) w[b] = m25_max(w[a], w[b]);
#include "iak_median25_net.inc"
#undef M25_CE
#undef M25_MIN
#undef M25_MAX
                med[j] = w[12];
            }
            uint32_t o0 = __builtin_amdgcn_perm(med[1], med[0], 0x06040200u) & keep0;
            uint32_t o1 = __builtin_amdgcn_perm(med[3], med[2], 0x06040200u) & keep1;
            if (row < A || row > H - A) o0 = o1 = 0;
            if (row < H) st8<IAK_NT_STENCIL_ST>(dst + (long long)row * W + col0, o0, o1);
        }
    }
}

// IAK_E_UNSUPPORTED when this path does not apply (the caller then takes the column-histogram kernel).
static int launch_median25_rows(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
                                hipStream_t stream) {
    if (W % 8 != 0 || W < 8 || (long long)W * ((long long)H + 16) >= (1LL << 31)) return IAK_E_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(in) % 4 != 0 || reinterpret_cast<uintptr_t>(out) % 8 != 0 || in_stride % 4 != 0 ||
        out_stride % 8 != 0)
        return IAK_E_UNSUPPORTED;
    const long long items = (long long)(W >> 3) * ((H + 7) / 8);
    hipLaunchKernelGGL(median25_rows_kernel, dim3(blocks_x(items, 256, B, 1 << 20), B), dim3(256), 0, stream, in, out, W, H,
                       in_stride, out_stride);
    IAK_LAUNCH_CHECK();
    return 0;
}

}  // namespace iak
