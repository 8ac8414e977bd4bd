// Square box filters of 3x3 up to 9x9 whose per-tap truncation commutes with the sum (see
// iak_box.hpp: the filter is then the integer sum of floor(pixel * w) over the window), with
// everything in registers.
//
// A lane owns 8 pixels x 8 rows of the output, as in the 8-pixel 3x3 kernels: it loads the
// 8 + 2a input rows once (one 16-byte buffer load each: columns col0-4 .. col0+11, which hold
// the +-4 columns a 9-wide window reaches) and walks down them.  A row's 16 terms are kept as
// 16-bit pairs in both alignments -- E[k] = (t[2k], t[2k+1]), O[k] = (t[2k+1], t[2k+2]) -- so
// the horizontal window sums of an output pair are plain 32-bit adds of whole registers (sums
// stay below 2^16, so nothing crosses the halves); the vertical sum slides: add the new row's
// horizontal sums, take away those of the row that left.  No LDS tile, no barrier.
// The terms are the pixels themselves for weight 1 (the all-ones average filter) and come
// from a 256-entry table otherwise.  Average filter: sum / divisor by a multiplication and a
// shift that the host has verified for every possible sum, low byte (imageAnalysis.cu:487:
// no clamp); linear filter: min(sum, 255).  Border pixels (row < a, row > H-a, col < s,
// col > W-s) are 0; taps outside the image read 0 through the buffer descriptor (flat view:
// columns past a row's end are the next row's first ones).  W % 8 == 0, images 4-byte and
// outputs 8-byte aligned.
#pragma once
#include "iak_bitmask.hpp"
#include "iak_common.hpp"
#include "iak_stencil_ops.hpp"

namespace iak {

struct BoxTerms {
    uint16_t t[256];  // floor(pixel * w)
};

template <bool AVG, int A, bool IDENT>
__global__ __launch_bounds__(256) void box_rows_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int W, int H,
                                                       long long in_stride, long long out_stride, uint32_t div_mul,
                                                       int div_shift, BoxTerms terms) {
    constexpr int S = A, KH = 2 * A + 1, ROWS = 8 + 2 * A;
    __shared__ uint16_t lut[256];
    if (!IDENT) {
        lut[threadIdx.x] = terms.t[threadIdx.x];
        __syncthreads();
    }
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
        u32x4 v[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++) v[r] = load16_flat(rsrc, (y0 - A + r) * W + col0 - 4, N);
        // the border rule by column: output x survives iff S - col0 <= x <= W - S - col0
        uint32_t keep = 0xffu;
        {
            const int lo = S - col0, hi = W - S - col0;
            if (lo > 0) keep &= lo >= 8 ? 0u : (0xffu << lo);
            if (hi < 7) keep &= hi < 0 ? 0u : ((1u << (hi + 1)) - 1u);
        }
        const uint32_t keep0 = spread4(keep & 0xfu), keep1 = spread4((keep >> 4) & 0xfu);
        uint32_t ring[KH][4];
        uint32_t V[4] = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            uint32_t E[8], O[7];
            if (IDENT) {
#pragma unroll
                for (int k = 0; k < 8; k++) E[k] = __builtin_amdgcn_perm(0u, v[r][k >> 1], (k & 1) ? 0x0c030c02u : 0x0c010c00u);
            } else {
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    const uint32_t x = v[r][w];
                    const uint32_t t0 = lut[x & 0xffu], t1 = lut[(x >> 8) & 0xffu], t2 = lut[(x >> 16) & 0xffu], t3 = lut[x >> 24];
                    E[2 * w] = t0 | (t1 << 16);
                    E[2 * w + 1] = t2 | (t3 << 16);
                }
            }
#pragma unroll
            for (int k = 0; k < 7; k++) O[k] = __builtin_amdgcn_alignbit(E[k + 1], E[k], 16);
            // horizontal sums of output pairs (2m, 2m+1): the pairs starting at positions 2m+4-S .. 2m+4+S
            uint32_t hp[4];
#pragma unroll
            for (int m = 0; m < 4; m++) {
                uint32_t sum = 0;
#pragma unroll
                for (int j = -S; j <= S; j++) {
                    const int q = 2 * m + 4 + j;
                    sum += (q & 1) ? O[(q - 1) >> 1] : E[q >> 1];
                }
                hp[m] = sum;
            }
            // vertical: the last KH rows
#pragma unroll
            for (int m = 0; m < 4; m++) {
                V[m] += hp[m];
                if (r >= KH) V[m] -= ring[r % KH][m];
                ring[r % KH][m] = hp[m];
            }
            if (r >= 2 * A) {
                const int row = y0 + r - 2 * A;
                uint32_t R[4];
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    if (AVG) {
                        // (the product as unsigned: it may use all 32 bits)
                        const uint32_t q0 = (uint32_t)__umul24(V[m] & 0xffffu, div_mul) >> div_shift;
                        const uint32_t q1 = (uint32_t)__umul24(V[m] >> 16, div_mul) >> div_shift;
                        R[m] = (q0 & 0xffu) | ((q1 & 0xffu) << 16);
                    } else {
                        typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
                        const u16x2 a = __builtin_bit_cast(u16x2, V[m]), lim = {255, 255};
                        R[m] = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(a, lim));
                    }
                }
                uint32_t o0 = __builtin_amdgcn_perm(R[1], R[0], 0x06040200u) & keep0;
                uint32_t o1 = __builtin_amdgcn_perm(R[3], R[2], 0x06040200u) & keep1;
                if (row < A || row > H - A) o0 = o1 = 0;
                if (row < H) st8<IAK_NT_STENCIL_ST>(dst + (long long)row * W + col0, o0, o1);
            }
        }
    }
}

// (mul, shift) with (sum * mul) >> shift == sum / d for every sum in [0, maxsum], mul < 2^24 and
// sum * mul < 2^32 -- checked value by value; false if there is no such pair.
inline bool box_rows_divider(uint32_t d, uint32_t maxsum, uint32_t *mul, int *shift) {
    for (int k = 31; k >= 0; k--) {
        const unsigned long long m = ((1ull << k) + d - 1) / d;
        if (m >= (1ull << 24) || m * maxsum >= (1ull << 32)) continue;
        bool ok = true;
        for (uint32_t s = 0; s <= maxsum && ok; s++) ok = (uint32_t)((s * m) >> k) == s / d;
        if (ok) {
            *mul = (uint32_t)m;
            *shift = k;
            return true;
        }
    }
    return false;
}

// A square kW x kW window (kW odd, <= 9) of one weight w for which the box path is exact
// (box_sum_exact, checked by the caller).  IAK_E_UNSUPPORTED when this path does not apply.
static int launch_box_rows(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
                           int kW, int kH, float w, int divisor, bool avg, hipStream_t stream) {
    const int a = (kH - 1) / 2;
    if (kW != kH || (kW & 1) == 0 || a < 1 || a > 4 || W % 8 != 0 || W < 8 || (long long)W * ((long long)H + 16) >= (1LL << 31))
        return IAK_E_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(in) % 4 != 0 || reinterpret_cast<uintptr_t>(out) % 8 != 0 || in_stride % 4 != 0 ||
        out_stride % 8 != 0)
        return IAK_E_UNSUPPORTED;
    // remembered for the last filter seen: the divider search tries every sum
    struct Memo { float w; int k, divisor; bool avg, ok; uint32_t mul; int shift; BoxTerms terms; };
    static thread_local Memo memo = {0.0f, 0, 0, false, false, 0, 0, {}};
    if (!(memo.w == w && memo.k == kW && memo.divisor == divisor && memo.avg == avg)) {
        Memo m{w, kW, divisor, avg, true, 1, 0, {}};
        uint32_t tmax = 0;
        for (int px = 0; px < 256; px++) {
            m.terms.t[px] = (uint16_t)(uint32_t)((float)px * w);
            if (m.terms.t[px] > tmax) tmax = m.terms.t[px];
        }
        if (avg) m.ok = divisor > 0 && box_rows_divider((uint32_t)divisor, tmax * (uint32_t)(kW * kH), &m.mul, &m.shift);
        memo = m;
    }
    if (!memo.ok) return IAK_E_UNSUPPORTED;
    const bool ident = w == 1.0f;
    const long long items = (long long)(W >> 3) * ((H + 7) / 8);
    const dim3 grid(blocks_x(items, 256, B, 1 << 20), B);
#define BR_LAUNCH(AV, AA, ID)                                                                                          \
    hipLaunchKernelGGL((box_rows_kernel<AV, AA, ID>), grid, dim3(256), 0, stream, in, out, W, H, in_stride, out_stride, \
                       memo.mul, memo.shift, memo.terms)
#define BR_CASE(AA)                                                                        \
    case AA:                                                                               \
        if (avg) { if (ident) BR_LAUNCH(true, AA, true); else BR_LAUNCH(true, AA, false); }   \
        else { if (ident) BR_LAUNCH(false, AA, true); else BR_LAUNCH(false, AA, false); }     \
        break;
    switch (a) {
        BR_CASE(1) BR_CASE(2) BR_CASE(3) BR_CASE(4)
    }
#undef BR_CASE
#undef BR_LAUNCH
    IAK_LAUNCH_CHECK();
    return 0;
}

}  // namespace iak
