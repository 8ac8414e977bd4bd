// All-ones medians of up to 49 taps (5x5, 7x7, 3x7 ...) by bit-sliced radix selection.
//
// medianFilter (imageAnalysis.cu:641-680) sorts the window and takes element [taps / 2].  The same element is
// found from the most significant bit down: among the window's pixels that still agree with the bits decided so
// far (the candidates), count those whose next bit is 0; if the sought rank lies within them the median's bit is
// 0 and the others drop out, otherwise it is 1, the rank moves on by that count and the zeros drop out.  Eight
// rounds, whatever the pixel values are.
//
// Every quantity of that recurrence is one BIT per (output pixel, window tap), so a lane carries 32 horizontally
// adjacent output pixels in the bits of its registers: `cand[tap]` is one register per tap, the count is a
// carry-save adder network on registers (three-input logic ops: a full adder is two instructions for 32 pixels),
// rank and count are six registers each, and a whole level costs about 260 instructions for 32 pixels where
// the column-histogram kernel spends 138 per pixel.
//
// The image reaches that form as eight bit planes in LDS: a workgroup loads the byte tile it needs (on the flat
// view of the image, so a window column past a row's end is the next row's pixel and offsets outside the image
// read 0, as in the reference), each lane transposes 32 pixels to eight plane words (byte permutes and bit
// exchanges between registers), and the tap (dx, dy) of a lane's 32 pixels is a funnel shift of plane
// words of row y + dy.  The median's eight bit words go back to bytes by the same transpose.
#pragma once
#include "iak_common.hpp"

namespace iak {

#define MB_TW 256  // output columns per workgroup (eight lanes of 32 pixels)
#define MB_TH 32   // output rows per workgroup
#define MB_WORDS (MB_TW / 32 + 2)  // plane words per tile row: the tile's columns and 32 on either side

// Three-input logic as one instruction: TT is the function's truth table, i.e. the expression evaluated on
// a = 0xf0, b = 0xcc, c = 0xaa.
template <int TT>
__device__ __forceinline__ uint32_t mb_op3(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, TT);
}
// 32 pixels (bytes of px[0..7], pixel 4 i + j = byte j of px[i]) <-> eight plane words (plane[b] bit p = bit b of
// pixel p).  An element sits at (register i2 i1 i0, bit position j1 j0 b2 b1 b0) and belongs at (register b2 b1 b0,
// position i2 i1 i0 j1 j0): five exchanges of a register-index bit with a position bit, each on four register
// pairs -- two of whole bytes (one byte permute per register), three of nibbles / bit pairs / bits (a shift and a
// bit select per register) -- leave plane b in register b1 b0 b2.  Every exchange is its own inverse, so the way
// back is the same steps in the opposite order.
template <int SH>
__device__ __forceinline__ void mb_exchange(uint32_t &lo, uint32_t &hi) {
    // position bit log2(SH) of `lo` <-> the register bit that tells lo from hi
    constexpr uint32_t M = SH == 4 ? 0x0f0f0f0fu : (SH == 2 ? 0x33333333u : 0x55555555u);
    const uint32_t l = mb_op3<0xca>(M << SH, hi << SH, lo);  // lo keeps its M bits, takes hi's M bits above them
    hi = mb_op3<0xca>(M, lo >> SH, hi);
    lo = l;
}
__device__ __forceinline__ void mb_bytes_to_planes(const uint32_t px[8], uint32_t plane[8]) {
    uint32_t r[8], t[8];
#pragma unroll
    for (int i = 0; i < 4; i++) {  // register bit 2 <-> byte-index bit 1
        t[i] = __builtin_amdgcn_perm(px[i + 4], px[i], 0x05040100u);
        t[i + 4] = __builtin_amdgcn_perm(px[i + 4], px[i], 0x07060302u);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {  // register bit 1 <-> byte-index bit 0
        if (i & 2) continue;
        r[i] = __builtin_amdgcn_perm(t[i + 2], t[i], 0x06020400u);
        r[i + 2] = __builtin_amdgcn_perm(t[i + 2], t[i], 0x07030501u);
    }
#pragma unroll
    for (int i = 0; i < 8; i += 2) mb_exchange<4>(r[i], r[i + 1]);  // register bit 0 <-> pixel bit 2
#pragma unroll
    for (int i = 0; i < 4; i++) mb_exchange<2>(r[i], r[i + 4]);     // register bit 2 <-> pixel bit 1
#pragma unroll
    for (int i = 0; i < 8; i++)
        if (!(i & 2)) mb_exchange<1>(r[i], r[i + 2]);              // register bit 1 <-> pixel bit 0
#pragma unroll
    for (int b = 0; b < 8; b++) plane[b] = r[((b >> 1) & 1) * 4 + (b & 1) * 2 + (b >> 2)];
}
__device__ __forceinline__ void mb_planes_to_bytes(const uint32_t plane[8], uint32_t px[8]) {
    uint32_t r[8], t[8];
#pragma unroll
    for (int b = 0; b < 8; b++) r[((b >> 1) & 1) * 4 + (b & 1) * 2 + (b >> 2)] = plane[b];
#pragma unroll
    for (int i = 0; i < 8; i++)
        if (!(i & 2)) mb_exchange<1>(r[i], r[i + 2]);
#pragma unroll
    for (int i = 0; i < 4; i++) mb_exchange<2>(r[i], r[i + 4]);
#pragma unroll
    for (int i = 0; i < 8; i += 2) mb_exchange<4>(r[i], r[i + 1]);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (i & 2) continue;
        t[i] = __builtin_amdgcn_perm(r[i + 2], r[i], 0x06020400u);
        t[i + 2] = __builtin_amdgcn_perm(r[i + 2], r[i], 0x07030501u);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        px[i] = __builtin_amdgcn_perm(t[i + 4], t[i], 0x05040100u);
        px[i + 4] = __builtin_amdgcn_perm(t[i + 4], t[i], 0x07060302u);
    }
}

// A number per pixel, bit-sliced: bit i of the number of pixel p is bit p of b[i].  MAXV: the largest value it
// can hold (its width follows).
constexpr int mb_width(int maxv) { return maxv <= 0 ? 1 : (maxv < 2 ? 1 : 1 + mb_width(maxv >> 1)); }
template <int MAXV>
struct MbNum {
    static constexpr int W = mb_width(MAXV);
    uint32_t b[W];
};
__device__ __forceinline__ void mb_full_add(uint32_t x, uint32_t y, uint32_t z, uint32_t &sum, uint32_t &carry) {
    sum = mb_op3<0x96>(x, y, z);    // x ^ y ^ z
    carry = mb_op3<0xe8>(x, y, z);  // majority
}
// the count of up to seven one-bit inputs
template <int N>
__device__ __forceinline__ MbNum<N> mb_count(const uint32_t *z) {
    static_assert(N >= 1 && N <= 7, "one window row");
    MbNum<N> r;
    if (N == 1) {
        r.b[0] = z[0];
    } else if (N == 2) {
        r.b[0] = z[0] ^ z[1];
        r.b[1] = z[0] & z[1];
    } else if (N == 3) {
        mb_full_add(z[0], z[1], z[2], r.b[0], r.b[1]);
    } else if (N <= 5) {
        uint32_t s1, c1, c2;
        mb_full_add(z[0], z[1], z[2], s1, c1);
        mb_full_add(s1, z[3], N == 5 ? z[4] : 0u, r.b[0], c2);
        r.b[1] = c1 ^ c2;
        r.b[2] = c1 & c2;
    } else {
        uint32_t s1, c1, s2, c2, c3;
        mb_full_add(z[0], z[1], z[2], s1, c1);
        mb_full_add(z[3], z[4], z[5], s2, c2);
        mb_full_add(s1, s2, N == 7 ? z[6] : 0u, r.b[0], c3);
        mb_full_add(c1, c2, c3, r.b[1], r.b[2]);
    }
    return r;
}
// x + y, ripple carry over the narrower width and half adders above it
template <int A, int B>
__device__ __forceinline__ MbNum<A + B> mb_add(const MbNum<A> &x, const MbNum<B> &y) {
    MbNum<A + B> r;
    constexpr int WA = MbNum<A>::W, WB = MbNum<B>::W, WR = MbNum<A + B>::W;
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < WR; i++) {
        const uint32_t xi = i < WA ? x.b[i] : 0u, yi = i < WB ? y.b[i] : 0u;
        if (i == 0) {
            r.b[0] = xi ^ yi;
            carry = xi & yi;
        } else {
            uint32_t s, c;
            mb_full_add(xi, yi, carry, s, c);
            r.b[i] = s;
            carry = c;
        }
    }
    return r;
}
// the sum of KH row counts of at most KW each, as a balanced tree
template <int KW, int KH, int FIRST = 0, int N = KH>
struct MbTree {
    static __device__ __forceinline__ MbNum<KW * N> sum(const MbNum<KW> *rows) {
        constexpr int L = N / 2;
        return mb_add(MbTree<KW, KH, FIRST, L>::sum(rows), MbTree<KW, KH, FIRST + L, N - L>::sum(rows));
    }
};
template <int KW, int KH, int FIRST>
struct MbTree<KW, KH, FIRST, 1> {
    static __device__ __forceinline__ MbNum<KW> sum(const MbNum<KW> *rows) { return rows[FIRST]; }
};

template <int KW, int KH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KW * KH > 35 ? 3 : 4))) void bitslice_median_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int W,
                                                              int H, long long in_stride, long long out_stride, int tiles_x) {
    constexpr int TAPS = KW * KH, s = (KW - 1) / 2, a = (KH - 1) / 2;
    constexpr int ROWS = MB_TH + KH - 1;
    static_assert(TAPS <= 63 && KW <= 7 && KH <= 15, "bit-sliced median: window too large");
    __shared__ uint32_t planes[8 * ROWS * MB_WORDS];  // [bit][tile row][word]
    const int N = W * H;                              // the launcher keeps W * (H + 64) below 2^31
    const uint8_t *src = in + (long long)blockIdx.y * in_stride;
    uint8_t *dst = out + (long long)blockIdx.y * out_stride;
    // offsets outside [0, N) -- negative ones as unsigned -- read 0: the reference's out-of-range taps; everything is
    // on the flat view of the image (W % 4 == 0 and aligned rows: every dword lies wholly inside or outside)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, N, 0x00020000);
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int x0 = tx * MB_TW, y0 = ty * MB_TH;

    // (A) the tile as bit planes: a task is 32 pixels of one tile row
    for (int task = threadIdx.x; task < ROWS * MB_WORDS; task += 256) {
        const int R = task / MB_WORDS, wj = task - R * MB_WORDS;
        const int f = (y0 - a + R) * W + x0 - 32 + 32 * wj;  // flat offset, a multiple of 4
        uint32_t px[8];
        if (f < 0 && f > -32) {
            // the 32 pixels straddle the image's first one: dword by dword (those in front of it read 0)
#pragma unroll
            for (int i = 0; i < 8; i++)
                px[i] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)min((uint32_t)(f + 4 * i), (uint32_t)N), 0, 0);
        } else {
            const uint32_t o = min((uint32_t)f, (uint32_t)N);  // clamped in a register (no register + immediate split)
            const u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)o, 0, 0);
            const u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)min(o + 16u, (uint32_t)N), 0, 0);
            px[0] = v0[0]; px[1] = v0[1]; px[2] = v0[2]; px[3] = v0[3];
            px[4] = v1[0]; px[5] = v1[1]; px[6] = v1[2]; px[7] = v1[3];
        }
        uint32_t pl[8];
        mb_bytes_to_planes(px, pl);
#pragma unroll
        for (int b = 0; b < 8; b++) planes[(b * ROWS + R) * MB_WORDS + wj] = pl[b];
    }
    __syncthreads();

    // (B) a lane = 32 output pixels: row y0 + r, columns x0 + 32 g .. + 31
    const int r = threadIdx.x >> 3, g = threadIdx.x & 7;
    const int row = y0 + r, col0 = x0 + 32 * g;
    if (row >= H || col0 >= W) return;
    uint32_t cand[TAPS];
#pragma unroll
    for (int q = 0; q < TAPS; q++) cand[q] = 0xffffffffu;
    constexpr int RW = mb_width(TAPS);  // width of rank and count
    uint32_t rank[RW];
#pragma unroll
    for (int i = 0; i < RW; i++) rank[i] = (((TAPS / 2 + 1) >> i) & 1) ? 0xffffffffu : 0u;  // sorted[taps / 2], counted from 1
    // The levels run as a loop (not unrolled: one level's registers, not eight levels' worth); the median's bit words
    // move along med[] so that no register is indexed by the level.
    uint32_t med[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t *pb = planes + (7 * ROWS + r) * MB_WORDS + g;  // words g, g+1, g+2 = columns col0-32 .. col0+63
#pragma unroll 1
    for (int b = 7; b >= 0; b--, pb -= ROWS * MB_WORDS) {
        uint32_t z[TAPS];
        MbNum<KW> rows[KH];
#pragma unroll
        for (int dy = 0; dy < KH; dy++) {
            const uint32_t w0 = pb[dy * MB_WORDS], w1 = pb[dy * MB_WORDS + 1], w2 = pb[dy * MB_WORDS + 2];
#pragma unroll
            for (int dx = 0; dx < KW; dx++) {
                const int d = dx - s;  // bit p of `bits` = bit b of the pixel at column col0 + p + d
                const uint32_t bits = d == 0 ? w1 : (d < 0 ? __builtin_amdgcn_alignbit(w1, w0, 32 + d) : __builtin_amdgcn_alignbit(w2, w1, d));
                z[dy * KW + dx] = mb_op3<0x30>(cand[dy * KW + dx], bits, bits);  // cand & ~bits: candidates whose bit is 0
            }
            rows[dy] = mb_count<KW>(z + dy * KW);
        }
        const MbNum<TAPS> zeros = MbTree<KW, KH>::sum(rows);
        // zeros < rank: the median's bit is 1 and the rank moves on by the zeros (a borrow chain and a subtraction)
        uint32_t lt = 0, borrow = 0, diff[RW];
#pragma unroll
        for (int i = 0; i < RW; i++) {
            const uint32_t zi = zeros.b[i], ri = rank[i];
            lt = mb_op3<0x8e>(zi, ri, lt);          // (~z & r) | (~(z ^ r) & lt)
            diff[i] = mb_op3<0x96>(ri, zi, borrow);
            borrow = mb_op3<0x8e>(ri, zi, borrow);  // (~r & z) | (~(r ^ z) & borrow)
        }
#pragma unroll
        for (int i = 7; i > 0; i--) med[i] = med[i - 1];
        med[0] = lt;  // after the last level med[i] is the median's bit i
        if (b == 0) break;  // the last bit is decided: nothing left to update
#pragma unroll
        for (int i = 0; i < RW; i++) rank[i] = mb_op3<0xe4>(diff[i], rank[i], lt);  // lt ? diff : rank
#pragma unroll
        for (int q = 0; q < TAPS; q++) cand[q] = mb_op3<0x64>(cand[q], z[q], lt);  // bit 1: the ones stay (cand & ~z); bit 0: the zeros (z)
    }
    // border rule (imageAnalysis.cu:649): whole rows, and the columns of the image's two edges
    uint32_t keep = (row < a || row > H - a) ? 0u : 0xffffffffu;
    if (col0 < s) keep &= 0xffffffffu << (s - col0);
    if (col0 + 31 > W - s) keep &= (W - s - col0 >= 0) ? (0xffffffffu >> (31 - (W - s - col0))) : 0u;
#pragma unroll
    for (int b = 0; b < 8; b++) med[b] &= keep;
    uint32_t px[8];
    mb_planes_to_bytes(med, px);
    uint8_t *q = dst + (long long)row * W + col0;
    if (col0 + 32 <= W) {
        st16(q, u32x4{px[0], px[1], px[2], px[3]});
        st16(q + 16, u32x4{px[4], px[5], px[6], px[7]});
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (col0 + 4 * i < W) *reinterpret_cast<uint32_t *>(q + 4 * i) = px[i];  // W % 4 == 0
    }
}

// kW x kH all-ones median with at most 49 taps, kW <= 7; images whose rows start at multiples of 4 bytes.
// IAK_E_UNSUPPORTED otherwise.
static int launch_bitslice_median(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
                                  int kW, int kH, hipStream_t s) {
    if (W % 4 != 0 || reinterpret_cast<uintptr_t>(in) % 4 != 0 || reinterpret_cast<uintptr_t>(out) % 4 != 0 || in_stride % 4 != 0 ||
        out_stride % 4 != 0 || (long long)W * ((long long)H + 64) >= (1LL << 31))
        return IAK_E_UNSUPPORTED;
    const int tiles_x = (W + MB_TW - 1) / MB_TW, tiles_y = (H + MB_TH - 1) / MB_TH;
    const dim3 grid((unsigned)(tiles_x * tiles_y), B);
#define MB_CASE(KW_, KH_)                                                                                                  \
    if (kW == KW_ && kH == KH_) {                                                                                          \
        hipLaunchKernelGGL((bitslice_median_kernel<KW_, KH_>), grid, dim3(256), 0, s, in, out, W, H, in_stride, out_stride, \
                           tiles_x);                                                                                      \
        IAK_LAUNCH_CHECK();                                                                                               \
        return 0;                                                                                                         \
    }
    MB_CASE(7, 7)
    MB_CASE(5, 5)
    MB_CASE(5, 7)
    MB_CASE(7, 5)
    MB_CASE(3, 5)
    MB_CASE(5, 3)
    MB_CASE(3, 7)
    MB_CASE(7, 3)
#undef MB_CASE
    return IAK_E_UNSUPPORTED;
}

}  // namespace iak
