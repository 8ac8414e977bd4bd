// Box filters whose taps depend on the running sum (see iak_box.hpp), bit-sliced: 32 pixels per register.
//
// For such a weight w every pixel is one of a few classes: f = floor(pixel * w) is 0..3, the pixel is "special"
// or not (its tap adds f to an empty running sum and f + 1 to any other), and it contributes (f >= 1) or not.
// linearFilter's result (imageAnalysis.cu:517-524: row-major taps, the sum truncated after every tap) is then
//     sum f + #special - #(specials up to and including the first contributing tap)      if a tap contributes,
//     0                                                                                  otherwise,
// clamped to 255.  All of it is counting of one-bit properties over the window, so a lane carries 32 horizontally
// adjacent pixels in the bits of its registers, numbers are kept as bit planes (bit i of the number of pixel p =
// bit p of word i) and added with three-input logic (a full adder is two instructions for 32 pixels):
//
//  * a pixel row reaches that form by an 8 x 8 bit transpose of its bytes; its class bits are comparisons of the
//    eight planes with the (at most three) thresholds of f and the (at most four) special values;
//  * per image row and pixel, over the window's KW columns (neighbouring words by funnel shifts): FS = sum of
//    f + special, SP = #special, G = #specials in front of and at the row's first contributing tap, C = any
//    contributing tap;
//  * a lane walks UP its strip of rows.  The window that starts at row t has sum FS = sliding sum (the row that
//    leaves is remembered in registers), "specials that earn nothing" = C(t) ? G(t) : SP(t) + that of the window
//    starting at t + 1 (right whenever the window holds a contributing tap at all), and "holds a contributing tap"
//    = rows since the last row with C < KH.  No branch depends on pixel data.
//
// Which lane has which word of which strip: iak_bitrows.hpp.
#pragma once
#include "iak_bitrows.hpp"
#include "iak_median_bits.hpp"  // included from iak_box.hpp, behind BoxSpecial

namespace iak {

struct BsbDesc {
    uint32_t thr[3];      // pixel >= thr[k]  <=>  floor(pixel * w) > k   (256: no pixel)
    uint32_t special[4];  // the special pixel values (256: unused)
    BitRows g;
};

// the count of up to 15 one-bit inputs
template <int N>
__device__ __forceinline__ MbNum<N> bsb_count(const uint32_t *z) {
    if constexpr (N <= 7) {
        return mb_count<N>(z);
    } else {
        return mb_add(mb_count<7>(z), bsb_count<N - 7>(z + 7));
    }
}
// s += x and s -= x on bit planes, modulo 2^WS
template <int WS, int WX>
__device__ __forceinline__ void bsb_add_into(uint32_t (&s)[WS], const uint32_t (&x)[WX]) {
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < WS; i++) {
        if (i == 0) {
            carry = s[0] & x[0];
            s[0] ^= x[0];
        } else if (i < WX) {
            const uint32_t t = mb_op3<0x96>(s[i], x[i], carry);
            carry = mb_op3<0xe8>(s[i], x[i], carry);
            s[i] = t;
        } else {
            const uint32_t t = s[i] ^ carry;
            carry &= s[i];
            s[i] = t;
        }
    }
}
template <int WS, int WX>
__device__ __forceinline__ void bsb_sub_from(uint32_t (&s)[WS], const uint32_t (&x)[WX]) {
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < WS; i++) {
        if (i == 0) {
            borrow = ~s[0] & x[0];
            s[0] ^= x[0];
        } else if (i < WX) {
            const uint32_t t = mb_op3<0x96>(s[i], x[i], borrow);
            borrow = mb_op3<0x8e>(s[i], x[i], borrow);  // (~s & x) | (~(s ^ x) & borrow)
            s[i] = t;
        } else {
            const uint32_t t = s[i] ^ borrow;
            borrow &= ~s[i];
            s[i] = t;
        }
    }
}
// pixel >= c and pixel == c on the eight planes of 32 pixels, c uniform
__device__ __forceinline__ uint32_t bsb_ge(const uint32_t (&pl)[8], uint32_t c) {
    if (c > 255u) return 0u;
    uint32_t ge = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t m = 0u - ((c >> i) & 1u);
        ge = mb_op3<0x8e>(m, pl[i], ge);  // bit of c set: pl & ge, clear: pl | ge
    }
    return ge;
}
__device__ __forceinline__ uint32_t bsb_eq(const uint32_t (&pl)[8], uint32_t c) {
    if (c > 255u) return 0u;
    uint32_t eq = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t m = 0u - ((c >> i) & 1u);
        eq = mb_op3<0x90>(eq, pl[i], m);  // eq & ~(pl ^ m)
    }
    return eq;
}

// 32 pixels at flat offset f (a multiple of 4; outside [0, N) they read 0)
__device__ __forceinline__ void bsb_load(__amdgpu_buffer_rsrc_t rsrc, int f, int N, uint32_t (&px)[8]) {
    if (f < 0 && f > -32) {
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
}

template <int KW, int KH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4))) void bitslice_box_kernel(
    const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int W, int H, long long in_stride, long long out_stride, BsbDesc d) {
    constexpr int s = (KW - 1) / 2, a = (KH - 1) / 2;
    constexpr int FSW = mb_width(4 * KW), CW = mb_width(KW), SUMW = mb_width(4 * KW * KH), BEFW = mb_width(KW * KH);
    static_assert(KW <= 15 && KH <= 15 && SUMW <= 10 && BEFW <= SUMW, "bit-sliced box: window");
    int image, strip, wj;
    bool stores;
    if (!d.g.locate(blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), threadIdx.x & 63, &image, &strip, &wj, &stores)) return;
    const int N = W * H;  // the launcher keeps W * (H + 64) below 2^31
    const uint8_t *src = in + (long long)image * in_stride;
    uint8_t *dst = out + (long long)image * out_stride;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, N, 0x00020000);
    const int col0 = 32 * wj;
    const int ys = min(strip, d.g.strips) * d.g.rows;  // lanes without a strip read below the image: zeros
    // columns of the image's two edges are zero in the output (imageAnalysis.cu:509)
    uint32_t keep = 0xffffffffu;
    if (col0 < s) keep &= col0 + 31 < s ? 0u : 0xffffffffu << (s - col0);
    if (col0 + 31 > W - s) keep &= (W - s - col0 >= 0) ? (0xffffffffu >> (31 - (W - s - col0))) : 0u;

    uint32_t ring[KH][FSW];
#pragma unroll
    for (int j = 0; j < KH; j++)
#pragma unroll
        for (int i = 0; i < FSW; i++) ring[j][i] = 0u;
    uint32_t sum[SUMW], bef[BEFW], dist[4];
#pragma unroll
    for (int i = 0; i < SUMW; i++) sum[i] = 0u;
#pragma unroll
    for (int i = 0; i < BEFW; i++) bef[i] = 0u;
#pragma unroll
    for (int i = 0; i < 4; i++) dist[i] = 0xffffffffu;

    const int steps = d.g.rows + KH - 1;
    int r = ys + d.g.rows - 1 + a;  // the input row of the step: the bottom row of the strip's last window first
    uint32_t px[8];
    bsb_load(rsrc, r * W + col0, N, px);

    auto step = [&](uint32_t (&slot)[FSW], bool emit) {
        // this row's classes; the next row's pixels are on their way meanwhile
        uint32_t pl[8];
        mb_bytes_to_planes(px, pl);
        bsb_load(rsrc, (r - 1) * W + col0, N, px);
        // (the constants pass through an empty asm so that their 40 bit masks are made here, by scalar
        // instructions, instead of living in scalar registers across the loop, which spills them)
        uint32_t c0 = d.thr[0], c1 = d.thr[1], c2 = d.thr[2], v0 = d.special[0], v1 = d.special[1], v2 = d.special[2], v3 = d.special[3];
        asm volatile("" : "+s"(c0), "+s"(c1), "+s"(c2), "+s"(v0), "+s"(v1), "+s"(v2), "+s"(v3));
        const uint32_t ge1 = bsb_ge(pl, c0), ge2 = bsb_ge(pl, c1), ge3 = bsb_ge(pl, c2);
        uint32_t f0[3], f1[3], sp[3];
        f0[1] = mb_op3<0x96>(ge1, ge2, ge3);
        f1[1] = ge2;
        sp[1] = bsb_eq(pl, v0) | bsb_eq(pl, v1) | bsb_eq(pl, v2) | bsb_eq(pl, v3);
        f0[0] = word_from_left(f0[1]); f0[2] = word_from_right(f0[1]);
        f1[0] = word_from_left(f1[1]); f1[2] = word_from_right(f1[1]);
        sp[0] = word_from_left(sp[1]); sp[2] = word_from_right(sp[1]);
        // the row's window quantities
        uint32_t zf0[KW], zf1[KW], zs[KW], zg[KW];
        uint32_t seen = 0u;  // a contributing tap in front of this one
#pragma unroll
        for (int dx = 0; dx < KW; dx++) {
            const int dd = dx - s;
            auto at = [&](const uint32_t (&w)[3]) {
                return dd == 0 ? w[1] : (dd < 0 ? __builtin_amdgcn_alignbit(w[1], w[0], 32 + dd) : __builtin_amdgcn_alignbit(w[2], w[1], dd));
            };
            zf0[dx] = at(f0);
            zf1[dx] = at(f1);
            zs[dx] = at(sp);
            zg[dx] = zs[dx] & ~seen;
            seen = mb_op3<0xfe>(seen, zf0[dx], zf1[dx]);
        }
        const MbNum<KW> cf0 = bsb_count<KW>(zf0), cf1 = bsb_count<KW>(zf1), cs = bsb_count<KW>(zs), g = bsb_count<KW>(zg);
        const MbNum<2 * KW> t = mb_add(cf0, cs);
        MbNum<KW> th;  // t >> 1
#pragma unroll
        for (int i = 0; i < CW; i++) th.b[i] = i + 1 < MbNum<2 * KW>::W ? t.b[i + 1] : 0u;
        const MbNum<2 * KW> up = mb_add(th, cf1);
        uint32_t fs[FSW];
        fs[0] = t.b[0];
#pragma unroll
        for (int i = 1; i < FSW; i++) fs[i] = i - 1 < MbNum<2 * KW>::W ? up.b[i - 1] : 0u;
        // the window that starts at this row
        bsb_sub_from<SUMW, FSW>(sum, slot);
#pragma unroll
        for (int i = 0; i < FSW; i++) slot[i] = fs[i];
        bsb_add_into<SUMW, FSW>(sum, fs);
        {
            uint32_t nb[BEFW];
#pragma unroll
            for (int i = 0; i < BEFW; i++) nb[i] = bef[i];
            bsb_add_into<BEFW, CW>(nb, cs.b);
#pragma unroll
            for (int i = 0; i < BEFW; i++) bef[i] = i < CW ? mb_op3<0xca>(seen, g.b[i], nb[i]) : (nb[i] & ~seen);  // C ? G : before + SP
        }
        {   // rows since the last one with a contributing tap, 15 = never; 0 for this row if it has one
            const uint32_t c1 = dist[0], c2 = c1 & dist[1], c3 = c2 & dist[2], full = c3 & dist[3];
            const uint32_t n0 = ~dist[0] | full, n1 = (dist[1] ^ c1) | full, n2 = (dist[2] ^ c2) | full, n3 = (dist[3] ^ c3) | full;
            dist[0] = n0 & ~seen; dist[1] = n1 & ~seen; dist[2] = n2 & ~seen; dist[3] = n3 & ~seen;
        }
        if (!emit) return;
        const int y = r + a;
        uint32_t any = 0u;  // dist < KH
#pragma unroll
        for (int i = 0; i < 4; i++) any = ((KH >> i) & 1) ? (~dist[i] | any) : (~dist[i] & any);
        uint32_t val[SUMW];
#pragma unroll
        for (int i = 0; i < SUMW; i++) val[i] = sum[i];
        bsb_sub_from<SUMW, BEFW>(val, bef);
        uint32_t over = 0u;  // above 255
#pragma unroll
        for (int i = 8; i < SUMW; i++) over |= val[i];
        uint32_t mask = any & keep;
        if (y < a || y > H - a) mask = 0u;
        uint32_t o8[8], ob[8];
#pragma unroll
        for (int i = 0; i < 8; i++) o8[i] = i < SUMW ? mb_op3<0xa8>(val[i], over, mask) : (over & mask);  // (val | over) & mask
        mb_planes_to_bytes(o8, ob);
        if (stores && y < H) {
            uint8_t *q = dst + (long long)y * W + col0;
            if (col0 + 32 <= W) {
                st16<IAK_NT_STENCIL_ST>(q, u32x4{ob[0], ob[1], ob[2], ob[3]});
                st16<IAK_NT_STENCIL_ST>(q + 16, u32x4{ob[4], ob[5], ob[6], ob[7]});
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++)
                    if (col0 + 4 * i < W) *reinterpret_cast<uint32_t *>(q + 4 * i) = ob[i];  // W % 4 == 0
            }
        }
    };
    // KH steps per turn of the loop, so that the remembered rows are registers with fixed names
    for (int k0 = 0; k0 < steps; k0 += KH) {
#pragma unroll
        for (int j = 0; j < KH; j++) {
            if (k0 + j < steps) {
                step(ring[j], k0 + j >= KH - 1);
                r--;
            }
        }
    }
}

// KW x KW box (KW = 5 .. 13) of weight w with sum-dependent taps (BOX_SPECIAL), linear filter only; images whose
// rows start at multiples of 4 bytes.  IAK_E_UNSUPPORTED otherwise (15 x 15 would keep its remembered rows in
// scratch memory: the loop is too long to unroll).
static int launch_bitslice_box(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
                               int kW, int kH, float w, const BoxSpecial &sp, hipStream_t s) {
    if (kW != kH || W % 4 != 0 || reinterpret_cast<uintptr_t>(in) % 4 != 0 || reinterpret_cast<uintptr_t>(out) % 4 != 0 ||
        in_stride % 4 != 0 || out_stride % 4 != 0 || (long long)W * ((long long)H + 64) >= (1LL << 31))
        return IAK_E_UNSUPPORTED;
    BsbDesc d;
    for (int k = 0; k < 3; k++) d.thr[k] = 256;
    for (int px = 255; px >= 0; px--) {  // floor(pixel * w) is monotone in the pixel
        const int f = (int)((float)px * w);
        if (f > 3) return IAK_E_UNSUPPORTED;
        for (int k = 0; k < f; k++) d.thr[k] = (uint32_t)px;
    }
    for (int k = 0; k < BOX_MAX_SPECIAL; k++) d.special[k] = sp.value[k];
    // Rows per strip.  The kernel is bound by its vector instructions, a workgroup is four waves, one per SIMD of
    // its CU, and a CU holds three workgroups of this size: the time is about (workgroups of the busiest CU) x
    // (steps of a strip), rows + KH - 1 with two more for the start.  Up to one round of resident workgroups the
    // busiest CU has a whole number of them; beyond that the dispatcher evens the CUs out.
    int rows = 32;
    double best = 1e30;
    for (int r_ = 8; r_ <= 48; r_++) {
        const long long wgs = ((long long)BitRows::waves_per_image(W, H, r_) * B + 3) / 4;
        const double per_cu = wgs <= 3 * 256 ? (double)((wgs + 255) / 256) : (double)wgs / 256.0;
        const double cost = per_cu * (r_ + kH + 1);
        if (cost < best) { best = cost; rows = r_; }
    }
    if (const char *e = getenv("IAK_BSB_ROWS")) rows = atoi(e) > 0 && atoi(e) <= 48 ? atoi(e) : rows;  // for measurements
    d.g = BitRows::make(W, H, B, rows);
    const dim3 grid(d.g.grid_blocks());
#define BSB_CASE(K)                                                                                                       \
    if (kW == K) {                                                                                                        \
        hipLaunchKernelGGL((bitslice_box_kernel<K, K>), grid, dim3(256), 0, s, in, out, W, H, in_stride, out_stride, d);   \
        IAK_LAUNCH_CHECK();                                                                                               \
        return 0;                                                                                                         \
    }
    BSB_CASE(11)
    BSB_CASE(5)
    BSB_CASE(7)
    BSB_CASE(9)
    BSB_CASE(13)
#undef BSB_CASE
    return IAK_E_UNSUPPORTED;
}

}  // namespace iak
