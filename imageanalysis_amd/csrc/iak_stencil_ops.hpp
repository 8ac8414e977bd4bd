// The 3x3 operators of the 8-pixels-per-lane stencil path, shared by the stand-alone stencil
// kernel (one 16-byte buffer load per row) and the executor's fused kernel (rows staged in LDS).
//
// Operator interface:
//   typedef ... Row;                                   a prepared input row kept in registers
//   void prepare(const Raw16&, row, col0, W, H, Row&)  once per loaded row
//   void compute(const Row& up, const Row& mid, const Row& dn, uint32_t o[2])   8 output pixels
//   static const bool kBorderZero   false only for dilation (its border pixels can be reached
//                                   by a neighbour's window)
//   static const bool kManyRegisters  the fused kernel needs more than 64 registers per lane with it
// Raw16 holds bytes col0-4 .. col0+11 of one image row on the FLAT view of the image
// (imageAnalysis.cu:514-521: taps are read at i*W+j, so the pixel after a row's last one is
// the next row's first; flat indices outside the image read 0).
#pragma once
#include "iak_common.hpp"

namespace iak {

// three-input integer ops are single instructions on gfx950; the asm pins them
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t umax3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_max3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ int imed3(int a, int b, int c) {
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// low bytes of four values -> one packed word, three v_perm_b32 (no masking needed)
__device__ __forceinline__ uint32_t pack4_lo(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3) {
    uint32_t p01 = __builtin_amdgcn_perm(b1, b0, 0x0c0c0400u);
    uint32_t p23 = __builtin_amdgcn_perm(b3, b2, 0x0c0c0400u);
    return __builtin_amdgcn_perm(p23, p01, 0x05040100u);
}

// floor(sqrt(q)) & 255 for 0 <= q <= 2*255^2, as (uchar)sqrtf((float)q) gives it
// (imageAnalysis.cu:552).  The hardware square root is good to 1 ulp (3e-5 at 360); the
// true root of a non-square q stays at least 1/(2*361) below the next integer, so adding
// 0.001 before truncating repairs a perfect square that came out one ulp low and never
// lifts a non-square over an integer.
__device__ __forceinline__ uint32_t isqrt_small(uint32_t q) {
    return (uint32_t)(int)(__builtin_amdgcn_sqrtf((float)q) + 0.001f);
}

// bit k of the result = byte k of x is non-zero
// byte b of the result = 1 if byte b of x is non-zero, else 0
__device__ __forceinline__ uint32_t nonzero_bytes(uint32_t x) {
    const uint32_t t = ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x;  // bit 7 of each byte = byte != 0
    return (t >> 7) & 0x01010101u;
}
// bit b of the result = byte b of x is non-zero (a byte-wise dot product with 1, 2, 4, 8 gathers the flags)
__device__ __forceinline__ uint32_t nonzero4(uint32_t x) {
    return __builtin_amdgcn_udot4(nonzero_bytes(x), 0x08040201u, 0u, false);
}
// the same for 16 bytes: bit 4 * w + b = byte b of x[w] is non-zero
__device__ __forceinline__ uint32_t nonzero16(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3) {
    const uint32_t lo = __builtin_amdgcn_udot4(nonzero_bytes(x1), 0x80402010u, nonzero4(x0), false);
    const uint32_t hi = __builtin_amdgcn_udot4(nonzero_bytes(x3), 0x80402010u, nonzero4(x2), false);
    return lo | (hi << 8);
}
// 4 bits -> 4 bytes of 0x00 / 0xff
__device__ __forceinline__ uint32_t spread4(uint32_t bits4) {
    uint32_t t = (bits4 * 0x00204081u) & 0x01010101u;
    return (t << 8) - t;
}
struct Raw16 {
    uint32_t d[4];  // bytes col0-4 .. col0+11
};

// pixel j in [-4, 11] relative to col0
template <int J>
__device__ __forceinline__ uint32_t rpx(const Raw16 &q) {
    return (q.d[(J + 4) >> 2] >> (8 * ((J + 4) & 3))) & 0xffu;
}

// unpacked pixels col0-1 .. col0+8
struct PRow {
    uint32_t p[10];
};
__device__ __forceinline__ void unpack10(const Raw16 &q, PRow &u) {
    u.p[0] = q.d[0] >> 24;
    u.p[1] = q.d[1] & 0xffu;
    u.p[2] = (q.d[1] >> 8) & 0xffu;
    u.p[3] = (q.d[1] >> 16) & 0xffu;
    u.p[4] = q.d[1] >> 24;
    u.p[5] = q.d[2] & 0xffu;
    u.p[6] = (q.d[2] >> 8) & 0xffu;
    u.p[7] = (q.d[2] >> 16) & 0xffu;
    u.p[8] = q.d[2] >> 24;
    u.p[9] = q.d[3] & 0xffu;
}


struct Unpacked8 {
    typedef PRow Row;
    __device__ __forceinline__ void prepare(const Raw16 &q, int, int, int, int, PRow &u) const { unpack10(q, u); }
};

template <class T>
__device__ __forceinline__ void pack8(const T *res, uint32_t o[2]) {
    o[0] = pack4_lo((uint32_t)res[0], (uint32_t)res[1], (uint32_t)res[2], (uint32_t)res[3]);
    o[1] = pack4_lo((uint32_t)res[4], (uint32_t)res[5], (uint32_t)res[6], (uint32_t)res[7]);
}

struct FRow8 {
    float f[10];
};

// linear / average, float weights: `int sum; sum += uchar * float` of the reference
// (imageAnalysis.cu:517-524) truncates after every tap; (int)x == truncf(x) for |x| < 2^31,
// so the sum stays an integral float.
// FUSED: each tap as one fused multiply-add (contraction mode: nvcc's default build of the reference)
template <bool AVERAGE, bool FUSED = false>
struct Lin8Float {
    static const bool kBorderZero = true;
    static const bool kManyRegisters = true;
    typedef FRow8 Row;
    float k[9];
    __device__ __forceinline__ static float tap(float sum, float px, float w) {
        return truncf(FUSED ? __builtin_fmaf(px, w, sum) : sum + px * w);
    }
    __device__ __forceinline__ void prepare(const Raw16 &q, int, int, int, int, FRow8 &f) const {
        f.f[0] = (float)(q.d[0] >> 24);
#pragma unroll
        for (int w = 0; w < 2; w++) {
            f.f[1 + 4 * w] = (float)(q.d[1 + w] & 0xffu);
            f.f[2 + 4 * w] = (float)((q.d[1 + w] >> 8) & 0xffu);
            f.f[3 + 4 * w] = (float)((q.d[1 + w] >> 16) & 0xffu);
            f.f[4 + 4 * w] = (float)(q.d[1 + w] >> 24);
        }
        f.f[9] = (float)(q.d[3] & 0xffu);
    }
    __device__ __forceinline__ void compute(const FRow8 &u, const FRow8 &m, const FRow8 &d, uint32_t o[2]) const {
        int res[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float sum = 0.0f;
            sum = tap(sum, u.f[j], k[0]);
            sum = tap(sum, u.f[j + 1], k[1]);
            sum = tap(sum, u.f[j + 2], k[2]);
            sum = tap(sum, m.f[j], k[3]);
            sum = tap(sum, m.f[j + 1], k[4]);
            sum = tap(sum, m.f[j + 2], k[5]);
            sum = tap(sum, d.f[j], k[6]);
            sum = tap(sum, d.f[j + 1], k[7]);
            sum = tap(sum, d.f[j + 2], k[8]);
            const int s = (int)sum;
            res[j] = AVERAGE ? s / 9 : imed3(s, 0, 255);
        }
        pack8(res, o);
    }
};

// integral weights with every partial sum below 2^23: 24-bit integer multiply-adds, identical results
template <bool AVERAGE>
struct Lin8Int : Unpacked8 {
    static const bool kBorderZero = true;
    static const bool kManyRegisters = false;
    int k[9];
    __device__ __forceinline__ void compute(const PRow &u, const PRow &m, const PRow &d, uint32_t o[2]) const {
        int res[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            int s = __mul24((int)u.p[j], k[0]);
            s += __mul24((int)u.p[j + 1], k[1]);
            s += __mul24((int)u.p[j + 2], k[2]);
            s += __mul24((int)m.p[j], k[3]);
            s += __mul24((int)m.p[j + 1], k[4]);
            s += __mul24((int)m.p[j + 2], k[5]);
            s += __mul24((int)d.p[j], k[6]);
            s += __mul24((int)d.p[j + 1], k[7]);
            s += __mul24((int)d.p[j + 2], k[8]);
            res[j] = AVERAGE ? s / 9 : imed3(s, 0, 255);
        }
        pack8(res, o);
    }
};

// Non-negative weights that are multiples of 2^-q.  With the running sum an integer, adding
// a non-negative product and truncating adds floor(product): every tap contributes
// (pixel * weight*2^q) >> q on its own.  The launcher takes this path only where each float
// operation of Lin8Float would have been exact, so the results are identical.
template <bool AVERAGE>
struct Lin8Fixed : Unpacked8 {
    static const bool kBorderZero = true;
    static const bool kManyRegisters = false;
    uint32_t k[9];  // weight * 2^q
    int q;
    __device__ __forceinline__ void compute(const PRow &u, const PRow &m, const PRow &d, uint32_t o[2]) const {
        int res[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t s = (__umul24(u.p[j], k[0]) >> q) + (__umul24(u.p[j + 1], k[1]) >> q) +
                         (__umul24(u.p[j + 2], k[2]) >> q) + (__umul24(m.p[j], k[3]) >> q) +
                         (__umul24(m.p[j + 1], k[4]) >> q) + (__umul24(m.p[j + 2], k[5]) >> q) +
                         (__umul24(d.p[j], k[6]) >> q) + (__umul24(d.p[j + 1], k[7]) >> q) +
                         (__umul24(d.p[j + 2], k[8]) >> q);
            res[j] = AVERAGE ? (int)s / 9 : (int)min(s, 255u);
        }
        pack8(res, o);
    }
};

// Symmetric smoothing kernels [a b a; b c b; a b a] whose weights are 0 or 2^-s and sum to at
// most 1 (binomial / Gaussian 3x3, small box filters): every tap contributes pixel >> s
// (see Lin8Fixed) and the nine terms can never exceed 255, so the whole filter is byte-wise
// SWAR on packed words -- shift+mask per weight, plain 32-bit adds (no carry can cross a
// byte), byte-aligned windows for the +-1 column taps -- about 4 instructions per pixel and
// no unpacking, clamping or re-packing.  Chosen by the launcher; results are identical to
// the float path because every float operation there is exact for such weights.
struct SwarRow {
    uint32_t a[4], b[4];  // (pixel >> sa), (pixel >> sb) for bytes col0-4 .. col0+11
    uint32_t c[2];        // (pixel >> sc) for the chunk itself
};

struct Lin8Swar {
    static const bool kBorderZero = true;
    static const bool kManyRegisters = false;
    typedef SwarRow Row;
    int sa, sb, sc;            // shifts; 8 means the weight is zero
    __device__ __forceinline__ static uint32_t plane(uint32_t x, int s) {
        return (x >> s) & ((0xffu >> s) * 0x01010101u);  // s == 8 gives 0
    }
    __device__ __forceinline__ void prepare(const Raw16 &q, int, int, int, int, SwarRow &r) const {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            r.a[i] = plane(q.d[i], sa);
            r.b[i] = plane(q.d[i], sb);
        }
        r.c[0] = plane(q.d[1], sc);
        r.c[1] = plane(q.d[2], sc);
    }
    __device__ __forceinline__ void compute(const SwarRow &u, const SwarRow &m, const SwarRow &d, uint32_t o[2]) const {
        uint32_t s1[4];  // terms read at columns j-1 and j+1: corners of the outer rows, edges of the middle row
#pragma unroll
        for (int i = 0; i < 4; i++) s1[i] = u.a[i] + d.a[i] + m.b[i];
        // terms read at column j: edges of the outer rows, centre
        const uint32_t s0a = u.b[1] + d.b[1] + m.c[0];
        const uint32_t s0b = u.b[2] + d.b[2] + m.c[1];
        o[0] = __builtin_amdgcn_alignbyte(s1[1], s1[0], 3) + __builtin_amdgcn_alignbyte(s1[2], s1[1], 1) + s0a;
        o[1] = __builtin_amdgcn_alignbyte(s1[2], s1[1], 3) + __builtin_amdgcn_alignbyte(s1[3], s1[2], 1) + s0b;
    }
};

// Sobel on half-precision pairs: every instruction of the gradient part works on two pixels, and
// the masks (imageAnalysis.cu:560-561) are applied separably -- horizontally once per input
// row, vertically per output row:
//   hd[j] = p[j+1] - p[j-1],  hs[j] = p[j-1] + 2 p[j] + p[j+1]      (per input row, kept)
//   gx = hd(up) + 2 hd(mid) + hd(down),  gy = hs(down) - hs(up)
// A row is unpacked to pairs in two alignments, O_k = (p[2k-1], p[2k]), E_k = (p[2k], p[2k+1])
// (p[j] = pixel col0+j), so the +-1 column taps of the output pair (2k, 2k+1) are whole
// registers: hd = O[k+1] - O[k], hs = O[k] + 2 E[k] + O[k+1].  Both responses are clamped to
// [0,255] (linearFilter's clamp) before the magnitude (imageAnalysis.cu:552).
//
// The arithmetic is float16 on SUBNORMALS: a pixel byte b zero-extended to 16 bits is the
// half-precision number b * 2^-24, so unpacking costs nothing, and every value here is an integer
// of magnitude <= 1020 times 2^-24 -- exact in half precision (gradual underflow, 10-bit
// mantissa).  The squares are summed in float32 by the two-term dot product, together with
// 0.5 * 2^-48: y = sqrt(q + 0.5) * 2^-24 then lies at least 0.0006 inside (r, r + 1) for
// r = floor(sqrt(q)), q <= 2 * 255^2, which the 1-ulp hardware square root cannot leave, and
// the fused y * 2^24 + (2^23 - 0.5) -- the addend is exact, being below 2^23, while every sum
// is above it, where floats are whole numbers -- rounds to 2^23 + r: the result's low mantissa
// byte is (uchar)sqrtf(q).
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

struct SobelRow {
    h16x2 hd[4];
    h16x2 hs[4];
};

struct Sobel8 {
    static const bool kBorderZero = true;
    static const bool kManyRegisters = true;
    typedef SobelRow Row;
    __device__ __forceinline__ static h16x2 pr(uint32_t hi_src, uint32_t lo_src, uint32_t sel) {
        return __builtin_bit_cast(h16x2, __builtin_amdgcn_perm(hi_src, lo_src, sel));
    }
    // min(x, top) and max(., 0) in one instruction: the clamp modifier bounds the result to [0, 1],
    // and top is below 1
    __device__ __forceinline__ static uint32_t min_clamp0(h16x2 x, h16x2 top) {
        uint32_t r;
        asm("v_pk_min_f16 %0, %1, %2 clamp" : "=v"(r) : "v"(__builtin_bit_cast(uint32_t, x)), "v"(__builtin_bit_cast(uint32_t, top)));
        return r;
    }
    __device__ __forceinline__ void prepare(const Raw16 &q, int, int, int, int, SobelRow &r) const {
        // raw byte index = pixel + 4; selector 0x0c = zero byte, 0-3 = second source, 4-7 = first source
        h16x2 o[5], e[4];
        o[0] = pr(q.d[1], q.d[0], 0x0c040c03u);  // p[-1], p[0]
        o[1] = pr(q.d[1], q.d[1], 0x0c020c01u);  // p[1], p[2]
        o[2] = pr(q.d[2], q.d[1], 0x0c040c03u);  // p[3], p[4]
        o[3] = pr(q.d[2], q.d[2], 0x0c020c01u);  // p[5], p[6]
        o[4] = pr(q.d[3], q.d[2], 0x0c040c03u);  // p[7], p[8]
        e[0] = pr(q.d[1], q.d[1], 0x0c010c00u);  // p[0], p[1]
        e[1] = pr(q.d[1], q.d[1], 0x0c030c02u);  // p[2], p[3]
        e[2] = pr(q.d[2], q.d[2], 0x0c010c00u);  // p[4], p[5]
        e[3] = pr(q.d[2], q.d[2], 0x0c030c02u);  // p[6], p[7]
        const h16x2 two = {(_Float16)2, (_Float16)2};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            r.hd[k] = o[k + 1] - o[k];
            r.hs[k] = __builtin_elementwise_fma(e[k], two, o[k]) + o[k + 1];
        }
    }
    __device__ __forceinline__ void compute(const SobelRow &u, const SobelRow &m, const SobelRow &d, uint32_t o[2]) const {
        const h16x2 two = {(_Float16)2, (_Float16)2};
        const h16x2 top = __builtin_bit_cast(h16x2, 0x00ff00ffu);  // 255 * 2^-24 twice
        // 0.5 in units of 2^-48, added by the dot products.  They are written as asm text in their three-address
        // form: the compiler selects the two-address form, which overwrites its addend, and re-creates the constant
        // with a move in front of every one of them (an instruction per pixel).  This chip needs three wait states
        // between a dot product and a different kind of instruction that reads its result, and the compiler cannot
        // see into asm text: the two dot products of a pixel pair share one block that ends with those wait states.
        const float half_q = 0x1p-49f;
        const float magic = 8388607.5f;                            // 2^23 - 0.5: one binade below the sums
        uint32_t res[8];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            h16x2 gx = __builtin_elementwise_fma(m.hd[k], two, u.hd[k]) + d.hd[k];
            h16x2 gy = d.hs[k] - u.hs[k];
            const uint32_t GX = min_clamp0(gx, top), GY = min_clamp0(gy, top);
            // (gx, gy) of one pixel side by side, then gx*gx + gy*gy + 0.5 in one dot product
            const h16x2 g0 = __builtin_bit_cast(h16x2, __builtin_amdgcn_perm(GY, GX, 0x05040100u));
            const h16x2 g1 = __builtin_bit_cast(h16x2, __builtin_amdgcn_perm(GY, GX, 0x07060302u));
            float q0, q1;
            asm("v_dot2_f32_f16 %0, %2, %2, %4\n\t"
                "v_dot2_f32_f16 %1, %3, %3, %4\n\t"
                "s_nop 2"
                : "=&v"(q0), "=&v"(q1)
                : "v"(__builtin_bit_cast(uint32_t, g0)), "v"(__builtin_bit_cast(uint32_t, g1)), "v"(half_q));
            const float y0 = __builtin_amdgcn_sqrtf(q0);
            const float y1 = __builtin_amdgcn_sqrtf(q1);
            res[2 * k] = __builtin_bit_cast(uint32_t, __builtin_fmaf(y0, 0x1p24f, magic));
            res[2 * k + 1] = __builtin_bit_cast(uint32_t, __builtin_fmaf(y1, 0x1p24f, magic));
        }
        pack8(res, o);
    }
};

// Compass filter on 16-bit pairs (two pixels per instruction, rows unpacked as for Sobel8).
// With the per-row horizontal terms
//   D[j] = p[j+1] - p[j-1],   R[j] = p[j-1] + p[j] + p[j+1],   C[j] = p[j]
// and V = R(down) - R(up), E = D(down) + D(up), F = E + D(mid), the four mask responses
// (imageAnalysis.cu:599-602) are
//   s0 = F + D(mid)   s1 = V + F   s2 = V + C(down) - C(up)   s3 = V - F
// (east, south-east, south, south-west; expand the masks to check).  The running sum is carried
// from mask to mask as in the reference (:607-618): sum = |s0|, then sum = |sum + s_k|, and the
// result is the largest sum seen, low byte.  Everything stays within +-4080: int16 is exact.
struct CompassRow {
    s16x2 D[4];
    s16x2 R[4];
    s16x2 C[4];
};

struct Compass8 {
    static const bool kBorderZero = true;
    static const bool kManyRegisters = true;
    typedef CompassRow Row;
    __device__ __forceinline__ static s16x2 pr(uint32_t hi_src, uint32_t lo_src, uint32_t sel) {
        return __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(hi_src, lo_src, sel));
    }
    __device__ __forceinline__ static s16x2 abs2(s16x2 x) {
        const s16x2 zero = {0, 0};
        return __builtin_elementwise_max(x, zero - x);
    }
    __device__ __forceinline__ void prepare(const Raw16 &q, int, int, int, int, CompassRow &r) const {
        s16x2 o[5], e[4];  // O_k = (p[2k-1], p[2k]), E_k = (p[2k], p[2k+1]); selectors as in Sobel8
        o[0] = pr(q.d[1], q.d[0], 0x0c040c03u);
        o[1] = pr(q.d[1], q.d[1], 0x0c020c01u);
        o[2] = pr(q.d[2], q.d[1], 0x0c040c03u);
        o[3] = pr(q.d[2], q.d[2], 0x0c020c01u);
        o[4] = pr(q.d[3], q.d[2], 0x0c040c03u);
        e[0] = pr(q.d[1], q.d[1], 0x0c010c00u);
        e[1] = pr(q.d[1], q.d[1], 0x0c030c02u);
        e[2] = pr(q.d[2], q.d[2], 0x0c010c00u);
        e[3] = pr(q.d[2], q.d[2], 0x0c030c02u);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            r.D[k] = o[k + 1] - o[k];
            r.R[k] = o[k] + e[k] + o[k + 1];
            r.C[k] = e[k];
        }
    }
    __device__ __forceinline__ void compute(const CompassRow &u, const CompassRow &m, const CompassRow &d, uint32_t o[2]) const {
        s16x2 mx[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const s16x2 V = d.R[k] - u.R[k];
            const s16x2 F = d.D[k] + u.D[k] + m.D[k];
            const s16x2 s0 = F + m.D[k], s1 = V + F, s2 = V + (d.C[k] - u.C[k]), s3 = V - F;
            s16x2 sum = abs2(s0);
            s16x2 best = sum;
            sum = abs2(sum + s1); best = __builtin_elementwise_max(best, sum);
            sum = abs2(sum + s2); best = __builtin_elementwise_max(best, sum);
            sum = abs2(sum + s3); best = __builtin_elementwise_max(best, sum);
            mx[k] = best;
        }
        // low byte of each 16-bit result, four pixels per word
        o[0] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, mx[1]), __builtin_bit_cast(uint32_t, mx[0]), 0x06040200u);
        o[1] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, mx[3]), __builtin_bit_cast(uint32_t, mx[2]), 0x06040200u);
    }
};

// Linear / average filter with integral weights whose sums fit 16 bits (255 * sum |w| <= 32767):
// two pixels per instruction on 16-bit pairs, rows unpacked as for Sobel8.  Integer arithmetic
// is exact, so the per-tap truncation of the reference changes nothing and the order of the
// taps does not matter.
struct PkRow {
    s16x2 o[5], e[4];  // O_k = (p[2k-1], p[2k]), E_k = (p[2k], p[2k+1])
};
template <bool AVERAGE>
struct Lin8Pk16 {
    static const bool kBorderZero = true;
    static const bool kManyRegisters = false;
    typedef PkRow Row;
    s16x2 k[9];  // each weight in both halves
    __device__ __forceinline__ static s16x2 pr(uint32_t hi_src, uint32_t lo_src, uint32_t sel) {
        return __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(hi_src, lo_src, sel));
    }
    __device__ __forceinline__ void prepare(const Raw16 &q, int, int, int, int, PkRow &r) const {
        r.o[0] = pr(q.d[1], q.d[0], 0x0c040c03u);
        r.o[1] = pr(q.d[1], q.d[1], 0x0c020c01u);
        r.o[2] = pr(q.d[2], q.d[1], 0x0c040c03u);
        r.o[3] = pr(q.d[2], q.d[2], 0x0c020c01u);
        r.o[4] = pr(q.d[3], q.d[2], 0x0c040c03u);
        r.e[0] = pr(q.d[1], q.d[1], 0x0c010c00u);
        r.e[1] = pr(q.d[1], q.d[1], 0x0c030c02u);
        r.e[2] = pr(q.d[2], q.d[2], 0x0c010c00u);
        r.e[3] = pr(q.d[2], q.d[2], 0x0c030c02u);
    }
    __device__ __forceinline__ void compute(const PkRow &u, const PkRow &m, const PkRow &d, uint32_t o[2]) const {
        s16x2 sum[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            s16x2 s = u.o[p] * k[0];
            s += u.e[p] * k[1];
            s += u.o[p + 1] * k[2];
            s += m.o[p] * k[3];
            s += m.e[p] * k[4];
            s += m.o[p + 1] * k[5];
            s += d.o[p] * k[6];
            s += d.e[p] * k[7];
            s += d.o[p + 1] * k[8];
            sum[p] = s;
        }
        if (AVERAGE) {
            int res[8];
#pragma unroll
            for (int p = 0; p < 4; p++) {
                res[2 * p] = (int)sum[p].x / 9;  // imageAnalysis.cu:487: no clamp, the low byte is kept
                res[2 * p + 1] = (int)sum[p].y / 9;
            }
            pack8(res, o);
        } else {
            const s16x2 zero = {0, 0}, top = {255, 255};
#pragma unroll
            for (int p = 0; p < 4; p++) sum[p] = __builtin_elementwise_min(__builtin_elementwise_max(sum[p], zero), top);
            o[0] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, sum[1]), __builtin_bit_cast(uint32_t, sum[0]), 0x06040200u);
            o[1] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, sum[3]), __builtin_bit_cast(uint32_t, sum[2]), 0x06040200u);
        }
    }
};

// 3x3 median, all weights 1: sort each of the 10 columns once (min3 / med3 / max3), then per
// output the max of the three lows, the median of the three middles, the min of the three
// highs, and the median of those.
struct Median8 : Unpacked8 {
    static const bool kBorderZero = true;
    static const bool kManyRegisters = true;
    __device__ __forceinline__ void compute(const PRow &u, const PRow &m, const PRow &d, uint32_t o[2]) const {
        uint32_t lo[10], md[10], hi[10];
#pragma unroll
        for (int i = 0; i < 10; i++) {
            lo[i] = umin3(u.p[i], m.p[i], d.p[i]);
            md[i] = umed3(u.p[i], m.p[i], d.p[i]);
            hi[i] = umax3(u.p[i], m.p[i], d.p[i]);
        }
        uint32_t res[8];
#pragma unroll
        for (int j = 0; j < 8; j++)
            res[j] = umed3(umax3(lo[j], lo[j + 1], lo[j + 2]), umed3(md[j], md[j + 1], md[j + 2]),
                           umin3(hi[j], hi[j + 1], hi[j + 2]));
        pack8(res, o);
    }
};

// Weighted 3x3 median, every weight in [0, 255]: the nine keys (pixel << 8 | weight) are sorted
// with a 25-comparator network (min/max pairs; ties between equal pixels do not matter), the
// weights are summed along the sorted order, and the answer is the pixel of the first key at
// which the running weight exceeds kernSum/2 -- element kernSum/2 of the reference's replicated,
// sorted window (imageAnalysis.cu:653-677).
struct WMedian8 : Unpacked8 {
    static const bool kBorderZero = true;
    static const bool kManyRegisters = true;
    uint32_t w[9];
    uint32_t rank;  // kernSum / 2
    __device__ __forceinline__ static void ce(uint32_t &a, uint32_t &b) {
        const uint32_t lo = min(a, b), hi = max(a, b);
        a = lo;
        b = hi;
    }
    __device__ __forceinline__ void compute(const PRow &u, const PRow &m, const PRow &d, uint32_t o[2]) const {
        uint32_t res[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t k[9] = {(u.p[j] << 8) | w[0], (u.p[j + 1] << 8) | w[1], (u.p[j + 2] << 8) | w[2],
                             (m.p[j] << 8) | w[3], (m.p[j + 1] << 8) | w[4], (m.p[j + 2] << 8) | w[5],
                             (d.p[j] << 8) | w[6], (d.p[j + 1] << 8) | w[7], (d.p[j + 2] << 8) | w[8]};
            ce(k[0], k[3]); ce(k[1], k[7]); ce(k[2], k[5]); ce(k[4], k[8]);
            ce(k[0], k[7]); ce(k[2], k[4]); ce(k[3], k[8]); ce(k[5], k[6]);
            ce(k[0], k[2]); ce(k[1], k[3]); ce(k[4], k[5]); ce(k[7], k[8]);
            ce(k[1], k[4]); ce(k[3], k[6]); ce(k[5], k[7]);
            ce(k[0], k[1]); ce(k[2], k[4]); ce(k[3], k[5]); ce(k[6], k[8]);
            ce(k[2], k[3]); ce(k[4], k[5]); ce(k[6], k[7]);
            ce(k[1], k[2]); ce(k[3], k[4]); ce(k[5], k[6]);
            uint32_t c[8];
            c[0] = k[0] & 0xffu;
#pragma unroll
            for (int i = 1; i < 8; i++) c[i] = c[i - 1] + (k[i] & 0xffu);
            uint32_t key = k[8];  // the whole weight is always enough
#pragma unroll
            for (int i = 7; i >= 0; i--) key = c[i] > rank ? k[i] : key;
            res[j] = key >> 8;
        }
        pack8(res, o);
    }
};

// morphology on bit masks: bit i of a row mask = pixel col0-1+i (i in [0,10)) is non-zero / is a source
struct MRow8 {
    uint32_t bits;
    uint32_t c[2];  // the packed pixels of the chunk (erosion keeps the grey value)
};
__device__ __forceinline__ uint32_t row_bits8(const Raw16 &q) {
    return ((q.d[0] >> 24) != 0 ? 1u : 0u) | (nonzero4(q.d[1]) << 1) | (nonzero4(q.d[2]) << 5) |
           ((q.d[3] & 0xffu) != 0 ? (1u << 9) : 0u);
}

struct Erode8 {
    static const bool kBorderZero = true;
    static const bool kManyRegisters = false;
    typedef MRow8 Row;
    uint32_t mask;  // bit n set: structuring element n is non-zero
    __device__ __forceinline__ void prepare(const Raw16 &q, int, int, int, int, MRow8 &r) const {
        r.bits = row_bits8(q);
        r.c[0] = q.d[1];
        r.c[1] = q.d[2];
    }
    __device__ __forceinline__ void compute(const MRow8 &u, const MRow8 &m, const MRow8 &d, uint32_t o[2]) const {
        uint32_t ok = 0xffu;  // output j looks at bits j, j+1, j+2 of each row
        if (mask & 1u) ok &= u.bits;
        if (mask & 2u) ok &= u.bits >> 1;
        if (mask & 4u) ok &= u.bits >> 2;
        if (mask & 8u) ok &= m.bits;
        if (mask & 16u) ok &= m.bits >> 1;
        if (mask & 32u) ok &= m.bits >> 2;
        if (mask & 64u) ok &= d.bits;
        if (mask & 128u) ok &= d.bits >> 1;
        if (mask & 256u) ok &= d.bits >> 2;
        // a zero pixel stays zero whatever ok says: the AND with the pixel itself does it
        o[0] = m.c[0] & spread4(ok & 0xfu);
        o[1] = m.c[1] & spread4((ok >> 4) & 0xfu);
    }
};

struct Dilate8 {
    static const bool kBorderZero = false;
    static const bool kManyRegisters = false;
    typedef MRow8 Row;
    uint32_t mask;
    // Source pixels are those that scatter in the reference: foreground and not on the border
    // (row >= 1, col >= 1 for a 3x3 window).  The left halo bit of chunk 0 is the previous
    // row's last pixel, the right halo bit of the last chunk the next row's first.
    __device__ __forceinline__ void prepare(const Raw16 &q, int row, int col0, int W, int, MRow8 &r) const {
        uint32_t b = row_bits8(q);
        if (row < 1) b = 0;
        if (col0 == 0) {
            b &= ~2u;                   // column 0
            if (row - 1 < 1) b &= ~1u;  // left halo = (row-1, W-1)
        }
        const int e = W - col0 + 1;   // the bit of flat pixel "column W" = (row+1, 0): column 0, never a source
        if (e <= 9) b &= ~(1u << e);  // (bits above it only reach outputs past the row's end)
        r.bits = b;
    }
    __device__ __forceinline__ void compute(const MRow8 &u, const MRow8 &m, const MRow8 &d, uint32_t o[2]) const {
        // output x takes element n=(di+1)*3+(dj+1) from source x-(di,dj): the window is mirrored
        uint32_t hit = 0;
        if (mask & 1u) hit |= d.bits >> 2;   // di=-1,dj=-1 -> source (row+1, col+1)
        if (mask & 2u) hit |= d.bits >> 1;
        if (mask & 4u) hit |= d.bits;
        if (mask & 8u) hit |= m.bits >> 2;
        if (mask & 16u) hit |= m.bits >> 1;
        if (mask & 32u) hit |= m.bits;
        if (mask & 64u) hit |= u.bits >> 2;  // di=+1,dj=-1 -> source (row-1, col+1)
        if (mask & 128u) hit |= u.bits >> 1;
        if (mask & 256u) hit |= u.bits;
        o[0] = spread4(hit & 0xfu);
        o[1] = spread4((hit >> 4) & 0xfu);
    }
};

// ---- host side: which exact-arithmetic path a 3x3 linear kernel admits ----------------------
enum LinKind { LIN_INT, LIN_FIXED, LIN_FLOAT, LIN_SWAR, LIN_PK16 };

struct LinPlan {
    LinKind kind;
    int ik[9];
    uint32_t fk[9];
    int q;
    int sa, sb, sc;  // LIN_SWAR
};

// weight -> shift if the weight is 0 (8) or 2^-s with s in [0,7]; -1 otherwise
inline int pow2_shift(float w) {
    if (w == 0.0f) return 8;
    for (int s = 0; s <= 7; s++)
        if (w == 1.0f / (float)(1 << s)) return s;
    return -1;
}

// [a b a; b c b; a b a] with weights 0 or 2^-s and 4a + 4b + c <= 1
inline bool plan_swar(const float *k, LinPlan &p) {
    if (!(k[0] == k[2] && k[0] == k[6] && k[0] == k[8] && k[1] == k[3] && k[1] == k[5] && k[1] == k[7])) return false;
    const int sa = pow2_shift(k[0]), sb = pow2_shift(k[1]), sc = pow2_shift(k[4]);
    if (sa < 0 || sb < 0 || sc < 0) return false;
    if (4.0 * k[0] + 4.0 * k[1] + k[4] > 1.0) return false;
    p.kind = LIN_SWAR;
    p.sa = sa; p.sb = sb; p.sc = sc;
    return true;
}

// Which exact-arithmetic path a 3x3 kernel admits (see Lin8Int / Lin8Fixed / Lin8Float).
inline LinPlan plan_linear3(const float *kernel) {
    LinPlan p{};
    bool integral = true;
    double mag = 0;
    for (int i = 0; i < 9; i++) {
        integral = integral && (kernel[i] == (float)(int)kernel[i]) && kernel[i] > -65536.f && kernel[i] < 65536.f;
        mag += 255.0 * (kernel[i] < 0 ? -kernel[i] : kernel[i]);
    }
    if (integral && mag < 8388608.0) {  // every partial sum exact in float and within 24 bits
        p.kind = mag <= 32767.0 ? LIN_PK16 : LIN_INT;
        for (int i = 0; i < 9; i++) p.ik[i] = (int)kernel[i];
        return p;
    }
    // weights >= 0, all multiples of 2^-q (q <= 12), and 255 * (sum of weights) * 2^q < 2^24:
    // every product and partial sum of the float path is then exact
    bool ok = true;
    int q = 0;
    double total = 0;
    for (int i = 0; i < 9 && ok; i++) {
        const float w = kernel[i];
        if (!(w >= 0.0f) || w > 4096.0f) { ok = false; break; }
        total += w;
        int qi = 0;
        float t = w;
        while (t != (float)(int)t && qi <= 12) { t *= 2.0f; qi++; }
        if (qi > 12) ok = false;
        if (qi > q) q = qi;
    }
    if (ok && q > 0 && total * 255.0 * (double)(1 << q) < 16777216.0) {
        p.kind = LIN_FIXED;
        p.q = q;
        for (int i = 0; i < 9; i++) p.fk[i] = (uint32_t)(kernel[i] * (float)(1 << q));
        return p;
    }
    p.kind = LIN_FLOAT;
    return p;
}


// Runs `launch(op)` with the operator the plan selects (SWAR only for the clamping filter).
// fused: contraction mode.  Only the general float operator differs: the others are chosen for weights whose
// products are exact, where a fused multiply-add and a multiply followed by an add give the same bits.
template <bool AVG, class L>
inline int with_linear3_op(const float *kernel, L launch, bool fused = false) {
    LinPlan p = plan_linear3(kernel);
    if (!AVG && p.kind != LIN_INT && p.kind != LIN_PK16 && plan_swar(kernel, p)) {
        Lin8Swar op;
        op.sa = p.sa; op.sb = p.sb; op.sc = p.sc;
        return launch(op);
    }
    if (p.kind == LIN_PK16) {
        Lin8Pk16<AVG> op;
        for (int i = 0; i < 9; i++) op.k[i] = s16x2{(short)p.ik[i], (short)p.ik[i]};
        return launch(op);
    }
    if (p.kind == LIN_INT) {
        Lin8Int<AVG> op;
        for (int i = 0; i < 9; i++) op.k[i] = p.ik[i];
        return launch(op);
    }
    if (p.kind == LIN_FIXED) {
        Lin8Fixed<AVG> op;
        for (int i = 0; i < 9; i++) op.k[i] = p.fk[i];
        op.q = p.q;
        return launch(op);
    }
    if (fused) {
        Lin8Float<AVG, true> op;
        for (int i = 0; i < 9; i++) op.k[i] = kernel[i];
        return launch(op);
    }
    Lin8Float<AVG> op;
    for (int i = 0; i < 9; i++) op.k[i] = kernel[i];
    return launch(op);
}

// 3x3 structuring element -> bit mask (bit n: element n is non-zero)
inline uint32_t se_mask3(const int *se) {
    uint32_t mask = 0;
    for (int i = 0; i < 9; i++)
        if (se[i] != 0) mask |= 1u << i;
    return mask;
}

// the 8-pixel path needs 4-byte aligned rows and chunks, and 32-bit flat offsets
// the 8-pixel 3x3 path: any alignment, widths from 8 up (32-bit flat offsets inside the kernel)
inline bool fast8_ok(const uint8_t *, const uint8_t *, int W, int H, long long, long long) {
    return W >= 8 && (long long)W * H < (1LL << 30);
}

}  // namespace iak
