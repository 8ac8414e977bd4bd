// A run of per-pixel operations ("chain") as data: the stage list the executor plans, the
// per-image tables its prologue kernel builds, and the 16-pixel application routine shared by
// the stand-alone chain pass and the kernel that fuses a chain into the following 3x3 stencil.
#pragma once
#include "iak_common.hpp"
#include "iak_device_fns.hpp"

namespace iak {

enum { C_EQUALIZE = 1, C_OTSU = 2, C_CONST_LUT = 3, C_GAUSS = 4, C_SP = 5, C_KMEANS = 6 };
#define CHAIN_MAX 6

struct ChainStage {
    int type;
    float mean, stddev;    // C_GAUSS
    const uint8_t *plane;  // C_GAUSS: rv plane, C_SP: keep-mask plane, C_KMEANS: int32 initial centroids [B][2]
    const uint8_t *plane2; // C_SP: salt-mask plane
    int lut_slot;          // C_CONST_LUT: index into ChainDesc::const_lut
    // C_SP, where the noise is sparse enough: the noisy pixel indices as a sorted list (2 * index + salt) and,
    // per 16-pixel chunk of the image, the first entry at or behind it ([N / 16 + 1])
    const uint32_t *sp_ptr, *sp_list;
};
struct ChainDesc {
    int n = 0;
    ChainStage st[CHAIN_MAX];
    uint8_t const_lut[2][256];  // at most two constant tables per chain
    // > 0: the first compose_n stages are look-up stages and the table prologue leaves their composition
    // (stage compose_n - 1 of stage ... of stage 0) in the slot of stage compose_n - 1
    int compose_n = 0;
};

// LDS image of one image's tables
struct ChainTables {
    uint8_t lut[CHAIN_MAX][256];   // LUT stages: the table; Gaussian stages: the amplitude
    uint8_t nanf[CHAIN_MAX][256];  // Gaussian stages: 0xff where the amplitude was NaN (output 0)
    int any_nan[CHAIN_MAX];
};

// 256 lanes (or more) load the tables of image b; ends with a barrier
__device__ __forceinline__ void chain_load_tables(ChainTables &t, const ChainDesc &d, const uint16_t *tables_b) {
    if (threadIdx.x < CHAIN_MAX) t.any_nan[threadIdx.x] = 0;
    __syncthreads();
    if (threadIdx.x < 256) {
        for (int s = 0; s < d.n; s++) {
            if (d.st[s].type == C_SP) continue;
            const uint32_t e = tables_b[s * 256 + threadIdx.x];
            t.lut[s][threadIdx.x] = (uint8_t)e;
            t.nanf[s][threadIdx.x] = (e & 0x100u) ? 0xff : 0;
            if (d.st[s].type == C_GAUSS && (e & 0x100u)) t.any_nan[s] = 1;
        }
    }
    __syncthreads();
}

// four look-ups through a 256-byte LDS table, result packed (one extract + one LDS read per pixel)
__device__ __forceinline__ uint32_t lookup4(const uint8_t *t, uint32_t x) {
    const uint32_t b0 = t[x & 0xffu], b1 = t[(x >> 8) & 0xffu], b2 = t[(x >> 16) & 0xffu], b3 = t[x >> 24];
    const uint32_t p01 = __builtin_amdgcn_perm(b1, b0, 0x0c0c0400u);
    const uint32_t p23 = __builtin_amdgcn_perm(b3, b2, 0x0c0c0400u);
    return __builtin_amdgcn_perm(p23, p01, 0x05040100u);
}
// byte-wise (a + n) mod 256 on packed words: no carry crosses a byte
__device__ __forceinline__ uint32_t addbytes(uint32_t a, uint32_t n) {
    return ((a & 0x7f7f7f7fu) + (n & 0x7f7f7f7fu)) ^ ((a ^ n) & 0x80808080u);
}

// bit s = stage s is a Gaussian stage whose table holds NaN amplitudes; uniform over the
// workgroup, read once after chain_load_tables so the per-chunk code tests a scalar
__device__ __forceinline__ uint32_t chain_nan_bits(const ChainTables &t) {
    uint32_t bits = 0;
#pragma unroll
    for (int s = 0; s < CHAIN_MAX; s++) bits |= t.any_nan[s] ? (1u << s) : 0u;
    return __builtin_amdgcn_readfirstlane(bits);
}

// the whole chain on 16 pixels whose flat offset in the image is `off` (16-byte aligned planes)
__device__ __forceinline__ u32x4 chain_apply16(u32x4 v, long long off, const ChainDesc &d, const ChainTables &t,
                                               uint32_t nan_bits) {
    for (int s = 0; s < d.n; s++) {
        const int type = d.st[s].type;
        if (type == C_GAUSS) {
            const u32x4 r = ld16(d.st[s].plane + off);
            if ((nan_bits >> s) & 1u) {  // degenerate variance: rare, uniform per image
#pragma unroll
                for (int w = 0; w < 4; w++) v[w] = addbytes(v[w], lookup4(t.lut[s], r[w])) & ~lookup4(t.nanf[s], r[w]);
            } else {
#pragma unroll
                for (int w = 0; w < 4; w++) v[w] = addbytes(v[w], lookup4(t.lut[s], r[w]));
            }
        } else if (type == C_SP) {
            // salt and pepper as two byte masks (0x00 / 0xff): one three-input logic op per four pixels
            const u32x4 keep = ld16(d.st[s].plane + off);
            const u32x4 salt = ld16(d.st[s].plane2 + off);
#pragma unroll
            for (int w = 0; w < 4; w++) v[w] = (v[w] & keep[w]) | salt[w];
        } else {
#pragma unroll
            for (int w = 0; w < 4; w++) v[w] = lookup4(t.lut[s], v[w]);
        }
    }
    return v;
}

// Two chunks at once, stage by stage: the plane loads of both chunks are issued together, so a
// lane waits for one trip to the cache per stage instead of one per stage and chunk.
__device__ __forceinline__ void chain_apply16x2(u32x4 &v0, u32x4 &v1, long long off0, long long off1, const ChainDesc &d,
                                                const ChainTables &t, uint32_t nan_bits) {
    for (int s = 0; s < d.n; s++) {
        const int type = d.st[s].type;
        if (type == C_GAUSS) {
            const u32x4 r0 = ld16(d.st[s].plane + off0);
            const u32x4 r1 = ld16(d.st[s].plane + off1);
            if ((nan_bits >> s) & 1u) {
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    v0[w] = addbytes(v0[w], lookup4(t.lut[s], r0[w])) & ~lookup4(t.nanf[s], r0[w]);
                    v1[w] = addbytes(v1[w], lookup4(t.lut[s], r1[w])) & ~lookup4(t.nanf[s], r1[w]);
                }
            } else {
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    v0[w] = addbytes(v0[w], lookup4(t.lut[s], r0[w]));
                    v1[w] = addbytes(v1[w], lookup4(t.lut[s], r1[w]));
                }
            }
        } else if (type == C_SP) {
            const u32x4 keep0 = ld16(d.st[s].plane + off0), keep1 = ld16(d.st[s].plane + off1);
            const u32x4 salt0 = ld16(d.st[s].plane2 + off0), salt1 = ld16(d.st[s].plane2 + off1);
#pragma unroll
            for (int w = 0; w < 4; w++) {
                v0[w] = (v0[w] & keep0[w]) | salt0[w];
                v1[w] = (v1[w] & keep1[w]) | salt1[w];
            }
        } else {
#pragma unroll
            for (int w = 0; w < 4; w++) {
                v0[w] = lookup4(t.lut[s], v0[w]);
                v1[w] = lookup4(t.lut[s], v1[w]);
            }
        }
    }
}

// one pixel (unaligned tails)
__device__ __forceinline__ uint32_t chain_apply1(uint32_t p, long long off, const ChainDesc &d, const ChainTables &t) {
    for (int s = 0; s < d.n; s++) {
        const int type = d.st[s].type;
        if (type == C_GAUSS) {
            const uint32_t r = d.st[s].plane[off];
            p = t.nanf[s][r] ? 0u : ((p + t.lut[s][r]) & 0xffu);
        } else if (type == C_SP) {
            p = (p & d.st[s].plane[off]) | d.st[s].plane2[off];
        } else {
            p = t.lut[s][p];
        }
    }
    return p;
}

}  // namespace iak
