// Shared host/device helpers for the gfx950 kernels behind include/iak.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/iak.h"

#define IAK_HIP(call)                          \
    do {                                       \
        hipError_t e__ = (call);               \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

// launch check: a bad configuration surfaces here instead of at the next sync
#define IAK_LAUNCH_CHECK()                      \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

namespace iak {

// iak_set_contraction's switch (iak_host.cpp): the reference's a * b + c expressions as fused multiply-adds
extern std::atomic<int> g_contraction;
static inline bool contraction() { return g_contraction.load() != 0; }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Blocks along x for a batch launch with gridDim.y = B.  A memory-bound pass wants a few
// thousand workgroups in total (256 CUs x 8) and no more: the rest is grid-stride.
static inline int blocks_x(long long work_items_per_image, int threads, int B, int target_total = 4096) {
    long long need = (work_items_per_image + threads - 1) / threads;
    long long per_image = (target_total + B - 1) / B;
    if (per_image < 1) per_image = 1;
    if (need < per_image) per_image = need;
    if (per_image < 1) per_image = 1;
    return (int)per_image;
}

// byte k of a packed word
__device__ __forceinline__ uint32_t byte_of(uint32_t w, int k) { return (w >> (8 * k)) & 0xffu; }

__device__ __forceinline__ uint32_t pack4(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3) {
    return (b0 & 0xffu) | ((b1 & 0xffu) << 8) | ((b2 & 0xffu) << 16) | (b3 << 24);
}

// 16-byte global load/store on a 16-byte aligned address
// Cache hints per call site (1 = nontemporal), chosen by measurement on the 500-image
// pipeline: the RGB planes are read once and never again (nontemporal); the grayscale image
// is stored normally, which lets the next pass find part of it still on-die; later passes
// store nontemporally; the shared random planes and every stencil's overlapping row reads stay
// cached.
#ifndef IAK_NT_GRAY_LD
#define IAK_NT_GRAY_LD 1
#endif
#ifndef IAK_NT_GRAY_ST
#define IAK_NT_GRAY_ST 0
#endif
#ifndef IAK_NT_CHAIN_LD
#define IAK_NT_CHAIN_LD 0
#endif
#ifndef IAK_NT_CHAIN_ST
#define IAK_NT_CHAIN_ST 1
#endif
#ifndef IAK_NT_STENCIL_LD
#define IAK_NT_STENCIL_LD 0
#endif
#ifndef IAK_NT_STENCIL_ST
#define IAK_NT_STENCIL_ST 1
#endif
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// 16 bytes from / to GLOBAL memory at any alignment: still one dwordx4 instruction (the device
// handles unaligned vector memory accesses), so images at odd addresses and batches of images
// with an odd number of pixels take the same 16-pixel paths as aligned ones.
typedef u32x4 u32x4_any __attribute__((aligned(1)));
template <bool NT = false>
__device__ __forceinline__ u32x4 ld16(const uint8_t *p) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const u32x4_any *>(p));
    return *reinterpret_cast<const u32x4_any *>(p);
}
template <bool NT = false>
__device__ __forceinline__ void st16(uint8_t *p, u32x4 v) {
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4_any *>(p));
    else *reinterpret_cast<u32x4_any *>(p) = v;
}
template <bool NT = false>
__device__ __forceinline__ void st8(uint8_t *p, uint32_t lo, uint32_t hi) {
    u32x2 v = {lo, hi};
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x2 *>(p));
    else *reinterpret_cast<u32x2 *>(p) = v;
}

// 8 bytes to an address of any alignment (one store instruction: the device handles unaligned
// vector memory accesses)
struct __attribute__((packed, aligned(1))) u32x2_unaligned { uint32_t lo, hi; };
__device__ __forceinline__ void st8_unaligned(uint8_t *p, uint32_t lo, uint32_t hi) {
    u32x2_unaligned v;
    v.lo = lo;
    v.hi = hi;
    *reinterpret_cast<u32x2_unaligned *>(p) = v;
}

// float -> unsigned char where the reference leaves the cast undefined: truncate
// toward zero to int, keep the low byte; NaN gives 0.
__device__ __forceinline__ uint32_t f2u8(float v) {
    if (v != v) return 0u;
    return ((uint32_t)(int)v) & 0xffu;
}

}  // namespace iak
