// The kernel that runs a chain of per-pixel operations and the 3x3 stencil behind it in one
// pass, staging the chain's output in LDS.
#pragma once
#include "iak_chain.hpp"
#include "iak_stencil_ops.hpp"

namespace iak {

// ---------------------------------------------------------------------------------------
// Chain fused into the following 3x3 stencil.  A workgroup stages a tile of the CHAIN'S OUTPUT
// in LDS -- (TH+2) rows x (TW+32) columns on the flat view of the image, every pixel pushed
// through the chain exactly once (plus the thin halo) -- and then runs the stencil operator of
// iak_stencil_ops.hpp on rows read from LDS instead of from memory.  The intermediate image
// is never written: one read of the chain's input and one write of the stencil's output.
// Pixels outside the image are staged as 0 (the reference's out-of-range taps), not as
// chain(0).
// ---------------------------------------------------------------------------------------
#define CS_THREADS 512
template <class Op>
__global__ __launch_bounds__(CS_THREADS) void chain_stencil8_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                             int W, int H, long long in_stride, long long out_stride,
                                                             ChainDesc d, const uint16_t *__restrict__ tables, int TW,
                                                             int TH, int tiles_x, Op op) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int pitch = TW + 32;
    const int tile_bytes = (TH + 2) * pitch;
    uint8_t *tile = smem;
    ChainTables &t = *reinterpret_cast<ChainTables *>(smem + tile_bytes);  // tile_bytes is a multiple of 16
    const int b = blockIdx.y;
    chain_load_tables(t, d, tables + (long long)b * CHAIN_MAX * 256);
    const uint32_t nan_bits = chain_nan_bits(t);
    const uint8_t *src = in + (long long)b * in_stride;
    uint8_t *dst = out + (long long)b * out_stride;
    const long long N = (long long)W * H;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int x0 = tx * TW, y0 = ty * TH;

    // phase 1: stage chain(input) for rows y0-1 .. y0+TH, columns x0-16 .. x0+TW+15.
    // Chunk c of the tile is (row c / cpr, 16-byte column c % cpr); a lane takes chunks
    // tid, tid + T, tid + 2T, ... two at a time.  The division is done once: afterwards row,
    // column, flat image offset and tile offset advance by additions (32-bit: W*H < 2^31).
    const int cpr = pitch >> 4;  // 16-byte chunks per tile row
    const int nchunks = (TH + 2) * cpr;
    const int step_r = CS_THREADS / cpr, step_k = CS_THREADS - step_r * cpr;  // one stride of T chunks
    const int Ni = (int)N;
    int r0 = (int)threadIdx.x / cpr, k0 = (int)threadIdx.x - r0 * cpr;
    int f0 = (y0 - 1 + r0) * W + (x0 - 16 + 16 * k0);  // flat offset of the chunk in the image (may be < 0 or >= N)
    int t0 = r0 * pitch + 16 * k0;                     // its offset in the tile
    const int df_row = W - 16 * cpr, dt_row = pitch - 16 * cpr;  // correction when the column wraps to the next row
    for (int c = threadIdx.x; c < nchunks; c += 2 * CS_THREADS) {
        // second chunk of the pair: one stride further
        int k1 = k0 + step_k, f1 = f0 + step_r * W + 16 * step_k, t1 = t0 + step_r * pitch + 16 * step_k;
        if (k1 >= cpr) { k1 -= cpr; f1 += df_row; t1 += dt_row; }
        const bool in1 = c + CS_THREADS < nchunks;
        const bool ok0 = f0 >= 0 && f0 < Ni;  // W % 16 == 0: a chunk is wholly inside or outside
        const bool ok1 = in1 && f1 >= 0 && f1 < Ni;
        const int g0 = ok0 ? f0 : 0, g1 = ok1 ? f1 : 0;
        u32x4 v0 = ld16<IAK_NT_CHAIN_LD>(src + g0);
        u32x4 v1 = ld16<IAK_NT_CHAIN_LD>(src + g1);
        v0 = chain_apply16(v0, g0, d, t, nan_bits);
        v1 = chain_apply16(v1, g1, d, t, nan_bits);
        const u32x4 zero = {0, 0, 0, 0};
        *reinterpret_cast<u32x4 *>(tile + t0) = ok0 ? v0 : zero;
        if (in1) *reinterpret_cast<u32x4 *>(tile + t1) = ok1 ? v1 : zero;
        // first chunk of the next pair: two strides from this one
        k0 = k1 + step_k; f0 = f1 + step_r * W + 16 * step_k; t0 = t1 + step_r * pitch + 16 * step_k;
        if (k0 >= cpr) { k0 -= cpr; f0 += df_row; t0 += dt_row; }
    }
    __syncthreads();

    // phase 2: the stencil on 8-pixel chunks x groups of 8 rows, rows coming from the tile
    const int cols = TW >> 3;
    const int items = cols * (TH >> 3);
    for (int it = threadIdx.x; it < items; it += CS_THREADS) {
        const int rg = it / cols, cx = it - rg * cols;
        const int col0 = x0 + (cx << 3);
        const int yb = y0 + (rg << 3);
        if (col0 >= W || yb >= H) continue;
        const uint8_t *base = tile + (rg << 3) * pitch + ((cx << 3) + 12);  // column col0-4 of tile row 8*rg
        Raw16 raw[10];
#pragma unroll
        for (int r = 0; r < 10; r++) {
            const uint32_t *q = reinterpret_cast<const uint32_t *>(base + r * pitch);
            raw[r].d[0] = q[0]; raw[r].d[1] = q[1]; raw[r].d[2] = q[2]; raw[r].d[3] = q[3];
        }
        typename Op::Row up, mid, dn;
        op.prepare(raw[0], yb - 1, col0, W, H, up);
        op.prepare(raw[1], yb, col0, W, H, mid);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int row = yb + r;
            op.prepare(raw[r + 2], row + 1, col0, W, H, dn);
            uint32_t o[2];
            op.compute(up, mid, dn, o);
            if (Op::kBorderZero) {
                if (row < 1) o[0] = o[1] = 0;
                if (col0 == 0) o[0] &= 0xffffff00u;
            }
            if (row < H) st8<IAK_NT_STENCIL_ST>(dst + (long long)row * W + col0, o[0], o[1]);
            up = mid;
            mid = dn;
        }
    }
}

// The same pass for images no wider than one tile.  The +-1 column taps of a row's first and
// last pixel are then pixels of the neighbouring rows (the flat view), which the tile holds anyway:
// the tile is the flat range of rows y0-1 .. y0+TH with one 16-byte chunk before and one behind, no
// halo columns, and chunk c of the tile is chunk c of that range -- no row / column arithmetic, about
// 4% fewer chunks, and for the bench geometry every lane stages exactly four.
template <class Op>
__global__ __launch_bounds__(CS_THREADS) void chain_stencil8_flat_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                                  int W, int H, long long in_stride, long long out_stride,
                                                                  ChainDesc d, const uint16_t *__restrict__ tables, int TH,
                                                                  Op op) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tile_bytes = (TH + 2) * W + 32;
    uint8_t *tile = smem;
    ChainTables &t = *reinterpret_cast<ChainTables *>(smem + tile_bytes);  // tile_bytes is a multiple of 16
    const int b = blockIdx.y;
    chain_load_tables(t, d, tables + (long long)b * CHAIN_MAX * 256);
    const uint32_t nan_bits = chain_nan_bits(t);
    const uint8_t *src = in + (long long)b * in_stride;
    uint8_t *dst = out + (long long)b * out_stride;
    const int Ni = W * H;
    const int y0 = blockIdx.x * TH;

    // phase 1: chunk c of the tile is the 16 pixels at flat offset fbase + 16 c (W % 16 == 0: a chunk
    // is wholly inside or outside the image; outside is staged as 0)
    const int nchunks = tile_bytes >> 4;
    const int fbase = (y0 - 1) * W - 16;
    const int T = blockDim.x;  // 512, or 256 for the operators that need many registers (see the launcher)
    for (int c = threadIdx.x; c < nchunks; c += 2 * T) {
        const int c1 = c + T;
        const int f0 = fbase + 16 * c, f1 = fbase + 16 * c1;
        const bool in1 = c1 < nchunks;
        const bool ok0 = f0 >= 0 && f0 < Ni;
        const bool ok1 = in1 && f1 >= 0 && f1 < Ni;
        u32x4 v0 = {0, 0, 0, 0}, v1 = {0, 0, 0, 0};
        if (ok0 || ok1) {  // whole waves skip the rows below the image in the last tile
            const int g0 = ok0 ? f0 : 0, g1 = ok1 ? f1 : 0;
            v0 = ld16<IAK_NT_CHAIN_LD>(src + g0);
            v1 = ld16<IAK_NT_CHAIN_LD>(src + g1);
            v0 = chain_apply16(v0, g0, d, t, nan_bits);
            v1 = chain_apply16(v1, g1, d, t, nan_bits);
            const u32x4 zero = {0, 0, 0, 0};
            v0 = ok0 ? v0 : zero;
            v1 = ok1 ? v1 : zero;
        }
        *reinterpret_cast<u32x4 *>(tile + 16 * c) = v0;
        if (in1) *reinterpret_cast<u32x4 *>(tile + 16 * c1) = v1;
    }
    __syncthreads();

    // phase 2: the stencil on 8-pixel chunks x groups of 8 rows, rows coming from the tile
    const int cols = W >> 3;
    const int items = cols * (TH >> 3);
    for (int it = threadIdx.x; it < items; it += T) {
        const int rg = it / cols, cx = it - rg * cols;
        const int col0 = cx << 3;
        const int yb = y0 + (rg << 3);
        if (yb >= H) continue;
        const uint8_t *base = tile + 16 + (rg << 3) * W + (col0 - 4);  // column col0-4 of image row yb-1
        Raw16 raw[10];
#pragma unroll
        for (int r = 0; r < 10; r++) {
            const uint32_t *q = reinterpret_cast<const uint32_t *>(base + r * W);
            raw[r].d[0] = q[0]; raw[r].d[1] = q[1]; raw[r].d[2] = q[2]; raw[r].d[3] = q[3];
        }
        typename Op::Row up, mid, dn;
        op.prepare(raw[0], yb - 1, col0, W, H, up);
        op.prepare(raw[1], yb, col0, W, H, mid);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int row = yb + r;
            op.prepare(raw[r + 2], row + 1, col0, W, H, dn);
            uint32_t o[2];
            op.compute(up, mid, dn, o);
            if (Op::kBorderZero) {
                if (row < 1) o[0] = o[1] = 0;
                if (col0 == 0) o[0] &= 0xffffff00u;
            }
            if (row < H) st8<IAK_NT_STENCIL_ST>(dst + (long long)row * W + col0, o[0], o[1]);
            up = mid;
            mid = dn;
        }
    }
}

// ---------------------------------------------------------------------------------------
// The flat pass for chains of the shape  [look-up stages]  [Gaussian noise]  [salt-and-pepper]
// (each part optional, in this order) -- the shape of the reference configurations' noise runs
// (test/configs/input.json: equalisation / quantisation, then noise, then a filter).  Known at
// launch, the shape needs no stage interpreter: the leading look-up stages arrive composed into ONE
// table (chain_prep_kernel), the pixel and random-draw loads of a lane's next two chunks are
// requested before the current two are looked up, and salt-and-pepper -- a property of the pixel
// INDEX that touches about 2 % of the pixels at the configurations' level 50 -- is not two mask
// planes read for every pixel but a sorted list of the noisy pixel indices: each wave overwrites the
// few listed bytes inside the 1024-pixel ranges it has just staged (LDS operations of one wave keep
// their order, so no barrier is needed between the two).
// ---------------------------------------------------------------------------------------
struct LgsDesc {
    int lut_slot;             // table slot of the composed look-up stage, -1: none
    int gauss_slot;           // table slot of the Gaussian amplitude, -1: none
    const uint8_t *rv;        // Gaussian draw per pixel index
    const uint32_t *sp_ptr;   // [N/16 + 1]: first list entry at or behind 16-pixel chunk i; null: no salt-and-pepper
    const uint32_t *sp_list;  // noisy pixel indices in ascending order, 2 * index + (1 = salt, 0 = pepper)
};

// low bytes of four sums -> one packed word
__device__ __forceinline__ uint32_t lgs_pack(uint32_t s0, uint32_t s1, uint32_t s2, uint32_t s3) {
    const uint32_t p01 = __builtin_amdgcn_perm(s1, s0, 0x0c0c0400u);
    const uint32_t p23 = __builtin_amdgcn_perm(s3, s2, 0x0c0c0400u);
    return __builtin_amdgcn_perm(p23, p01, 0x05040100u);
}

// look-up and amplitude add on four packed pixels: out byte k = (lut[v_k] + amp[r_k]) mod 256
template <bool LUT, bool GAUSS>
__device__ __forceinline__ uint32_t lgs_word(uint32_t v, uint32_t r, const uint8_t *lut, const uint8_t *amp) {
    if (LUT && GAUSS) {
        const uint32_t s0 = (uint32_t)lut[v & 0xffu] + amp[r & 0xffu];
        const uint32_t s1 = (uint32_t)lut[(v >> 8) & 0xffu] + amp[(r >> 8) & 0xffu];
        const uint32_t s2 = (uint32_t)lut[(v >> 16) & 0xffu] + amp[(r >> 16) & 0xffu];
        const uint32_t s3 = (uint32_t)lut[v >> 24] + amp[r >> 24];
        return lgs_pack(s0, s1, s2, s3);
    }
    if (LUT) return lookup4(lut, v);
    if (GAUSS) return addbytes(v, lookup4(amp, r));
    return v;
}

// The look-up + amplitude case: byte k of the result is (lut[v_k] + amp[r_k]) mod 256, added straight into its
// place by a sub-dword add (dst_sel:BYTE_k keeps the sum's low byte), so the eight bytes fetched from LDS need no
// packing.  A sub-dword write must not be followed at once by an instruction that reads the same register (one
// wait state on this part, which the compiler cannot insert into asm text): hence the s_nop after each.
__device__ __forceinline__ uint32_t lgs_word_sdwa(uint32_t v, uint32_t r, const uint8_t *lut, const uint8_t *amp) {
    const uint32_t e0 = lut[v & 0xffu], e1 = lut[(v >> 8) & 0xffu], e2 = lut[(v >> 16) & 0xffu], e3 = lut[v >> 24];
    const uint32_t a0 = amp[r & 0xffu], a1 = amp[(r >> 8) & 0xffu], a2 = amp[(r >> 16) & 0xffu], a3 = amp[r >> 24];
    uint32_t o;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
        "s_nop 0\n\t"
        "v_add_u32_sdwa %0, %3, %4 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "s_nop 0\n\t"
        "v_add_u32_sdwa %0, %5, %6 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "s_nop 0\n\t"
        "v_add_u32_sdwa %0, %7, %8 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "s_nop 0"
        : "=&v"(o)
        : "v"(e0), "v"(a0), "v"(e1), "v"(a1), "v"(e2), "v"(a2), "v"(e3), "v"(a3));
    return o;
}

template <bool LUT, bool GAUSS>
__device__ __forceinline__ u32x4 lgs_chunk(u32x4 v, u32x4 r, const uint8_t *lut, const uint8_t *amp, const uint8_t *nanf,
                                           bool any_nan) {
    u32x4 y;
    if (LUT && GAUSS) {
#pragma unroll
        for (int w = 0; w < 4; w++) y[w] = lgs_word_sdwa(v[w], r[w], lut, amp);
    } else {
#pragma unroll
        for (int w = 0; w < 4; w++) y[w] = lgs_word<LUT, GAUSS>(v[w], r[w], lut, amp);
    }
    if (GAUSS && any_nan) {  // degenerate variance: rare, uniform per image
#pragma unroll
        for (int w = 0; w < 4; w++) y[w] &= ~lookup4(nanf, r[w]);
    }
    return y;
}

// Per image, as chain_prep_kernel leaves them for this pass: the composed look-up table, the Gaussian
// amplitudes, the "amplitude was NaN" flags as 0x00 / 0xff bytes, and one word that is non-zero if any flag is set.
#define LGS_TAB_BYTES 1024
#define LGS_TAB_LUT 0
#define LGS_TAB_AMP 256
#define LGS_TAB_NAN 512
#define LGS_TAB_ANY 768

// Phase 1 of the pass: wave w stages the contiguous chunks [w * cpw, (w + 1) * cpw) of the tile, 64 at a time,
// with the loads of the next two sets of 64 in flight while one is looked up.  Offsets are 32-bit onto buffer
// descriptors that cover exactly the image: chunks outside it load zeros without a test (and are cleared after
// the look-up in the few workgroups that have any).
template <bool LUT, bool GAUSS>
__device__ __forceinline__ void lgs_stage_tile(__amdgpu_buffer_rsrc_t rs_px, __amdgpu_buffer_rsrc_t rs_rv, uint8_t *tile,
                                               int nchunks, int cpw, int fbase, int Ni, const uint32_t *__restrict__ sp_ptr,
                                               const uint32_t *__restrict__ sp_list, const uint8_t *lut, const uint8_t *amp,
                                               const uint8_t *nanf, bool any_nan) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int c_begin = wave * cpw;
    const int c_end = c_begin + cpw < nchunks ? c_begin + cpw : nchunks;
    const int K = (c_end - c_begin + 63) >> 6;  // sets of 64 chunks (uniform)
    const bool edge_tile = fbase < 0 || fbase + 16 * nchunks > Ni;  // uniform: the tile reaches past the image
    // salt-and-pepper: the list entries inside this wave's range of the image.  The two offsets are asked for
    // first and the entries themselves behind the first pixel loads, so that neither wait holds those up.
    uint32_t sp_lo = 0, sp_hi = 0, e0 = 0xffffffffu, e1 = 0xffffffffu;
    if (sp_ptr) {
        const int NC = Ni >> 4, cbase = fbase >> 4;  // fbase is a multiple of 16 (it may be negative)
        int a = cbase + c_begin, b = cbase + c_end;
        a = a < 0 ? 0 : (a > NC ? NC : a);
        b = b < 0 ? 0 : (b > NC ? NC : b);
        sp_lo = sp_ptr[a];
        sp_hi = sp_ptr[b];
    }
    struct Set {
        u32x4 v, r;
    };
    int voff = fbase + 16 * (c_begin + lane);  // flat offset of this lane's chunk of set 0; + 1024 per set
    auto request = [&](Set &x, int k) {
        // lanes past the wave's range ask for an offset outside every image: zeros, no traffic
        const int o = (c_begin + 64 * k + lane < c_end) ? voff + 1024 * k : -16;
        x.v = __builtin_amdgcn_raw_buffer_load_b128(rs_px, o, 0, IAK_NT_CHAIN_LD ? 2 : 0);
        if (GAUSS) x.r = __builtin_amdgcn_raw_buffer_load_b128(rs_rv, o, 0, 0);
    };
    auto stage = [&](const Set &x, int k) {
        u32x4 y = lgs_chunk<LUT, GAUSS>(x.v, x.r, lut, amp, nanf, any_nan);
        const int c = c_begin + 64 * k + lane;
        if (c < c_end) {
            *reinterpret_cast<u32x4 *>(tile + 16 * c) = y;
            if (edge_tile) {
                const int o = voff + 1024 * k;
                const u32x4 zero = {0, 0, 0, 0};
                if (o < 0 || o >= Ni) *reinterpret_cast<u32x4 *>(tile + 16 * c) = zero;  // outside the image: 0, not chain(0)
            }
        }
    };
    Set A, B;
    request(A, 0);
    if (K > 1) request(B, 1);
    if (sp_ptr) {
        if (sp_lo + lane < sp_hi) e0 = sp_list[sp_lo + lane];
        if (sp_lo + 64 + lane < sp_hi) e1 = sp_list[sp_lo + 64 + lane];
    }
    for (int k = 0; k < K; k += 2) {
        stage(A, k);
        if (k + 2 < K) request(A, k + 2);
        if (k + 1 < K) {
            stage(B, k + 1);
            if (k + 3 < K) request(B, k + 3);
        }
    }
    if (sp_ptr) {
        // the wave's own stores above come first: LDS operations of one wave keep their order
        if (e0 != 0xffffffffu) tile[(int)(e0 >> 1) - fbase] = (e0 & 1u) ? 255 : 0;
        if (e1 != 0xffffffffu) tile[(int)(e1 >> 1) - fbase] = (e1 & 1u) ? 255 : 0;
        for (uint32_t i = sp_lo + 128 + lane; i < sp_hi; i += 64) {  // denser noise: the rest of the range
            const uint32_t e = sp_list[i];
            tile[(int)(e >> 1) - fbase] = (e & 1u) ? 255 : 0;
        }
    }
}

template <class Op>
__global__ __launch_bounds__(CS_THREADS) __attribute__((amdgpu_waves_per_eu(Op::kManyRegisters ? 4 : 8))) void lgs_stencil8_flat_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                                int W, int H, long long in_stride, long long out_stride,
                                                                LgsDesc d, const uint32_t *__restrict__ sp_ptr,
                                                                const uint32_t *__restrict__ sp_list,
                                                                const uint8_t *__restrict__ lgs_tab, int TH, int cpw,
                                                                uint32_t cols_magic, Op op) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tile_bytes = (TH + 2) * W + 32;
    uint8_t *tile = smem;
    uint8_t *lut = smem + tile_bytes;  // tile_bytes is a multiple of 16
    uint8_t *amp = lut + 256;
    uint8_t *nanf = amp + 256;
    const int b = blockIdx.y;
    const uint8_t *tb = lgs_tab + (long long)b * LGS_TAB_BYTES;
    const uint8_t *src = in + (long long)b * in_stride;
    uint8_t *dst = out + (long long)b * out_stride;
    const int Ni = W * H;
    const int y0 = blockIdx.x * TH;
    const int T = blockDim.x;  // 512, or 256 for the operators that need many registers (see the launcher)
    const __amdgpu_buffer_rsrc_t rs_px = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, Ni, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_rv = __builtin_amdgcn_make_buffer_rsrc((void *)d.rv, 0, d.rv ? Ni : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)dst, 0, Ni, 0x00020000);

    // the image's tables: 64 (look-up) + 64 (amplitudes) dwords, the NaN flags only where there are any
    const bool any_nan = *reinterpret_cast<const uint32_t *>(tb + LGS_TAB_ANY) != 0;  // uniform
    {
        const int n = any_nan ? 192 : 128;
        if ((int)threadIdx.x < n)
            reinterpret_cast<uint32_t *>(lut)[threadIdx.x] = reinterpret_cast<const uint32_t *>(tb)[threadIdx.x];
    }
    __syncthreads();

    // phase 1: chunk c of the tile is the 16 pixels at flat offset fbase + 16 c
    const int nchunks = tile_bytes >> 4;
    const int fbase = (y0 - 1) * W - 16;
    if (d.lut_slot >= 0) {
        if (d.gauss_slot >= 0) lgs_stage_tile<true, true>(rs_px, rs_rv, tile, nchunks, cpw, fbase, Ni, sp_ptr, sp_list, lut, amp, nanf, any_nan);
        else lgs_stage_tile<true, false>(rs_px, rs_rv, tile, nchunks, cpw, fbase, Ni, sp_ptr, sp_list, lut, amp, nanf, any_nan);
    } else {
        if (d.gauss_slot >= 0) lgs_stage_tile<false, true>(rs_px, rs_rv, tile, nchunks, cpw, fbase, Ni, sp_ptr, sp_list, lut, amp, nanf, any_nan);
        else lgs_stage_tile<false, false>(rs_px, rs_rv, tile, nchunks, cpw, fbase, Ni, sp_ptr, sp_list, lut, amp, nanf, any_nan);
    }
    __syncthreads();

    // phase 2: the stencil on 8-pixel chunks x groups of 8 rows, rows coming from the tile.  Stores go through a
    // descriptor of the output image, which drops the rows below the image's last one.
    const int cols = W >> 3;
    const int items = cols * (TH >> 3);
    for (int it = threadIdx.x; it < items; it += T) {
        const int rg = (int)__umulhi((uint32_t)it, cols_magic);  // it / cols (cols_magic = ceil(2^32 / cols), it < 2^16)
        const int cx = it - (int)__umul24(rg, cols);
        const int col0 = cx << 3;
        const int yb = y0 + (rg << 3);
        if (yb >= H) continue;
        const uint8_t *base = tile + 16 + (int)__umul24(rg << 3, W) + (col0 - 4);  // column col0-4 of image row yb-1
        Raw16 raw[10];
#pragma unroll
        for (int r = 0; r < 10; r++) {
            const uint32_t *q = reinterpret_cast<const uint32_t *>(base);
            raw[r].d[0] = q[0]; raw[r].d[1] = q[1]; raw[r].d[2] = q[2]; raw[r].d[3] = q[3];
            base += W;
        }
        const uint32_t first_col = col0 == 0 ? 0xffffff00u : 0xffffffffu;
        int ooff = (int)__umul24(yb, W) + col0;
        typename Op::Row up, mid, dn;
        op.prepare(raw[0], yb - 1, col0, W, H, up);
        op.prepare(raw[1], yb, col0, W, H, mid);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int row = yb + r;
            op.prepare(raw[r + 2], row + 1, col0, W, H, dn);
            uint32_t o[2];
            op.compute(up, mid, dn, o);
            if (Op::kBorderZero) {
                if (r == 0 && row < 1) o[0] = o[1] = 0;
                o[0] &= first_col;
            }
            const u32x2 ov = {o[0], o[1]};
            __builtin_amdgcn_raw_buffer_store_b64(ov, rs_out, ooff, 0, IAK_NT_STENCIL_ST ? 2 : 0);
            ooff += W;
            up = mid;
            mid = dn;
        }
    }
}

struct TileGeom {
    int TW, TH, tiles_x, tiles_y;
    size_t lds;
    bool flat;
};

// Tile: as wide as the image up to 1024 columns; height 8k rows with the tile under 40 KiB
// (four workgroups per CU) and k picked so that (TW/8)*k work items fill the workgroup's lanes evenly.
static TileGeom tile_geometry(int W, int H) {
    TileGeom g;
    g.TW = W <= 1024 ? W : 1024;
    g.tiles_x = (W + g.TW - 1) / g.TW;
    g.flat = g.tiles_x == 1;
    const int pitch = g.flat ? g.TW : g.TW + 32;
    const int extra = g.flat ? 32 : 0;
    int kmax = (((40960 - extra) / pitch) - 2) / 8;
    if (kmax < 1) kmax = 1;
    if (kmax > 8) kmax = 8;
    int best = kmax;
    double best_eff = -1;
    for (int k = kmax; k >= 1 && k >= kmax - 3; k--) {
        const int items = (g.TW / 8) * k;
        const double eff = (double)items / ((double)CS_THREADS * ((items + CS_THREADS - 1) / CS_THREADS));
        if (eff > best_eff + 1e-9) { best_eff = eff; best = k; }
    }
    g.TH = 8 * best;
    g.tiles_y = (H + g.TH - 1) / g.TH;
    g.lds = (size_t)(g.TH + 2) * pitch + extra + sizeof(ChainTables);
    return g;
}

// in: B images at stride in_stride, 16-byte aligned; W % 16 == 0; out 8-byte aligned.
// tables: [B][CHAIN_MAX][256] as chain_prep_kernel lays them out.
template <class Op>
static int launch_chain_stencil(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride,
                                long long out_stride, const ChainDesc &d, const uint16_t *tables, const Op &op,
                                hipStream_t s) {
    const TileGeom g = tile_geometry(W, H);
    // An operator that needs more than 64 registers per lane leaves room for one 512-lane workgroup
    // per CU, whose two phases then run one after the other with nothing to overlap them; as 256-lane
    // workgroups the same number of waves belongs to four workgroups in different phases.
    if (g.flat)
        hipLaunchKernelGGL((chain_stencil8_flat_kernel<Op>), dim3(g.tiles_y, B), dim3(Op::kManyRegisters ? 256 : CS_THREADS),
                           g.lds, s, in, out, W, H, in_stride, out_stride, d, tables, g.TH, op);
    else
        hipLaunchKernelGGL((chain_stencil8_kernel<Op>), dim3(g.tiles_x * g.tiles_y, B), dim3(CS_THREADS), g.lds, s, in, out, W, H,
                           in_stride, out_stride, d, tables, g.TW, g.TH, g.tiles_x, op);
    IAK_LAUNCH_CHECK();
    return 0;
}

// The chain as an LgsDesc, if it has that shape: look-up stages (composed by the table prologue into
// the slot of the last one: ChainDesc::compose_n), then at most one Gaussian stage, then at most one
// salt-and-pepper stage that has its sparse list.
static bool lgs_from_chain(const ChainDesc &c, LgsDesc *d) {
    int i = 0;
    while (i < c.n && c.st[i].type != C_GAUSS && c.st[i].type != C_SP) i++;
    d->lut_slot = i > 0 ? i - 1 : -1;
    d->gauss_slot = -1;
    d->rv = nullptr;
    d->sp_ptr = d->sp_list = nullptr;
    if (i < c.n && c.st[i].type == C_GAUSS) {
        d->gauss_slot = i;
        d->rv = c.st[i].plane;
        i++;
    }
    if (i < c.n && c.st[i].type == C_SP && c.st[i].sp_ptr) {
        d->sp_ptr = c.st[i].sp_ptr;
        d->sp_list = c.st[i].sp_list;
        i++;
    }
    return i == c.n && c.n > 0;
}

// in as launch_chain_stencil; lgs_tab: [B][LGS_TAB_BYTES] as chain_prep_kernel lays it out; flat geometry only
// (one tile across the image)
template <class Op>
static int launch_lgs_stencil(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
                              const LgsDesc &d, const uint8_t *lgs_tab, const Op &op, hipStream_t s) {
    const TileGeom g = tile_geometry(W, H);
    if (!g.flat) return IAK_E_UNSUPPORTED;
    const int T = Op::kManyRegisters ? 256 : CS_THREADS;
    const int nchunks = ((g.TH + 2) * W + 32) >> 4;
    const int cpw = (nchunks + T / 64 - 1) / (T / 64);  // chunks per wave
    const int cols = W >> 3;
    if (cols * (g.TH >> 3) >= 65536) return IAK_E_UNSUPPORTED;  // the kernel's division by multiplication
    const uint32_t cols_magic = (uint32_t)(((1ULL << 32) + cols - 1) / cols);
    hipLaunchKernelGGL((lgs_stencil8_flat_kernel<Op>), dim3(g.tiles_y, B), dim3(T), g.lds, s, in, out, W, H, in_stride,
                       out_stride, d, d.sp_ptr, d.sp_list, lgs_tab, g.TH, cpw, cols_magic, op);
    IAK_LAUNCH_CHECK();
    return 0;
}

}  // namespace iak
