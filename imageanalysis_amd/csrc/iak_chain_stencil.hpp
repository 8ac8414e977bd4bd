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


}  // namespace iak
