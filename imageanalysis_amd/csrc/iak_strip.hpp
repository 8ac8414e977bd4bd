// A run of up to three 3x3 stencils (optionally behind a chain of per-pixel operations) in ONE
// pass: a workgroup owns a strip of TH full-width rows, stages the strip plus S halo rows on
// either side in LDS (chain applied on the way in), and runs the stencils one after the other
// between two LDS buffers; only the last one stores to memory.  Each intermediate image is
// neither written nor read back.
//
// Everything is on the flat view of the image: the LDS buffers hold whole rows back to back
// (pitch = W), so a tap one past a row's end is the next row's first pixel exactly as in
// memory; rows outside the image are zeros, which is what a tap past the end reads.  A stage's
// output rows are its input rows minus one on either side, so after S stages the strip's own
// TH rows remain.  Two reads fall just outside a stage's input rows.  The bytes before the
// first row only feed the column -1 taps of column-0 outputs, which every operator here sets
// to zero.  The byte after the last row is the column W tap of the last column of the last
// output row, i.e. column 0 of the row after the buffer: for the first stage it is staged
// from memory with the rest; for later stages it is the previous stencil's column-0 output,
// which is zero, so a zeroed guard word is exact.  Dilation breaks both arguments (it writes
// column 0 and it scatters across the row end), so the planner keeps it out of these runs.
#pragma once
#include "iak_chain.hpp"
#include "iak_stencil_ops.hpp"

#include <cstring>

namespace iak {

enum { SK_SWAR, SK_INT, SK_INT_AVG, SK_FIXED, SK_FIXED_AVG, SK_FLOAT, SK_FLOAT_AVG, SK_MEDIAN, SK_SOBEL, SK_COMPASS,
       SK_ERODE };
#define SK_MAX_STAGES 3
#define SK_THREADS 256
#define SK_LDS_BUDGET 40960

struct StripOp {
    int kind;
    int q, sa, sb, sc;
    uint32_t mask;
    int ik[9];
    uint32_t xk[9];
    float fk[9];
};
struct StripDesc {
    int S;   // stencils in the run
    int TH;  // rows per strip
    int n;   // output rows per work item
    StripOp op[SK_MAX_STAGES];
};

// One stencil over the rows a stage owes: in_rows points at the first input row (rb_in) of the
// LDS buffer; outputs are rows rb_in+1 .. rb_in+R, written to out_rows (LDS, first output row)
// or, for the last stage, to the image in memory.
template <class Op>
__device__ __forceinline__ void strip_stage(const Op &op, const uint8_t *in_rows, int rb_in, uint8_t *out_rows,
                                            uint8_t *gdst, int R, int n, int W, int H) {
    const int cols = W >> 3;
    const int G = (R + n - 1) / n;
    const int items = cols * G;
    const int rb_out = rb_in + 1;
    if (!gdst && threadIdx.x < 8) {  // the guard words on either side of the rows produced here
        uint32_t *guard = reinterpret_cast<uint32_t *>(threadIdx.x < 4 ? out_rows - 16 : out_rows + R * W);
        guard[threadIdx.x & 3] = 0;
    }
    for (int it = threadIdx.x; it < items; it += SK_THREADS) {
        const int g = it / cols, cx = it - g * cols;
        const int col0 = cx << 3;
        int r0 = g * n;
        if (r0 + n > R) r0 = R - n;  // the last group overlaps its neighbour: same values written twice
        const int yb = rb_out + r0;
        const uint8_t *base = in_rows + r0 * W + col0 - 4;  // row yb-1, column col0-4
        Raw16 raw[10];
#pragma unroll
        for (int r = 0; r < 10; r++) {
            if (r < n + 2) {
                const uint32_t *q = reinterpret_cast<const uint32_t *>(base + r * W);
                raw[r].d[0] = q[0]; raw[r].d[1] = q[1]; raw[r].d[2] = q[2]; raw[r].d[3] = q[3];
            }
        }
        typename Op::Row up, mid, dn;
        op.prepare(raw[0], yb - 1, col0, W, H, up);
        op.prepare(raw[1], yb, col0, W, H, mid);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (r < n) {
                const int row = yb + r;
                op.prepare(raw[r + 2], row + 1, col0, W, H, dn);
                uint32_t o[2];
                op.compute(up, mid, dn, o);
                if (Op::kBorderZero) {
                    if (row < 1) o[0] = o[1] = 0;
                    if (col0 == 0) o[0] &= 0xffffff00u;
                }
                const bool inside = row >= 0 && row < H;
                if (gdst) {
                    if (inside) st8<IAK_NT_STENCIL_ST>(gdst + (long long)row * W + col0, o[0], o[1]);
                } else {
                    if (!inside) o[0] = o[1] = 0;  // rows outside the image read as zeros downstream
                    *reinterpret_cast<uint2 *>(out_rows + r0 * W + r * W + col0) = make_uint2(o[0], o[1]);
                }
                up = mid;
                mid = dn;
            }
        }
    }
}

__device__ __forceinline__ void strip_dispatch(const StripOp &so, const uint8_t *in_rows, int rb_in, uint8_t *out_rows,
                                               uint8_t *gdst, int R, int n, int W, int H) {
    switch (so.kind) {
    case SK_SWAR: {
        Lin8Swar op;
        op.sa = so.sa; op.sb = so.sb; op.sc = so.sc;
        strip_stage(op, in_rows, rb_in, out_rows, gdst, R, n, W, H);
        break;
    }
    case SK_INT: {
        Lin8Int<false> op;
        for (int i = 0; i < 9; i++) op.k[i] = so.ik[i];
        strip_stage(op, in_rows, rb_in, out_rows, gdst, R, n, W, H);
        break;
    }
    case SK_INT_AVG: {
        Lin8Int<true> op;
        for (int i = 0; i < 9; i++) op.k[i] = so.ik[i];
        strip_stage(op, in_rows, rb_in, out_rows, gdst, R, n, W, H);
        break;
    }
    case SK_FIXED: {
        Lin8Fixed<false> op;
        for (int i = 0; i < 9; i++) op.k[i] = so.xk[i];
        op.q = so.q;
        strip_stage(op, in_rows, rb_in, out_rows, gdst, R, n, W, H);
        break;
    }
    case SK_FIXED_AVG: {
        Lin8Fixed<true> op;
        for (int i = 0; i < 9; i++) op.k[i] = so.xk[i];
        op.q = so.q;
        strip_stage(op, in_rows, rb_in, out_rows, gdst, R, n, W, H);
        break;
    }
    case SK_FLOAT: {
        Lin8Float<false> op;
        for (int i = 0; i < 9; i++) op.k[i] = so.fk[i];
        strip_stage(op, in_rows, rb_in, out_rows, gdst, R, n, W, H);
        break;
    }
    case SK_FLOAT_AVG: {
        Lin8Float<true> op;
        for (int i = 0; i < 9; i++) op.k[i] = so.fk[i];
        strip_stage(op, in_rows, rb_in, out_rows, gdst, R, n, W, H);
        break;
    }
    case SK_MEDIAN: strip_stage(Median8(), in_rows, rb_in, out_rows, gdst, R, n, W, H); break;
    case SK_SOBEL: strip_stage(Sobel8(), in_rows, rb_in, out_rows, gdst, R, n, W, H); break;
    case SK_COMPASS: strip_stage(Compass8(), in_rows, rb_in, out_rows, gdst, R, n, W, H); break;
    case SK_ERODE: {
        Erode8 op;
        op.mask = so.mask;
        strip_stage(op, in_rows, rb_in, out_rows, gdst, R, n, W, H);
        break;
    }
    }
}

__global__ __launch_bounds__(SK_THREADS) void strip_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int W,
                                                           int H, long long in_stride, long long out_stride, ChainDesc d,
                                                           const uint16_t *__restrict__ tables, StripDesc sd) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int S = sd.S, TH = sd.TH;
    const int sizeA = (TH + 2 * S) * W + 32, sizeB = (S > 1) ? (TH + 2 * S - 2) * W + 32 : 0;
    uint8_t *bufA = smem, *bufB = smem + sizeA;
    ChainTables &t = *reinterpret_cast<ChainTables *>(smem + sizeA + sizeB);
    const int b = blockIdx.y;
    const int y0 = blockIdx.x * TH;
    const uint8_t *src = in + (long long)b * in_stride;
    uint8_t *dst = out + (long long)b * out_stride;
    const long long N = (long long)W * H;
    if (d.n) chain_load_tables(t, d, tables + (long long)b * CHAIN_MAX * 256);
    const uint32_t nan_bits = d.n ? chain_nan_bits(t) : 0u;

    // stage the strip: rows y0-S .. y0+TH+S-1 plus 16 bytes on either side, chain applied
    const long long flat0 = (long long)(y0 - S) * W - 16;
    const int nchunks = sizeA >> 4;
    for (int c = threadIdx.x; c < nchunks; c += 2 * SK_THREADS) {
        const int c1 = c + SK_THREADS;
        const long long f0 = flat0 + 16LL * c, f1 = flat0 + 16LL * c1;
        const bool ok0 = f0 >= 0 && f0 < N;  // W % 16 == 0: a chunk is wholly inside or outside
        const bool ok1 = c1 < nchunks && f1 >= 0 && f1 < N;
        const long long g0 = ok0 ? f0 : 0, g1 = ok1 ? f1 : 0;
        u32x4 v0 = ld16<IAK_NT_CHAIN_LD>(src + g0);
        u32x4 v1 = ld16<IAK_NT_CHAIN_LD>(src + g1);
        if (d.n) {
            v0 = chain_apply16(v0, g0, d, t, nan_bits);
            v1 = chain_apply16(v1, g1, d, t, nan_bits);
        }
        const u32x4 zero = {0, 0, 0, 0};
        *reinterpret_cast<u32x4 *>(bufA + 16 * c) = ok0 ? v0 : zero;
        if (c1 < nchunks) *reinterpret_cast<u32x4 *>(bufA + 16 * c1) = ok1 ? v1 : zero;
    }
    for (int s = 0; s < S; s++) {
        __syncthreads();
        const uint8_t *inb = ((s & 1) ? bufB : bufA) + 16;
        uint8_t *outb = ((s & 1) ? bufA : bufB) + 16;
        const int rb_in = y0 - (S - s);
        const int R = TH + 2 * (S - 1 - s);
        const bool last = s == S - 1;
        strip_dispatch(sd.op[s], inb, rb_in, last ? nullptr : outb, last ? dst : nullptr, R, sd.n, W, H);
    }
}

// rows per strip for a run of S stencils on images W wide; 0 when a useful strip does not fit
static int strip_rows(int W, int S) {
    if (W % 16 != 0 || S < 1 || S > SK_MAX_STAGES) return 0;
    const long long fixed = (long long)(4 * S - 2) * W + 64 + (long long)sizeof(ChainTables);
    const long long room = SK_LDS_BUDGET - fixed;
    if (room <= 0) return 0;
    int th = (int)(room / (2LL * W));
    th &= ~3;
    if (th > 32) th = 32;
    return th >= 8 ? th : 0;
}

static size_t strip_lds_bytes(int W, const StripDesc &sd) {
    const size_t sizeA = (size_t)(sd.TH + 2 * sd.S) * W + 32;
    const size_t sizeB = sd.S > 1 ? (size_t)(sd.TH + 2 * sd.S - 2) * W + 32 : 0;
    return sizeA + sizeB + sizeof(ChainTables);
}

static int launch_strip(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
                        const ChainDesc &d, const uint16_t *tables, const StripDesc &sd, hipStream_t s) {
    const int strips = (H + sd.TH - 1) / sd.TH;
    hipLaunchKernelGGL(strip_kernel, dim3((unsigned)strips, B), dim3(SK_THREADS), strip_lds_bytes(W, sd), s, in, out, W, H,
                       in_stride, out_stride, d, tables, sd);
    IAK_LAUNCH_CHECK();
    return 0;
}

// host: the strip form of a 3x3 operation
static bool strip_op_from(const iak_op &o, StripOp *so) {
    std::memset(so, 0, sizeof *so);
    switch (o.type) {
    case IAK_OP_LINEAR: case IAK_OP_AVERAGE: {
        if (o.kW != 3 || o.kH != 3) return false;
        const bool avg = o.type == IAK_OP_AVERAGE;
        LinPlan p = plan_linear3(o.fkernel);
        if (!avg && p.kind != LIN_INT && plan_swar(o.fkernel, p)) {
            so->kind = SK_SWAR;
            so->sa = p.sa; so->sb = p.sb; so->sc = p.sc;
        } else if (p.kind == LIN_INT) {
            so->kind = avg ? SK_INT_AVG : SK_INT;
            for (int i = 0; i < 9; i++) so->ik[i] = p.ik[i];
        } else if (p.kind == LIN_FIXED) {
            so->kind = avg ? SK_FIXED_AVG : SK_FIXED;
            for (int i = 0; i < 9; i++) so->xk[i] = p.fk[i];
            so->q = p.q;
        } else {
            so->kind = avg ? SK_FLOAT_AVG : SK_FLOAT;
            for (int i = 0; i < 9; i++) so->fk[i] = o.fkernel[i];
        }
        return true;
    }
    case IAK_OP_MEDIAN:
        if (o.kW != 3 || o.kH != 3) return false;
        for (int i = 0; i < 9; i++)
            if (o.ikernel[i] != 1) return false;
        so->kind = SK_MEDIAN;
        return true;
    case IAK_OP_SOBEL: so->kind = SK_SOBEL; return true;
    case IAK_OP_COMPASS: so->kind = SK_COMPASS; return true;
    case IAK_OP_ERODE:
        if (o.kW != 3 || o.kH != 3) return false;
        so->kind = SK_ERODE;
        so->mask = se_mask3(o.ikernel);
        return true;
    default: return false;
    }
}

}  // namespace iak
