// Batched pipeline executor: the device-resident replacement of the per-image, per-operation
// loop in the reference's driver (test/test.cu:266-564).  A plan is made once
// (iak_pipeline_create); a run enqueues a fixed list of launch groups on one stream and
// touches no allocator and no host synchronisation.
//
// Fused mode merges what the operation list allows without changing any result:
//   * grayscale + per-image histogram are one pass;
//   * a run of per-pixel operations (equalisation, quantisation, thresholds, Gaussian noise,
//     salt-and-pepper) is one read-modify-write pass.  The 256-entry work between them
//     (cdf map, Otsu threshold, mean/std, amplitude table) is done by a one-workgroup-per-
//     image prologue; a histogram needed after a table look-up is obtained by pushing the
//     previous histogram through the table, not by another pass over the image.
#include "iak_common.hpp"
#include "iak_device_fns.hpp"
#include "iak_block.hpp"
#include "iak_chain.hpp"
#include "iak_hist.hpp"
#include "iak_chain_stencil.hpp"
#include "iak_binary_stencil.hpp"
#include "iak_stencil_ops.hpp"

#include <new>
#include <atomic>
#include <string>
#include <vector>

namespace iak {

// dataset histogram: sum[v] += sum_b hist[b][v]; workgroup `block` folds images 8 block .. 8 block + 7
#define ACC_IMAGES 8
__device__ __forceinline__ void hist_accumulate_block(const uint32_t *__restrict__ hist, int B,
                                                      unsigned long long *__restrict__ sum, int block) {
    const int v = threadIdx.x;
    const int b0 = block * ACC_IMAGES;
    unsigned long long s = 0;
#pragma unroll
    for (int i = 0; i < ACC_IMAGES; i++)
        if (b0 + i < B) s += hist[(long long)(b0 + i) * 256 + v];
    if (s) atomicAdd(&sum[v], s);
}

// Per image (one workgroup of 256 lanes, lane v = grey value v): walk the chain once on the
// histogram, producing every stage's table.  tables: [B][CHAIN_MAX][256] uint16 (LUT stages
// use the low byte).  Everything that is order-free is done across lanes; the two float
// accumulations whose order the reference fixes (the cdf, the variance sum) are added up by
// one lane in bin order from per-lane terms.
// hist_out (may be null, may be hist_in): the histogram of the run's OUTPUT, for runs made of
// look-up stages only -- it is the input's histogram pushed through the tables, so the consumer
// behind the run needs no counting pass.
// Workgroups past the last image (acc_sum != nullptr) fold the data-set histogram instead, eight
// images each: a launch of their own would cost more than the work.
__global__ __launch_bounds__(256) void chain_prep_kernel(const uint32_t *hist_in, int numPixels, ChainDesc d,
                                                         uint16_t *__restrict__ tables, uint32_t *hist_out = nullptr,
                                                         int n_images = 0x7fffffff, const uint32_t *acc_hist = nullptr,
                                                         unsigned long long *acc_sum = nullptr, uint8_t *lgs_tab = nullptr) {
    if ((int)blockIdx.x >= n_images) {  // uniform over the workgroup
        hist_accumulate_block(acc_hist, n_images, acc_sum, (int)blockIdx.x - n_images);
        return;
    }
    __shared__ uint32_t h[256];  // current histogram, bin 0 empty
    __shared__ uint32_t h2[256];
    __shared__ BlockScratch bs;
    float *pdf = bs.fbuf;
    long long *red = bs.red;
    __shared__ float ms[2];
    __shared__ uint16_t stage_lut[256];
    const int b = blockIdx.x;
    const int v = threadIdx.x;
    uint32_t composed = (uint32_t)v;  // value v after the look-up stages so far (d.compose_n > 0)
    h[v] = hist_in ? hist_in[(long long)b * 256 + v] : 0u;
    __syncthreads();
    uint16_t *tb = tables + (long long)b * CHAIN_MAX * 256;
    // lgs_tab: the byte tables of lgs_stencil8_flat_kernel as well (composed look-up, amplitudes, NaN flags)
    uint8_t *lt = lgs_tab ? lgs_tab + (long long)b * LGS_TAB_BYTES : nullptr;
    if (lt && v == 0) *reinterpret_cast<uint32_t *>(lt + LGS_TAB_ANY) = 0u;
    for (int s = 0; s < d.n; s++) {
        const ChainStage st = d.st[s];
        if (st.type == C_SP) continue;
        const uint32_t hv = (v == 0) ? 0u : h[v];
        if (st.type == C_GAUSS) {
            if (st.mean == -1 || st.stddev == -1) {
                block_mean_std(hv, numPixels, bs);
                if (v == 0) {
                    ms[0] = bs.ms[0];
                    ms[1] = bs.ms[1] * bs.ms[1];  // imageAnalysis.cu:861 passes stdDev * stdDev
                }
            } else if (v == 0) {
                ms[0] = st.mean;
                ms[1] = st.stddev * st.stddev;
            }
            __syncthreads();
            const uint16_t entry = gauss_entry(v, ms[0], ms[1]);
            tb[s * 256 + v] = entry;
            if (lt) {
                lt[LGS_TAB_AMP + v] = (uint8_t)entry;
                lt[LGS_TAB_NAN + v] = (entry & 0x100u) ? 0xff : 0;
                const int any = __syncthreads_or(entry & 0x100u);
                if (v == 0) *reinterpret_cast<uint32_t *>(lt + LGS_TAB_ANY) = any ? 1u : 0u;
            }
            __syncthreads();
            continue;  // position-dependent from here on: the planner never asks for h again
        }
        // look-up stages: build the table, then push the histogram through it
        uint32_t lut;
        if (st.type == C_EQUALIZE) {
            // equalizeHistogram (imageAnalysis.cu:155-176): divisions in parallel, the float32
            // running sum strictly in bin order
            pdf[v] = (float)(int)hv / (float)numPixels;
            __syncthreads();
            // the running sum goes to a second array: with no store into pdf[] the 256 LDS reads are
            // independent of one another and pipeline, leaving only the chain of additions serial
            float *cdf_out = reinterpret_cast<float *>(bs.ibuf);
            if (v == 0) {
                float cdf = 0.0f;
#pragma unroll 16
                for (int i = 0; i < 256; i++) {
                    cdf = (i > 0) ? (pdf[i] + cdf) : pdf[i];
                    cdf_out[i] = cdf;
                }
            }
            __syncthreads();
            lut = (uint32_t)(int)(cdf_out[v] * 255) & 0xffu;
        } else if (st.type == C_OTSU) {
            block_otsu(hv, numPixels, bs);
            lut = (v > bs.thr) ? 255u : 0u;
        } else if (st.type == C_KMEANS) {
            // kMeansThresholding (imageAnalysis.cu:281-379) on the histogram; lane v = value v.
            // Label 1 = not nearer to centroid 1 (:298-302); class sums wrap as unsigned int.
            const long long nz = block_sum_ll((long long)hv, red);
            const uint32_t cnt = (v == 0) ? (uint32_t)((long long)numPixels - nz) : hv;
            const int32_t *ini = reinterpret_cast<const int32_t *>(st.plane) + 2 * (long long)b;
            int c0 = (uint8_t)ini[0], c1 = (uint8_t)ini[1];
            int rounds = 0;
            bool changed, one;
            do {
                one = !(abs(c0 - v) > abs(c1 - v));
                const uint32_t s0 = (uint32_t)block_sum_ll(one ? 0 : (long long)((uint32_t)v * cnt), red);
                const uint32_t s1 = (uint32_t)block_sum_ll(one ? (long long)((uint32_t)v * cnt) : 0, red);
                const int n0 = (int)block_sum_ll(one ? 0 : (long long)cnt, red);
                const int n1 = (int)block_sum_ll(one ? (long long)cnt : 0, red);
                const int m0 = (int)f2u8(floorf((float)s0 / (float)n0));  // :332; empty class: NaN -> 0
                const int m1 = (int)f2u8(floorf((float)s1 / (float)n1));
                rounds++;
                changed = (c0 != m0) || (c1 != m1);
                c0 = m0;
                c1 = m1;
            } while (changed && rounds < 100);
            lut = one ? 255u : 0u;
        } else {
            lut = d.const_lut[st.lut_slot][v];
        }
        if (s < d.compose_n) {
            // the run's look-up stages as one table: stage s applied to what the stages before made of v
            stage_lut[v] = (uint16_t)lut;
            __syncthreads();
            composed = stage_lut[composed];
            tb[s * 256 + v] = (uint16_t)(s == d.compose_n - 1 ? composed : lut);
            if (lt && s == d.compose_n - 1) lt[LGS_TAB_LUT + v] = (uint8_t)composed;
        } else {
            tb[s * 256 + v] = (uint16_t)lut;
        }
        h2[v] = 0;
        // zeros are not in h: bin 0 holds numPixels - sum(others)
        const long long nonzero = block_sum_ll((long long)hv, red);
        const uint32_t cnt = (v == 0) ? (uint32_t)((long long)numPixels - nonzero) : hv;
        __syncthreads();
        if (cnt) atomicAdd(&h2[lut], cnt);
        __syncthreads();
        h[v] = (v == 0) ? 0u : h2[v];
        __syncthreads();
    }
    if (hist_out) hist_out[(long long)b * 256 + v] = h[v];
}

// salt-and-pepper code plane (0 keep, 1 pepper, 2 salt) -> the two byte masks of (pixel & keep) | salt
__global__ __launch_bounds__(256) void sp_masks_kernel(const uint8_t *__restrict__ code, uint8_t *__restrict__ keep,
                                                       uint8_t *__restrict__ salt, long long N) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const uint8_t c = code[i];
    keep[i] = c == 0 ? 0xff : 0x00;
    salt[i] = c == 2 ? 0xff : 0x00;
}

// HIST: also accumulates the histogram of the output (hist[b][256], zeroed by the launcher, bin 0
// left empty), for the consumer behind this run.
// (with the histogram the workgroup is 1024 lanes: its 32 KiB table would otherwise cut the CU down to
// 16 waves)
template <bool HIST = false>
__global__ __launch_bounds__(HIST ? 1024 : 256) void chain_apply_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                          long long N, long long in_stride, long long out_stride,
                                                          ChainDesc d, const uint16_t *__restrict__ tables,
                                                          uint32_t *__restrict__ hist = nullptr) {
    __shared__ ChainTables t;
    __shared__ uint32_t sh[HIST ? 256 * HIST_COLS : 1];
    const int hcol = threadIdx.x & (HIST_COLS - 1);
    if (HIST) hist_zero(sh);
    const int b = blockIdx.y;
    chain_load_tables(t, d, tables + (long long)b * CHAIN_MAX * 256);
    const uint32_t nan_bits = chain_nan_bits(t);
    const uint8_t *src = in + (long long)b * in_stride;
    uint8_t *dst = out + (long long)b * out_stride;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nthreads = (long long)gridDim.x * blockDim.x;
    {
        // two chunks per iteration: their loads are issued together, so a wave always has two
        // independent request / look-up chains in flight
        const long long nchunks = N >> 4;
        for (long long c = tid; c < nchunks; c += 2 * nthreads) {
            const long long off0 = c << 4;
            const bool two = c + nthreads < nchunks;
            const long long off1 = two ? ((c + nthreads) << 4) : off0;
            u32x4 y0 = ld16<IAK_NT_CHAIN_LD>(src + off0);
            u32x4 y1 = ld16<IAK_NT_CHAIN_LD>(src + off1);
            chain_apply16x2(y0, y1, off0, off1, d, t, nan_bits);
            st16<IAK_NT_CHAIN_ST>(dst + off0, y0);
            if (two) st16<IAK_NT_CHAIN_ST>(dst + off1, y1);
            if (HIST) {
#pragma unroll
                for (int w = 0; w < 4; w++) hist_add4(sh, y0[w], hcol);
                if (two) {
#pragma unroll
                    for (int w = 0; w < 4; w++) hist_add4(sh, y1[w], hcol);
                }
            }
        }
    }
    const long long scalar_from = (N >> 4) << 4;
    for (long long x = scalar_from + tid; x < N; x += nthreads) {
        const uint32_t y = chain_apply1(src[x], x, d, t) & 0xffu;
        dst[x] = (uint8_t)y;
        if (HIST) atomicAdd(&sh[y * HIST_COLS + hcol], 1u);
    }
    if (HIST) hist_flush(sh, hist + (long long)b * 256);
}

__global__ __launch_bounds__(256) void hist_accumulate_kernel(const uint32_t *__restrict__ hist, int B,
                                                              unsigned long long *__restrict__ sum) {
    hist_accumulate_block(hist, B, sum, blockIdx.x);
}

}  // namespace iak

namespace iak {
int stencil3_with_hist(const iak_op &o, const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride,
                       long long out_stride, uint32_t *hist, void *stream, bool hist_cleared, bool fused);  // iak_stencil.hip
int linear_with_contraction(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
                            const float *kernel, int kW, int kH, int mode, bool fused, void *stream);  // iak_stencil.hip
int morph_chain3(const iak_op *ops, int n, const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride,
                 long long out_stride, uint32_t *hist, const uint16_t *tables, void *stream, bool hist_cleared);  // iak_stencil.hip
// iak_gray_hist / iak_channel_hist (method 2) into a histogram the caller has cleared already
int gray_hist_cleared(const uint8_t *rgb, uint8_t *gray, uint32_t *hist, int W, int H, int B, long long rgb_stride,
                      long long gray_stride, int method, void *stream);  // iak_pointwise.hip
}

using namespace iak;

// ---- plan ---------------------------------------------------------------------------------
namespace {

struct OpCopy {
    iak_op op;
    int kmeans_index = -1;  // IAK_OP_KMEANS: its number among the pipeline's k-means operations
    std::vector<float> f;
    std::vector<int> i;
};

enum GroupKind { G_GRAYHIST, G_HIST, G_CHAIN, G_OP, G_CHAIN_OP, G_FEATURES, G_MORPH };

struct Group {
    GroupKind kind;
    std::string name;
    long long alg_bytes;  // algorithmic bytes per pixel of the operations in the group
    int op_index = -1;    // G_OP; G_MORPH: the first operation of the run
    int n_ops = 1;        // G_MORPH: operations in the run
    ChainDesc chain;      // G_CHAIN, G_CHAIN_OP; G_MORPH (n == 0: no threshold in front)
    bool chain_needs_hist = false;
    bool emit_hist = false;  // G_CHAIN, G_OP: the pass also leaves the histogram of its output
    bool hist_from_tables = false;  // G_CHAIN of look-up stages only: that histogram comes from the table prologue
    bool hist_tracked = false;      // G_CHAIN: the prologue has the input's histogram and every stage carries it on
    bool binary = false;     // G_CHAIN_OP: the run is one threshold, the stencil is done by table (bt)
    BinTable bt;
    bool lgs = false;        // G_CHAIN_OP: the run has the look-ups / Gaussian / salt-and-pepper shape (lgsd)
    LgsDesc lgsd;
};

// operators the fused chain+stencil kernel can take over (the 8-pixel 3x3 family)
bool is_fusable_stencil(const iak_op &o) {
    switch (o.type) {
    case IAK_OP_LINEAR: case IAK_OP_AVERAGE: case IAK_OP_DILATE: case IAK_OP_ERODE: return o.kW == 3 && o.kH == 3;
    case IAK_OP_MEDIAN: {
        if (o.kW != 3 || o.kH != 3) return false;
        for (int i = 0; i < 9; i++)
            if (o.ikernel[i] != 1) return false;
        return true;
    }
    case IAK_OP_SOBEL: case IAK_OP_COMPASS: return true;
    default: return false;
    }
}

bool is_pointwise(int t) {
    return t == IAK_OP_EQUALIZE || t == IAK_OP_GAUSS || t == IAK_OP_SALT_PEPPER || t == IAK_OP_THRESHOLD ||
           t == IAK_OP_OTSU || t == IAK_OP_QUANTIZE || t == IAK_OP_KMEANS;
}
bool needs_hist(const iak_op &o) {
    return o.type == IAK_OP_EQUALIZE || o.type == IAK_OP_OTSU || o.type == IAK_OP_KMEANS ||
           (o.type == IAK_OP_GAUSS && (o.mean == -1 || o.stddev == -1));
}
bool hist_survives(int t) {  // histogram after the stage follows from the one before it
    return t == IAK_OP_EQUALIZE || t == IAK_OP_THRESHOLD || t == IAK_OP_OTSU || t == IAK_OP_QUANTIZE ||
           t == IAK_OP_KMEANS;
}
const char *op_name(int t) {
    static const char *n[] = {"?", "equalize", "gauss", "salt_pepper", "linear", "average", "median", "sobel",
                              "compass", "dilate", "erode", "threshold", "otsu", "quantize", "kmeans", "features"};
    return (t >= 1 && t <= 15) ? n[t] : "?";
}
// stage-wise compulsory traffic per pixel: every stage reads its input and writes its output;
// the noise stages also read one plane byte
long long op_alg_bytes(int t) {
    if (t == IAK_OP_GAUSS || t == IAK_OP_SALT_PEPPER) return 3;
    if (t == IAK_OP_FEATURES) return 1;  // as built here: one read of the image (plus a histogram pass when needed)
    return 2;
}

}  // namespace

#define HIST_RING 8
struct iak_pipeline {
    int W = 0, H = 0, maxB = 0, mode = 0;
    bool fused_math = false;  // contraction mode as it was when the pipeline was created (iak_set_contraction)
    int input = IAK_INPUT_LUMINANCE;
    long long N = 0;
    std::vector<OpCopy> ops;
    std::vector<Group> groups;
    // device memory
    uint8_t *bufA = nullptr, *bufB = nullptr;
    // Grayscale histograms, [maxB][256] per run.  The counting pass adds into a cleared table, and a
    // clear in front of every run is a launch of its own (4.5 us); so runs take turns through
    // HIST_RING tables that one fill clears together, when the last clean one has been handed out.
    // The table for an intermediate's histogram (hist2) comes from the same ring: its first user in a run
    // finds it clear.
    uint32_t *hist = nullptr;       // this run's
    uint32_t *hist_ring = nullptr;  // [HIST_RING][2][maxB][256]
    bool hist2_clear = false;       // hist2 has not been written in this run
    int ring_next = 0, ring_clean = 0;
    hipStream_t ring_stream = nullptr;  // the stream the tables were cleared on
    uint32_t *hist2 = nullptr;  // [maxB][256] histogram of an intermediate (this run's)
    int32_t *map = nullptr;     // [maxB][256]
    int32_t *thr = nullptr;     // [maxB]
    uint16_t *tables = nullptr; // [maxB][CHAIN_MAX][256]
    uint8_t *lgs_tab = nullptr; // [maxB][LGS_TAB_BYTES]
    uint32_t *x0 = nullptr, *x1 = nullptr;
    uint8_t *rv = nullptr;
    // k-means: the libc rand() state the centroid draws continue from run to run, and the draws
    // of the current run, [n_kmeans][B][2]
    int n_kmeans = 0;
    uint32_t *rand_state = nullptr;
    int32_t *kmeans_init = nullptr;
    int run_B = 0;
    // feature extraction: workspace ([maxB] slices of feat_ws_per_image bytes) and the bound output
    int n_features = 0;
    uint8_t *feat_ws = nullptr;
    long long feat_ws_per_image = 0;
    float *features_out = nullptr;
    // level -> planes: the code plane (0 keep, 1 pepper, 2 salt) followed, plane_pitch bytes apart,
    // by the keep-mask and salt-mask planes the fused chain passes read
    std::vector<std::pair<int, uint8_t *>> sp;
    long long plane_pitch = 0;
    // level -> the same noise as a sorted list of the noisy pixel indices (2 * index + salt) behind a
    // per-16-pixel-chunk offset array; only for levels whose noise is sparse (see lgs_stencil8_flat_kernel)
    struct SpSparse {
        int level;
        uint32_t *ptr, *list;
    };
    std::vector<SpSparse> sp_sparse;
    // concurrency: the batch is cut into `lanes` sub-batches that run on their own streams, so
    // that bandwidth-bound and issue-bound kernels of different sub-batches overlap
    int lanes = 1;
    std::vector<hipStream_t> aux;      // lanes - 1 extra streams
    std::vector<hipEvent_t> fork_join; // 1 fork + (lanes - 1) join events
    int timed_images = 0;
    // timing: `slots` sets of (2 * groups + 1) events, run r records into set r % slots: event 0
    // at the start, then per group a split event (between its helper launch and its main
    // kernel, when it has a helper) and an end event.  split[g]: 0 no helper, 1 helper first,
    // 2 main kernel first.
    std::vector<int> split;
    int slots = 0;
    int every = 1;            // only every `every`-th run records events
    long long runs_seen = 0;  // runs since set_timing
    std::vector<hipEvent_t> ev;
    long long runs_timed = 0;

    uint8_t *sp_plane(int level) const {
        for (auto &e : sp)
            if (e.first == level) return e.second;
        return nullptr;
    }
    const SpSparse *sp_sparse_of(int level) const {
        for (auto &e : sp_sparse)
            if (e.level == level) return &e;
        return nullptr;
    }
};

static void free_all(iak_pipeline *p) {
    void *ptrs[] = {p->bufA, p->bufB, p->hist_ring, p->map, p->thr, p->tables, p->lgs_tab, p->x0, p->x1, p->rv,
                    p->rand_state, p->kmeans_init, p->feat_ws};
    for (void *q : ptrs) (void)hipFree(q);
    for (auto &e : p->sp) (void)hipFree(e.second);
    for (auto &e : p->sp_sparse) {
        (void)hipFree(e.ptr);
        (void)hipFree(e.list);
    }
    for (auto e : p->ev) (void)hipEventDestroy(e);
    for (auto e : p->fork_join) (void)hipEventDestroy(e);
    for (auto st : p->aux) (void)hipStreamDestroy(st);
}

// The 512 results of a 3x3 operation on 0 / 255 pixels, from the operation's own kernel run once
// on the image of all patterns (synchronous; pipeline creation only).
static int eval_binary_table(const iak_op &o, BinTable *bt) {
    std::vector<uint8_t> host((size_t)BT_W * BT_H, 0);
    binary_patterns_image(host.data());
    const long long n = (long long)BT_W * BT_H;
    uint8_t *din = nullptr, *dout = nullptr;
    int rc = (int)hipMalloc((void **)&din, (size_t)n);
    if (rc == 0) rc = (int)hipMalloc((void **)&dout, (size_t)n);
    if (rc == 0) rc = (int)hipMemcpy(din, host.data(), (size_t)n, hipMemcpyHostToDevice);
    if (rc == 0) {
        switch (o.type) {
        case IAK_OP_LINEAR: rc = iak_linear(din, dout, BT_W, BT_H, 1, n, n, o.fkernel, 3, 3, IAK_LINEAR_CLAMP, nullptr); break;
        case IAK_OP_AVERAGE: rc = iak_linear(din, dout, BT_W, BT_H, 1, n, n, o.fkernel, 3, 3, IAK_LINEAR_AVERAGE, nullptr); break;
        case IAK_OP_MEDIAN: rc = iak_median(din, dout, BT_W, BT_H, 1, n, n, o.ikernel, 3, 3, nullptr); break;
        case IAK_OP_SOBEL: rc = iak_sobel(din, dout, BT_W, BT_H, 1, n, n, nullptr); break;
        case IAK_OP_COMPASS: rc = iak_compass(din, dout, BT_W, BT_H, 1, n, n, nullptr); break;
        case IAK_OP_ERODE: rc = iak_erode(din, dout, BT_W, BT_H, 1, n, n, o.ikernel, 3, 3, nullptr); break;
        default: rc = IAK_E_UNSUPPORTED; break;
        }
    }
    if (rc == 0) rc = (int)hipMemcpy(host.data(), dout, (size_t)n, hipMemcpyDeviceToHost);
    if (rc == 0) binary_table_from_output(host.data(), bt);
    (void)hipFree(din);
    (void)hipFree(dout);
    return rc;
}

// The salt-and-pepper code plane of a level as the list form the shaped chain pass reads (host work, pipeline
// creation only).  Levels whose noise touches more than one pixel in eight, and images the flat pass does not
// take, keep the mask planes only.
static int make_sp_sparse(iak_pipeline *p, int level, const uint8_t *code_plane) {
    const long long N = p->N;
    if (p->W % 16 != 0 || !tile_geometry(p->W, p->H).flat || N >= (1LL << 30)) return 0;
    std::vector<uint8_t> code((size_t)N);
    IAK_HIP(hipMemcpy(code.data(), code_plane, (size_t)N, hipMemcpyDeviceToHost));
    const long long nc = N / 16;
    std::vector<uint32_t> ptr((size_t)nc + 1), list;
    for (long long c = 0; c < nc; c++) {
        ptr[(size_t)c] = (uint32_t)list.size();
        for (int k = 0; k < 16; k++) {
            const long long i = 16 * c + k;
            if (code[(size_t)i]) list.push_back((uint32_t)(2 * i) | (code[(size_t)i] == 2 ? 1u : 0u));
        }
    }
    ptr[(size_t)nc] = (uint32_t)list.size();
    if ((long long)list.size() * 8 > N) return 0;
    iak_pipeline::SpSparse e{level, nullptr, nullptr};
    IAK_HIP(hipMalloc((void **)&e.ptr, sizeof(uint32_t) * ptr.size()));
    p->sp_sparse.push_back(e);  // owned from here on (free_all)
    IAK_HIP(hipMalloc((void **)&p->sp_sparse.back().list, sizeof(uint32_t) * (list.size() + 1)));
    IAK_HIP(hipMemcpy(p->sp_sparse.back().ptr, ptr.data(), sizeof(uint32_t) * ptr.size(), hipMemcpyHostToDevice));
    if (!list.empty())
        IAK_HIP(hipMemcpy(p->sp_sparse.back().list, list.data(), sizeof(uint32_t) * list.size(), hipMemcpyHostToDevice));
    return 0;
}

// Is "the Sobel of a 0 / 255 image is non-zero" what the bit-mask run computes for it (iak_morph_bits.hpp: one
// of the two gradients positive)?  Checked once per process against the Sobel kernel's own 512 results.
static bool sobel_as_bits_verified() {
    if (getenv("IAK_NO_MORPH_BITS")) return false;
    static std::atomic<int> state{-1};  // -1 not tried, 0 no, 1 yes
    if (state.load() >= 0) return state.load() == 1;
    iak_op sobel{};
    sobel.type = IAK_OP_SOBEL;
    BinTable bt;
    if (eval_binary_table(sobel, &bt) != 0) return false;  // not decided: try again with the next pipeline
    bool same = true;
    for (int idx = 0; idx < 512 && same; idx++) {
        auto px = [&](int i, int j) { return (idx >> (3 * i + j)) & 1; };  // row offset i - 1, column offset j - 1
        const int right = px(0, 2) + 2 * px(1, 2) + px(2, 2), left = px(0, 0) + 2 * px(1, 0) + px(2, 0);
        const int below = px(2, 0) + 2 * px(2, 1) + px(2, 2), above = px(0, 0) + 2 * px(0, 1) + px(0, 2);
        same = (bt.v[idx] != 0) == (right > left || below > above);
    }
    state.store(same ? 1 : 0);
    return same;
}

static int plan(iak_pipeline *p) {
    p->groups.clear();
    Group g0;
    g0.kind = G_GRAYHIST;
    g0.name = "gray+hist";
    g0.alg_bytes = 4 + 1;
    p->groups.push_back(g0);
    const int n = (int)p->ops.size();
    bool hist_known = true;  // of the current image, in p->hist (after gray) or p->hist2
    int i = 0;
    while (i < n) {
        const iak_op &o = p->ops[i].op;
        if (p->mode != IAK_PIPE_STAGEWISE && is_pointwise(o.type)) {
            Group g;
            g.kind = G_CHAIN;
            g.alg_bytes = 0;
            g.chain.n = 0;
            g.chain.compose_n = 0;
            int luts = 0;
            bool chain_hist = hist_known;  // histogram tracked through the chain
            bool first = true;
            while (i < n && is_pointwise(p->ops[i].op.type) && g.chain.n < CHAIN_MAX) {
                const iak_op &q = p->ops[i].op;
                if (needs_hist(q) && !chain_hist) {
                    if (!first) break;  // needs a histogram of a noisy intermediate: new pass
                    g.chain_needs_hist = true;  // a histogram pass on the chain's input comes first
                    chain_hist = true;
                }
                if ((q.type == IAK_OP_THRESHOLD || q.type == IAK_OP_QUANTIZE) && luts >= 2) break;
                ChainStage st{};
                switch (q.type) {
                case IAK_OP_EQUALIZE: st.type = C_EQUALIZE; break;
                case IAK_OP_OTSU: st.type = C_OTSU; break;
                case IAK_OP_KMEANS:
                    st.type = C_KMEANS;
                    st.lut_slot = p->ops[i].kmeans_index;  // which k-means operation of the pipeline
                    break;
                case IAK_OP_THRESHOLD:
                    st.type = C_CONST_LUT;
                    st.lut_slot = luts;
                    for (int v = 0; v < 256; v++) g.chain.const_lut[luts][v] = (v > q.iparam) ? 255 : 0;
                    luts++;
                    break;
                case IAK_OP_QUANTIZE:
                    st.type = C_CONST_LUT;
                    st.lut_slot = luts;
                    for (int v = 0; v < 256; v++) {
                        uint8_t r = (uint8_t)v;
                        for (int l = 0; l < q.iparam; l++)
                            if (v >= q.ikernel[l * 3] && v < q.ikernel[l * 3 + 1]) {
                                r = (uint8_t)q.ikernel[l * 3 + 2];
                                break;
                            }
                        g.chain.const_lut[luts][v] = r;
                    }
                    luts++;
                    break;
                case IAK_OP_GAUSS:
                    st.type = C_GAUSS;
                    st.mean = q.mean;
                    st.stddev = q.stddev;
                    st.plane = p->rv;
                    break;
                case IAK_OP_SALT_PEPPER:
                    st.type = C_SP;
                    st.plane = p->sp_plane(q.iparam) + p->plane_pitch;
                    st.plane2 = st.plane + p->plane_pitch;
                    if (const auto *sparse = p->sp_sparse_of(q.iparam)) {
                        st.sp_ptr = sparse->ptr;
                        st.sp_list = sparse->list;
                    }
                    break;
                }
                g.chain.st[g.chain.n++] = st;
                g.alg_bytes += op_alg_bytes(q.type);
                if (!g.name.empty()) g.name += "+";
                g.name += op_name(q.type);
                if (!hist_survives(q.type)) chain_hist = false;
                first = false;
                i++;
            }
            g.hist_tracked = chain_hist;
            if (g.chain_needs_hist) {
                Group gh;
                gh.kind = G_HIST;
                gh.name = "hist";
                gh.alg_bytes = 1;
                p->groups.push_back(gh);
            }
            // one threshold (fixed and not negative, or Otsu) in front of 3x3 erosions / dilations: the
            // run works on "above the threshold" bits and the mask image is never written
            auto is_morph3 = [&](int k) {
                const iak_op &q = p->ops[k].op;
                return (q.type == IAK_OP_ERODE || q.type == IAK_OP_DILATE) && q.kW == 3 && q.kH == 3;
            };
            const bool one_threshold = g.chain.n == 1 && (p->ops[i - 1].op.type == IAK_OP_OTSU ||
                                                          (p->ops[i - 1].op.type == IAK_OP_THRESHOLD && p->ops[i - 1].op.iparam >= 0));
            // A Sobel between the threshold and the erosions / dilations joins the run when only "is it non-zero"
            // of its result is ever looked at: a dilation behind it in the run, rows of whole 32-pixel words (the
            // pass that takes it), and its 512 results on 0 / 255 neighbourhoods -- from the Sobel kernel
            // itself -- non-zero exactly where the run's formula says so.
            auto is_stage = [&](int k) {
                return is_morph3(k) || (p->ops[k].op.type == IAK_OP_SOBEL && p->W % 32 == 0 && sobel_as_bits_verified());
            };
            int run = 0;
            if (p->mode == IAK_PIPE_FUSED && one_threshold && p->W % 16 == 0 && (long long)p->W * ((long long)p->H + 8) < (1LL << 30)) {
                while (i + run < n && run < 3 && is_stage(i + run)) run++;
                for (int k = 0; k < run; k++) {
                    if (p->ops[i + k].op.type != IAK_OP_SOBEL) continue;
                    bool dilate_behind = false;
                    for (int q = k + 1; q < run; q++) dilate_behind = dilate_behind || p->ops[i + q].op.type == IAK_OP_DILATE;
                    if (!dilate_behind) run = k;  // the run ends in front of it
                }
            }
            if (run >= 1) {
                g.kind = G_MORPH;
                g.op_index = i;
                g.n_ops = run;
                for (int k = 0; k < run; k++) {
                    g.alg_bytes += op_alg_bytes(p->ops[i + k].op.type);
                    g.name += std::string("+") + op_name(p->ops[i + k].op.type);
                }
                i += run;
            } else
            // a single stencil right after the chain reads the chain's output from LDS instead of memory
            if (p->mode == IAK_PIPE_FUSED && i < n && p->W % 16 == 0 && (long long)p->W * p->H < (1LL << 30) && is_fusable_stencil(p->ops[i].op)) {
                g.kind = G_CHAIN_OP;
                g.op_index = i;
                // one threshold (fixed and not negative, or Otsu) in front of an operation that zeroes
                // its border: the operation becomes a look-up in a table of its 512 possible results
                const iak_op &th = p->ops[i - 1].op;
                g.binary = g.chain.n == 1 && p->ops[i].op.type != IAK_OP_DILATE &&
                           (th.type == IAK_OP_OTSU || (th.type == IAK_OP_THRESHOLD && th.iparam >= 0));
                g.alg_bytes += op_alg_bytes(p->ops[i].op.type);
                g.name += std::string("+") + op_name(p->ops[i].op.type);
                i++;
                // look-ups, then Gaussian noise, then sparse salt-and-pepper, on an image one tile wide: the
                // pass that knows this shape; the table prologue composes the look-up stages for it
                if (!g.binary && tile_geometry(p->W, p->H).flat && !getenv("IAK_NO_LGS") && lgs_from_chain(g.chain, &g.lgsd)) {
                    g.lgs = true;
                    g.chain.compose_n = g.lgsd.lut_slot + 1;
                }
            }
            p->groups.push_back(g);
            hist_known = false;  // conservative: a later consumer re-measures
            continue;
        }
        // a run of 3x3 erosions / dilations is one pass on bit masks
        if (p->mode == IAK_PIPE_FUSED && p->W % 8 == 0 &&
            (long long)p->W * ((long long)p->H + 8) < (1LL << 30)) {
            auto is_morph3 = [&](int k) {
                const iak_op &q = p->ops[k].op;
                return (q.type == IAK_OP_ERODE || q.type == IAK_OP_DILATE) && q.kW == 3 && q.kH == 3;
            };
            int run = 0;
            while (i + run < n && run < 3 && is_morph3(i + run)) run++;
            if (run >= 2) {
                Group gm;
                gm.kind = G_MORPH;
                gm.op_index = i;
                gm.n_ops = run;
                gm.alg_bytes = 0;
                gm.chain.n = 0;
                for (int k = 0; k < run; k++) {
                    gm.alg_bytes += op_alg_bytes(p->ops[i + k].op.type);
                    if (k) gm.name += "+";
                    gm.name += op_name(p->ops[i + k].op.type);
                }
                p->groups.push_back(gm);
                hist_known = false;
                i += run;
                continue;
            }
        }
        Group g;
        g.kind = o.type == IAK_OP_FEATURES ? G_FEATURES : G_OP;
        g.op_index = i;
        g.name = op_name(o.type);
        g.alg_bytes = op_alg_bytes(o.type) + (needs_hist(o) ? 1 : 0);
        if (g.kind == G_FEATURES) {  // the image and what is known about its histogram stay as they are
            if (!hist_known) {  // the features start from the histogram of what they look at
                Group gh;
                gh.kind = G_HIST;
                gh.name = "hist";
                gh.alg_bytes = 1;
                p->groups.push_back(gh);
                hist_known = true;
            }
            p->groups.push_back(g);
            i++;
            continue;
        }
        p->groups.push_back(g);
        hist_known = false;
        i++;
    }
    // A histogram pass right behind a per-pixel run or a 3x3 stencil is folded into that pass
    // (the pixels are in registers there): the producer also leaves the histogram of its output.
    if (p->mode != IAK_PIPE_STAGEWISE) {
        for (size_t k = 1; k < p->groups.size(); k++) {
            if (p->groups[k].kind != G_HIST) continue;
            Group &prev = p->groups[k - 1];
            const bool stencil_ok = prev.kind == G_OP && is_fusable_stencil(p->ops[prev.op_index].op) && p->W >= 8 &&
                                    (long long)p->W * p->H < (1LL << 30);
            if (prev.kind != G_CHAIN && prev.kind != G_MORPH && !stencil_ok) continue;
            bool luts_only = prev.kind == G_CHAIN && prev.hist_tracked;
            for (int q = 0; luts_only && q < prev.chain.n; q++)
                luts_only = prev.chain.st[q].type != C_GAUSS && prev.chain.st[q].type != C_SP;
            if (luts_only) {
                prev.hist_from_tables = true;  // no counting at all
            } else {
                prev.emit_hist = true;
                prev.alg_bytes += 1;
            }
            prev.name += "+hist";
            p->groups.erase(p->groups.begin() + (long)k);
            k--;
        }
    }
    if ((int)p->groups.size() > IAK_PIPE_MAX_STAGES) return IAK_E_TOO_LARGE;
    return 0;
}

extern "C" {

int iak_pipeline_create(iak_pipeline **out, int W, int H, int maxB, const iak_op *ops, int nops,
                        unsigned long long seed, int mode) {
    if (!out || W <= 0 || H <= 0 || maxB <= 0 || nops < 0 || (nops > 0 && !ops)) return IAK_E_ARG;
    if (mode != IAK_PIPE_STAGEWISE && mode != IAK_PIPE_FUSED && mode != IAK_PIPE_FUSED_POINTWISE) return IAK_E_ARG;
    iak_pipeline *p = new (std::nothrow) iak_pipeline();
    if (!p) return (int)hipErrorOutOfMemory;
    p->W = W; p->H = H; p->maxB = maxB; p->mode = mode;
    p->fused_math = contraction();
    p->N = (long long)W * H;
    bool any_noise = false;
    for (int i = 0; i < nops; i++) {
        OpCopy c;
        c.op = ops[i];
        const iak_op &o = ops[i];
        int taps = o.kW * o.kH;
        switch (o.type) {
        case IAK_OP_LINEAR: case IAK_OP_AVERAGE:
            if (!o.fkernel || o.kW < 1 || o.kH < 1) { delete p; return IAK_E_ARG; }
            if (taps > IAK_MAX_TAPS) { delete p; return IAK_E_TOO_LARGE; }
            c.f.assign(o.fkernel, o.fkernel + taps);
            break;
        case IAK_OP_MEDIAN: case IAK_OP_DILATE: case IAK_OP_ERODE:
            if (!o.ikernel || o.kW < 1 || o.kH < 1) { delete p; return IAK_E_ARG; }
            if (taps > IAK_MAX_TAPS) { delete p; return IAK_E_TOO_LARGE; }
            c.i.assign(o.ikernel, o.ikernel + taps);
            break;
        case IAK_OP_QUANTIZE:
            if (o.iparam < 0 || (o.iparam > 0 && !o.ikernel)) { delete p; return IAK_E_ARG; }
            c.i.assign(o.ikernel, o.ikernel + 3 * o.iparam);
            break;
        case IAK_OP_GAUSS: case IAK_OP_SALT_PEPPER: any_noise = true; break;
        case IAK_OP_EQUALIZE: case IAK_OP_SOBEL: case IAK_OP_COMPASS: case IAK_OP_THRESHOLD: case IAK_OP_OTSU: break;
        case IAK_OP_KMEANS: c.kmeans_index = p->n_kmeans++; break;
        case IAK_OP_FEATURES: p->n_features++; break;
        default: delete p; return IAK_E_ARG;
        }
        p->ops.push_back(std::move(c));
    }
    for (auto &c : p->ops) {  // re-point at the copies
        c.op.fkernel = c.f.empty() ? nullptr : c.f.data();
        c.op.ikernel = c.i.empty() ? nullptr : c.i.data();
    }
    const size_t img = (size_t)p->N * (size_t)maxB;
#define TRY(call)                                   \
    do {                                            \
        hipError_t e__ = (call);                    \
        if (e__ != hipSuccess) {                    \
            free_all(p);                            \
            delete p;                               \
            return (int)e__;                        \
        }                                           \
    } while (0)
    TRY(hipMalloc((void **)&p->bufA, img));
    TRY(hipMalloc((void **)&p->bufB, img));
    TRY(hipMalloc((void **)&p->hist_ring, sizeof(uint32_t) * 256 * (size_t)maxB * 2 * HIST_RING));
    p->hist = p->hist_ring;
    p->hist2 = p->hist_ring + 256 * (size_t)maxB;
    TRY(hipMalloc((void **)&p->map, sizeof(int32_t) * 256 * (size_t)maxB));
    TRY(hipMalloc((void **)&p->thr, sizeof(int32_t) * (size_t)maxB));
    TRY(hipMalloc((void **)&p->tables, sizeof(uint16_t) * 256 * CHAIN_MAX * (size_t)maxB));
    TRY(hipMalloc((void **)&p->lgs_tab, (size_t)LGS_TAB_BYTES * (size_t)maxB));
    if (p->n_kmeans) {
        TRY(hipMalloc((void **)&p->rand_state, sizeof(uint32_t) * IAK_GLIBC_RAND_WORDS));
        TRY(hipMalloc((void **)&p->kmeans_init, sizeof(int32_t) * 2 * (size_t)p->n_kmeans * (size_t)maxB));
        int rc = iak_glibc_rand_seed(p->rand_state, 1u, nullptr);  // a program that never calls srand
        if (rc != 0) { free_all(p); delete p; return rc; }
    }
    if (p->n_features) {
        p->feat_ws_per_image = iak_features_workspace_bytes(W, H, 1);
        TRY(hipMalloc((void **)&p->feat_ws, (size_t)p->feat_ws_per_image * (size_t)maxB));
    }
    if (any_noise) {
        // setupRandomness (imageAnalysis.cu:740-746): once per pipeline, outside the runs
        TRY(hipMalloc((void **)&p->x0, sizeof(uint32_t) * (size_t)p->N));
        TRY(hipMalloc((void **)&p->x1, sizeof(uint32_t) * (size_t)p->N));
        TRY(hipMalloc((void **)&p->rv, (size_t)p->N));
        int rc = iak_rng_planes(p->x0, p->x1, p->N, seed, nullptr);
        if (rc == 0) rc = iak_rv_plane(p->x0, p->rv, p->N, nullptr);
        for (auto &c : p->ops)
            if (rc == 0 && c.op.type == IAK_OP_SALT_PEPPER && !p->sp_plane(c.op.iparam)) {
                uint8_t *pl = nullptr;
                p->plane_pitch = (p->N + 255) & ~255LL;
                TRY(hipMalloc((void **)&pl, 3 * (size_t)p->plane_pitch));
                p->sp.emplace_back(c.op.iparam, pl);
                rc = iak_sp_plane(p->x0, p->x1, pl, p->N, c.op.iparam, nullptr);
                if (rc == 0) {
                    hipLaunchKernelGGL(sp_masks_kernel, dim3((unsigned)((p->N + 255) / 256)), dim3(256), 0, nullptr, pl,
                                       pl + p->plane_pitch, pl + 2 * p->plane_pitch, p->N);
                    rc = (int)hipGetLastError();
                }
                if (rc == 0) rc = make_sp_sparse(p, c.op.iparam, pl);
            }
        if (rc == 0) rc = (int)hipStreamSynchronize(nullptr);
        if (rc != 0) { free_all(p); delete p; return rc; }
    }
#undef TRY
    int rc = plan(p);
    for (auto &g : p->groups)
        if (rc == 0 && g.kind == G_CHAIN_OP && g.binary) rc = eval_binary_table(p->ops[g.op_index].op, &g.bt);
    if (rc != 0) { free_all(p); delete p; return rc; }
    *out = p;
    return 0;
}

int iak_pipeline_destroy(iak_pipeline *p) {
    if (!p) return 0;
    free_all(p);
    delete p;
    return 0;
}

int iak_pipeline_set_timing_every(iak_pipeline *p, int every) {
    if (!p || every < 1) return IAK_E_ARG;
    p->every = every;
    return 0;
}

int iak_pipeline_set_timing(iak_pipeline *p, int slots) {
    if (!p || slots < 0 || slots > 1024) return IAK_E_ARG;
    p->runs_seen = 0;
    for (auto e : p->ev) (void)hipEventDestroy(e);
    p->ev.clear();
    p->slots = 0;
    p->runs_timed = 0;
    const size_t per = 2 * p->groups.size() + 1;
    p->split.assign(p->groups.size(), 0);
    for (size_t i = 0; i < per * (size_t)slots; i++) {
        hipEvent_t e;
        IAK_HIP(hipEventCreate(&e));
        p->ev.push_back(e);
    }
    p->slots = slots;
    return 0;
}

int iak_pipeline_timings(iak_pipeline *p, int slot, float *ms, float *main_ms, const char **names,
                         long long *alg_bytes_per_px, int cap) {
    if (!p) return IAK_E_ARG;
    const int n = (int)p->groups.size();
    if (ms || main_ms) {
        if (slot < 0 || slot >= p->slots || slot >= p->runs_timed) return IAK_E_ARG;
        const hipEvent_t *ev = p->ev.data() + (size_t)slot * (2 * n + 1);
        IAK_HIP(hipEventSynchronize(ev[2 * n]));
        for (int i = 0; i < n && i < cap; i++) {
            const hipEvent_t e0 = ev[2 * i], es = ev[2 * i + 1], e1 = ev[2 * i + 2];
            if (ms) IAK_HIP(hipEventElapsedTime(&ms[i], e0, e1));
            if (main_ms) {
                if (p->split[i] == 1) IAK_HIP(hipEventElapsedTime(&main_ms[i], es, e1));
                else if (p->split[i] == 2) IAK_HIP(hipEventElapsedTime(&main_ms[i], e0, es));
                else IAK_HIP(hipEventElapsedTime(&main_ms[i], e0, e1));
            }
        }
    }
    for (int i = 0; i < n && i < cap; i++) {
        if (names) names[i] = p->groups[i].name.c_str();
        if (alg_bytes_per_px) alg_bytes_per_px[i] = p->groups[i].alg_bytes;
    }
    return n;
}

// per-image scratch of one sub-batch
struct Slice {
    uint8_t *bufA, *bufB;
    uint32_t *hist, *hist2;
    int32_t *map, *thr;
    uint16_t *tables;
    uint8_t *lgs_tab;
};

static Slice slice_of(const iak_pipeline *p, int b0) {
    Slice s;
    s.bufA = p->bufA + (size_t)b0 * (size_t)p->N;
    s.bufB = p->bufB + (size_t)b0 * (size_t)p->N;
    s.hist = p->hist + (size_t)b0 * 256;
    s.hist2 = p->hist2 + (size_t)b0 * 256;
    s.map = p->map + (size_t)b0 * 256;
    s.thr = p->thr + b0;
    s.tables = p->tables + (size_t)b0 * CHAIN_MAX * 256;
    s.lgs_tab = p->lgs_tab + (size_t)b0 * LGS_TAB_BYTES;
    return s;
}

static int run_single_op(iak_pipeline *p, const Slice &sl, const iak_op &o, const uint8_t *src, uint8_t *dst, int B,
                         const int32_t *kmeans_init, void *stream) {
    const int W = p->W, H = p->H;
    const long long N = p->N;
    switch (o.type) {
    case IAK_OP_EQUALIZE: {
        int rc = iak_hist(src, sl.hist2, N, B, N, stream);
        if (rc) return rc;
        rc = iak_eq_map(sl.hist2, sl.map, (int)N, B, stream);
        if (rc) return rc;
        return iak_apply_map(src, dst, N, B, N, N, sl.map, 256, stream);
    }
    case IAK_OP_GAUSS: {
        const bool from_image = (o.mean == -1 || o.stddev == -1);
        int rc = 0;
        if (from_image) {
            rc = iak_hist(src, sl.hist2, N, B, N, stream);
            if (rc) return rc;
        }
        rc = iak_gauss_table(sl.hist2, (int)N, from_image ? B : 1, o.mean, o.stddev, sl.tables, stream);
        if (rc) return rc;
        return iak_gauss_noise(src, dst, N, B, N, N, p->rv, sl.tables, from_image ? 256 : 0, stream);
    }
    case IAK_OP_SALT_PEPPER: return iak_salt_pepper(src, dst, N, B, N, N, p->sp_plane(o.iparam), stream);
    case IAK_OP_LINEAR: return linear_with_contraction(src, dst, W, H, B, N, N, o.fkernel, o.kW, o.kH, IAK_LINEAR_CLAMP, p->fused_math, stream);
    case IAK_OP_AVERAGE: return linear_with_contraction(src, dst, W, H, B, N, N, o.fkernel, o.kW, o.kH, IAK_LINEAR_AVERAGE, p->fused_math, stream);
    case IAK_OP_MEDIAN: return iak_median(src, dst, W, H, B, N, N, o.ikernel, o.kW, o.kH, stream);
    case IAK_OP_SOBEL: return iak_sobel(src, dst, W, H, B, N, N, stream);
    case IAK_OP_COMPASS: return iak_compass(src, dst, W, H, B, N, N, stream);
    case IAK_OP_DILATE: return iak_dilate(src, dst, W, H, B, N, N, o.ikernel, o.kW, o.kH, stream);
    case IAK_OP_ERODE: return iak_erode(src, dst, W, H, B, N, N, o.ikernel, o.kW, o.kH, stream);
    case IAK_OP_THRESHOLD: return iak_threshold(src, dst, N, B, N, N, o.iparam, stream);
    case IAK_OP_OTSU: {
        int rc = iak_hist(src, sl.hist2, N, B, N, stream);
        if (rc) return rc;
        rc = iak_otsu_threshold(sl.hist2, sl.thr, (int)N, B, stream);
        if (rc) return rc;
        return iak_threshold_per_image(src, dst, N, B, N, N, sl.thr, stream);
    }
    case IAK_OP_QUANTIZE: return iak_quantize(src, dst, N, B, N, N, o.ikernel, o.iparam, stream);
    case IAK_OP_KMEANS: {
        int rc = iak_hist(src, sl.hist2, N, B, N, stream);
        if (rc) return rc;
        rc = iak_kmeans2_map(sl.hist2, (int)N, B, kmeans_init, sl.map, nullptr, stream);
        if (rc) return rc;
        return iak_apply_map(src, dst, N, B, N, N, sl.map, 256, stream);
    }
    }
    return IAK_E_ARG;
}

// initial centroids of k-means operation k for the sub-batch that starts at image b0
static const int32_t *kmeans_of(const iak_pipeline *p, int k, int b0) {
    return k < 0 ? nullptr : p->kmeans_init + 2 * ((size_t)k * (size_t)p->run_B + (size_t)b0);
}
// the chain as the kernels of this sub-batch see it: k-means stages get their centroid slice
static ChainDesc chain_for(const iak_pipeline *p, const ChainDesc &c, int b0) {
    ChainDesc d = c;
    for (int k = 0; k < d.n; k++)
        if (d.st[k].type == C_KMEANS) d.st[k].plane = reinterpret_cast<const uint8_t *>(kmeans_of(p, d.st[k].lut_slot, b0));
    return d;
}

static int run_range(iak_pipeline *p, const uint8_t *rgb, uint8_t *out, int b0, int B, unsigned long long *hist_sum,
                     uint32_t *gray_hist, hipStream_t s, const hipEvent_t *ev) {
    const long long N = p->N;
    const int ng = (int)p->groups.size();
    const Slice sl = slice_of(p, b0);
    rgb += (size_t)b0 * 3 * (size_t)N;
    out += (size_t)b0 * (size_t)N;
    const bool out_unaligned = p->W % 16 == 0 && !aligned16(out);
    // image-producing groups alternate between the two buffers; the last one writes `out`
    int producers = 0;
    for (auto &g : p->groups)
        if (g.kind != G_HIST && g.kind != G_FEATURES) producers++;
    int produced = 0;
    const uint8_t *cur = nullptr;
    const uint32_t *cur_hist = nullptr;
    uint32_t *ghist = gray_hist ? gray_hist + (size_t)b0 * 256 : sl.hist;
    if (ev) IAK_HIP(hipEventRecord(ev[0], s));
    unsigned long long *pending_sum = nullptr;
    // hist2 of this slice is still as the ring's fill left it: its first writer need not clear it
    bool hist2_clear = p->hist2_clear;
    auto take_hist2 = [&]() {
        const bool was = hist2_clear;
        hist2_clear = false;
        return was;
    };
    auto launch_prep = [&](const ChainDesc &cd, uint32_t *hist_out, bool lgs = false) {
        const int extra = pending_sum ? (B + ACC_IMAGES - 1) / ACC_IMAGES : 0;
        hipLaunchKernelGGL(chain_prep_kernel, dim3(B + extra), dim3(256), 0, s, cur_hist, (int)N, cd, sl.tables, hist_out, B,
                           (const uint32_t *)ghist, pending_sum, lgs ? sl.lgs_tab : (uint8_t *)nullptr);
        pending_sum = nullptr;
    };
    for (int gi = 0; gi < ng; gi++) {
        const Group &g = p->groups[gi];
        uint8_t *dst = nullptr;
        bool copy_out = false;
        if (g.kind != G_HIST && g.kind != G_FEATURES) {
            produced++;
            dst = (produced == producers) ? out : ((produced & 1) ? sl.bufA : sl.bufB);
            // The fused passes want 16-byte aligned images (W % 16 == 0 plans only).  A caller's `out`
            // that is not: the last pass writes the executor's own buffer and a copy follows.
            if (dst == out && out_unaligned) {
                dst = (produced & 1) ? sl.bufA : sl.bufB;
                copy_out = true;
            }
        }
        int rc = 0;
        switch (g.kind) {
        case G_GRAYHIST:
            if (!gray_hist)  // the executor's own table: cleared ahead of time (iak_pipeline_run)
                rc = p->input <= IAK_INPUT_AVERAGE
                         ? gray_hist_cleared(rgb, dst, ghist, p->W, p->H, B, 3 * N, N, (p->input == IAK_INPUT_LUMINANCE && p->fused_math) ? 3 : p->input, s)
                         : gray_hist_cleared(rgb + (long long)(p->input - IAK_INPUT_RED) * N, dst, ghist, p->W, p->H, B, 3 * N, N, 2, s);
            else if (p->input <= IAK_INPUT_AVERAGE) {
                rc = (int)hipMemsetAsync(ghist, 0, sizeof(uint32_t) * 256 * (size_t)B, s);  // the caller's array: cleared per run
                if (rc == 0)
                    rc = gray_hist_cleared(rgb, dst, ghist, p->W, p->H, B, 3 * N, N,
                                           (p->input == IAK_INPUT_LUMINANCE && p->fused_math) ? 3 : p->input, s);
            } else
                rc = iak_channel_hist(rgb, dst, ghist, p->W, p->H, B, 3 * N, N, p->input - IAK_INPUT_RED, s);
            if (ev) p->split[gi] = 0;
            // the data-set histogram is folded by the next table prologue's spare workgroups, or at the end
            pending_sum = rc == 0 ? hist_sum : nullptr;
            cur_hist = ghist;
            break;
        case G_HIST:
            rc = iak_hist(cur, sl.hist2, N, B, N, s);
            hist2_clear = false;
            cur_hist = sl.hist2;
            break;
        case G_CHAIN: {
            const ChainDesc cd = chain_for(p, g.chain, b0);
            bool wants = false;
            for (int k = 0; k < g.chain.n; k++)
                wants = wants || g.chain.st[k].type == C_EQUALIZE || g.chain.st[k].type == C_OTSU || g.chain.st[k].type == C_KMEANS ||
                        (g.chain.st[k].type == C_GAUSS && (g.chain.st[k].mean == -1 || g.chain.st[k].stddev == -1));
            if (wants && !cur_hist) return IAK_E_UNSUPPORTED;  // planner invariant
            // hist_from_tables: a workgroup reads its image's old histogram before it writes the new one,
            // so hist2 can be both
            launch_prep(cd, g.hist_from_tables ? sl.hist2 : (uint32_t *)nullptr);
            if (g.hist_from_tables) hist2_clear = false;
            IAK_LAUNCH_CHECK();
            if (ev) {
                p->split[gi] = 1;
                IAK_HIP(hipEventRecord(ev[2 * gi + 1], s));
            }
            if (g.emit_hist) {
                // few, large workgroups: each pays a 32 KiB table and a 256-atomic merge.  The table
                // prologue above has consumed the old histogram (stream order), so hist2 can be reused.
                if (!take_hist2()) IAK_HIP(hipMemsetAsync(sl.hist2, 0, sizeof(uint32_t) * 256 * (size_t)B, s));
                dim3 grid(blocks_x((N + 31) / 32, 1024, B, 1024), B);
                hipLaunchKernelGGL((chain_apply_kernel<true>), grid, dim3(1024), 0, s, cur, dst, N, N, N, cd, sl.tables, sl.hist2);
                IAK_LAUNCH_CHECK();
                cur_hist = sl.hist2;
                break;
            }
            dim3 grid(blocks_x((N + 31) / 32, 256, B, 32768), B);
            hipLaunchKernelGGL((chain_apply_kernel<false>), grid, dim3(256), 0, s, cur, dst, N, N, N, cd, sl.tables);
            IAK_LAUNCH_CHECK();
            cur_hist = g.hist_from_tables ? sl.hist2 : nullptr;
            break;
        }
        case G_CHAIN_OP: {
            const ChainDesc cd = chain_for(p, g.chain, b0);
            bool wants = false;
            for (int k = 0; k < g.chain.n; k++)
                wants = wants || g.chain.st[k].type == C_EQUALIZE || g.chain.st[k].type == C_OTSU || g.chain.st[k].type == C_KMEANS ||
                        (g.chain.st[k].type == C_GAUSS && (g.chain.st[k].mean == -1 || g.chain.st[k].stddev == -1));
            if (wants && !cur_hist) return IAK_E_UNSUPPORTED;  // planner invariant
            if (!aligned16(cur) || (reinterpret_cast<uintptr_t>(dst) & 7u)) return IAK_E_UNSUPPORTED;
            for (int k = 0; k < g.chain.n; k++)
                if (g.chain.st[k].plane && g.chain.st[k].type != C_KMEANS && !aligned16(g.chain.st[k].plane))
                    return IAK_E_UNSUPPORTED;
            launch_prep(cd, nullptr, g.lgs);
            IAK_LAUNCH_CHECK();
            if (ev) {
                p->split[gi] = 1;
                IAK_HIP(hipEventRecord(ev[2 * gi + 1], s));
            }
            const iak_op &o = p->ops[g.op_index].op;
            if (g.binary) {
                rc = launch_binary_table_stencil(cur, dst, p->W, p->H, B, N, N, sl.tables, g.bt, s);
                cur_hist = nullptr;
                break;
            }
            auto go = [&](const auto &op) {
                if (g.lgs) return launch_lgs_stencil(cur, dst, p->W, p->H, B, N, N, g.lgsd, sl.lgs_tab, op, s);
                return launch_chain_stencil(cur, dst, p->W, p->H, B, N, N, cd, sl.tables, op, s);
            };
            switch (o.type) {
            case IAK_OP_LINEAR: rc = with_linear3_op<false>(o.fkernel, go, p->fused_math); break;
            case IAK_OP_AVERAGE: rc = with_linear3_op<true>(o.fkernel, go, p->fused_math); break;
            case IAK_OP_MEDIAN: rc = go(Median8()); break;
            case IAK_OP_SOBEL: rc = go(Sobel8()); break;
            case IAK_OP_COMPASS: rc = go(Compass8()); break;
            case IAK_OP_ERODE: { Erode8 op; op.mask = se_mask3(o.ikernel); rc = go(op); break; }
            case IAK_OP_DILATE: { Dilate8 op; op.mask = se_mask3(o.ikernel); rc = go(op); break; }
            default: rc = IAK_E_UNSUPPORTED; break;
            }
            cur_hist = nullptr;
            break;
        }
        case G_FEATURES:
            if (!p->features_out) return IAK_E_ARG;
            rc = iak_features(cur, p->W, p->H, B, N, cur_hist, p->features_out + (size_t)b0 * IAK_N_FEATURES,
                              p->feat_ws + (size_t)b0 * (size_t)p->feat_ws_per_image, s);
            break;
        case G_MORPH: {
            iak_op run[3];
            for (int k = 0; k < g.n_ops; k++) run[k] = p->ops[g.op_index + k].op;
            const uint16_t *thr_tables = nullptr;
            if (g.chain.n) {  // the threshold in front of the run: its table first
                const ChainDesc cd = chain_for(p, g.chain, b0);
                if (cd.st[0].type == C_OTSU && !cur_hist) return IAK_E_UNSUPPORTED;  // planner invariant
                launch_prep(cd, nullptr);
                IAK_LAUNCH_CHECK();
                if (ev) {
                    p->split[gi] = 1;
                    IAK_HIP(hipEventRecord(ev[2 * gi + 1], s));
                }
                thr_tables = sl.tables;
            }
            rc = morph_chain3(run, g.n_ops, cur, dst, p->W, p->H, B, N, N, g.emit_hist ? sl.hist2 : nullptr, thr_tables, s,
                              g.emit_hist ? take_hist2() : false);
            if (rc == IAK_E_UNSUPPORTED && !thr_tables && dst != sl.bufA && dst != sl.bufB && (cur == sl.bufA || cur == sl.bufB)) {
                // the caller's output buffer is not aligned for the one-pass kernel: one operation
                // at a time through the two intermediate buffers
                const uint8_t *from = cur;
                rc = 0;
                for (int k = 0; k < g.n_ops && rc == 0; k++) {
                    uint8_t *to = (k == g.n_ops - 1) ? dst : (from == sl.bufA ? sl.bufB : sl.bufA);
                    rc = run_single_op(p, sl, run[k], from, to, B, nullptr, s);
                    hist2_clear = false;
                    from = to;
                }
                if (rc == 0 && g.emit_hist) rc = iak_hist(dst, sl.hist2, N, B, N, s);
            }
            cur_hist = g.emit_hist ? sl.hist2 : nullptr;
            break;
        }
        case G_OP:
            if (g.emit_hist) {
                rc = stencil3_with_hist(p->ops[g.op_index].op, cur, dst, p->W, p->H, B, N, N, sl.hist2, s, take_hist2(), p->fused_math);
                if (rc == IAK_E_UNSUPPORTED) {  // buffers not aligned for the 8-pixel path: two passes
                    rc = run_single_op(p, sl, p->ops[g.op_index].op, cur, dst, B, nullptr, s);
                    if (rc == 0) rc = iak_hist(dst, sl.hist2, N, B, N, s);
                }
                cur_hist = sl.hist2;
                break;
            }
            rc = run_single_op(p, sl, p->ops[g.op_index].op, cur, dst, B,
                               kmeans_of(p, p->ops[g.op_index].kmeans_index, b0), s);
            hist2_clear = false;  // some operations count into it
            cur_hist = nullptr;
            break;
        }
        if (rc) return rc;
        if (copy_out) IAK_HIP(hipMemcpyAsync(out, dst, (size_t)B * (size_t)N, hipMemcpyDeviceToDevice, s));
        if (dst) cur = dst;
        if (ev) IAK_HIP(hipEventRecord(ev[2 * gi + 2], s));
    }
    if (pending_sum) {  // no table prologue behind the grayscale pass took it along
        hipLaunchKernelGGL(hist_accumulate_kernel, dim3((B + ACC_IMAGES - 1) / ACC_IMAGES), dim3(256), 0, s, ghist, B, pending_sum);
        IAK_LAUNCH_CHECK();
    }
    return 0;
}

// Bytes per pixel a group has to move as it is executed: what its passes read from and write to
// image-sized arrays.  Intermediates a fused group keeps on chip do not count, nor do the noise
// planes (one image's worth, shared by the whole batch).
static long long group_moved_bytes(const iak_pipeline *p, const Group &g) {
    switch (g.kind) {
    case G_GRAYHIST: return p->input <= IAK_INPUT_AVERAGE ? 4 : 2;
    case G_HIST: case G_FEATURES: return 1;
    case G_CHAIN: case G_CHAIN_OP: case G_MORPH: return 2;
    case G_OP: {
        const iak_op &o = p->ops[g.op_index].op;
        return 2 + ((needs_hist(o) && !g.emit_hist) ? 1 : 0);  // stage-wise: a counting pass of its own first
    }
    }
    return 0;
}

int iak_pipeline_group_bytes(iak_pipeline *p, long long *moved_bytes_per_px, int cap) {
    if (!p) return IAK_E_ARG;
    const int n = (int)p->groups.size();
    for (int i = 0; i < n && i < cap; i++)
        if (moved_bytes_per_px) moved_bytes_per_px[i] = group_moved_bytes(p, p->groups[i]);
    return n;
}

int iak_pipeline_set_lanes(iak_pipeline *p, int lanes) {
    if (!p || lanes < 1 || lanes > 8) return IAK_E_ARG;
    while ((int)p->aux.size() < lanes - 1) {
        hipStream_t st;
        IAK_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        p->aux.push_back(st);
    }
    while ((int)p->fork_join.size() < lanes) {
        hipEvent_t e;
        IAK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        p->fork_join.push_back(e);
    }
    p->lanes = lanes;
    return 0;
}

int iak_pipeline_set_input(iak_pipeline *p, int input) {
    if (!p || input < IAK_INPUT_LUMINANCE || input > IAK_INPUT_BLUE) return IAK_E_ARG;
    p->input = input;
    p->groups[0].name = input <= IAK_INPUT_AVERAGE ? "gray+hist" : "channel+hist";
    p->groups[0].alg_bytes = input <= IAK_INPUT_AVERAGE ? 4 + 1 : 2 + 1;
    return 0;
}

int iak_pipeline_bind_features(iak_pipeline *p, float *features) {
    if (!p) return IAK_E_ARG;
    p->features_out = features;
    return 0;
}

int iak_pipeline_skip_images(iak_pipeline *p, long long images, void *stream) {
    if (!p || images < 0) return IAK_E_ARG;
    if (!p->n_kmeans) return 0;  // nothing in this pipeline depends on how many images came before
    while (images > 0) {  // draw and drop, one batch's worth at a time (kmeans_init holds maxB images)
        const int n = images > p->maxB ? p->maxB : (int)images;
        const int rc = iak_glibc_rand_centroids(p->rand_state, n, p->n_kmeans, p->kmeans_init, stream);
        if (rc) return rc;
        images -= n;
    }
    return 0;
}

int iak_pipeline_timed_images(iak_pipeline *p) { return p ? p->timed_images : IAK_E_ARG; }

int iak_pipeline_run(iak_pipeline *p, const uint8_t *rgb, uint8_t *out, int B, unsigned long long *hist_sum,
                     uint32_t *gray_hist, void *stream) {
    if (!p || !rgb || !out || B <= 0 || B > p->maxB) return IAK_E_ARG;
    hipStream_t s = as_stream(stream);
    const int ng = (int)p->groups.size();
    const bool timed = p->slots && (p->runs_seen++ % p->every) == 0;
    const hipEvent_t *ev = timed ? p->ev.data() + (size_t)(p->runs_timed % p->slots) * (2 * ng + 1) : nullptr;
    const int lanes = (p->lanes > B) ? B : p->lanes;
    p->run_B = B;
    if (!gray_hist) {
        // Next clean histogram table.  When none is left -- or the caller has changed streams, and the
        // order of this run after that clear is no longer given -- all of them are cleared by one fill on
        // this stream, behind every earlier run that still reads one.
        if (p->ring_clean == 0 || p->ring_stream != s) {
            IAK_HIP(hipMemsetAsync(p->hist_ring, 0, sizeof(uint32_t) * 256 * (size_t)p->maxB * 2 * HIST_RING, s));
            p->ring_clean = HIST_RING;
            p->ring_next = 0;
            p->ring_stream = s;
        }
        p->hist = p->hist_ring + (size_t)p->ring_next * 2 * 256 * (size_t)p->maxB;
        p->hist2 = p->hist + 256 * (size_t)p->maxB;
        p->ring_next++;
        p->ring_clean--;
    }
    p->hist2_clear = !gray_hist;
    if (p->n_kmeans) {  // this run's rand() draws, two per image per k-means operation, in image order
        int rc = iak_glibc_rand_centroids(p->rand_state, B, p->n_kmeans, p->kmeans_init, stream);
        if (rc) return rc;
    }
    if (lanes <= 1) {
        int rc = run_range(p, rgb, out, 0, B, hist_sum, gray_hist, s, ev);
        if (rc) return rc;
        p->timed_images = B;
    } else {
        // fork: every extra stream starts after the work already queued on the caller's stream
        IAK_HIP(hipEventRecord(p->fork_join[0], s));
        const int per = (B + lanes - 1) / lanes;
        for (int l = 1; l < lanes; l++) {
            const int b0 = l * per;
            const int nb = (b0 + per <= B) ? per : (B - b0);
            if (nb <= 0) continue;
            hipStream_t sx = p->aux[l - 1];
            IAK_HIP(hipStreamWaitEvent(sx, p->fork_join[0], 0));
            int rc = run_range(p, rgb, out, b0, nb, hist_sum, gray_hist, sx, nullptr);
            if (rc) return rc;
            IAK_HIP(hipEventRecord(p->fork_join[l], sx));
        }
        int rc = run_range(p, rgb, out, 0, per, hist_sum, gray_hist, s, ev);  // sub-batch 0 carries the timing events
        if (rc) return rc;
        p->timed_images = per;
        for (int l = 1; l < lanes; l++)  // join
            if (l * per < B) IAK_HIP(hipStreamWaitEvent(s, p->fork_join[l], 0));
    }
    if (ev) p->runs_timed++;
    return 0;
}

}  // extern "C"
