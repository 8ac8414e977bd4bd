/*
 * ia_oracle.c -- CPU restatement of imageAnalysisLib's per-pixel path.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED except for the reference's host-only
 * functions (see ia_oracle.h).
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
 * Every loop below is "one logical thread per pixel, tid in [0,N)": the
 * reference's swapped <<<threadsPerBlock, blocksPerGrid>>> arguments
 * (imageAnalysis.cu:49-55 and every other wrapper) only cap the image size it
 * can launch, not what a thread computes.
 */
#include "ia_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* Contraction mode.  The reference's two expressions of the form a * b + c whose product is not exact -- the
 * luminance (imageAnalysis.cu:22) and the filter accumulation (:521) -- are compiled by nvcc, which by
 * default (-fmad=true) contracts such an expression into one fused multiply-add with a single rounding.  Off
 * (the default here): every operation rounds on its own.  On: the two sites are fused the way LLVM's
 * combiner fuses them (the left product of a sum of two products first; then a product into the sum it is
 * added to).  Neither setting is pinned by anything the reference ships: the switch exists so that the two
 * possible builds can be compared. */
static int g_contract = 0;
void iao_set_contraction(int on) { g_contract = on != 0; }
int iao_get_contraction(void) { return g_contract; }

/* float -> unsigned char: C leaves out-of-range values undefined; the
 * reference relies on it (imageAnalysis.cu:552,844).  Fixed here as
 * truncate toward zero to int, then modulo 256; NaN -> 0. */
static inline uint8_t f2u8(float v) {
    if (v != v) return 0;
    return (uint8_t)(int)v;
}

static inline int tap(const uint8_t *img, long N, long idx) {
    return (idx >= 0 && idx < N) ? img[idx] : 0;
}

static inline int is_border(int row, int col, int W, int H, int a, int s) {
    return row < a || row > (H - a) || col < s || col > (W - s);
}

/* ------------------------------------------------------------------ colour */

int iao_gray(const uint8_t *rgb, uint8_t *gray, int W, int H, int method) {
    long N = (long)W * H;
    if (method == 0) {
        /* imageAnalysis.cu:22 -- literals are double, so the sum is double. */
        for (long t = 0; t < N; t++) {
            double y;
            if (g_contract) {
                y = fma(.07, (double)rgb[t + 2 * N], fma(.21, (double)rgb[t], .72 * rgb[t + N]));
            } else {
                double r = .21 * rgb[t];
                double g = .72 * rgb[t + N];
                double b = .07 * rgb[t + 2 * N];
                y = (r + g) + b;
            }
            gray[t] = (uint8_t)y; /* 0 <= y < 256 always */
        }
        return 0;
    }
    if (method == 1) {
        /* imageAnalysis.cu:33 */
        for (long t = 0; t < N; t++)
            gray[t] = (uint8_t)((rgb[t] + rgb[t + N] + rgb[t + 2 * N]) / 3);
        return 0;
    }
    return -1; /* imageAnalysis.cu:66-68: prints and leaves the image unset */
}

/* ------------------------------------------------- histogram / LUT family */

void iao_hist(const uint8_t *img, long N, int *hist) {
    memset(hist, 0, 256 * sizeof(int));
    for (long t = 0; t < N; t++) {
        int v = img[t];
        if (v != 0) hist[v]++; /* imageAnalysis.cu:136 */
    }
}

void iao_eq_map(const int *hist, int *map, int numPixels) {
    /* imageAnalysis.cu:159-174 */
    float pdf[256];
    float cdf[256];
    for (int i = 0; i < 256; i++) {
        pdf[i] = hist[i] / (float)numPixels;
        cdf[i] = pdf[i];
        if (i > 0) cdf[i] = cdf[i] + cdf[i - 1];
        map[i] = (int)(cdf[i] * 255);
    }
}

void iao_apply_map(const uint8_t *in, uint8_t *out, long N, const int *map) {
    for (long t = 0; t < N; t++) out[t] = (uint8_t)map[in[t]]; /* :183 */
}

void iao_quantize(const uint8_t *in, uint8_t *out, long N, const int *levels, int numLevels) {
    for (long t = 0; t < N; t++) {
        uint8_t p = in[t];
        uint8_t o = p;
        for (int i = 0; i < numLevels; i++) { /* :767-775 */
            if (p >= levels[i * 3] && p < levels[i * 3 + 1]) {
                o = (uint8_t)levels[i * 3 + 2];
                break;
            } else {
                o = p;
            }
        }
        out[t] = o;
    }
}

void iao_threshold(const uint8_t *in, uint8_t *out, long N, int threshold) {
    for (long t = 0; t < N; t++) out[t] = (in[t] > threshold) ? 255 : 0; /* :210-213 */
}

static int otsu_from_hist(const int *hist, long long sum_ll, int totalPixels) {
    /* imageAnalysis.cu:255-277.  `sum` is an int there. */
    int sum = (int)sum_ll;
    float varMax = 0;
    int threshold = 0;
    int backgroundSum = 0;
    int numForegroundPixels = 0;
    int numBackgroundPixels = 0;
    for (int i = 0; i < 256; i++) {
        numBackgroundPixels += hist[i];
        if (numBackgroundPixels == 0) continue;
        numForegroundPixels = totalPixels - numBackgroundPixels;
        if (numForegroundPixels == 0) continue;
        backgroundSum += i * hist[i];
        float backgroundMean = backgroundSum / numBackgroundPixels;                 /* int division */
        float foregroundMean = (sum - backgroundSum) / numForegroundPixels;         /* int division */
        float bcv = (float)numBackgroundPixels * (float)numForegroundPixels *
                    (backgroundMean - foregroundMean) * (backgroundMean - foregroundMean);
        if (bcv > varMax) {
            varMax = bcv;
            threshold = i;
        }
    }
    return threshold;
}

int iao_otsu_from_hist(const int *hist, long long sum, int numPixels) { return otsu_from_hist(hist, sum, numPixels); }

int iao_otsu(const uint8_t *in, uint8_t *out, int W, int H) {
    long N = (long)W * H;
    int hist[256];
    iao_hist(in, N, hist);
    long long sum = 0; /* exact; the reference's float Thrust sum is order-unspecified (:250-254) */
    for (long t = 0; t < N; t++) sum += in[t];
    int thr = otsu_from_hist(hist, sum, (int)N);
    if (out) iao_threshold(in, out, N, thr);
    return thr;
}

void iao_mean_std_from_hist(const long long *cnt, long N, float *mean, float *stddev) {
    long long sum = 0;
    for (int v = 0; v < 256; v++) sum += cnt[v] * v;
    iao_mean_std_from_hist_sum(cnt, N, sum, mean, stddev);
}

void iao_mean_std_from_hist_sum(const long long *cnt, long N, long long sum, float *mean, float *stddev) {
    float m = (float)sum / (float)N; /* :116 */
    double acc = 0.0;
    for (int v = 0; v < 256; v++) {
        double d = (double)((float)v - m); /* :102 pow(data-mean,2) on float data */
        acc += (double)cnt[v] * (d * d);
    }
    float variance = (float)acc / (float)(N - 1); /* :123 */
    *mean = m;
    *stddev = sqrtf(variance); /* :124 */
}

void iao_mean_std(const uint8_t *in, long N, float *mean, float *stddev) {
    long long cnt[256];
    memset(cnt, 0, sizeof cnt);
    for (long t = 0; t < N; t++) cnt[in[t]]++;
    iao_mean_std_from_hist(cnt, N, mean, stddev);
}

/* ---------------------------------------------------------------- XORWOW
 * Restated from the published cuRAND device API (curand_kernel.h, CUDA toolkit,
 * version not pinned by the reference): curand_init(seed, subsequence, 0) salts
 * the Marsaglia xorwow start state with the seed, then jumps subsequence*2^67
 * steps; curand() is one xorwow step plus a Weyl counter d += 362437.  The jump
 * is linear over GF(2) on the 160-bit v state; d is unchanged by it because
 * 362437 * 2^67 == 0 mod 2^32.
 */
typedef struct {
    uint32_t v[5];
    uint32_t d;
} xw_state;

static void xw_seed(xw_state *s, unsigned long long seed) {
    uint32_t s0 = ((uint32_t)seed) ^ 0xaad26b49u;
    uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    uint32_t t0 = 1099087573u * s0;
    uint32_t t1 = 2591861531u * s1;
    s->d = 6615241u + t1 + t0;
    s->v[0] = 123456789u + t0;
    s->v[1] = 362436069u ^ t0;
    s->v[2] = 521288629u + t1;
    s->v[3] = 88675123u ^ t1;
    s->v[4] = 5783321u + t0;
}

static inline void xw_step_v(uint32_t *v) {
    uint32_t t = v[0] ^ (v[0] >> 2);
    v[0] = v[1];
    v[1] = v[2];
    v[2] = v[3];
    v[3] = v[4];
    v[4] = (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1));
}

static inline uint32_t xw_next(xw_state *s) {
    xw_step_v(s->v);
    s->d += 362437u;
    return s->v[4] + s->d;
}

/* m: 160 rows x 5 words; row r = image of basis bit r. */
static void xw_matvec(const uint32_t *m, const uint32_t *v, uint32_t *out) {
    uint32_t r[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 32; j++)
            if (v[i] & (1u << j)) {
                const uint32_t *row = m + (size_t)(i * 32 + j) * 5;
                for (int k = 0; k < 5; k++) r[k] ^= row[k];
            }
    memcpy(out, r, sizeof r);
}

static void xw_matsq(const uint32_t *m, uint32_t *out) {
    uint32_t tmp[800];
    for (int r = 0; r < 160; r++) xw_matvec(m, m + (size_t)r * 5, tmp + (size_t)r * 5);
    memcpy(out, tmp, sizeof tmp);
}

static void xw_one_step_matrix(uint32_t *m) {
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 32; j++) {
            uint32_t v[5] = {0, 0, 0, 0, 0};
            v[i] = 1u << j;
            xw_step_v(v);
            memcpy(m + (size_t)(i * 32 + j) * 5, v, sizeof v);
        }
}

void iao_xorwow_jump_matrix(int log2steps, uint32_t *m800) {
    xw_one_step_matrix(m800);
    for (int i = 0; i < log2steps; i++) xw_matsq(m800, m800);
}

/* jump matrices for 2^(67+k) steps, k = 0..31, built once */
static uint32_t g_jump[32][800];
static volatile int g_jump_ready = 0;

static void xw_prepare(void) {
    if (g_jump_ready) return;
#pragma omp critical(iao_xw_prepare)
    {
        if (!g_jump_ready) {
            iao_xorwow_jump_matrix(67, g_jump[0]);
            for (int k = 1; k < 32; k++) xw_matsq(g_jump[k - 1], g_jump[k]);
            g_jump_ready = 1;
        }
    }
}

static void xw_init(xw_state *s, unsigned long long seed, unsigned int subsequence) {
    xw_prepare();
    xw_seed(s, seed);
    for (int k = 0; k < 32; k++)
        if (subsequence & (1u << k)) xw_matvec(g_jump[k], s->v, s->v);
}

void iao_xorwow_pair(unsigned long long seed, unsigned int tid, uint32_t *x0, uint32_t *x1) {
    xw_state s;
    xw_init(&s, seed, tid);
    *x0 = xw_next(&s);
    *x1 = xw_next(&s);
}

void iao_xorwow_planes(unsigned long long seed, long N, uint32_t *x0, uint32_t *x1) {
    xw_prepare();
#pragma omp parallel for schedule(static)
    for (long t = 0; t < N; t++) iao_xorwow_pair(seed, (unsigned int)t, &x0[t], &x1[t]);
}

void iao_xorwow_stream(unsigned long long seed, unsigned int subsequence, int n, uint32_t *out) {
    xw_state s;
    xw_init(&s, seed, subsequence);
    for (int i = 0; i < n; i++) out[i] = xw_next(&s);
}

float iao_uniform(uint32_t x) {
    /* curand_uniform: x * 2^-32 + 2^-33 in float.  The factor is 2^-32 exactly, so the product is exact and a
     * fused multiply-add (nvcc's default) gives the same bits: contraction mode does not touch it. */
    float f = (float)x * 2.3283064e-10f;
    return f + (2.3283064e-10f / 2.0f);
}

/* ------------------------------------------------------------------ noise */

void iao_salt_pepper(const uint8_t *in, uint8_t *out, long N,
                     const uint32_t *x0, const uint32_t *x1, int level) {
    for (long t = 0; t < N; t++) {
        /* :724 float * double -> double -> int */
        int randVal = (int)(iao_uniform(x0[t]) * (level - 0 + .999999));
        if (randVal == level) {
            float sop = iao_uniform(x1[t]); /* :726 */
            out[t] = (sop > .5) ? 255 : 0;
        } else {
            out[t] = in[t];
        }
    }
}

/* exp for the noise amplitude, in double, from + - * only, so that the same
 * operation sequence gives the same bits on any IEEE machine.  Accurate to
 * ~1e-16 relative, i.e. correctly rounded to float except in astronomically
 * rare double-rounding ties. */
static double portable_exp(double x) {
    if (x != x) return x;
    if (x > 709.0) return HUGE_VAL;
    if (x < -708.0) return 0.0;
    const double INV_LN2 = 1.4426950408889634074;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double kf = x * INV_LN2;
    long k = (long)(kf + (kf < 0 ? -0.5 : 0.5));
    double dk = (double)k;
    double r = (x - dk * LN2_HI) - dk * LN2_LO;
    /* Taylor to r^13/13!, Horner */
    double p = 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    if (k < -1022) return 0.0;
    if (k > 1023) return HUGE_VAL;
    union {
        uint64_t u;
        double d;
    } sc;
    sc.u = (uint64_t)(k + 1023) << 52;
    return p * sc.d;
}

/* entry: low 8 bits amplitude, 0x100 = the amplitude was NaN (output 0). */
static void gauss_table16(float mean, float variance, uint16_t *tab, int use_libm) {
    for (int rv = 0; rv < 256; rv++) {
        /* :841 round(255 * exp(-1.0f * (((randVal-mean)*(randVal-mean)) / (2*variance)))) */
        float d = (float)rv - mean;
        float q = (d * d) / (2 * variance);
        float x = -1.0f * q;
        float e = use_libm ? expf(x) : (float)portable_exp((double)x);
        float noise = roundf(255 * e);
        if (noise != noise)
            tab[rv] = 0x100;
        else
            tab[rv] = (uint16_t)(uint8_t)(int)noise;
    }
}

void iao_gauss_table(float mean, float variance, uint8_t *table256, int use_libm) {
    uint16_t t16[256];
    gauss_table16(mean, variance, t16, use_libm);
    for (int i = 0; i < 256; i++) table256[i] = (uint8_t)t16[i];
}

void iao_gauss_noise(const uint8_t *in, uint8_t *out, long N,
                     const uint32_t *x0, float meanPar, float stdDevPar) {
    float mean = meanPar, stdDev = stdDevPar;
    if (mean == -1 || stdDev == -1) iao_mean_std(in, N, &mean, &stdDev); /* :858-860 */
    uint16_t tab[256];
    gauss_table16(mean, stdDev * stdDev, tab, 0); /* :861 */
    for (long t = 0; t < N; t++) {
        int randVal = (int)(iao_uniform(x0[t]) * (255 - 0 + .999999)); /* :837 */
        uint16_t e = tab[randVal];
        /* :844 (uchar)(image + noise): image+noise is exact in float, <= 510 */
        out[t] = (e & 0x100) ? 0 : f2u8((float)in[t] + (float)(e & 0xff));
    }
}

/* --------------------------------------------------------------- stencils */

static inline int accumulate(const uint8_t *in, long N, int W, int row, int col, int a, int s,
                             const float *kernel) {
    /* :517-524  int sum; sum += uchar * float  ==> sum = (int)((float)sum + (float)px*k) */
    int sum = 0;
    int k = 0;
    for (int i = row - a; i <= row + a; i++)
        for (int j = col - s; j <= col + s; j++) {
            float acc;
            if (g_contract) {
                acc = fmaf((float)tap(in, N, (long)i * W + j), kernel[k], (float)sum);
            } else {
                float prod = (float)tap(in, N, (long)i * W + j) * kernel[k];
                acc = (float)sum + prod;
            }
            sum = (int)acc;
            k++;
        }
    return sum;
}

void iao_linear(const uint8_t *in, uint8_t *out, int W, int H, const float *kernel, int kW, int kH) {
    long N = (long)W * H;
    int a = (kH - 1) / 2, s = (kW - 1) / 2;
    for (long t = 0; t < N; t++) {
        int row = (int)(t / W), col = (int)(t % W);
        if (is_border(row, col, W, H, a, s)) {
            out[t] = 0;
            continue;
        }
        int sum = accumulate(in, N, W, row, col, a, s, kernel);
        if (sum > 255) sum = 255;
        else if (sum < 0) sum = 0;
        out[t] = (uint8_t)sum;
    }
}

void iao_average(const uint8_t *in, uint8_t *out, int W, int H, const float *kernel, int kW, int kH) {
    long N = (long)W * H;
    int a = (kH - 1) / 2, s = (kW - 1) / 2;
    for (long t = 0; t < N; t++) {
        int row = (int)(t / W), col = (int)(t % W);
        if (is_border(row, col, W, H, a, s)) {
            out[t] = 0;
            continue;
        }
        int sum = accumulate(in, N, W, row, col, a, s, kernel);
        out[t] = (uint8_t)(sum / (kW * kH)); /* :487 no clamp */
    }
}

int iao_median(const uint8_t *in, uint8_t *out, int W, int H, const int *weights, int kW, int kH) {
    if (kW < 1 || kH < 1 || !(kW & 1) || !(kH & 1)) return -1;
    int kernSum = 0;
    for (int k = 0; k < kW * kH; k++) {
        if (weights[k] < 0) return -1;
        kernSum += weights[k];
    }
    if (kernSum <= 0) return -1;
    long N = (long)W * H;
    int a = (kH - 1) / 2, s = (kW - 1) / 2;
    int *vals = (int *)malloc(sizeof(int) * (size_t)kernSum);
    for (long t = 0; t < N; t++) {
        int row = (int)(t / W), col = (int)(t % W);
        if (is_border(row, col, W, H, a, s)) {
            out[t] = 0;
            continue;
        }
        int n = 0, k = 0;
        for (int i = row - a; i <= row + a; i++)
            for (int j = col - s; j <= col + s; j++) {
                int v = tap(in, N, (long)i * W + j);
                for (int z = 0; z < weights[k]; z++) vals[n++] = v; /* :656-658 intent */
                k++;
            }
        /* :667-675 insertion sort, ascending */
        for (int i = 1; i < n; i++) {
            int key = vals[i], j = i - 1;
            while (j >= 0 && vals[j] > key) {
                vals[j + 1] = vals[j];
                j--;
            }
            vals[j + 1] = key;
        }
        out[t] = (uint8_t)vals[kernSum / 2]; /* :676 */
    }
    free(vals);
    return 0;
}

void iao_sobel(const uint8_t *in, uint8_t *out, int W, int H) {
    /* :559-561 masks, :585-587 two clamped linear filters then the combine */
    static const float sx[9] = {-1, 0, 1, -2, 0, 2, -1, 0, 1};
    static const float sy[9] = {-1, -2, -1, 0, 0, 0, 1, 2, 1};
    long N = (long)W * H;
    uint8_t *gx = (uint8_t *)malloc((size_t)N);
    uint8_t *gy = (uint8_t *)malloc((size_t)N);
    iao_linear(in, gx, W, H, sx, 3, 3);
    iao_linear(in, gy, W, H, sy, 3, 3);
    for (long t = 0; t < N; t++) {
        int q = (gx[t] * gx[t]) + (gy[t] * gy[t]);
        out[t] = f2u8(sqrtf((float)q)); /* :552 */
    }
    free(gx);
    free(gy);
}

void iao_compass(const uint8_t *in, uint8_t *out, int W, int H) {
    static const float f[4][9] = {{-1, 0, 1, -2, 0, 2, -1, 0, 1},
                                  {-2, -1, 0, -1, 0, 1, 0, 1, 2},
                                  {-1, -2, -1, 0, 0, 0, 1, 2, 1},
                                  {0, -1, -2, 1, 0, -1, 2, 1, 0}};
    long N = (long)W * H;
    for (long t = 0; t < N; t++) {
        int row = (int)(t / W), col = (int)(t % W);
        if (row < 1 || row > (H - 1) || col < 1 || col > (W - 1)) { /* :604 */
            out[t] = 0;
            continue;
        }
        int sum = 0, maxSum = 0;
        for (int m = 0; m < 4; m++) {
            int k = 0; /* the reference lets k run on to 35 (out of bounds); restart per mask */
            for (int i = row - 1; i <= row + 1; i++)
                for (int j = col - 1; j <= col + 1; j++) {
                    float acc = (float)sum + (float)tap(in, N, (long)i * W + j) * f[m][k];
                    sum = (int)acc;
                    k++;
                }
            sum = sum > -sum ? sum : -sum; /* :617; sum is NOT reset for the next mask */
            maxSum = sum > maxSum ? sum : maxSum;
        }
        out[t] = (uint8_t)maxSum; /* :624 no clamp */
    }
}

void iao_dilate(const uint8_t *in, uint8_t *out, int W, int H, const int *se, int kW, int kH) {
    /* :381-404 as a scatter where any thread's 255 beats every 0. */
    long N = (long)W * H;
    int a = (kH - 1) / 2, s = (kW - 1) / 2;
    memset(out, 0, (size_t)N);
    for (long t = 0; t < N; t++) {
        int row = (int)(t / W), col = (int)(t % W);
        if (is_border(row, col, W, H, a, s)) continue;
        if (in[t] == 0) continue;
        int k = 0;
        for (int i = row - a; i <= row + a; i++)
            for (int j = col - s; j <= col + s; j++) {
                long x = (long)i * W + j;
                if (se[k] != 0 && x >= 0 && x < N) out[x] = 255;
                k++;
            }
    }
}

void iao_dilate_seq(const uint8_t *in, uint8_t *out, int W, int H, const int *se, int kW, int kH) {
    long N = (long)W * H;
    int a = (kH - 1) / 2, s = (kW - 1) / 2;
    memset(out, 0, (size_t)N); /* cudaMalloc'd memory is not zeroed; every pixel is written below */
    for (long t = 0; t < N; t++) {
        int row = (int)(t / W), col = (int)(t % W);
        if (is_border(row, col, W, H, a, s)) {
            out[t] = 0;
        } else if (in[t] > 0) {
            int k = 0;
            for (int i = row - a; i <= row + a; i++)
                for (int j = col - s; j <= col + s; j++) {
                    long x = (long)i * W + j;
                    if (x >= 0 && x < N) out[x] = se[k] ? 255 : 0;
                    k++;
                }
        } else {
            out[t] = in[t];
        }
    }
}

void iao_erode(const uint8_t *in, uint8_t *out, int W, int H, const int *se, int kW, int kH) {
    long N = (long)W * H;
    int a = (kH - 1) / 2, s = (kW - 1) / 2;
    for (long t = 0; t < N; t++) {
        int row = (int)(t / W), col = (int)(t % W);
        if (is_border(row, col, W, H, a, s)) {
            out[t] = 0;
        } else if (in[t] > 0) {
            int cond = 1, k = 0;
            for (int i = row - a; i <= row + a; i++)
                for (int j = col - s; j <= col + s; j++) {
                    if (se[k] != 0) cond &= (tap(in, N, (long)i * W + j) != 0); /* :435-437 */
                    k++;
                }
            out[t] = cond ? in[t] : 0; /* :441-444 keeps the grey level */
        } else {
            out[t] = in[t];
        }
    }
}

/* ------------------------------------------------- segmentation / features */

int iao_centroids_changed(uint8_t *cent, int k) {
    /* centroidsHaveChanged, imageAnalysis.cu:281-293: cent = k current values, then k new ones */
    int changed = 0;
    for (int i = 0; i < k; i++)
        if ((cent[i] ^ cent[i + k]) != 0) {
            cent[i] = cent[i + k];
            changed = 1;
        }
    return changed;
}

int iao_random_from_rand(int lo, int hi, int rand_value) {
    /* random(), imageAnalysis.cu:39-41, on one value of libc rand() (RAND_MAX = 2^31 - 1) */
    return lo + rand_value / (2147483647 / (hi - lo + 1) + 1);
}

int iao_array_sum(const int *a, int n) {
    /* arraySum, imageAnalysis.cu:682-688 */
    int sum = 0;
    for (int i = 0; i < n; i++) sum += a[i];
    return sum;
}

int iao_kmeans2(const uint8_t *in, uint8_t *out, long N, int c0, int c1) {
    uint8_t cent[4] = {(uint8_t)c0, (uint8_t)c1, 0, 0};
    uint8_t *labels = (uint8_t *)malloc((size_t)N);
    int count = 0, changed;
    do {
        unsigned int sums[2] = {0, 0};
        int counts[2] = {0, 0};
        for (long t = 0; t < N; t++) { /* :298-302 label 0 = nearer to centroid 1 */
            labels[t] = (abs(cent[0] - in[t]) > abs(cent[1] - in[t])) ? 0 : 1;
            sums[labels[t]] += in[t];
            counts[labels[t]]++;
        }
        for (int i = 0; i < 2; i++) /* :332 */
            cent[i + 2] = f2u8(floorf(sums[i] / (float)counts[i]));
        count++;
        changed = iao_centroids_changed(cent, 2); /* :283-291 */
    } while (changed && count < 100);
    for (long t = 0; t < N; t++) out[t] = labels[t] == 0 ? 0 : 255; /* :342-347 */
    free(labels);
    return count;
}

int iao_hist_mode(const int *hist) {
    /* calcMode, imageMachineLearning.cu:341-351 */
    int maxVal = 0, mode = -1;
    for (int i = 0; i < 256; i++)
        if (maxVal < hist[i]) {
            maxVal = hist[i];
            mode = i;
        }
    return mode;
}
int iao_hist_area(const int *hist, int numPixels) {
    /* calculateCellArea, imageMachineLearning.cu:353-360 */
    return hist[255] < numPixels - hist[255] ? hist[255] : numPixels - hist[255];
}
int iao_hist_perimeter(const int *hist) { return hist[255]; /* calculateCellPerimeter, :362-366 */ }
void iao_hist_range(const int *hist, int *rmin, int *rmax) {
    /* calcMinAndMaxOfRange, imageMachineLearning.cu:379-391, from -1 / -1 (:406-407) */
    *rmin = -1;
    *rmax = -1;
    for (int i = 0; i < 256; i++)
        if (hist[i] != 0 && *rmin == -1) *rmin = i;
    for (int i = 255; i >= 0; i--)
        if (hist[i] != 0 && *rmin != -1 && *rmax == -1) *rmax = i;
}

void iao_features(const uint8_t *in, int W, int H, float *feat) {
    /* imageMachineLearning.cu:393-415 */
    long N = (long)W * H;
    int hist[256];
    float mean, sd;
    iao_mean_std(in, N, &mean, &sd);
    uint8_t *t1 = (uint8_t *)malloc((size_t)N);
    uint8_t *t2 = (uint8_t *)malloc((size_t)N);
    iao_otsu(in, t1, W, H);
    iao_hist(t1, N, hist);
    int area = iao_hist_area(hist, (int)N); /* :399-400 */
    iao_compass(t1, t2, W, H);
    iao_hist(t2, N, hist);
    int perimeter = iao_hist_perimeter(hist); /* :401-403 */
    iao_hist(in, N, hist);
    int mode = iao_hist_mode(hist); /* :405 */
    int rmin, rmax;
    iao_hist_range(hist, &rmin, &rmax); /* :406-408 */
    feat[0] = mean;
    feat[1] = sd;
    feat[2] = (float)area;
    feat[3] = (float)perimeter;
    feat[4] = (float)mode;
    feat[5] = (float)rmin;
    feat[6] = (float)rmax;
    free(t1);
    free(t2);
}

/* ------------------------------------------------------------- pipelines */

static void run_ops_one(const uint8_t *rgb, uint8_t *out, int W, int H, const iao_op *ops, int nops,
                        const uint32_t *x0, const uint32_t *x1, int *hist, uint8_t *tmp, const int *centroids,
                        float *features, int input) {
    long N = (long)W * H;
    uint8_t *cur = out, *nxt = tmp;
    int kmeans_seen = 0;
    if (input <= 1) iao_gray(rgb, cur, W, H, input); /* test.cu:276-287 */
    else memcpy(cur, rgb + (size_t)(input - 2) * N, (size_t)N); /* extractSingleColorChannel, test.cu:288-318 */
    iao_hist(cur, N, hist);
    for (int i = 0; i < nops; i++) {
        const iao_op *op = &ops[i];
        if (op->type == IAO_OP_FEATURES) { /* test.cu:533-555: the image stays as it is */
            if (features) iao_features(cur, W, H, features);
            continue;
        }
        switch (op->type) {
        case IAO_OP_KMEANS: /* test.cu:519-532; the two random(0,256) draws of this call */
            iao_kmeans2(cur, nxt, N, centroids[2 * kmeans_seen], centroids[2 * kmeans_seen + 1]);
            kmeans_seen++;
            break;
        case IAO_OP_EQUALIZE: { /* test.cu:404-419 */
            int h[256], map[256];
            iao_hist(cur, N, h);
            iao_eq_map(h, map, (int)N);
            iao_apply_map(cur, nxt, N, map);
            break;
        }
        case IAO_OP_GAUSS: iao_gauss_noise(cur, nxt, N, x0, op->mean, op->stddev); break;
        case IAO_OP_SALT_PEPPER: iao_salt_pepper(cur, nxt, N, x0, x1, op->iparam); break;
        case IAO_OP_LINEAR: iao_linear(cur, nxt, W, H, op->fkernel, op->kW, op->kH); break;
        case IAO_OP_AVERAGE: iao_average(cur, nxt, W, H, op->fkernel, op->kW, op->kH); break;
        case IAO_OP_MEDIAN: iao_median(cur, nxt, W, H, op->ikernel, op->kW, op->kH); break;
        case IAO_OP_SOBEL: iao_sobel(cur, nxt, W, H); break;
        case IAO_OP_COMPASS: iao_compass(cur, nxt, W, H); break;
        case IAO_OP_DILATE: iao_dilate(cur, nxt, W, H, op->ikernel, op->kW, op->kH); break;
        case IAO_OP_ERODE: iao_erode(cur, nxt, W, H, op->ikernel, op->kW, op->kH); break;
        case IAO_OP_THRESHOLD: iao_threshold(cur, nxt, N, op->iparam); break;
        case IAO_OP_OTSU: iao_otsu(cur, nxt, W, H); break;
        case IAO_OP_QUANTIZE: iao_quantize(cur, nxt, N, op->ikernel, op->iparam); break;
        default: memcpy(nxt, cur, (size_t)N); break;
        }
        uint8_t *sw = cur;
        cur = nxt;
        nxt = sw;
    }
    if (cur != out) memcpy(out, cur, (size_t)N);
}

void iao_pipeline(const uint8_t *rgb, uint8_t *out, int W, int H, int B, const iao_op *ops, int nops,
                  const uint32_t *x0, const uint32_t *x1, long long *hist_sum, int threads) {
    iao_pipeline_ex(rgb, out, W, H, B, ops, nops, x0, x1, hist_sum, threads, NULL, NULL, 0);
}

void iao_pipeline_ex(const uint8_t *rgb, uint8_t *out, int W, int H, int B, const iao_op *ops, int nops,
                     const uint32_t *x0, const uint32_t *x1, long long *hist_sum, int threads, const int *centroids,
                     float *features, int input) {
    long N = (long)W * H;
    int nk = 0;
    for (int i = 0; i < nops; i++) nk += ops[i].type == IAO_OP_KMEANS;
    if (hist_sum) memset(hist_sum, 0, 256 * sizeof(long long));
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        uint8_t *tmp = (uint8_t *)malloc((size_t)N);
        long long local[256];
        memset(local, 0, sizeof local);
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < B; b++) {
            int hist[256];
            run_ops_one(rgb + (size_t)b * 3 * N, out + (size_t)b * N, W, H, ops, nops, x0, x1, hist, tmp,
                        centroids ? centroids + (size_t)b * 2 * nk : NULL, features ? features + (size_t)b * 7 : NULL,
                        input);
            for (int v = 0; v < 256; v++) local[v] += hist[v];
        }
        if (hist_sum) {
#pragma omp critical(iao_hist_merge)
            for (int v = 0; v < 256; v++) hist_sum[v] += local[v];
        }
        free(tmp);
    }
}


/* ---- k-nearest neighbours (imageMachineLearning.cu:29-339) ----------------------------
 * PARITY UNPINNED, more than elsewhere: the reference's selection kernel races, reads one
 * distance per thread slot (:109-112) and lets thread 1 vote; its cross-validation passes
 * `dataset + trainSize` as the test rows (:275-276; rows are numAttributes floats wide).
 * Restated here as what the code is evidently after: the k nearest rows in (distance, index)
 * order, the vote exactly as written (:54-90), the test fold = the last tenth of the rotated
 * rows.  NUM_CLASSES is 6 (:212); rows whose class falls outside are left out of the
 * confusion matrix instead of written out of bounds. */
static int knn_before(float d1, int i1, float d2, int i2) {
    int n1 = d1 != d1, n2 = d2 != d2;
    if (n1 != n2) return n2;
    if (!n1 && d1 != d2) return d1 < d2;
    return i1 < i2;
}
static int knn_vote_ref(const int *indexes, const float *train, int k, int A) {
    int classVotes[32];
    int duplicate = 0, finalClass = 0, mostVotes = 0;
    for (int i = 0; i < 32; i++) classVotes[i] = 0;
    for (int i = 0; i < k; i++) {
        int c = (int)train[(long)indexes[i] * A + A - 1];
        if (c >= 0 && c < 32) classVotes[c] += 1;
    }
    for (int i = 0; i < 32; i++)
        if (classVotes[i] > mostVotes) {
            finalClass = i;
            mostVotes = classVotes[i];
        }
    for (int i = 0; i < 32; i++)
        if (classVotes[i] == mostVotes && classVotes[i] > 0) duplicate = 1;
    if (duplicate && (k - 1) > 0) return knn_vote_ref(indexes, train, k - 1, A); /* :83-88 */
    return finalClass;
}
void iao_knn_predict(const float *train, const float *test, int numTrain, int numTest, int A, int k, int *pred) {
    if (k > numTrain) k = numTrain;
    float *dist = (float *)malloc(sizeof(float) * (size_t)numTrain);
    int *order = (int *)malloc(sizeof(int) * (size_t)numTrain);
    for (int t = 0; t < numTest; t++) {
        for (int j = 0; j < numTrain; j++) { /* computeDistances :29-43 */
            float sum = 0;
            for (int a = 0; a < A - 1; a++)
                sum += (test[(long)t * A + a] - train[(long)j * A + a]) * (test[(long)t * A + a] - train[(long)j * A + a]);
            dist[j] = (float)sqrt(sum);
            order[j] = j;
        }
        for (int r = 0; r < k; r++) { /* selection of the r-th nearest */
            int m = r;
            for (int j = r + 1; j < numTrain; j++)
                if (knn_before(dist[order[j]], order[j], dist[order[m]], order[m])) m = j;
            int tmp = order[r];
            order[r] = order[m];
            order[m] = tmp;
        }
        pred[t] = knn_vote_ref(order, train, k, A);
    }
    free(dist);
    free(order);
}

#define IAO_NUM_CLASSES 6
void iao_confusion_metrics(const int *pred, const float *rows, int A, int n, int *confusion, float *accuracy,
                           float *precision, float *recall) {
    /* computeConfusionMatrix / computeAccuracy / computePrecisionAndRecall,
     * imageMachineLearning.cu:214-255.  confusion: IAO_NUM_CLASSES^2 ints, overwritten. */
    memset(confusion, 0, sizeof(int) * IAO_NUM_CLASSES * IAO_NUM_CLASSES);
    for (int i = 0; i < n; i++) { /* :214-227 */
        int tc = (int)rows[(long)i * A + A - 1], pc = pred[i];
        if (tc >= 0 && tc < IAO_NUM_CLASSES && pc >= 0 && pc < IAO_NUM_CLASSES) confusion[tc * IAO_NUM_CLASSES + pc]++;
    }
    int ok = 0;
    for (int i = 0; i < IAO_NUM_CLASSES; i++) ok += confusion[i * IAO_NUM_CLASSES + i];
    *accuracy = ok / (float)n; /* :229-236 */
    float tp = 0, tr = 0;
    for (int i = 0; i < IAO_NUM_CLASSES; i++) { /* :239-255 */
        int row = 0, col = 0;
        for (int j = 0; j < IAO_NUM_CLASSES; j++) {
            col += confusion[i + j * IAO_NUM_CLASSES];
            row += confusion[i * IAO_NUM_CLASSES + j];
        }
        int t = confusion[i * IAO_NUM_CLASSES + i];
        tp += (row > 0) ? t / (float)row : 0;
        tr += (col > 0) ? t / (float)col : 0;
    }
    *precision = tp / IAO_NUM_CLASSES;
    *recall = tr / IAO_NUM_CLASSES;
}

float iao_knn_crossval(const float *dataset, int numInstances, int A, int k, float *precision, float *recall,
                       float *fold_accuracy) {
    int numToRotate = (int)(numInstances * .1); /* :262 */
    int trainSize = numInstances - numToRotate;
    float *rot = (float *)malloc(sizeof(float) * (size_t)numInstances * A);
    int *pred = (int *)malloc(sizeof(int) * (size_t)(numToRotate > 0 ? numToRotate : 1));
    double totalAccuracy = 0.0;
    float totalPrecision = 0, totalRecall = 0;
    for (int fold = 0; fold < 10; fold++) {
        long shift = (long)numToRotate * A * fold; /* std::rotate to the left, :264 */
        long total = (long)numInstances * A;
        for (long i = 0; i < total; i++) rot[i] = dataset[(i + shift) % total];
        float accuracy = 0, prec = 0, rec = 0;
        if (numToRotate > 0) {
            const float *test = rot + (long)trainSize * A;
            iao_knn_predict(rot, test, trainSize, numToRotate, A, k, pred);
            int confusion[IAO_NUM_CLASSES * IAO_NUM_CLASSES];
            iao_confusion_metrics(pred, test, A, numToRotate, confusion, &accuracy, &prec, &rec);
        }
        if (fold_accuracy) fold_accuracy[fold] = accuracy;
        totalAccuracy += (double)accuracy;
        totalPrecision += prec;
        totalRecall += rec;
    }
    free(rot);
    free(pred);
    *precision = totalPrecision / 10.0;
    *recall = totalRecall / 10.0;
    return (float)(totalAccuracy / 10.0); /* :337-338 */
}
