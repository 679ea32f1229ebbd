/*
 * oracle/rnnt_oracle.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-precision CPU restatement of the RNN-Transducer loss as
 * the reference (1ytic/warp-rnnt v0.7.0) computes it.  It exists so that the
 * HIP kernels in warp_rnnt_amd/csrc can be checked against something that
 * follows the reference's *operation order* (fp32 association matters at the
 * 1e-4 level on long lattices).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product path (warp_rnnt / warp_rnnt_amd) never
 * imports, links or falls back to anything in oracle/.
 *
 * Parity pin: this file is checked against every golden vector the
 * reference's own tests hold for the path (tests/golden/reference_vectors.json,
 * transcribed data from pytorch_binding/warp_rnnt/test.py:34-188,214-257) by
 * tests/test_oracle.py.  The reference's arithmetic itself is CUDA and cannot
 * be compiled or run in this image (no nvcc, no CUDA device), so there is no
 * oracle/_ref build; see DESIGN.md "Oracle".
 *
 * Each function cites the reference lines it restates (paths relative to the
 * reference checkout).  Nothing here is copied: the reference is a set of CUDA
 * kernels organised around 32-lane warps and global spin locks, this is a
 * sequential double loop.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* log(exp(a)+exp(b)) exactly as core.cu:26-39 / core_gather.cu:22-35 do it:
 * max + log1pf(expf(-|a-b|)), no cutoff, NaN when both are -inf. */
static inline float lse(float a, float b) {
    float mx, diff;
    if (a > b) { mx = a; diff = b - a; } else { mx = b; diff = a - b; }
    return mx + log1pf(expf(diff));
}

typedef struct {
    const float *lp;    /* (N,T,U,V) row-major */
    const int *labels;  /* (N,U-1) or NULL when gathered */
    int T, U, V;
    int blank;          /* >=0: dense layout; -1: gathered layout (V==2, ch0 blank, ch1 label) */
} lp_view;

/* dense indexing: core.cu:84-87,116-120 ; gathered: core_gather.cu:81,91,110-112 */
static inline float lp_blank(const lp_view *v, int n, int t, int u) {
    size_t cell = ((size_t)n * v->T + t) * v->U + u;
    return v->lp[cell * v->V + (v->blank < 0 ? 0 : v->blank)];
}
static inline float lp_label(const lp_view *v, int n, int t, int u) {
    size_t cell = ((size_t)n * v->T + t) * v->U + u;
    if (v->blank < 0) return v->lp[cell * v->V + 1];
    return v->lp[cell * v->V + v->labels[(size_t)n * (v->U - 1) + u]];
}

/* Inclusive Hillis-Steele scan over up to 32 values, same round structure as
 * the shuffle loop at core_gather.cu:93-99 (i = 1,2,4,8,16; lane d adds the
 * pre-round value of lane d-i when i <= d). */
static void warp_scan32(float *b, int cnt) {
    float prev[32];
    for (int i = 1; i < 32; i *= 2) {
        memcpy(prev, b, sizeof(float) * (size_t)cnt);
        for (int d = i; d < cnt; ++d) b[d] += prev[d - i];
    }
}

/*
 * One utterance: alphas and betas, row-major (T,U) slices.
 *   scan_mode 0: first column / last column accumulated serially
 *   scan_mode 1: accumulated in 32-frame tiles with the reference's
 *                shuffle-scan association (core_gather.cu:86-104,187-205)
 * Recurrences: core_gather.cu:62-64 (alpha[0,0]), :76-84 (row 0),
 * :106-126 (interior: lse(alpha[t-1,u]+blank, alpha[t,u-1]+label)),
 * :163-165 (beta corner), :177-185 (last row), :207-227 (interior).
 */
static void sweep_one(const lp_view *v, int n, int Tn, int Un, int scan_mode,
                      float *al, float *be) {
    const int U = v->U;
    /* ---- alphas ---- */
    al[0] = 0.0f;
    for (int u = 1; u < Un; ++u)
        al[u] = al[u - 1] + lp_label(v, n, 0, u - 1);
    if (scan_mode == 0) {
        for (int t = 1; t < Tn; ++t)
            al[(size_t)t * U] = al[(size_t)(t - 1) * U] + lp_blank(v, n, t - 1, 0);
    } else {
        for (int p = 0; p + 1 < Tn; p += 32) {
            float b[32];
            int cnt = 0;
            for (int d = 0; d < 32 && p + d + 1 < Tn; ++d, ++cnt)
                b[d] = lp_blank(v, n, p + d, 0);
            warp_scan32(b, cnt);
            float carry = al[(size_t)p * U];
            for (int d = 0; d < cnt; ++d)
                al[(size_t)(p + d + 1) * U] = carry + b[d];
        }
    }
    for (int t = 1; t < Tn; ++t)
        for (int u = 1; u < Un; ++u) {
            float skip = al[(size_t)(t - 1) * U + u] + lp_blank(v, n, t - 1, u);
            float emit = al[(size_t)t * U + u - 1] + lp_label(v, n, t, u - 1);
            al[(size_t)t * U + u] = lse(skip, emit);
        }
    /* ---- betas ---- */
    const int T1 = Tn - 1, U1 = Un - 1;
    be[(size_t)T1 * U + U1] = lp_blank(v, n, T1, U1);
    for (int u = U1 - 1; u >= 0; --u)
        be[(size_t)T1 * U + u] = be[(size_t)T1 * U + u + 1] + lp_label(v, n, T1, u);
    if (scan_mode == 0) {
        for (int t = T1 - 1; t >= 0; --t)
            be[(size_t)t * U + U1] = be[(size_t)(t + 1) * U + U1] + lp_blank(v, n, t, U1);
    } else {
        /* tile g covers reversed frames tt = p+d+1, i.e. t = T1 - tt */
        for (int p = 0; p + 1 < Tn; p += 32) {
            float b[32];
            int cnt = 0;
            for (int d = 0; d < 32 && p + d + 1 < Tn; ++d, ++cnt)
                b[d] = lp_blank(v, n, T1 - (p + d + 1), U1);
            warp_scan32(b, cnt);
            float carry = be[(size_t)(T1 - p) * U + U1];
            for (int d = 0; d < cnt; ++d)
                be[(size_t)(T1 - (p + d + 1)) * U + U1] = carry + b[d];
        }
    }
    for (int t = T1 - 1; t >= 0; --t)
        for (int u = U1 - 1; u >= 0; --u) {
            float skip = be[(size_t)(t + 1) * U + u] + lp_blank(v, n, t, u);
            float emit = be[(size_t)t * U + u + 1] + lp_label(v, n, t, u);
            be[(size_t)t * U + u] = lse(skip, emit);
        }
}

/*
 * Full op: alphas, betas, grads, costs for a minibatch.
 *   log_probs (N,T,U,V) fp32 row-major; blank == -1 means the gathered layout
 *   (V must be 2), otherwise dense with labels (N,U-1).
 *   grads has the shape of log_probs and is fully written (zeros where the
 *   reference relies on zero-initialisation, binding.cpp:58).
 *   mismatch[n] (optional) = 1 when the forward/backward guard of
 *   core_gather.cu:341-354 fired for sample n.
 * Gradient formulas: core_gather.cu:248-284 (blank), :286-319 (label, with
 * the FastEmit factor evaluated in double as `(1. + lambda) * a`), costs and
 * guard :321-357.  Dense-layout write order (blank kernel first, label kernel
 * second, so a label equal to blank overwrites): core.cu:382-393.
 * Returns 0, or -1 on bad arguments.
 */
int oracle_rnnt_loss(const float *log_probs, const int *labels, const int *xn,
                     const int *yn, int N, int T, int U, int V, int blank,
                     float fastemit_lambda, int scan_mode, float *alphas,
                     float *betas, float *grads, float *costs, int *mismatch) {
    if (blank < 0 && V != 2) return -1;
    if (blank >= V) return -1;
    lp_view v = {log_probs, labels, T, U, V, blank};
    const size_t TU = (size_t)T * U;
    memset(grads, 0, sizeof(float) * (size_t)N * TU * V);

#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int n = 0; n < N; ++n) {
        const int Tn = xn[n], Un = yn[n] + 1;
        float *al = alphas + (size_t)n * TU;
        float *be = betas + (size_t)n * TU;
        float *g = grads + (size_t)n * TU * V;
        sweep_one(&v, n, Tn, Un, scan_mode, al, be);
        const float b00 = be[0];
        const int bch = blank < 0 ? 0 : blank;
        /* blank gradients */
        for (int t = 0; t < Tn; ++t)
            for (int u = 0; u < Un; ++u) {
                if (t == Tn - 1 && u < Un - 1) continue;
                float a = al[(size_t)t * U + u];
                if (t < Tn - 1) a += be[(size_t)(t + 1) * U + u];
                a = expf(a + lp_blank(&v, n, t, u) - b00);
                g[((size_t)t * U + u) * V + bch] = -a;
            }
        /* label gradients (written after the blank ones, as the reference launches them) */
        for (int t = 0; t < Tn; ++t)
            for (int u = 0; u < Un - 1; ++u) {
                float a = al[(size_t)t * U + u] + be[(size_t)t * U + u + 1];
                a = expf(a + lp_label(&v, n, t, u) - b00);
                a = (float)((1. + fastemit_lambda) * a);
                int lch = blank < 0 ? 1 : labels[(size_t)n * (U - 1) + u];
                g[((size_t)t * U + u) * V + lch] = -a;
            }
        /* cost + forward/backward consistency guard */
        float a = al[(size_t)(Tn - 1) * U + (Un - 1)] + lp_blank(&v, n, Tn - 1, Un - 1);
        float b = b00;
        float ratio = fabsf(a - b) / fabsf(fmaxf(a, b));
        int bad = ratio > 0.001f;
        if (bad) {
            memset(g, 0, sizeof(float) * TU * V);
            b = (a + b) / 2.0f;
        }
        if (mismatch) mismatch[n] = bad;
        costs[n] = -b;
    }
    return 0;
}

/* fp32 log-softmax over the last axis the way torch's CPU/CUDA kernels
 * associate it: (x - max) - log(sum(exp(x - max))).  Caller side of the path:
 * pytorch_binding/benchmark.py:65,70 ; README.md:59. */
void oracle_log_softmax(const float *x, float *out, long rows, int V) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (long r = 0; r < rows; ++r) {
        const float *xi = x + (size_t)r * V;
        float *oi = out + (size_t)r * V;
        float mx = xi[0];
        for (int k = 1; k < V; ++k) mx = xi[k] > mx ? xi[k] : mx;
        float s = 0.0f;
        for (int k = 0; k < V; ++k) s += expf(xi[k] - mx);
        float ls = logf(s);
        for (int k = 0; k < V; ++k) oi[k] = (xi[k] - mx) - ls;
    }
}

/* The gather prologue of warp_rnnt/__init__.py:118-128: channel 0 = blank,
 * channel 1 = labels[n,u] for u < U-1 and blank for the last column. */
void oracle_gather(const float *log_probs, const int *labels, float *out,
                   int N, int T, int U, int V, int blank) {
    for (int n = 0; n < N; ++n)
        for (int t = 0; t < T; ++t)
            for (int u = 0; u < U; ++u) {
                size_t cell = ((size_t)n * T + t) * U + u;
                int l = (u < U - 1) ? labels[(size_t)n * (U - 1) + u] : blank;
                out[cell * 2 + 0] = log_probs[cell * V + blank];
                out[cell * 2 + 1] = log_probs[cell * V + l];
            }
}

void oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
