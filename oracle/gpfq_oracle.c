/*
 * gpfq_oracle.c -- CPU restatement of the reference's greedy per-neuron quantizer.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (quantized_neural_networks_amd/)
 * may import, link or call this file; it exists so tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg have an independent checker for the HIP kernels.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function below against
 * the .npz files under tests/golden, which tools/gen_golden.py produced by running the reference's own
 * functions (scripts/quantized_network.py) under its era interpreter (numpy 1.26 legacy casting).
 *
 * The reference is Python; every mixed f32/f64 step of its NumPy expressions is spelled out
 * here as an explicit C cast so the result does not depend on any interpreter's promotion
 * rules.  Build with -ffp-contract=off: the f32 products and the
 * f32 subtraction of the update must round separately, exactly as NumPy's ufuncs do.
 *
 * Reference lines followed (all in /root/reference/scripts/quantized_network.py):
 *   gpfq_oracle_nearest        <- _bit_round_parallel            :40-57
 *   gpfq_oracle_step (static)  <- _quantize_weight_parallel      :59-89
 *   gpfq_oracle_neuron         <- _quantize_neuron_parallel      :91-121
 *                                 _quantize_filter2D_parallel_jit:185-233 (same recurrence)
 *   gpfq_oracle_layer          <- _quantize_layer_parallel       :523-574 (fan-out + Q assembly)
 *   gpfq_oracle_median_abs     <- median(abs(W.flatten()))       :544, :831
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* alphabet[argmin(abs(alphabet - t))] (:57): |a_k - t| evaluated in f64, FIRST index of the
 * minimum wins (np.argmin).  NaN distances: np.argmin returns the first NaN; with every
 * distance NaN that is index 0, which is what the strict '<' scan below returns too. */
int gpfq_oracle_nearest(double t, const double *alphabet, int M)
{
    int best = 0;
    double dbest = fabs(alphabet[0] - t);
    for (int k = 1; k < M; ++k) {
        double d = fabs(alphabet[k] - t);
        if (d < dbest) { dbest = d; best = k; }
    }
    return best;
}

/* index of an exact 0.0 in the alphabet (odd sizes), else -1: rule (i) returns the literal 0
 * (:84) whether or not it is an alphabet member. */
static int zero_index(const double *alphabet, int M)
{
    int z = -1;
    for (int k = 0; k < M; ++k) if (alphabet[k] == 0.0) z = k;
    return z;
}

/* scipy.linalg.norm(x_f32, 2) (:83, :89): BLAS snrm2 -> f32-rounded Euclidean norm handed back
 * as a Python float.  Verified against the oracle interpreter: equals float32(sqrt(sum_f64 x^2)). */
float gpfq_oracle_norm32(const float *x, long m)
{
    double s = 0.0;
    for (long i = 0; i < m; ++i) s += (double)x[i] * (double)x[i];
    return (float)sqrt(s);
}

void gpfq_oracle_row_norms(const float *Xq, long N, long m, long ld, float *nrm32)
{
    for (long t = 0; t < N; ++t) nrm32[t] = gpfq_oracle_norm32(Xq + t * ld, m);
}

/* One weight (:59-89) followed by the residual update (:119).  Returns the alphabet index
 * (or zero_idx for the literal 0) and stores the f64 value in *qval. */
static int gpfq_oracle_step(float w, double *u, const float *X, const float *Xq, long m,
                            const double *alphabet, int M, int zero_idx, double *qval)
{
    int k;
    double q;
    float nrm = gpfq_oracle_norm32(Xq, m);
    if ((double)nrm < 1e-16) {                         /* :83-84 */
        k = zero_idx;
        q = 0.0;
    } else {
        double d0 = 0.0;                               /* dot(X_tilde, u): f32 upcast, f64 dot (:86) */
        for (long i = 0; i < m; ++i) d0 += (double)Xq[i] * u[i];
        if (fabs(d0) < 1e-10) {                        /* :86-87, w upcast exactly to f64 */
            k = gpfq_oracle_nearest((double)w, alphabet, M);
        } else {                                       /* :89 */
            double d1 = 0.0;
            for (long i = 0; i < m; ++i) {
                float p = w * X[i];                    /* w * X: f32 product            */
                double v = u[i] + (double)p;           /* u + (...): f64                */
                d1 += (double)Xq[i] * v;
            }
            double denom = (double)nrm * (double)nrm;  /* (f32-rounded norm) ** 2 in f64 */
            k = gpfq_oracle_nearest(d1 / denom, alphabet, M);
        }
        q = alphabet[k];
    }
    /* u += w[t]*wX[t,:] - q[t]*qX[t,:]  (:119) under legacy casting: both products and the
     * subtraction are float32 (q is first rounded to f32), the accumulation is f64. */
    float q32 = (float)q;
    for (long i = 0; i < m; ++i) {
        float p = w * X[i];
        float r = q32 * Xq[i];
        float d = p - r;
        u[i] += (double)d;
    }
    *qval = q;
    return k;
}

/* One neuron / one (channel, filter) pair.  w has stride wstride (a column of the Keras
 * [N][C] kernel has stride C).  X, Xq: feature-major [N][ld] f32 rows of length m.
 * Outputs (any may be NULL): q f64[N], idx i16[N] (alphabets of up to 2^15 members), u f64[m] final residual. */
void gpfq_oracle_neuron(const float *w, long wstride, const float *X, const float *Xq,
                        long N, long m, long ld, const double *alphabet, int M,
                        double *q, int16_t *idx, double *u_out)
{
    int zi = zero_index(alphabet, M);
    double *u = (double *)calloc((size_t)(m > 0 ? m : 1), sizeof(double));   /* zeros(m) (:115) */
    for (long t = 0; t < N; ++t) {
        double qv;
        int k = gpfq_oracle_step(w[t * wstride], u, X + t * ld, Xq + t * ld, m, alphabet, M, zi, &qv);
        if (q) q[t] = qv;
        if (idx) idx[t] = (int16_t)k;
    }
    if (u_out) memcpy(u_out, u, (size_t)m * sizeof(double));
    free(u);
}

/* Neurons [j0, j1) of a Dense layer, W in Keras layout [N][C].  Outputs are neuron-major
 * ([j1-j0][N]) so each task writes one contiguous row; resid[j] = ||u_final||_2 (f64).
 * nthreads <= 0 -> all available cores.  Mirrors the process-pool fan-out (:549-567). */
void gpfq_oracle_layer(const float *W, long N, long C, long j0, long j1,
                       const float *X, const float *Xq, long m, long ld,
                       const double *alphabet, int M,
                       double *Q, int16_t *idx, double *resid, int nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (long j = j0; j < j1; ++j) {
        double *u = (double *)malloc((size_t)(m > 0 ? m : 1) * sizeof(double));
        gpfq_oracle_neuron(W + j, C, X, Xq, N, m, ld, alphabet, M,
                           Q ? Q + (j - j0) * N : NULL, idx ? idx + (j - j0) * N : NULL, u);
        if (resid) {
            double s = 0.0;
            for (long i = 0; i < m; ++i) s += u[i] * u[i];
            resid[j - j0] = sqrt(s);
        }
        free(u);
    }
}

int gpfq_oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* np.median(np.abs(W.flatten())) on float32 data (:544): float32 result; for an even count the
 * float32 mean of the two middle values (np.mean of two f32 -> f32 add then divide by 2). */
static int cmp_f32(const void *a, const void *b)
{
    float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

float gpfq_oracle_median_abs(const float *W, long n)
{
    if (n <= 0) return NAN;
    float *a = (float *)malloc((size_t)n * sizeof(float));
    for (long i = 0; i < n; ++i) a[i] = fabsf(W[i]);
    qsort(a, (size_t)n, sizeof(float), cmp_f32);
    float med;
    if (n & 1) med = a[n / 2];
    else {
        float s = a[n / 2 - 1] + a[n / 2];
        med = s / 2.0f;
    }
    free(a);
    return med;
}

/* Plain memoryless scalar quantization of a whole kernel (drivers' MSQ baseline,
 * quantize_pretrained_mlp.py:109): nearest(alphabet, (double)w) per weight. */
/* Interface revision of this file (2: int16 index outputs); oracle/__init__.py rebuilds a library that reports another. */
int gpfq_oracle_abi(void) { return 2; }

void gpfq_oracle_msq(const float *W, long n, const double *alphabet, int M, double *Q, int16_t *idx)
{
    for (long i = 0; i < n; ++i) {
        int k = gpfq_oracle_nearest((double)W[i], alphabet, M);
        if (Q) Q[i] = alphabet[k];
        if (idx) idx[i] = (int16_t)k;
    }
}
