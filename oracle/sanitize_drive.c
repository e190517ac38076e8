/* TEST INFRASTRUCTURE (like everything under oracle/): a sanitizer pass over the oracle's entry points -- random layers, ragged sizes, zero rows,
 * m = 0, partial neuron ranges, NULL outputs.  Built and run by tests/test_oracle_golden.py with -fsanitize=address,undefined (CPU only). */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
void gpfq_oracle_layer(const float *W, long N, long C, long j0, long j1, const float *X, const float *Xq, long m, long ld,
                       const double *alphabet, int M, double *Q, int16_t *idx, double *resid, int nthreads);
float gpfq_oracle_median_abs(const float *W, long n);
void gpfq_oracle_row_norms(const float *Xq, long N, long m, long ld, float *nrm32);
void gpfq_oracle_msq(const float *W, long n, const double *alphabet, int M, double *Q, int16_t *idx);
static float rnd(void) { return (float)rand() / (float)RAND_MAX - 0.5f; }
int main(void)
{
    srand(7);
    long shapes[][3] = {{1, 1, 1}, {7, 0, 3}, {33, 17, 5}, {64, 129, 9}, {20, 300, 40}, {5, 1000, 2}};
    for (unsigned s = 0; s < sizeof(shapes) / sizeof(shapes[0]); ++s) {
        long N = shapes[s][0], m = shapes[s][1], C = shapes[s][2], ld = m + 3;
        float *W = malloc(sizeof(float) * N * C), *X = malloc(sizeof(float) * N * ld + 4), *Xq = malloc(sizeof(float) * N * ld + 4);
        for (long i = 0; i < N * C; ++i) W[i] = rnd();
        for (long i = 0; i < N * ld; ++i) { X[i] = fmaxf(rnd(), 0.f); Xq[i] = fmaxf(X[i] + 0.1f * rnd(), 0.f); }
        if (N > 2) for (long i = 0; i < ld; ++i) Xq[2 * ld + i] = 0.f;
        for (int M = 2; M <= 17; M += 5) {
            double *A = malloc(sizeof(double) * M);
            double rad = 3.0 * (double)gpfq_oracle_median_abs(W, N * C);
            for (int k = 0; k < M; ++k) A[k] = rad * (-1.0 + 2.0 * k / (M - 1));
            double *Q = malloc(sizeof(double) * N * C), *res = malloc(sizeof(double) * C);
            int16_t *idx = malloc(sizeof(int16_t) * N * C);
            gpfq_oracle_layer(W, N, C, 0, C, X, Xq, m, ld, A, M, Q, idx, res, 2);
            gpfq_oracle_layer(W, N, C, C / 2, C, X, Xq, m, ld, A, M, NULL, idx, NULL, 1);
            gpfq_oracle_msq(W, N * C, A, M, Q, idx);
            float *n32 = malloc(sizeof(float) * N);
            gpfq_oracle_row_norms(Xq, N, m, ld, n32);
            free(n32); free(Q); free(res); free(idx); free(A);
        }
        free(W); free(X); free(Xq);
    }
    puts("oracle under ASan/UBSan: ok");
    return 0;
}
