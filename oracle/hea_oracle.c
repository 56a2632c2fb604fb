/*
 * hea_oracle.c -- CPU fp64 ORACLE (plain C + OpenMP over the batch).  TEST INFRASTRUCTURE ONLY.
 *
 * Host twin of the device C ABI in include/quanonet_hea.h: same arguments with HOST
 * pointers and no workspace/stream.  It restates, gate by gate, the reference algorithm
 *   circuit   : core/quantum_circuits_tq.py:65-104  (== core/quantum_circuits_ms.py:127-226)
 *   read-out  : core/quantum_circuits_tq.py:106-127 (== core/quantum_circuits_ms.py:28-39)
 *   gradient  : adjoint differentiation, the scheme behind MindQuantum's
 *               get_expectation_with_grad (core/quantum_circuits_ms.py:229-233); batch-parallel
 *               over CPU threads like mqvector [upstream].
 * It deliberately applies every RX/RY/RZ/CNOT separately (no fusion) so that it is an
 * independent statement from the fused HIP kernels.  Pinned by tests/test_oracle_golden.py
 * against the numpy restatement and the reference's known answers K1-K8.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXQ 14

typedef struct { double re, im; } cplx;

static inline void rx(cplx* s, int n, int q, double th) {
    const double c = cos(0.5 * th), sn = sin(0.5 * th);
    const long dim = 1L << n, m = 1L << q;
    for (long k = 0; k < dim; ++k) {
        if (k & m) continue;
        cplx a = s[k], b = s[k | m];
        /* [[c,-is],[-is,c]] */
        s[k].re = c * a.re + sn * b.im;     s[k].im = c * a.im - sn * b.re;
        s[k | m].re = c * b.re + sn * a.im; s[k | m].im = c * b.im - sn * a.re;
    }
}
static inline void ry(cplx* s, int n, int q, double th) {
    const double c = cos(0.5 * th), sn = sin(0.5 * th);
    const long dim = 1L << n, m = 1L << q;
    for (long k = 0; k < dim; ++k) {
        if (k & m) continue;
        cplx a = s[k], b = s[k | m];
        /* [[c,-s],[s,c]] */
        s[k].re = c * a.re - sn * b.re;     s[k].im = c * a.im - sn * b.im;
        s[k | m].re = sn * a.re + c * b.re; s[k | m].im = sn * a.im + c * b.im;
    }
}
static inline void rz(cplx* s, int n, int q, double th) {
    const double c = cos(0.5 * th), sn = sin(0.5 * th);
    const long dim = 1L << n, m = 1L << q;
    for (long k = 0; k < dim; ++k) {
        cplx a = s[k];
        if (k & m) { s[k].re = c * a.re - sn * a.im; s[k].im = c * a.im + sn * a.re; }   /* e^{+i th/2} */
        else       { s[k].re = c * a.re + sn * a.im; s[k].im = c * a.im - sn * a.re; }   /* e^{-i th/2} */
    }
}
static inline void cnot(cplx* s, int n, int control, int target) {
    const long dim = 1L << n, mc = 1L << control, mt = 1L << target;
    for (long k = 0; k < dim; ++k) {
        if ((k & mc) && !(k & mt)) { cplx t = s[k]; s[k] = s[k | mt]; s[k | mt] = t; }
    }
}
/* Im <lam| sigma_q |psi> */
static inline double im_inner(const cplx* lam, const cplx* psi, int n, int q, char pauli) {
    const long dim = 1L << n, m = 1L << q;
    double acc = 0.0;
    for (long k = 0; k < dim; ++k) {
        cplx l = lam[k], v;
        if (pauli == 'X') { v = psi[k ^ m]; }
        else if (pauli == 'Y') {
            cplx p = psi[k ^ m];
            if (k & m) { v.re = -p.im; v.im = p.re; }   /* (sigma_y psi)_1 = +i psi_0 */
            else       { v.re = p.im;  v.im = -p.re; }  /* (sigma_y psi)_0 = -i psi_1 */
        } else { v = psi[k]; if (k & m) { v.re = -v.re; v.im = -v.im; } }
        acc += l.re * v.im - l.im * v.re;               /* Im(conj(l) v) */
    }
    return acc;
}

static void run_forward(cplx* s, int n, int nb, const int32_t* enc, const int32_t* ld,
                        const double* xb, const double* w) {
    const long dim = 1L << n;
    memset(s, 0, sizeof(cplx) * dim);
    s[0].re = 1.0;
    long col = 0, blk = 0;
    for (int b = 0; b < nb; ++b) {
        for (int j = 0; j < enc[b]; ++j) rx(s, n, j % n, xb[col++]);
        for (int l = 0; l < ld[b]; ++l, ++blk) {
            const double* wb = w + blk * 3 * n;
            for (int i = 0; i < n; ++i) {
                ry(s, n, i, wb[0 * n + i]);
                rz(s, n, i, wb[1 * n + i]);
                ry(s, n, i, wb[2 * n + i]);
            }
            for (int i = 0; i < n; ++i) cnot(s, n, (i + 1) % n, i);
        }
    }
}

static inline double ham_k(long k, int n, double off, double co, const double* diag) {
    if (diag) return diag[k];
    return off + co * (double)(n - 2 * __builtin_popcountl((unsigned long)k));
}

/* hs = H s for H = off + co * sum_q P_q with P = X ('X') or Y ('Y'), written out Pauli by Pauli
 * (generate_simple_hamiltonian's `pauli`, core/quantum_circuits_ms.py:28-39).  Deliberately NOT the
 * basis-change trick the HIP kernels use, so the two stay independent statements. */
static void apply_pauli_ham(const cplx* s, cplx* hs, int n, double off, double co, int pauli) {
    const long dim = 1L << n;
    for (long k = 0; k < dim; ++k) {
        double re = off * s[k].re, im = off * s[k].im;
        for (int q = 0; q < n; ++q) {
            const long m = 1L << q;
            const cplx p = s[k ^ m];
            if (pauli == 1) { re += co * p.re; im += co * p.im; }
            else if (k & m) { re += co * -p.im; im += co * p.re; }     /* (sigma_y psi)_1 = +i psi_0 */
            else            { re += co * p.im;  im += co * -p.re; }    /* (sigma_y psi)_0 = -i psi_1 */
        }
        hs[k].re = re; hs[k].im = im;
    }
}

static int check(int n, int nb, const int32_t* enc, const int32_t* ld, long* E, long* blk) {
    if (n < 2 || n > MAXQ || nb < 0 || (nb > 0 && (!enc || !ld))) return -1;
    *E = 0; *blk = 0;
    for (int b = 0; b < nb; ++b) {
        if (enc[b] < 0 || ld[b] < 0) return -1;
        *E += enc[b]; *blk += ld[b];
    }
    return 0;
}

int qhea_oracle_forward(int n, int nb, const int32_t* enc, const int32_t* ld, int64_t B,
                        const double* x, const double* w, double off, double co,
                        const double* diag, int pauli, double* out, double* state_out) {
    long E, blk;
    if (check(n, nb, enc, ld, &E, &blk) || B < 0 || !out || (B > 0 && E > 0 && !x) || (blk > 0 && !w))
        return -1;
    if (pauli < 0 || pauli > 2 || (pauli && diag)) return -1;
    const long dim = 1L << n;
#pragma omp parallel
    {
        cplx* s = (cplx*)malloc(sizeof(cplx) * dim);
        cplx* hs = (cplx*)malloc(sizeof(cplx) * dim);
#pragma omp for schedule(static)
        for (int64_t b = 0; b < B; ++b) {
            run_forward(s, n, nb, enc, ld, x + b * E, w);
            double acc = 0.0;
            if (pauli == 0) {
                for (long k = 0; k < dim; ++k)
                    acc += ham_k(k, n, off, co, diag) * (s[k].re * s[k].re + s[k].im * s[k].im);
            } else {
                apply_pauli_ham(s, hs, n, off, co, pauli);
                for (long k = 0; k < dim; ++k) acc += s[k].re * hs[k].re + s[k].im * hs[k].im;
            }
            out[b] = acc;
            if (state_out) memcpy(state_out + b * dim * 2, s, sizeof(cplx) * dim);
        }
        free(s); free(hs);
    }
    return 0;
}

int qhea_oracle_backward(int n, int nb, const int32_t* enc, const int32_t* ld, int64_t B,
                         const double* x, const double* w, double off, double co,
                         const double* diag, int pauli, const double* g,
                         double* out, double* grad_x, double* grad_w) {
    long E, blk;
    if (check(n, nb, enc, ld, &E, &blk) || B < 0 || !g || !grad_x || !grad_w) return -1;
    if (pauli < 0 || pauli > 2 || (pauli && diag)) return -1;
    const long dim = 1L << n, P = blk * 3 * n;
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    double* part = (double*)calloc((size_t)nthreads * (size_t)(P > 0 ? P : 1), sizeof(double));
    if (!part) return -1;
#pragma omp parallel
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        double* gw = part + (size_t)tid * (size_t)P;
        cplx* s = (cplx*)malloc(sizeof(cplx) * dim);
        cplx* lam = (cplx*)malloc(sizeof(cplx) * dim);
#pragma omp for schedule(static)
        for (int64_t b = 0; b < B; ++b) {
            const double* xb = x + b * E;
            double* gxb = grad_x + b * E;
            run_forward(s, n, nb, enc, ld, xb, w);
            double acc = 0.0;
            if (pauli == 0) {
                for (long k = 0; k < dim; ++k) {
                    const double h = ham_k(k, n, off, co, diag);
                    acc += h * (s[k].re * s[k].re + s[k].im * s[k].im);
                    lam[k].re = g[b] * h * s[k].re;
                    lam[k].im = g[b] * h * s[k].im;
                }
            } else {
                apply_pauli_ham(s, lam, n, off, co, pauli);
                for (long k = 0; k < dim; ++k) {
                    acc += s[k].re * lam[k].re + s[k].im * lam[k].im;
                    lam[k].re *= g[b]; lam[k].im *= g[b];
                }
            }
            if (out) out[b] = acc;
            long col = E, sub = blk;
            for (int bb = nb - 1; bb >= 0; --bb) {
                for (int l = ld[bb] - 1; l >= 0; --l) {
                    --sub;
                    const double* wb = w + sub * 3 * n;
                    double* gwb = gw + sub * 3 * n;
                    for (int i = n - 1; i >= 0; --i) { cnot(s, n, (i + 1) % n, i); cnot(lam, n, (i + 1) % n, i); }
                    for (int i = n - 1; i >= 0; --i) {
                        gwb[2 * n + i] += im_inner(lam, s, n, i, 'Y');
                        ry(s, n, i, -wb[2 * n + i]); ry(lam, n, i, -wb[2 * n + i]);
                        gwb[1 * n + i] += im_inner(lam, s, n, i, 'Z');
                        rz(s, n, i, -wb[1 * n + i]); rz(lam, n, i, -wb[1 * n + i]);
                        gwb[0 * n + i] += im_inner(lam, s, n, i, 'Y');
                        ry(s, n, i, -wb[0 * n + i]); ry(lam, n, i, -wb[0 * n + i]);
                    }
                }
                for (int j = enc[bb] - 1; j >= 0; --j) {
                    --col;
                    gxb[col] = im_inner(lam, s, n, j % n, 'X');
                    rx(s, n, j % n, -xb[col]); rx(lam, n, j % n, -xb[col]);
                }
            }
        }
        free(s); free(lam);
    }
    for (long p = 0; p < P; ++p) {
        double acc = 0.0;
        for (int t = 0; t < nthreads; ++t) acc += part[(size_t)t * (size_t)P + p];
        grad_w[p] = acc;
    }
    free(part);
    return 0;
}

int qhea_oracle_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* number of OpenMP threads of the following calls (bench.py's 1-thread baseline leg); returns the previous maximum */
int qhea_oracle_set_threads(int n) {
#ifdef _OPENMP
    const int before = omp_get_max_threads();
    if (n > 0) omp_set_num_threads(n);
    return before;
#else
    (void)n;
    return 1;
#endif
}
