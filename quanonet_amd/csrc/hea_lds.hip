// hea_lds.hip -- LDS-resident variant of the HEA simulator for the largest qubit counts.
//
// The wave-resident kernels (hea_device.hpp) keep 2^(n-6) amplitudes per lane in VGPRs; with psi and
// lambda live that exceeds the 256 architectural VGPRs for n >= 11 (backward) / n = 12 (forward) and the
// compiler spills to scratch (measured 14.5 ms vs 0.27 ms going from n=10 to n=12 for the same depth).
// Here ONE 256-thread workgroup owns one sample and the state(s) live in LDS (n=12: 64 KB per state, two
// states in the backward kernel = 128 of the CU's 160 KB).  Gates are applied two qubits per pass (each
// thread loads a group of four amplitudes, applies both 2x2 updates in registers and stores), the CNOT
// ring as in-place conditional swaps.  Same circuit, same fused SU(2) tables, same adjoint recipe as the
// wave-resident kernels; bound by LDS bandwidth (~2 state passes per gate pair).
#include "hea_device.hpp"

namespace qhea {

namespace {

constexpr int kT = 256;                       // threads per sample

struct c2 { double x, y; };
__device__ __forceinline__ c2 ld(const double2* p, int i) { const double2 v = p[i]; return {v.x, v.y}; }
__device__ __forceinline__ void st(double2* p, int i, c2 v) { p[i] = make_double2(v.x, v.y); }

__device__ __forceinline__ int ins0(int p, int q) { return ((p >> q) << (q + 1)) | (p & ((1 << q) - 1)); }

// [[a,b],[-conj b, conj a]] on (p0,p1); u = (ar, ai, br, bi)
__device__ __forceinline__ void su2(c2& p0, c2& p1, double4 u) {
    const c2 a0 = p0, a1 = p1;
    p0.x = u.x * a0.x - u.y * a0.y + u.z * a1.x - u.w * a1.y;
    p0.y = u.x * a0.y + u.y * a0.x + u.z * a1.y + u.w * a1.x;
    p1.x = u.x * a1.x + u.y * a1.y - u.z * a0.x - u.w * a0.y;
    p1.y = u.x * a1.y - u.y * a1.x - u.z * a0.y + u.w * a0.x;
}
__device__ __forceinline__ double4 dagger(double4 u) { return make_double4(u.x, -u.y, -u.z, -u.w); }
// RX as an SU(2) in the same form: a = c, b = -i s  ->  (c, 0, 0, -s)
__device__ __forceinline__ double4 rx_su2(double2 cs) { return make_double4(cs.x, 0.0, 0.0, -cs.y); }

// Im<l|sigma|p> contributions of one pair (p0,p1),(l0,l1)
__device__ __forceinline__ void inner(c2 p0, c2 p1, c2 l0, c2 l1, double& X, double& Y, double& Z) {
    X += (l0.x * p1.y - l0.y * p1.x) + (l1.x * p0.y - l1.y * p0.x);
    Y += -(l0.x * p1.x + l0.y * p1.y) + (l1.x * p0.x + l1.y * p0.y);
    Z += (l0.x * p0.y - l0.y * p0.x) - (l1.x * p1.y - l1.y * p1.x);
}

// two independent one-qubit gates (qa < qb) in one pass over the state
__device__ __forceinline__ void pass2(double2* s, int n, int qa, int qb, double4 ua, double4 ub) {
    const int ng = 1 << (n - 2);
    for (int g = threadIdx.x; g < ng; g += kT) {
        const int i00 = ins0(ins0(g, qa), qb), i01 = i00 | (1 << qa), i10 = i00 | (1 << qb), i11 = i01 | (1 << qb);
        c2 a00 = ld(s, i00), a01 = ld(s, i01), a10 = ld(s, i10), a11 = ld(s, i11);
        su2(a00, a01, ua); su2(a10, a11, ua);
        su2(a00, a10, ub); su2(a01, a11, ub);
        st(s, i00, a00); st(s, i01, a01); st(s, i10, a10); st(s, i11, a11);
    }
    __syncthreads();
}
__device__ __forceinline__ void pass1(double2* s, int n, int q, double4 u) {
    const int np = 1 << (n - 1);
    for (int p = threadIdx.x; p < np; p += kT) {
        const int i0 = ins0(p, q), i1 = i0 | (1 << q);
        c2 a0 = ld(s, i0), a1 = ld(s, i1);
        su2(a0, a1, u);
        st(s, i0, a0); st(s, i1, a1);
    }
    __syncthreads();
}
// CNOT(control c, target t): swap amplitudes (c=1,t=0) <-> (c=1,t=1)
__device__ __forceinline__ void cnot_pass(double2* s, int n, int c, int t) {
    const int lo = c < t ? c : t, hi = c < t ? t : c;
    const int ng = 1 << (n - 2);
    for (int g = threadIdx.x; g < ng; g += kT) {
        const int i = ins0(ins0(g, lo), hi) | (1 << c);
        const double2 a = s[i], b = s[i | (1 << t)];
        s[i] = b; s[i | (1 << t)] = a;
    }
    __syncthreads();
}
__device__ __forceinline__ void ring(double2* s, int n, bool reverse) {
    if (!reverse) for (int i = 0; i < n; ++i) cnot_pass(s, n, (i + 1) % n, i);
    else          for (int i = n - 1; i >= 0; --i) cnot_pass(s, n, (i + 1) % n, i);
}

__device__ __forceinline__ double4 gate_u(const char* gates, int n, int sub, int q) {   // base variant of gate (sub,q)
    return *reinterpret_cast<const double4*>(gates + ((long)(sub + 1) * n + q) * kGateBytes);
}

__device__ void forward_lds(double2* psi, int n, const Runs& runs, const double2* __restrict__ cs_b,
                            const char* __restrict__ gates) {
    const int dim = 1 << n;
    for (int k = threadIdx.x; k < dim; k += kT) psi[k] = make_double2(k == 0 ? 1.0 : 0.0, 0.0);
    __syncthreads();
    int col = 0, sub = 0;
    for (int ri = 0; ri < runs.nruns; ++ri) {
        const int ne = runs.enc[ri], nld = runs.ld[ri];
        for (int rep = 0; rep < runs.count[ri]; ++rep) {
            for (int j = 0; j < ne; ) {                        // RX(x[col+j]) on wire j % n
                const int q = j % n;
                if (j + 1 < ne && q + 1 < n) {
                    pass2(psi, n, q, q + 1, rx_su2(cs_b[col + j]), rx_su2(cs_b[col + j + 1]));
                    j += 2;
                } else {
                    pass1(psi, n, q, rx_su2(cs_b[col + j]));
                    j += 1;
                }
            }
            col += ne;
            for (int l = 0; l < nld; ++l, ++sub) {
                int q = 0;
                for (; q + 1 < n; q += 2) pass2(psi, n, q, q + 1, gate_u(gates, n, sub, q), gate_u(gates, n, sub, q + 1));
                if (q < n) pass1(psi, n, q, gate_u(gates, n, sub, q));
                ring(psi, n, false);
            }
        }
    }
}

// readout-basis change on every qubit (see basis_change in hea_device.hpp): X -> RY(-pi/2), Y -> RX(+pi/2)
__device__ void basis_lds(double2* s, int n, int pauli, bool dag) {
    if (pauli == 0) return;
    const double r = 0.70710678118654752440, sr = dag ? -r : r;
    const double4 u = pauli == 1 ? make_double4(r, 0.0, sr, 0.0) : make_double4(r, 0.0, 0.0, -sr);
    int q = 0;
    for (; q + 1 < n; q += 2) pass2(s, n, q, q + 1, u, u);
    if (q < n) pass1(s, n, q, u);
}

__device__ __forceinline__ double ham_w(int k, int n, double off, double co, const double* __restrict__ diag) {
    return diag ? diag[k] : off + co * (double)(n - 2 * (int)__popc((unsigned)k));
}

// sum of one double per thread over the 256-thread block (fixed order), result in every thread
__device__ __forceinline__ double block_sum(double v, double* scratch /*[4]*/) {
    double t[1] = {v};
    lane_reduce<1, 6>(t, threadIdx.x & 63);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = t[0];
    __syncthreads();
    return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}

}  // namespace

__global__ __launch_bounds__(kT) void lds_fwd_kernel(int n, Runs runs, long B, int E, const double2* __restrict__ cs,
                                                     const char* __restrict__ gates, double off, double co,
                                                     const double* __restrict__ diag, int pauli,
                                                     double* __restrict__ out, double* __restrict__ state_out,
                                                     const double* __restrict__ bias) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2* psi = reinterpret_cast<double2*>(smem);
    double* scratch = reinterpret_cast<double*>(smem + ((size_t)16 << n));
    const long b = blockIdx.x;
    forward_lds(psi, n, runs, cs + b * E, gates);
    const int dim = 1 << n;
    if (state_out)
        for (int k = threadIdx.x; k < dim; k += kT) reinterpret_cast<double2*>(state_out)[(b << n) + k] = psi[k];
    basis_lds(psi, n, pauli, false);
    double acc = 0.0;
    for (int k = threadIdx.x; k < dim; k += kT) {
        const double2 a = psi[k];
        acc += ham_w(k, n, off, co, diag) * (a.x * a.x + a.y * a.y);
    }
    const double tot = block_sum(acc, scratch);
    if (threadIdx.x == 0) out[b] = tot + (bias ? bias[0] : 0.0);
}

// Backward: one row of `partial` per sample ([B][blk][kw]); grad_x written directly.
template <int KW>
__global__ __launch_bounds__(kT) void lds_bwd_kernel(int n, Runs runs, long B, int E, int blk,
                                                     const double2* __restrict__ cs, const char* __restrict__ gates,
                                                     double off, double co, const double* __restrict__ diag, int pauli,
                                                     const double* __restrict__ g, const double* __restrict__ state_in,
                                                     const double* __restrict__ y, const double* __restrict__ bias,
                                                     double inv_bt, double* __restrict__ out,
                                                     double* __restrict__ grad_x, double* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int dim = 1 << n;
    double2* psi = reinterpret_cast<double2*>(smem);
    double2* lam = psi + dim;
    double* scratch = reinterpret_cast<double*>(lam + dim);          // [4][KW]
    const long b = blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const double2* __restrict__ cs_b = cs + b * E;

    if (state_in) {
        for (int k = threadIdx.x; k < dim; k += kT) psi[k] = reinterpret_cast<const double2*>(state_in)[(b << n) + k];
        __syncthreads();
    } else {
        forward_lds(psi, n, runs, cs_b, gates);
    }
    basis_lds(psi, n, pauli, false);
    double acc = 0.0;
    for (int k = threadIdx.x; k < dim; k += kT) {
        const double2 a = psi[k];
        acc += ham_w(k, n, off, co, diag) * (a.x * a.x + a.y * a.y);
    }
    const double pred = block_sum(acc, scratch) + (bias ? bias[0] : 0.0);
    if (out && threadIdx.x == 0) out[b] = pred;
    const double gb = y ? 2.0 * (pred - y[b]) * inv_bt : g[b];
    for (int k = threadIdx.x; k < dim; k += kT) {
        const double2 a = psi[k];
        const double h = gb * ham_w(k, n, off, co, diag);
        lam[k] = make_double2(h * a.x, h * a.y);
    }
    __syncthreads();
    basis_lds(psi, n, pauli, true);
    basis_lds(lam, n, pauli, true);

    double* __restrict__ part_b = partial + b * (long)blk * KW;
    int col = E, sub = blk;
    for (int ri = runs.nruns - 1; ri >= 0; --ri) {
        const int ne = runs.enc[ri], nld = runs.ld[ri];
        for (int rep = 0; rep < runs.count[ri]; ++rep) {
            for (int l = nld - 1; l >= 0; --l) {
                --sub;
                ring(psi, n, true);
                ring(lam, n, true);
                double xyz[KW];
#pragma unroll
                for (int i = 0; i < KW; ++i) xyz[i] = 0.0;
                // qubit pairs (q, q+1); compile-time pair index so that the xyz slots stay in registers
                static_for<0, (KW / 3 + 1) / 2>([&](auto pidx) {
                    constexpr int qq = 2 * decltype(pidx)::value;
                    if (qq < n) {                                   // wave-uniform
                        const bool two = qq + 1 < n;
                        const double4 ua = gate_u(gates, n, sub, qq), ub = two ? gate_u(gates, n, sub, qq + 1) : ua;
                        const double4 uad = dagger(ua), ubd = dagger(ub);
                        double Xa = 0, Ya = 0, Za = 0, Xb = 0, Yb = 0, Zb = 0;
                        if (two) {
                            const int ng = 1 << (n - 2);
                            for (int gi = threadIdx.x; gi < ng; gi += kT) {
                                const int i00 = ins0(ins0(gi, qq), qq + 1), i01 = i00 | (1 << qq), i10 = i00 | (2 << qq),
                                          i11 = i01 | (2 << qq);
                                c2 p00 = ld(psi, i00), p01 = ld(psi, i01), p10 = ld(psi, i10), p11 = ld(psi, i11);
                                c2 l00 = ld(lam, i00), l01 = ld(lam, i01), l10 = ld(lam, i10), l11 = ld(lam, i11);
                                inner(p00, p01, l00, l01, Xa, Ya, Za); inner(p10, p11, l10, l11, Xa, Ya, Za);
                                inner(p00, p10, l00, l10, Xb, Yb, Zb); inner(p01, p11, l01, l11, Xb, Yb, Zb);
                                su2(p00, p01, uad); su2(p10, p11, uad); su2(p00, p10, ubd); su2(p01, p11, ubd);
                                su2(l00, l01, uad); su2(l10, l11, uad); su2(l00, l10, ubd); su2(l01, l11, ubd);
                                st(psi, i00, p00); st(psi, i01, p01); st(psi, i10, p10); st(psi, i11, p11);
                                st(lam, i00, l00); st(lam, i01, l01); st(lam, i10, l10); st(lam, i11, l11);
                            }
                        } else {
                            const int np = 1 << (n - 1);
                            for (int p = threadIdx.x; p < np; p += kT) {
                                const int i0 = ins0(p, qq), i1 = i0 | (1 << qq);
                                c2 p0 = ld(psi, i0), p1 = ld(psi, i1), l0 = ld(lam, i0), l1 = ld(lam, i1);
                                inner(p0, p1, l0, l1, Xa, Ya, Za);
                                su2(p0, p1, uad); su2(l0, l1, uad);
                                st(psi, i0, p0); st(psi, i1, p1); st(lam, i0, l0); st(lam, i1, l1);
                            }
                        }
                        __syncthreads();
                        if constexpr (3 * qq + 2 < KW) { xyz[3 * qq] = Xa; xyz[3 * qq + 1] = Ya; xyz[3 * qq + 2] = Za; }
                        if constexpr (3 * qq + 5 < KW) { xyz[3 * qq + 3] = Xb; xyz[3 * qq + 4] = Yb; xyz[3 * qq + 5] = Zb; }
                    }
                });
                // block sum of the KW per-thread values: wave butterfly, then the four waves through LDS
                lane_reduce<KW, 6>(xyz, lane);
                if (lane < KW) scratch[wv * KW + lane] = xyz[0];
                __syncthreads();
                if (threadIdx.x < KW)
                    part_b[(long)sub * KW + threadIdx.x] = (scratch[threadIdx.x] + scratch[KW + threadIdx.x]) +
                                                           (scratch[2 * KW + threadIdx.x] + scratch[3 * KW + threadIdx.x]);
                __syncthreads();
            }
            col -= ne;
            for (int j = ne - 1; j >= 0; --j) {                 // RX gates in reverse order, one per pass
                const int q = j % n;
                const double4 ud = dagger(rx_su2(cs_b[col + j]));
                double X = 0, Yd = 0, Zd = 0;
                const int np = 1 << (n - 1);
                for (int p = threadIdx.x; p < np; p += kT) {
                    const int i0 = ins0(p, q), i1 = i0 | (1 << q);
                    c2 p0 = ld(psi, i0), p1 = ld(psi, i1), l0 = ld(lam, i0), l1 = ld(lam, i1);
                    inner(p0, p1, l0, l1, X, Yd, Zd);
                    su2(p0, p1, ud); su2(l0, l1, ud);
                    st(psi, i0, p0); st(psi, i1, p1); st(lam, i0, l0); st(lam, i1, l1);
                }
                const double tot = block_sum(X, scratch);        // contains the barriers that order the passes
                if (threadIdx.x == 0) grad_x[b * E + col + j] = tot;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
size_t lds_fwd_smem(int n) { return ((size_t)16 << n) + 64; }
size_t lds_bwd_smem(int n) { return ((size_t)32 << n) + 4 * 64 * sizeof(double); }

int launch_lds_fwd(int n, long B, hipStream_t st, const FwdArgs& a) {
    const size_t smem = lds_fwd_smem(n);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(lds_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024 - 256) != hipSuccess) return QHEA_ELAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL(lds_fwd_kernel, dim3((unsigned)B), dim3(kT), smem, st, n, a.runs, a.B, a.E, a.cs, a.gates, a.off,
                       a.co, a.diag, a.pauli, a.out, a.state_out, a.bias);
    return QHEA_OK;
}

template <int KW>
static int launch_bwd_kw(int n, long B, hipStream_t st, const BwdArgs& a) {
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(lds_bwd_kernel<KW>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) != hipSuccess)
            return QHEA_ELAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL(lds_bwd_kernel<KW>, dim3((unsigned)B), dim3(kT), lds_bwd_smem(n), st, n, a.runs, a.B, a.E, a.blk,
                       a.cs, a.gates, a.off, a.co, a.diag, a.pauli, a.g, a.state_in, a.y, a.bias, a.inv_bt, a.out, a.grad_x,
                       a.partial);
    return QHEA_OK;
}

int launch_lds_bwd(int n, long B, hipStream_t st, const BwdArgs& a) {       // KW must equal padded_3n(n): reduce_kernel's row width
    switch (padded_3n(n)) {
        case 8: return launch_bwd_kw<8>(n, B, st, a);
        case 16: return launch_bwd_kw<16>(n, B, st, a);
        case 32: return launch_bwd_kw<32>(n, B, st, a);
        default: return launch_bwd_kw<64>(n, B, st, a);
    }
}

}  // namespace qhea
