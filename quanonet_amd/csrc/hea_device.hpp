// hea_device.hpp -- MI355X (gfx950 / CDNA4) batched HEA statevector simulator.
//
// Design ("wave-resident state"): one sample's 2^n complex fp64 amplitudes live in the
// VGPRs of ONE wavefront for the whole circuit.  Qubits 0..LB-1 (LB = min(n,6)) are
// mapped to lane-index bits, qubits LB..n-1 to a per-lane register index, and for n < 6
// a wave carries 64/2^n independent samples side by side.  A one-qubit gate on a lane
// qubit is a cross-lane exchange (DPP / ds_swizzle / ds_bpermute, no memory) followed by
// 8 fp64 FMAs per amplitude; on a register qubit it is pure in-lane arithmetic.  The state
// never touches LDS or HBM between gates; HBM traffic is inputs + outputs only.
//
// What is restated (reference file:line):
//   circuit     core/quantum_circuits_tq.py:79-104 (== core/quantum_circuits_ms.py:164-226)
//   read-out    core/quantum_circuits_tq.py:106-127
//   gradient    adjoint differentiation as behind MindQuantum's get_expectation_with_grad
//               (core/quantum_circuits_ms.py:229-233); replaces torch autograd through
//               TorchQuantum's per-gate bmm ops (solvers/solver_pt.py:235)
// Fusions (batch-invariant, rebuilt every call by prep_kernel):
//   RY(w2)*RZ(w1)*RY(w0) on one wire -> one SU(2) matrix  U = [[a, b], [-conj b, conj a]].
//   Its three angle gradients are linear in X,Y,Z = Im<lambda|sigma_{x,y,z}|psi> taken
//   after U; the 3x3 map is applied once per call in reduce_kernel.
//
// No MFMA: gate application is a strided 2x2 update, not a contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

#include "quanonet_hea.h"

namespace qhea {

constexpr int kWaves = 4;              // waves per workgroup
constexpr int kMaxRuns = 16;           // run-length-encoded (count, enc, ld) block list

struct Runs {
    int nruns;
    int count[kMaxRuns];
    int enc[kMaxRuns];
    int ld[kMaxRuns];
};

template <int N>
struct Cfg {
    static constexpr int LB = N < 6 ? N : 6;        // lane bits per sample
    static constexpr int RB = N - LB;               // register bits
    static constexpr int R = 1 << RB;               // amplitudes per lane
    static constexpr int SPW = 64 >> LB;            // samples per wave
    static constexpr int LANES = 1 << LB;           // lanes per sample
    static constexpr int KW = (3 * N <= 8) ? 8 : (3 * N <= 16) ? 16 : (3 * N <= 32) ? 32 : 64;  // padded 3N
    static constexpr int KX = (N <= 2) ? 2 : (N <= 4) ? 4 : (N <= 8) ? 8 : 16;                   // padded N
};

__host__ __device__ constexpr int padded_3n(int n) {
    return (3 * n <= 8) ? 8 : (3 * n <= 16) ? 16 : (3 * n <= 32) ? 32 : 64;
}

// ---------------------------------------------------------------------------------------
// compile-time loops
// ---------------------------------------------------------------------------------------
template <int I, int END, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < END) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, END>(f);
    }
}
template <int I, int END, class F>
__device__ __forceinline__ void static_rfor(F&& f) {     // END-1 down to I
    if constexpr (I < END) {
        f(std::integral_constant<int, END - 1>{});
        static_rfor<I, END - 1>(f);
    }
}

// ---------------------------------------------------------------------------------------
// cross-lane exchange with lane ^ MASK (MASK a single bit, 1..32)
// ---------------------------------------------------------------------------------------
template <int MASK>
__device__ __forceinline__ int xchg_i32(int v) {
    if constexpr (MASK == 1) {
        return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, false);          // quad_perm [1,0,3,2]
    } else if constexpr (MASK == 2) {
        return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, false);          // quad_perm [2,3,0,1]
    } else if constexpr (MASK == 4) {
        int t = __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, false);        // row_half_mirror: ^7
        return __builtin_amdgcn_mov_dpp(t, 0x1B, 0xF, 0xF, false);          // quad_perm [3,2,1,0]: ^3
    } else if constexpr (MASK == 8) {
        return __builtin_amdgcn_mov_dpp(v, 0x128, 0xF, 0xF, false);         // row_ror:8
    } else if constexpr (MASK == 16) {
        return __builtin_amdgcn_ds_swizzle(v, 0x401F);                      // bit-mode xor 0x10
    } else {
        static_assert(MASK == 32, "single-bit lane mask expected");
        const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        return __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, v);
    }
}
template <int MASK>
__device__ __forceinline__ double xchg(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = xchg_i32<MASK>(lo);
    hi = xchg_i32<MASK>(hi);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_gather(double v, int src_lane_x4) {
    int lo = __builtin_amdgcn_ds_bpermute(src_lane_x4, __double2loint(v));
    int hi = __builtin_amdgcn_ds_bpermute(src_lane_x4, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// Transposing butterfly: K values per lane summed over lane bits [0, BITS).  On return the
// lane whose low bits are j (j < min(K, 2^BITS)) holds in v[i] the total of value (i << BITS) | j.
template <int K, int BITS, int T = 0>
__device__ __forceinline__ void lane_reduce(double (&v)[K], int lane) {
    if constexpr (T < BITS) {
        constexpr int C = (K >> T) > 0 ? (K >> T) : 1;       // live values before this step
        if constexpr (C > 1) {
            const bool up = (lane >> T) & 1;
#pragma unroll
            for (int i = 0; i < C / 2; ++i) {
                const double keep = up ? v[2 * i + 1] : v[2 * i];
                const double send = up ? v[2 * i] : v[2 * i + 1];
                v[i] = keep + xchg<(1 << T)>(send);
            }
        } else {
            v[0] += xchg<(1 << T)>(v[0]);
        }
        lane_reduce<K, BITS, T + 1>(v, lane);
    }
}

// ---------------------------------------------------------------------------------------
// gate application on the wave-resident state
// ---------------------------------------------------------------------------------------
// SU(2) gate [[a,b],[-conj b, conj a]] on qubit Q.  The adjoint is the same call with
// (ar,-ai,-br,-bi).
template <int N, int Q>
__device__ __forceinline__ void apply_su2(double (&re)[Cfg<N>::R], double (&im)[Cfg<N>::R],
                                          double ar, double ai, double br, double bi, int lane) {
    using C = Cfg<N>;
    if constexpr (Q < C::LB) {
        const double s = ((lane >> Q) & 1) ? -1.0 : 1.0;
        const double sai = s * ai, sbr = s * br;
#pragma unroll
        for (int r = 0; r < C::R; ++r) {
            const double pr = re[r], pi = im[r];
            const double qr = xchg<(1 << Q)>(pr), qi = xchg<(1 << Q)>(pi);
            re[r] = ar * pr - sai * pi + sbr * qr - bi * qi;
            im[r] = ar * pi + sai * pr + sbr * qi + bi * qr;
        }
    } else {
        constexpr int J = 1 << (Q - C::LB);
#pragma unroll
        for (int r0 = 0; r0 < C::R; ++r0) {
            if (r0 & J) continue;
            const int r1 = r0 | J;
            const double p0r = re[r0], p0i = im[r0], p1r = re[r1], p1i = im[r1];
            re[r0] = ar * p0r - ai * p0i + br * p1r - bi * p1i;
            im[r0] = ar * p0i + ai * p0r + br * p1i + bi * p1r;
            re[r1] = ar * p1r + ai * p1i - br * p0r - bi * p0i;     // conj(a) p1 - conj(b) p0
            im[r1] = ar * p1i - ai * p1r - br * p0i + bi * p0r;
        }
    }
}

// RX(theta) = [[c,-is],[-is,c]] with c = cos(theta/2), s = sin(theta/2); adjoint: s -> -s.
template <int N, int Q>
__device__ __forceinline__ void apply_rx(double (&re)[Cfg<N>::R], double (&im)[Cfg<N>::R],
                                         double c, double s) {
    using C = Cfg<N>;
    if constexpr (Q < C::LB) {
#pragma unroll
        for (int r = 0; r < C::R; ++r) {
            const double pr = re[r], pi = im[r];
            const double qr = xchg<(1 << Q)>(pr), qi = xchg<(1 << Q)>(pi);
            re[r] = c * pr + s * qi;
            im[r] = c * pi - s * qr;
        }
    } else {
        constexpr int J = 1 << (Q - C::LB);
#pragma unroll
        for (int r0 = 0; r0 < C::R; ++r0) {
            if (r0 & J) continue;
            const int r1 = r0 | J;
            const double p0r = re[r0], p0i = im[r0], p1r = re[r1], p1i = im[r1];
            re[r0] = c * p0r + s * p1i;
            im[r0] = c * p0i - s * p1r;
            re[r1] = c * p1r + s * p0i;
            im[r1] = c * p1i - s * p0r;
        }
    }
}

// CNOT(control=CQ, target=TQ): new[k] = old[k ^ (bit_CQ(k) << TQ)]
template <int N, int CQ, int TQ>
__device__ __forceinline__ void apply_cnot(double (&re)[Cfg<N>::R], double (&im)[Cfg<N>::R], int lane) {
    using C = Cfg<N>;
    if constexpr (CQ < C::LB && TQ < C::LB) {
        const bool on = (lane >> CQ) & 1;
#pragma unroll
        for (int r = 0; r < C::R; ++r) {
            const double qr = xchg<(1 << TQ)>(re[r]), qi = xchg<(1 << TQ)>(im[r]);
            re[r] = on ? qr : re[r];
            im[r] = on ? qi : im[r];
        }
    } else if constexpr (CQ < C::LB) {                 // control on a lane bit, target in registers
        constexpr int J = 1 << (TQ - C::LB);
        const bool on = (lane >> CQ) & 1;
#pragma unroll
        for (int r0 = 0; r0 < C::R; ++r0) {
            if (r0 & J) continue;
            const int r1 = r0 | J;
            const double a0r = re[r0], a0i = im[r0], a1r = re[r1], a1i = im[r1];
            re[r0] = on ? a1r : a0r;  im[r0] = on ? a1i : a0i;
            re[r1] = on ? a0r : a1r;  im[r1] = on ? a0i : a1i;
        }
    } else if constexpr (TQ < C::LB) {                 // control in registers, target on a lane bit
        constexpr int J = 1 << (CQ - C::LB);
#pragma unroll
        for (int r = 0; r < C::R; ++r) {
            if (!(r & J)) continue;
            re[r] = xchg<(1 << TQ)>(re[r]);
            im[r] = xchg<(1 << TQ)>(im[r]);
        }
    } else {                                           // both in registers: rename
        constexpr int JC = 1 << (CQ - C::LB), JT = 1 << (TQ - C::LB);
#pragma unroll
        for (int r = 0; r < C::R; ++r) {
            if ((r & JC) && !(r & JT)) {
                double t = re[r]; re[r] = re[r | JT]; re[r | JT] = t;
                t = im[r]; im[r] = im[r | JT]; im[r | JT] = t;
            }
        }
    }
}

// entangler ring: for i = 0..N-1 in order CNOT(control=(i+1)%N, target=i); REVERSE undoes it.
template <int N, bool REVERSE>
__device__ __forceinline__ void apply_ring(double (&re)[Cfg<N>::R], double (&im)[Cfg<N>::R],
                                           int lane, int ring_src_x4) {
    using C = Cfg<N>;
    if constexpr (C::RB == 0) {                        // whole state on lanes: one gather
        re[0] = lane_gather(re[0], ring_src_x4);
        im[0] = lane_gather(im[0], ring_src_x4);
    } else if constexpr (!REVERSE) {
        static_for<0, N>([&](auto i) { apply_cnot<N, (decltype(i)::value + 1) % N, decltype(i)::value>(re, im, lane); });
    } else {
        static_rfor<0, N>([&](auto i) { apply_cnot<N, (decltype(i)::value + 1) % N, decltype(i)::value>(re, im, lane); });
    }
}

// source lane (x4) of the composite ring permutation for the all-lane layout
template <int N>
__device__ __forceinline__ int ring_source(int lane, bool reverse) {
    int k = lane & ((1 << N) - 1);
    if (!reverse) {     // final[k] = old[f0(f1(...f_{N-1}(k)))]
        for (int i = N - 1; i >= 0; --i) k ^= ((k >> ((i + 1) % N)) & 1) << i;
    } else {            // inverse: f_{N-1}(...f0(k))
        for (int i = 0; i < N; ++i) k ^= ((k >> ((i + 1) % N)) & 1) << i;
    }
    return ((lane & ~((1 << N) - 1)) | k) << 2;
}

// per-lane partial sums of Im<lam|sigma|psi> for sigma = X,Y,Z on qubit Q
template <int N, int Q>
__device__ __forceinline__ void pauli_inner(const double (&pr)[Cfg<N>::R], const double (&pi)[Cfg<N>::R],
                                            const double (&lr)[Cfg<N>::R], const double (&li)[Cfg<N>::R],
                                            int lane, double& X, double& Y, double& Z) {
    using C = Cfg<N>;
    double x = 0.0, y = 0.0, z = 0.0;
    if constexpr (Q < C::LB) {
#pragma unroll
        for (int r = 0; r < C::R; ++r) {
            const double qr = xchg<(1 << Q)>(pr[r]), qi = xchg<(1 << Q)>(pi[r]);
            x += lr[r] * qi - li[r] * qr;
            y += lr[r] * qr + li[r] * qi;
            z += lr[r] * pi[r] - li[r] * pr[r];
        }
        const double s = ((lane >> Q) & 1) ? -1.0 : 1.0;
        X = x; Y = -s * y; Z = s * z;
    } else {
        constexpr int J = 1 << (Q - C::LB);
#pragma unroll
        for (int r0 = 0; r0 < C::R; ++r0) {
            if (r0 & J) continue;
            const int r1 = r0 | J;
            x += (lr[r0] * pi[r1] - li[r0] * pr[r1]) + (lr[r1] * pi[r0] - li[r1] * pr[r0]);
            y += -(lr[r0] * pr[r1] + li[r0] * pi[r1]) + (lr[r1] * pr[r0] + li[r1] * pi[r0]);
            z += (lr[r0] * pi[r0] - li[r0] * pr[r0]) - (lr[r1] * pi[r1] - li[r1] * pr[r1]);
        }
        X = x; Y = y; Z = z;
    }
}
template <int N, int Q>
__device__ __forceinline__ double pauli_x_inner(const double (&pr)[Cfg<N>::R], const double (&pi)[Cfg<N>::R],
                                                const double (&lr)[Cfg<N>::R], const double (&li)[Cfg<N>::R]) {
    using C = Cfg<N>;
    double x = 0.0;
    if constexpr (Q < C::LB) {
#pragma unroll
        for (int r = 0; r < C::R; ++r) {
            const double qr = xchg<(1 << Q)>(pr[r]), qi = xchg<(1 << Q)>(pi[r]);
            x += lr[r] * qi - li[r] * qr;
        }
    } else {
        constexpr int J = 1 << (Q - C::LB);
#pragma unroll
        for (int r0 = 0; r0 < C::R; ++r0) {
            if (r0 & J) continue;
            const int r1 = r0 | J;
            x += (lr[r0] * pi[r1] - li[r0] * pr[r1]) + (lr[r1] * pi[r0] - li[r1] * pr[r0]);
        }
    }
    return x;
}

template <int N>
__device__ __forceinline__ double ham_weight(int k, double off, double co, const double* __restrict__ diag) {
    if (diag) return diag[k];
    return off + co * (double)(N - 2 * (int)__popc((unsigned)k));
}

// ---------------------------------------------------------------------------------------
// forward sweep (shared by the forward and backward kernels)
// ---------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void forward_sweep(double (&re)[Cfg<N>::R], double (&im)[Cfg<N>::R],
                                              const Runs& runs, const double2* __restrict__ cs_b,
                                              const double4* __restrict__ U, int lane, int ring_fwd) {
    using C = Cfg<N>;
#pragma unroll
    for (int r = 0; r < C::R; ++r) { re[r] = 0.0; im[r] = 0.0; }
    if ((lane & (C::LANES - 1)) == 0) re[0] = 1.0;
    int col = 0, sub = 0;
    for (int ri = 0; ri < runs.nruns; ++ri) {
        const int ne = runs.enc[ri], nld = runs.ld[ri];
        for (int rep = 0; rep < runs.count[ri]; ++rep) {
            for (int j0 = 0; j0 < ne; j0 += N) {
                static_for<0, N>([&](auto q) {
                    if (j0 + decltype(q)::value < ne) {
                        const double2 c = cs_b[col + j0 + decltype(q)::value];
                        apply_rx<N, decltype(q)::value>(re, im, c.x, c.y);
                    }
                });
            }
            col += ne;
            for (int l = 0; l < nld; ++l, ++sub) {
                const double4* __restrict__ Us = U + (long)sub * N;
                static_for<0, N>([&](auto q) {
                    const double4 u = Us[decltype(q)::value];
                    apply_su2<N, decltype(q)::value>(re, im, u.x, u.y, u.z, u.w, lane);
                });
                apply_ring<N, false>(re, im, lane, ring_fwd);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(kWaves * 64) void fwd_kernel(Runs runs, long B, int E,
                                                          const double2* __restrict__ cs,
                                                          const double4* __restrict__ U,
                                                          double off, double co,
                                                          const double* __restrict__ diag,
                                                          double* __restrict__ out,
                                                          double* __restrict__ state_out) {
    using C = Cfg<N>;
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * kWaves + (threadIdx.x >> 6);
    const long b_raw = wave * C::SPW + (lane >> C::LB);
    const bool valid = b_raw < B;
    const long b = valid ? b_raw : B - 1;
    const int klow = lane & (C::LANES - 1);
    const int ring_fwd = ring_source<N < 6 ? N : 6>(lane, false);

    double re[C::R], im[C::R];
    forward_sweep<N>(re, im, runs, cs + b * E, U, lane, ring_fwd);

    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < C::R; ++r) {
        const int k = (r << C::LB) | klow;
        acc += ham_weight<N>(k, off, co, diag) * (re[r] * re[r] + im[r] * im[r]);
        if (state_out && valid) {
            double2* dst = reinterpret_cast<double2*>(state_out) + (b << N) + k;
            *dst = make_double2(re[r], im[r]);
        }
    }
    double v[1] = {acc};
    lane_reduce<1, C::LB>(v, lane);
    if (valid && klow == 0) out[b] = v[0];
}

template <int N>
__global__ __launch_bounds__(kWaves * 64) void bwd_kernel(Runs runs, long B, int E, int blk,
                                                          const double2* __restrict__ cs,
                                                          const double4* __restrict__ U,
                                                          double off, double co,
                                                          const double* __restrict__ diag,
                                                          const double* __restrict__ g,
                                                          const double* __restrict__ state_in,
                                                          double* __restrict__ out,
                                                          double* __restrict__ grad_x,
                                                          double* __restrict__ partial) {
    using C = Cfg<N>;
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * kWaves + (threadIdx.x >> 6);
    const long b_raw = wave * C::SPW + (lane >> C::LB);
    const bool valid = b_raw < B;
    const long b = valid ? b_raw : B - 1;
    const int klow = lane & (C::LANES - 1);
    const int ring_fwd = ring_source<N < 6 ? N : 6>(lane, false);
    const int ring_rev = ring_source<N < 6 ? N : 6>(lane, true);
    const double2* __restrict__ cs_b = cs + b * E;

    double pr[C::R], pi[C::R], lr[C::R], li[C::R];
    if (state_in) {
#pragma unroll
        for (int r = 0; r < C::R; ++r) {
            const double2 a = reinterpret_cast<const double2*>(state_in)[(b << N) + ((r << C::LB) | klow)];
            pr[r] = a.x; pi[r] = a.y;
        }
    } else {
        forward_sweep<N>(pr, pi, runs, cs_b, U, lane, ring_fwd);
    }

    const double gb = valid ? g[b] : 0.0;      // padding lanes carry lambda = 0: no gradient contribution
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < C::R; ++r) {
        const double h = ham_weight<N>((r << C::LB) | klow, off, co, diag);
        acc += h * (pr[r] * pr[r] + pi[r] * pi[r]);
        lr[r] = gb * h * pr[r];
        li[r] = gb * h * pi[r];
    }
    if (out) {
        double v[1] = {acc};
        lane_reduce<1, C::LB>(v, lane);
        if (valid && klow == 0) out[b] = v[0];
    }

    double* __restrict__ part_w = partial + wave * (long)blk * C::KW;
    int col = E, sub = blk;
    for (int ri = runs.nruns - 1; ri >= 0; --ri) {
        const int ne = runs.enc[ri], nld = runs.ld[ri];
        for (int rep = 0; rep < runs.count[ri]; ++rep) {
            for (int l = 0; l < nld; ++l) {
                --sub;
                apply_ring<N, true>(pr, pi, lane, ring_rev);
                apply_ring<N, true>(lr, li, lane, ring_rev);
                const double4* __restrict__ Us = U + (long)sub * N;
                double acc3[C::KW];
#pragma unroll
                for (int i = 0; i < C::KW; ++i) acc3[i] = 0.0;
                static_rfor<0, N>([&](auto q) {
                    const double4 u = Us[decltype(q)::value];
                    pauli_inner<N, decltype(q)::value>(pr, pi, lr, li, lane, acc3[3 * decltype(q)::value], acc3[3 * decltype(q)::value + 1],
                                            acc3[3 * decltype(q)::value + 2]);
                    apply_su2<N, decltype(q)::value>(pr, pi, u.x, -u.y, -u.z, -u.w, lane);
                    apply_su2<N, decltype(q)::value>(lr, li, u.x, -u.y, -u.z, -u.w, lane);
                });
                lane_reduce<C::KW, 6>(acc3, lane);
                if (lane < C::KW) part_w[(long)sub * C::KW + lane] = acc3[0];
            }
            col -= ne;
            const int nchunks = (ne + N - 1) / N;
            for (int ch = nchunks - 1; ch >= 0; --ch) {
                const int j0 = ch * N;
                double gx[C::KX];
#pragma unroll
                for (int i = 0; i < C::KX; ++i) gx[i] = 0.0;
                static_rfor<0, N>([&](auto q) {
                    if (j0 + decltype(q)::value < ne) {
                        const double2 c = cs_b[col + j0 + decltype(q)::value];
                        gx[decltype(q)::value] = pauli_x_inner<N, decltype(q)::value>(pr, pi, lr, li);
                        apply_rx<N, decltype(q)::value>(pr, pi, c.x, -c.y);
                        apply_rx<N, decltype(q)::value>(lr, li, c.x, -c.y);
                    }
                });
                lane_reduce<C::KX, C::LB>(gx, lane);
                // lane with klow == j holds value j (KX <= 2^LB for every N)
                if (valid && klow < N && j0 + klow < ne) grad_x[b * E + col + j0 + klow] = gx[0];
            }
        }
    }
}


// ---------------------------------------------------------------------------------------
// launch entry points; one translation unit per qubit count (hea_inst.hip, -DQHEA_N=n)
// ---------------------------------------------------------------------------------------
struct FwdArgs {
    Runs runs; long B; int E; const double2* cs; const double4* U; double off, co;
    const double* diag; double* out; double* state_out;
};
struct BwdArgs {
    Runs runs; long B; int E; int blk; const double2* cs; const double4* U; double off, co;
    const double* diag; const double* g; const double* state_in; double* out; double* grad_x; double* partial;
};

#define QHEA_FOR_EACH_N(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12)
#define QHEA_DECLARE(NN)                                              \
    void launch_fwd_##NN(dim3 grid, hipStream_t st, const FwdArgs& a); \
    void launch_bwd_##NN(dim3 grid, hipStream_t st, const BwdArgs& a);
QHEA_FOR_EACH_N(QHEA_DECLARE)
#undef QHEA_DECLARE

}  // namespace qhea
