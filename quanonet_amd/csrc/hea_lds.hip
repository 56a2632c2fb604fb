// hea_lds.hip -- workgroup-resident variant of the HEA simulator for the largest qubit counts (n = 10..12).
//
// The wave-resident kernels (hea_device.hpp, built for n <= 9) keep 2^(n-6) amplitudes per lane in VGPRs; with psi
// and lambda live that exceeds the 256 architectural VGPRs for n >= 11 (backward) / n = 12 (forward) and the compiler
// spills to scratch.  Here ONE workgroup of 2^(n-4) threads owns one sample; the state(s) rest in LDS
// (n = 12: 64 KB per state, psi + lambda = 128 of the CU's 160 KB) and every thread works on 16 (forward) or 8
// (backward) amplitudes at a time in registers:
//
//   * a gate layer (the n fused SU(2) gates of a sub-layer) is ceil(n/LG) passes, LG = 4 or 3; pass p loads, per
//     thread, the 2^LG amplitudes that differ in index bits A..A+LG-1 (A = LG p, or n-LG for a ragged last pass),
//     applies the gates of those qubits as in-register 2x2 updates and stores them back -- one LDS round trip
//     of the state per LG gates;
//   * a block's RX encodings are folded, per sample, into the fused gates of its first sub-layer
//     (U RX(theta) is again an SU(2) matrix), so they cost no layer of their own; their gradient is
//     n . (X,Y,Z) of that sub-layer's inner products with n = axis of U X U^dagger (rotated_x_axis);
//     only encodings beyond n per block, or a block without sub-layers, still run as RX layers;
//   * the CNOT ring is a permutation of the basis index that is linear over GF(2), so it costs no pass of
//     its own: the last pass of a forward sub-layer scatters its amplitudes to ring(k), the first pass of a
//     reverse sub-layer gathers from ring(k); per amplitude that is ONE xor with a compile-time constant
//     on top of a per-thread base computed once per kernel;
//   * LDS index swizzle (phys): for every pass the 16-byte accesses of 16 neighbouring lanes fall into 16
//     different bank quads (a thread's 16 amplitudes would otherwise sit 256 B apart for A = 0).
//
// Same circuit, same fused SU(2) table, same adjoint recipe (X,Y,Z inner products taken after the fused gate,
// mapped to the three angles in reduce_kernel) as the wave-resident kernels.  Bound by fp64 FMA issue
// (16 FMAs per amplitude pair and gate; 44 in the reverse sweep) plus the LDS round trip of each pass, which
// the barriers keep from overlapping with the arithmetic when only one workgroup fits a CU (backward, n = 12).
// Measured, cfg 5 (n = 12, Net40-2-20-2, B = 1024): forward 1.90 ms, forward + backward 8.47 ms (the previous
// gate-pair-per-pass LDS kernels: 6.6 / 21.8 ms).
#include "hea_device.hpp"

namespace qhea {

namespace {

struct c2 { double x, y; };

constexpr int kFwdLG = 4;      // gate qubits per pass (see LCfg): forward kernel
#ifndef QHEA_LDS_BWD_LG
#define QHEA_LDS_BWD_LG 3
#endif
constexpr int kBwdLG = QHEA_LDS_BWD_LG;      // backward kernel

// LG = log2(amplitudes a thread holds per pass) = gate qubits per pass.  Forward kernel: 4 (256 threads at n = 12,
// two workgroups per CU).  Backward kernel: 3 (512 threads at n = 12, 248 VGPRs): psi + lambda fill the LDS, so one
// workgroup per CU, and the larger workgroup gives two waves per SIMD at the price of a fourth pass per layer --
// measured the two cancel almost exactly (forward + backward, LG = 3 / 4: n = 10 193 / 200 us, n = 11 410 / 415 us
// per 12 sub-layers, cfg 5 8.46 / 8.47 ms), LG = 3 kept.
template <int N, int LG>
struct LCfg {
    static_assert(N >= 10 && N <= 12, "workgroup-resident kernels: n = 10..12");
    static_assert(LG == 3 || LG == 4, "3 or 4 gate qubits per pass");
    static constexpr int M = 1 << LG;                 // amplitudes per thread and pass
    static constexpr int T = 1 << (N - LG);           // threads per sample
    static constexpr int NW = T / 64;                 // waves
    static constexpr int NP = (N + LG - 1) / LG;      // passes per gate layer
    static constexpr int DIM = 1 << N;
    static constexpr int KW = Cfg<N>::KW;             // padded 3n = row width of `partial` (reduce_kernel)
    static constexpr size_t STATE_BYTES = (size_t)16 << N;
    static constexpr size_t SCRATCH_BYTES = (size_t)NW * 64 * sizeof(double) + 16 * sizeof(double4);   // sums + gate table
};
template <int N, int P, int LG>
struct Pass {
    static constexpr int Q0 = LG * P;                                  // gate qubits [Q0, Q1)
    static constexpr int Q1 = (LG * P + LG < N) ? LG * P + LG : N;
    static constexpr int A = (LG * P + LG <= N) ? LG * P : N - LG;     // lowest of the LG index bits held per thread
};

// Which passes need a workgroup barrier between them.  thread_part maps the lane bits of a thread (t bits 0..5) and
// the pass's LG local bits onto index bits [0, 6 + LG) whenever A <= 6, and the wave number onto the bits above: the
// passes with A <= 6 (LG = 3: A = 0, 3, 6; LG = 4: A = 0, 4) all work on the SAME 2^(6+LG) amplitudes per wave.  Between
// two of them only the wave's own LDS accesses have to stay in order -- which the LDS pipe does for one wave's
// instructions -- so the waves of a workgroup drift apart there and one wave's LDS round trip overlaps another's
// arithmetic (with a barrier after every pass all waves load, compute and store in step: no overlap with one
// workgroup per CU).  Only the last pass (index bits above 6 + LG) exchanges amplitudes between waves.
template <int N, int LG, int PA, int PB>
constexpr bool wave_local_passes() { return Pass<N, PA, LG>::A <= 6 && Pass<N, PB, LG>::A <= 6; }
template <bool WAVE_LOCAL>
__device__ __forceinline__ void pass_sync() {
    if constexpr (WAVE_LOCAL) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // compiler ordering only: stores before ...
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");       // ... the next pass's loads
    } else {
        __syncthreads();
    }
}

// LDS index swizzle (an involution, linear over GF(2)): whatever the pass, the 16-byte accesses of 16 neighbouring
// lanes fall into 16 different bank quads.  LG = 4: bits 0..3 ^= bits 4..7.  LG = 3 (passes at bits 0, 3, 6, 9 or a
// ragged last one): bits 0..2 ^= bits 4..6 and bit 3 ^= bit 6.
template <int LG>
__host__ __device__ constexpr int phys(int k) {
    return LG == 4 ? (k ^ ((k >> 4) & 15)) : (k ^ ((k >> 4) & 7) ^ (((k >> 6) & 1) << 3));
}
// CNOT ring as a map of basis indices: |k> -> |ring(k)>, CNOT(control (i+1)%n, target i) for i = 0..n-1 in order
template <int N>
__host__ __device__ constexpr int ring_dst(int k) {
    for (int i = 0; i < N; ++i) k ^= ((k >> ((i + 1) % N)) & 1) << i;
    return k;
}
// index bits of thread t for a pass with base bit A (the LG bits A..A+LG-1 are the per-thread local index j)
template <int A, int LG>
__device__ __forceinline__ int thread_part(int t) { return ((t >> A) << (A + LG)) | (t & ((1 << A) - 1)); }

// [[a,b],[-conj b, conj a]] on (p0,p1); u = (ar, ai, br, bi)
__device__ __forceinline__ void su2(c2& p0, c2& p1, const double4& u) {
    const c2 a0 = p0, a1 = p1;
    p0.x = u.x * a0.x - u.y * a0.y + u.z * a1.x - u.w * a1.y;
    p0.y = u.x * a0.y + u.y * a0.x + u.z * a1.y + u.w * a1.x;
    p1.x = u.x * a1.x + u.y * a1.y - u.z * a0.x - u.w * a0.y;
    p1.y = u.x * a1.y - u.y * a1.x - u.z * a0.y + u.w * a0.x;
}
__device__ __forceinline__ double4 dagger(double4 u) { return make_double4(u.x, -u.y, -u.z, -u.w); }
// RX as an SU(2) in the same form: a = c, b = -i s  ->  (c, 0, 0, -s)
__device__ __forceinline__ double4 rx_su2(double2 cs) { return make_double4(cs.x, 0.0, 0.0, -cs.y); }

// merge_rx / rotated_x_axis (the RX fold) are shared with the wave-resident kernels: hea_device.hpp

// Per-sample gate coefficients of a layer (folded RX) are computed ONCE per layer by threads 0..n-1 into a
// small LDS table and read back (broadcast) right before use.  Computing them inline in every thread made
// the compiler keep all of a layer's merged gates live and spill ~1500 registers (measured: 10x slower).
template <int N, class F>
__device__ __forceinline__ void set_gates(double4* gtab, F f) {
    if (threadIdx.x < N) gtab[threadIdx.x] = f((int)threadIdx.x);
    __syncthreads();
}
// Im<l|sigma|p> contributions of one pair (p0,p1),(l0,l1), accumulated as FMA chains: 4 fp64 instructions per product
// (written as sums of differences the compiler may not reassociate: mul, fma, mul, fma, add, add)
__device__ __forceinline__ void inner_x(const c2& p0, const c2& p1, const c2& l0, const c2& l1, double& X) {
    X = fma(l0.x, p1.y, X); X = fma(-l0.y, p1.x, X); X = fma(l1.x, p0.y, X); X = fma(-l1.y, p0.x, X);
}
__device__ __forceinline__ void inner(const c2& p0, const c2& p1, const c2& l0, const c2& l1, double& X, double& Y,
                                      double& Z) {
    inner_x(p0, p1, l0, l1, X);
    Y = fma(-l0.x, p1.x, Y); Y = fma(-l0.y, p1.y, Y); Y = fma(l1.x, p0.x, Y); Y = fma(l1.y, p0.y, Y);
    Z = fma(l0.x, p0.y, Z); Z = fma(-l0.y, p0.x, Z); Z = fma(-l1.x, p1.y, Z); Z = fma(l1.y, p1.x, Z);
}

// the 2^LG amplitudes of this thread for a pass with base bit A: local index j <-> index bits A..A+LG-1.
// `base` = phys(thread_part) (in place) or phys(ring(thread_part)) (through the ring); both maps are linear
// over GF(2), so amplitude j sits at base ^ constant_j.
template <int N, int A, bool RING, int LG>
__device__ __forceinline__ void load_group(const double2* s, int base, c2 (&v)[1 << LG]) {
    static_for<0, (1 << LG)>([&](auto jj) {
        constexpr int J = decltype(jj)::value;
        constexpr int CJ = RING ? phys<LG>(ring_dst<N>(J << A)) : phys<LG>(J << A);
        const double2 a = s[base ^ CJ];
        v[J].x = a.x; v[J].y = a.y;
    });
}
template <int N, int A, bool RING, int LG>
__device__ __forceinline__ void store_group(double2* s, int base, const c2 (&v)[1 << LG]) {
    static_for<0, (1 << LG)>([&](auto jj) {
        constexpr int J = decltype(jj)::value;
        constexpr int CJ = RING ? phys<LG>(ring_dst<N>(J << A)) : phys<LG>(J << A);
        s[base ^ CJ] = make_double2(v[J].x, v[J].y);
    });
}
template <int LBIT, int LG>
__device__ __forceinline__ void apply_group(c2 (&v)[1 << LG], const double4& u) {
    static_for<0, (1 << LG)>([&](auto jj) {
        constexpr int J = decltype(jj)::value;
        if constexpr (!(J & (1 << LBIT))) su2(v[J], v[J | (1 << LBIT)], u);
    });
}

template <int N, int LG>
struct Bases {                        // per-thread LDS index bases, computed once per kernel
    int plain[LCfg<N, LG>::NP];       // phys(thread_part<A_p>(t))
    int ring;                         // phys(ring(thread_part<A_last>(t)))
    __device__ __forceinline__ void init(int t) {
        static_for<0, LCfg<N, LG>::NP>([&](auto p) {
            constexpr int P = decltype(p)::value;
            plain[P] = phys<LG>(thread_part<Pass<N, P, LG>::A, LG>(t));
        });
        ring = phys<LG>(ring_dst<N>(thread_part<Pass<N, LCfg<N, LG>::NP - 1, LG>::A, LG>(t)));
    }
};

// One forward gate layer.  gate(q, u) fills u and returns whether qubit q has a gate (wave-uniform).
// RING: the last pass scatters through the CNOT ring.
template <int N, int LG, bool RING, class G>
__device__ __forceinline__ void fwd_layer(double2* s, const Bases<N, LG>& bs, G gate) {
    constexpr int NP = LCfg<N, LG>::NP;
    static_for<0, NP>([&](auto p) {
        constexpr int P = decltype(p)::value;
        using PS = Pass<N, P, LG>;
        c2 v[1 << LG];
        load_group<N, PS::A, false, LG>(s, bs.plain[P], v);
        static_for<PS::Q0, PS::Q1>([&](auto q) {
            constexpr int Q = decltype(q)::value;
            double4 u;
            if (gate(Q, u)) apply_group<Q - PS::A, LG>(v, u);
        });
        if constexpr (RING && P == NP - 1) {
            __syncthreads();                               // every thread holds its amplitudes: safe to permute
            store_group<N, PS::A, true, LG>(s, bs.ring, v);
            __syncthreads();
        } else {
            store_group<N, PS::A, false, LG>(s, bs.plain[P], v);
            // the next pass: P + 1, or pass 0 of the following layer
            pass_sync<wave_local_passes<N, LG, P, (P + 1 < NP ? P + 1 : 0)>()>();
        }
    });
}

// One reverse layer on psi and lambda: (ring^-1 via a gather in the first pass when RING), then per qubit the
// inner products Im<lam|sigma|psi> (XYZ: all three, else X only) followed by the adjoint gate on both states.
// gate(q, u) returns the ADJOINT coefficients.  acc: XYZ ? [3q..3q+2] : [q].
template <int N, int LG, bool RING, bool XYZ, class G, int NA>
__device__ __forceinline__ void bwd_layer(double2* psi, double2* lam, const Bases<N, LG>& bs, G gate, double (&acc)[NA]) {
    constexpr int NP = LCfg<N, LG>::NP;
    static_rfor<0, NP>([&](auto p) {
        constexpr int P = decltype(p)::value;
        using PS = Pass<N, P, LG>;
        c2 v[1 << LG], l[1 << LG];
        if constexpr (RING && P == NP - 1) {
            load_group<N, PS::A, true, LG>(psi, bs.ring, v);
            load_group<N, PS::A, true, LG>(lam, bs.ring, l);
            __syncthreads();                               // all gathers done before anything is overwritten
        } else {
            load_group<N, PS::A, false, LG>(psi, bs.plain[P], v);
            load_group<N, PS::A, false, LG>(lam, bs.plain[P], l);
        }
        static_for<PS::Q0, PS::Q1>([&](auto q) {
            constexpr int Q = decltype(q)::value;
            constexpr int LBIT = Q - PS::A;
            double4 u;
            if (gate(Q, u)) {
                static_for<0, (1 << LG)>([&](auto jj) {
                    constexpr int J = decltype(jj)::value;
                    if constexpr (!(J & (1 << LBIT))) {
                        constexpr int J1 = J | (1 << LBIT);
                        if constexpr (XYZ) inner(v[J], v[J1], l[J], l[J1], acc[3 * Q], acc[3 * Q + 1], acc[3 * Q + 2]);
                        else inner_x(v[J], v[J1], l[J], l[J1], acc[Q]);
                        su2(v[J], v[J1], u);
                        su2(l[J], l[J1], u);
                    }
                });
            }
        });
        store_group<N, PS::A, false, LG>(psi, bs.plain[P], v);
        store_group<N, PS::A, false, LG>(lam, bs.plain[P], l);
        // the next pass: P - 1, or the last pass of the following (earlier) layer
        pass_sync<(P > 0) && wave_local_passes<N, LG, P, (P > 0 ? P - 1 : 0)>()>();
    });
}

__device__ __forceinline__ double4 gate_u(const char* gates, int n, int sub, int q) {   // base variant of gate (sub,q)
    return *reinterpret_cast<const double4*>(gates + ((long)(sub + 1) * n + q) * kGateBytes);
}

template <int N>
__device__ __forceinline__ double ham_w(int k, double off, double co, const double* __restrict__ diag) {
    return diag ? diag[k] : off + co * (double)(N - 2 * (int)__popc((unsigned)k));
}

// sum of one double per thread over the workgroup (fixed order), result in every thread
template <int N, int LG>
__device__ __forceinline__ double block_sum(double v, double* scratch) {
    double t[1] = {v};
    lane_reduce<1, 6>(t, threadIdx.x & 63);
    if constexpr (LCfg<N, LG>::NW == 1) return t[0];
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = t[0];
    __syncthreads();
    double tot = scratch[0];
#pragma unroll
    for (int w = 1; w < LCfg<N, LG>::NW; ++w) tot += scratch[w];
    __syncthreads();                                       // scratch is reused by the caller
    return tot;
}

// readout-basis change on every qubit (see basis_change in hea_device.hpp): X -> RY(-pi/2), Y -> RX(+pi/2)
template <int N, int LG>
__device__ __forceinline__ void basis_lds(double2* s, const Bases<N, LG>& bs, int pauli, bool dag) {
    if (pauli == 0) return;
    const double r = 0.70710678118654752440, sr = dag ? -r : r;
    const double4 uc = pauli == 1 ? make_double4(r, 0.0, sr, 0.0) : make_double4(r, 0.0, 0.0, -sr);
    fwd_layer<N, LG, false>(s, bs, [&](int, double4& u) { u = uc; return true; });
}

template <int N, int LG>
__device__ __forceinline__ void forward_lds(double2* psi, double4* gtab, const Bases<N, LG>& bs, const Runs& runs,
                                            const double2* __restrict__ cs_b, const char* __restrict__ gates) {
    using L = LCfg<N, LG>;
#pragma unroll
    for (int j = 0; j < L::M; ++j) {
        const int p = threadIdx.x + j * L::T;
        psi[p] = make_double2(p == 0 ? 1.0 : 0.0, 0.0);              // |0..0>: phys(0) = 0
    }
    __syncthreads();
    int col = 0, sub = 0;
    for (int ri = 0; ri < runs.nruns; ++ri) {
        const int ne = runs.enc[ri], nld = runs.ld[ri];
        for (int rep = 0; rep < runs.count[ri]; ++rep) {
            // RX(x[col+j]) on wire j % n, n wires per layer; the last layer is folded into the first sub-layer
            const int nchunks = (ne + N - 1) / N;
            const int nsep = (nld > 0 && ne > 0) ? nchunks - 1 : nchunks;
            for (int ch = 0; ch < nsep; ++ch) {
                const int j0 = ch * N;
                const int m = (ne - j0) < N ? (ne - j0) : N;
                const double2* c = cs_b + col + j0;
                fwd_layer<N, LG, false>(psi, bs, [&](int q, double4& u) {
                    if (q >= m) return false;
                    u = rx_su2(c[q]);
                    return true;
                });
            }
            if (nsep < nchunks) {
                const int j0 = nsep * N;
                const int m = ne - j0;
                const double2* c = cs_b + col + j0;
                set_gates<N>(gtab, [&](int q) {
                    return merge_rx(gate_u(gates, N, sub, q), q < m ? c[q] : make_double2(1.0, 0.0));
                });
                fwd_layer<N, LG, true>(psi, bs, [&](int q, double4& u) { u = gtab[q]; return true; });
                ++sub;
            }
            col += ne;
            for (int l = (nsep < nchunks) ? 1 : 0; l < nld; ++l, ++sub)
                fwd_layer<N, LG, true>(psi, bs, [&](int q, double4& u) { u = gate_u(gates, N, sub, q); return true; });
        }
    }
}

}  // namespace

template <int N>
__global__ __launch_bounds__((LCfg<N, kFwdLG>::T)) void lds_fwd_kernel(Runs runs, long B, int E, const double2* __restrict__ cs,
                                                             const char* __restrict__ gates, double off, double co,
                                                             const double* __restrict__ diag, int pauli,
                                                             double* __restrict__ out, double* __restrict__ state_out,
                                                             const double* __restrict__ bias) {
    constexpr int LG = kFwdLG;
    using L = LCfg<N, LG>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2* psi = reinterpret_cast<double2*>(smem);
    double* scratch = reinterpret_cast<double*>(smem + L::STATE_BYTES);
    double4* gtab = reinterpret_cast<double4*>(scratch + L::NW * 64);
    const long b = blockIdx.x;
    Bases<N, LG> bs;
    bs.init(threadIdx.x);
    forward_lds<N, LG>(psi, gtab, bs, runs, cs + b * E, gates);
    if (state_out) {
#pragma unroll
        for (int j = 0; j < L::M; ++j) {
            const int k = threadIdx.x + j * L::T;
            reinterpret_cast<double2*>(state_out)[(b << N) + k] = psi[phys<LG>(k)];
        }
    }
    basis_lds<N, LG>(psi, bs, pauli, false);
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < L::M; ++j) {
        const int k = threadIdx.x + j * L::T;
        const double2 a = psi[phys<LG>(k)];
        acc += ham_w<N>(k, off, co, diag) * (a.x * a.x + a.y * a.y);
    }
    const double tot = block_sum<N, LG>(acc, scratch);
    if (threadIdx.x == 0) out[b] = tot + (bias ? bias[0] : 0.0);
}

// Backward: one row of `partial` per sample ([B][blk][KW]); grad_x written directly.
template <int N>
__global__ __launch_bounds__((LCfg<N, kBwdLG>::T)) void lds_bwd_kernel(Runs runs, long B, int E, int blk,
                                                             const double2* __restrict__ cs,
                                                             const char* __restrict__ gates, double off, double co,
                                                             const double* __restrict__ diag, int pauli,
                                                             const double* __restrict__ g,
                                                             const double* __restrict__ state_in,
                                                             const double* __restrict__ y,
                                                             const double* __restrict__ bias, double inv_bt,
                                                             double* __restrict__ out, double* __restrict__ grad_x,
                                                             double* __restrict__ partial) {
    constexpr int LG = kBwdLG;
    using L = LCfg<N, LG>;
    constexpr int KW = L::KW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2* psi = reinterpret_cast<double2*>(smem);
    double2* lam = psi + L::DIM;
    double* scratch = reinterpret_cast<double*>(smem + 2 * L::STATE_BYTES);        // [NW][64]
    double4* gtab = reinterpret_cast<double4*>(scratch + L::NW * 64);
    const long b = blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const double2* __restrict__ cs_b = cs + b * E;
    Bases<N, LG> bs;
    bs.init(threadIdx.x);

    if (state_in) {
#pragma unroll
        for (int j = 0; j < L::M; ++j) {
            const int k = threadIdx.x + j * L::T;
            psi[phys<LG>(k)] = reinterpret_cast<const double2*>(state_in)[(b << N) + k];
        }
        __syncthreads();
    } else {
        forward_lds<N, LG>(psi, gtab, bs, runs, cs_b, gates);
    }
    basis_lds<N, LG>(psi, bs, pauli, false);
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < L::M; ++j) {
        const int k = threadIdx.x + j * L::T;
        const double2 a = psi[phys<LG>(k)];
        acc += ham_w<N>(k, off, co, diag) * (a.x * a.x + a.y * a.y);
    }
    const double pred = block_sum<N, LG>(acc, scratch) + (bias ? bias[0] : 0.0);
    if (out && threadIdx.x == 0) out[b] = pred;
    const double gb = y ? 2.0 * (pred - y[b]) * inv_bt : g[b];
#pragma unroll
    for (int j = 0; j < L::M; ++j) {
        const int k = threadIdx.x + j * L::T;
        const double2 a = psi[phys<LG>(k)];
        const double h = gb * ham_w<N>(k, off, co, diag);
        lam[phys<LG>(k)] = make_double2(h * a.x, h * a.y);
    }
    __syncthreads();
    basis_lds<N, LG>(psi, bs, pauli, true);
    basis_lds<N, LG>(lam, bs, pauli, true);

    double* __restrict__ part_b = partial + b * (long)blk * KW;
    int col = E, sub = blk;
    for (int ri = runs.nruns - 1; ri >= 0; --ri) {
        const int ne = runs.enc[ri], nld = runs.ld[ri];
        const int nchunks = (ne + N - 1) / N;
        for (int rep = 0; rep < runs.count[ri]; ++rep) {
            const int nsep = (nld > 0 && ne > 0) ? nchunks - 1 : nchunks;      // see forward_lds
            const int m_merged = nsep < nchunks ? ne - nsep * N : 0;
            col -= ne;
            for (int l = nld - 1; l >= 0; --l) {
                --sub;
                const int mm = l == 0 ? m_merged : 0;           // encodings folded into this sub-layer's gates
                const double2* c = cs_b + col + nsep * N;
                double xyz[KW];
#pragma unroll
                for (int i = 0; i < KW; ++i) xyz[i] = 0.0;
                set_gates<N>(gtab, [&](int q) {
                    return dagger(merge_rx(gate_u(gates, N, sub, q), q < mm ? c[q] : make_double2(1.0, 0.0)));
                });
                bwd_layer<N, LG, true, true>(psi, lam, bs, [&](int q, double4& u) { u = gtab[q]; return true; }, xyz);
                // workgroup sum of the 3n per-thread values: wave butterfly, then the waves through LDS
                lane_reduce<KW, 6>(xyz, lane);               // lane i holds the wave total of value i
                if (lane < KW) scratch[wv * 64 + lane] = xyz[0];
                __syncthreads();
                if (threadIdx.x < KW) {
                    double tot = scratch[threadIdx.x];
#pragma unroll
                    for (int w = 1; w < L::NW; ++w) tot += scratch[w * 64 + threadIdx.x];
                    part_b[(long)sub * KW + threadIdx.x] = tot;
                }
                if (threadIdx.x < mm) {                      // encoding gradient of the folded RX on wire threadIdx.x
                    const int q = threadIdx.x;
                    double X = 0.0, Y = 0.0, Z = 0.0;
#pragma unroll
                    for (int w = 0; w < L::NW; ++w) {
                        X += scratch[w * 64 + 3 * q]; Y += scratch[w * 64 + 3 * q + 1]; Z += scratch[w * 64 + 3 * q + 2];
                    }
                    double nx, ny, nz;
                    rotated_x_axis(gate_u(gates, N, sub, q), nx, ny, nz);
                    grad_x[b * E + col + nsep * N + q] = nx * X + ny * Y + nz * Z;
                }
                __syncthreads();
            }
            for (int ch = nsep - 1; ch >= 0; --ch) {            // remaining RX layers in reverse order
                const int j0 = ch * N;
                const int m = (ne - j0) < N ? (ne - j0) : N;
                const double2* c = cs_b + col + j0;
                double gx[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) gx[i] = 0.0;
                bwd_layer<N, LG, false, false>(psi, lam, bs, [&](int q, double4& u) {
                    if (q >= m) return false;
                    u = dagger(rx_su2(c[q]));
                    return true;
                }, gx);
                lane_reduce<16, 6>(gx, lane);                // lane l holds the wave total of value l & 15
                if constexpr (L::NW == 1) {
                    if (lane < m) grad_x[b * E + col + j0 + lane] = gx[0];
                } else {
                    if (lane < 16) scratch[wv * 64 + lane] = gx[0];
                    __syncthreads();
                    if (threadIdx.x < m) {
                        double tot = scratch[threadIdx.x];
#pragma unroll
                        for (int w = 1; w < L::NW; ++w) tot += scratch[w * 64 + threadIdx.x];
                        grad_x[b * E + col + j0 + threadIdx.x] = tot;
                    }
                    __syncthreads();
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
namespace {

template <int N>
int launch_fwd_n(long B, hipStream_t st, const FwdArgs& a) {
    using L = LCfg<N, kFwdLG>;
    constexpr size_t smem = L::STATE_BYTES + L::SCRATCH_BYTES;
    // every launch (microseconds against a millisecond kernel): the attribute is per device, and a process may
    // drive more than one
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(lds_fwd_kernel<N>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return QHEA_ELAUNCH;
    hipLaunchKernelGGL(lds_fwd_kernel<N>, dim3((unsigned)B), dim3(L::T), smem, st, a.runs, a.B, a.E, a.cs, a.gates,
                       a.off, a.co, a.diag, a.pauli, a.out, a.state_out, a.bias);
    return QHEA_OK;
}

template <int N>
int launch_bwd_n(long B, hipStream_t st, const BwdArgs& a) {
    using L = LCfg<N, kBwdLG>;
    constexpr size_t smem = 2 * L::STATE_BYTES + L::SCRATCH_BYTES;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(lds_bwd_kernel<N>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return QHEA_ELAUNCH;
    hipLaunchKernelGGL(lds_bwd_kernel<N>, dim3((unsigned)B), dim3(L::T), smem, st, a.runs, a.B, a.E, a.blk, a.cs,
                       a.gates, a.off, a.co, a.diag, a.pauli, a.g, a.state_in, a.y, a.bias, a.inv_bt, a.out, a.grad_x,
                       a.partial);
    return QHEA_OK;
}

}  // namespace

bool lds_supported(int n) { return n >= 10 && n <= 12; }

int launch_lds_fwd(int n, long B, hipStream_t st, const FwdArgs& a) {
    switch (n) {
        case 10: return launch_fwd_n<10>(B, st, a);
        case 11: return launch_fwd_n<11>(B, st, a);
        case 12: return launch_fwd_n<12>(B, st, a);
        default: return QHEA_EUNSUPPORTED;
    }
}

int launch_lds_bwd(int n, long B, hipStream_t st, const BwdArgs& a) {
    switch (n) {
        case 10: return launch_bwd_n<10>(B, st, a);
        case 11: return launch_bwd_n<11>(B, st, a);
        case 12: return launch_bwd_n<12>(B, st, a);
        default: return QHEA_EUNSUPPORTED;
    }
}

}  // namespace qhea
