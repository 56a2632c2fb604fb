// hea_zyz.hpp -- second-generation kernels for n <= 5 qubits (the all-lane layout: one amplitude per lane,
// 64 >> n samples per wave), built on two facts measured in round 2:
//
//  * one wave alone on a SIMD issues ONE instruction per ~4.5-5.5 clocks whatever its kind, and at the headline
//    batch (1024 samples = 512 sample groups on 1024 SIMDs) every chain of gates is such a lone wave -- so the time
//    of a chain step is its instruction count.  The fused SU(2) gate of hea_device.hpp costs 8 fp64 operations and
//    4 cross-lane moves per gate.  Writing  RY(c) RZ(b) RY(a) = RZ(alpha) RY(theta) RZ(beta)  (ZYZ Euler form, batch
//    invariant, computed once per call) turns a sub-layer into  [diagonal] [RY on every qubit] [diagonal], and every
//    diagonal commutes through the CNOT ring as a permuted diagonal; so between two consecutive real rotation layers
//    there is exactly ONE per-lane complex phase (4 fp64 operations for the whole sub-layer) and an RY gate costs
//    4 fp64 operations + 4 moves: 24 fp64 operations per sub-layer instead of 40, 6 LDS coefficient reads instead of 10.
//    The per-sample RX encodings stay native RX gates (4 + 4) between two such diagonals.
//  * the (cos, sin)(x/2) of the per-sample encoding angles are computed INSIDE the circuit kernel into an LDS table
//    (by the workgroup's otherwise idle waves in the pipelined backward kernel): no 4.9 MB table written by a prep
//    kernel and read back, no refill stalls in the chains; the prep kernel that remains builds the batch-invariant
//    layer records only (181 KB at cfg 2).
//
// Gradients: the inner products X, Y, Z = Im<lambda|sigma_q|psi> are taken after the RY layer (position C, before
// the ring), where the true state is D_post psi_C with D_post = prod_q RZ(alpha_q); sigma_q conjugated by RZ(alpha_q)
// is a rotation of (X, Y) by alpha_q about Z, batch invariant, applied once per call in the reduce kernel
// (reduce_xyz_block, zyz = true) before the usual 3x3 map to the three angle gradients.  RX gradients are unchanged.
// Verified first in numpy against the oracle (1e-15, scripts/exp/zyz_prototype.py), then by the parity tests.
//
// Layer records (built by prep_zyz_kernel, hea_api.hip): one 1 KB record per layer in circuit order, a layer being
// one RX chunk (<= n encodings) or one ansatz sub-layer, plus a final record:
//   bytes [0, 16 * 2^n)          e^{i Phi_l(k)} as (cos, sin) for basis index k: the diagonal applied BEFORE layer l
//   bytes [512, 512 + 32 n)      ansatz layers: (cos(theta_q/2), -/+ sin(theta_q/2)) for lane-bit 0 / 1 of qubit q
// Reference: same circuit as hea_device.hpp (core/quantum_circuits_tq.py:65-127).
#pragma once
#include "hea_device.hpp"

namespace qhea {

constexpr int kRecBytes = 1024;
constexpr int kRecRy = 512;
constexpr int kZCsBytes = 20480;        // LDS bytes of (cos, sin) table per sample group: (64 >> n) * E * 16 must fit

__host__ __device__ inline bool zyz_eligible(int n, long E) {
    return n >= 2 && n <= 5 && (long)(64 >> n) * E * 16 <= kZCsBytes;
}
// layers of a block list: per block ceil(enc / n) RX chunks then `ld` sub-layers
__host__ __device__ inline int zyz_layer_count(const Runs& r, int n) {
    long L = 0;
    for (int i = 0; i < r.nruns; ++i) L += (long)r.count[i] * ((r.enc[i] + n - 1) / n + r.ld[i]);
    return (int)L;
}

// encoding angle of sample b, column e: given directly (circuit-level calls) or through the frequency layers
// (model-level calls; same arithmetic as prep_model_kernel)
struct AngleSrc {
    const double* x;          // [B, E] or nullptr
    EncDesc enc;
};
__device__ __forceinline__ double enc_angle(const AngleSrc& a, int E, long b, int e) {
    if (a.x) return a.x[b * E + e];
    const int si = e < a.enc.seg[0].ncols ? 0 : 1;
    const EncSeg& sg = a.enc.seg[si];
    const int ee = si ? e - a.enc.seg[0].ncols : e;
    const double v = sg.in[b * sg.width + ee % sg.width];
    return sg.w ? v * sg.w[ee] + sg.b[ee] : v * sg.scale;
}
// rows [s0, s0 + ns) x E of the group's table, spread over `nthreads` threads (tid of them); samples past the batch
// repeat the last one (their lanes carry lambda = 0)
__device__ __forceinline__ void fill_cs(double2* cs, const AngleSrc& src, int E, long b0, long B, int ns, int tid, int nthreads) {
    for (int s = 0; s < ns; ++s) {
        const long b = (b0 + s < B) ? b0 + s : B - 1;
        for (int e = tid; e < E; e += nthreads) {
            double sn, cn;
            sincos(0.5 * enc_angle(src, E, b, e), &sn, &cn);
            cs[s * E + e] = make_double2(cn, sn);
        }
    }
}

// (re, im) <- e^{+-i Phi} (re, im), d = (cos Phi, sin Phi)
template <bool DAGGER>
__device__ __forceinline__ void apply_phase(double& re, double& im, const double2& d) {
    const double s = DAGGER ? -d.y : d.y;
    const double nr = d.x * re - s * im;
    im = d.x * im + s * re;
    re = nr;
}
// RY(theta) = [[c, -s], [s, c]] on lane qubit Q; u = (c, sv) with sv = -s for lane-bit 0 and +s for lane-bit 1
template <int Q, bool DAGGER>
__device__ __forceinline__ void apply_ry(double& re, double& im, const double2& u) {
    const double qr = xchg<(1 << Q)>(re), qi = xchg<(1 << Q)>(im);
    const double sv = DAGGER ? -u.y : u.y;
    re = u.x * re + sv * qr;
    im = u.x * im + sv * qi;
}

// Layer records stream: global -> two stage registers (prefetch distance two layers) -> two-slot wave-private LDS
// ring -> this lane's coefficients, read one layer ahead.
template <int N>
struct LayerStream {
    rsrc_t rsrc;
    char* ring;                 // 2 x kRecBytes, wave-private
    unsigned lane16, dgoff;
    unsigned voff[N];
    int L;                      // records 0 .. L
    u32x4 st0, st1;             // record r travels in st[r & 1]
    double2 dg;                 // diagonal entry to apply next
    double2 ry[N];              // RY coefficients of the ansatz layer they were last read for

    __device__ __forceinline__ void init(const char* table, int bytes, char* ring_wave, int lane, int klow, int nlayers) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(table), 0, bytes, 0x00020000);
        ring = ring_wave;
        lane16 = (unsigned)lane * 16u;
        dgoff = (unsigned)klow * 16u;
        L = nlayers;
        static_for<0, N>([&](auto q) {
            constexpr int Q = decltype(q)::value;
            voff[Q] = kRecRy + Q * 32 + (((unsigned)lane >> Q) & 1u) * 16u;
        });
    }
    __device__ __forceinline__ u32x4 gload(int l) const {
        l = l < 0 ? 0 : (l > L ? L : l);
        return __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane16, l * kRecBytes, 0);
    }
    __device__ __forceinline__ char* buf(int l) const { return ring + ((l & 1) ? kRecBytes : 0); }
    __device__ __forceinline__ void park_and_fetch(int lp, int lf) {        // record lp: stage -> LDS; fetch record lf (same parity)
        if (lp & 1) { *reinterpret_cast<u32x4*>(buf(lp) + lane16) = st1; st1 = gload(lf); }
        else        { *reinterpret_cast<u32x4*>(buf(lp) + lane16) = st0; st0 = gload(lf); }
    }
    __device__ __forceinline__ double2 rd_diag(int l) const { return *reinterpret_cast<const double2*>(buf(l) + dgoff); }
    template <int Q>
    __device__ __forceinline__ double2 rd_ry(int l) const { return *reinterpret_cast<const double2*>(buf(l) + voff[Q]); }
    __device__ __forceinline__ void rd_all_ry(int l) { static_for<0, N>([&](auto q) { ry[decltype(q)::value] = rd_ry<decltype(q)::value>(l); }); }

    // walk with step D = +1 (forward) / -1 (reverse) starting at record l0: l0 parked and read, l0 + D and l0 + 2D in flight
    template <int D>
    __device__ __forceinline__ void prime(int l0) {
        if (l0 & 1) { st1 = gload(l0); *reinterpret_cast<u32x4*>(buf(l0) + lane16) = st1; st1 = gload(l0 + 2 * D); st0 = gload(l0 + D); }
        else        { st0 = gload(l0); *reinterpret_cast<u32x4*>(buf(l0) + lane16) = st0; st0 = gload(l0 + 2 * D); st1 = gload(l0 + D); }
        dg = rd_diag(l0);
        rd_all_ry(l0);
    }
    // top of the step that works on record l: bring record l + D into LDS, fetch record l + 3D
    template <int D>
    __device__ __forceinline__ void begin(int l) { park_and_fetch(l + D, l + 3 * D); }
};

// ---------------------------------------------------------------------------------------
// forward sweep
// ---------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void zyz_forward(double (&re)[1], double (&im)[1], const Runs& runs, LayerStream<N>& ls,
                                            const double2* __restrict__ csrow, int E, int lane, int klow, int ring_fwd) {
    re[0] = klow == 0 ? 1.0 : 0.0;
    im[0] = 0.0;
    int l = 0, col = 0;
    ls.template prime<1>(0);
    double2 nxt[N];                                   // (cos, sin) of the next RX chunk, read one block ahead
    auto prefetch = [&](int c) {
        static_for<0, N>([&](auto q) {
            constexpr int Q = decltype(q)::value;
            const int e = c + Q;
            nxt[Q] = csrow[e < E ? e : (E > 0 ? E - 1 : 0)];
        });
    };
    if (E > 0) prefetch(0);
    for (int ri = 0; ri < runs.nruns; ++ri) {
        const int ne = runs.enc[ri], nld = runs.ld[ri];
        for (int rep = 0; rep < runs.count[ri]; ++rep) {
            for (int j0 = 0; j0 < ne; j0 += N) {
                const int m = (ne - j0) < N ? (ne - j0) : N;
                ls.template begin<1>(l);
                apply_phase<false>(re[0], im[0], ls.dg);
                ls.dg = ls.rd_diag(l + 1);
                for_gates_below<N>(m, [&](auto q) {
                    constexpr int Q = decltype(q)::value;
                    apply_rx<N, Q>(re, im, nxt[Q].x, nxt[Q].y);
                });
                prefetch(col + j0 + m);               // columns are consumed in order: the next chunk starts here
                ls.rd_all_ry(l + 1);                  // an RX chunk uses none: read the next record's in one go
                ++l;
            }
            col += ne;
            for (int s = 0; s < nld; ++s) {
                ls.template begin<1>(l);
                apply_phase<false>(re[0], im[0], ls.dg);
                ls.dg = ls.rd_diag(l + 1);
                static_for<0, N>([&](auto q) {
                    constexpr int Q = decltype(q)::value;
                    apply_ry<Q, false>(re[0], im[0], ls.ry[Q]);
                    ls.ry[Q] = ls.template rd_ry<Q>(l + 1);
                });
                re[0] = lane_gather(re[0], ring_fwd);
                im[0] = lane_gather(im[0], ring_fwd);
                ++l;
            }
        }
    }
    apply_phase<false>(re[0], im[0], ls.dg);          // record L: what is still pending after the last layer
}

struct ZFwdArgs {
    Runs runs; long B; int E; const char* rec; int rec_bytes; int L; AngleSrc src; double off, co;
    const double* diag; int pauli; double* out; double* state_out; const double* bias;
};
struct ZBwdArgs {
    Runs runs; long B; int E; int blk; const char* rec; int rec_bytes; int L; AngleSrc src; double off, co;
    const double* diag; int pauli; const double* g; const double* state_in; const double* y; const double* bias;
    double inv_bt; double* out; double* grad_x; double* partial; int* status;
};

template <int N>
__global__ __launch_bounds__(kWaves * 64) void fwd_zyz_kernel(ZFwdArgs a) {
    using C = Cfg<N>;
    static_assert(C::R == 1, "all-lane layout");
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];                 // kWaves x SPW x E (cos, sin)
    __shared__ __attribute__((aligned(16))) char rec_ring[kWaves * 2 * kRecBytes];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long wave = (long)blockIdx.x * kWaves + wib;
    const long b_raw = wave * C::SPW + (lane >> C::LB);
    const bool valid = b_raw < a.B;
    const long b = valid ? b_raw : a.B - 1;
    const int klow = lane & (C::LANES - 1);
    const int ring_fwd = ring_source<N>(lane, false);

    double2* cs = reinterpret_cast<double2*>(dyn_lds) + (long)wib * C::SPW * a.E;
    fill_cs(cs, a.src, a.E, wave * C::SPW, a.B, C::SPW, lane, 64);                // wave-private: LDS is in-order per wave
    LayerStream<N> ls;
    ls.init(a.rec, a.rec_bytes, rec_ring + wib * 2 * kRecBytes, lane, klow, a.L);

    double re[1], im[1];
    zyz_forward<N>(re, im, a.runs, ls, cs + (lane >> C::LB) * a.E, a.E, lane, klow, ring_fwd);

    if (a.state_out && valid)
        reinterpret_cast<double2*>(a.state_out)[(b << N) + klow] = make_double2(re[0], im[0]);
    basis_change<N, false>(re, im, a.pauli, lane);
    double v[1] = {ham_weight<N>(klow, a.off, a.co, a.diag) * (re[0] * re[0] + im[0] * im[0])};
    lane_reduce<1, C::LB>(v, lane);
    if (valid && klow == 0) a.out[b] = v[0] + (a.bias ? a.bias[0] : 0.0);
}

// ---------------------------------------------------------------------------------------
// psi / lambda / sigma pipelined backward kernel, ZYZ form.  Roles, rings, counters and failure reporting as in
// bwd_tri_kernel (hea_device.hpp); a "step" is one layer (one ansatz sub-layer or one RX chunk).
// ---------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(128 + 64 * kSigmaWaves) void bwd_ztri_kernel(ZBwdArgs a) {
    using C = Cfg<N>;
    static_assert(C::R == 1 && kSigmaWaves == 2, "all-lane layout, two sigma waves");
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];                 // SPW x E (cos, sin)
    __shared__ __attribute__((aligned(16))) char rec_ring[2 * 2 * kRecBytes];
    __shared__ double2 psi_ring[kPairRing][64];
    __shared__ double2 lam_ring[kPairRing][64];
    __shared__ double2 psi_final[64];
    __shared__ TriSync sync;

    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // 0: psi, 1: lambda, 2..: sigma
    const long wave = blockIdx.x;                                                 // one sample group per workgroup
    const long b_raw = wave * C::SPW + (lane >> C::LB);
    const bool valid = b_raw < a.B;
    const long b = valid ? b_raw : a.B - 1;
    const int klow = lane & (C::LANES - 1);
    const int E = a.E;

    if (threadIdx.x == 0) {
        sync.psi_prod = 0; sync.lam_prod = 0; sync.ready = 0; sync.abort = 0;
        for (int w = 0; w < kSigmaWaves; ++w) sync.cursor[w] = w;
    }
    double2* cs = reinterpret_cast<double2*>(dyn_lds);
    fill_cs(cs, a.src, E, wave * C::SPW, a.B, C::SPW, (int)threadIdx.x, 128 + 64 * kSigmaWaves);   // all four waves
    __syncthreads();

    if (role < 2) {
        // ------------------------------------------------------------------ psi / lambda chains
        __builtin_amdgcn_s_setprio(3);
        const int ring_fwd = ring_source<N>(lane, false);
        const int ring_rev = ring_source<N>(lane, true);
        const double2* csrow = cs + (lane >> C::LB) * E;
        LayerStream<N> ls;
        ls.init(a.rec, a.rec_bytes, rec_ring + role * 2 * kRecBytes, lane, klow, a.L);
        double sr[1], si[1];
        int seen[kSigmaWaves];
#pragma unroll
        for (int w = 0; w < kSigmaWaves; ++w) seen[w] = 0;
        if (role == 0) {
            if (a.state_in) {
                const double2 s0 = reinterpret_cast<const double2*>(a.state_in)[(b << N) + klow];
                sr[0] = s0.x; si[0] = s0.y;
            } else {
                zyz_forward<N>(sr, si, a.runs, ls, csrow, E, lane, klow, ring_fwd);
            }
            psi_final[lane] = make_double2(sr[0], si[0]);
            __hip_atomic_store(&sync.ready, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            int seen_ready = 0;
            pair_wait_ge(&sync.ready, 1, &sync.abort, seen_ready);
            double fr[1] = {psi_final[lane].x}, fi[1] = {psi_final[lane].y};
            basis_change<N, false>(fr, fi, a.pauli, lane);
            const double h = ham_weight<N>(klow, a.off, a.co, a.diag);
            double gb;
            {
                double v[1] = {h * (fr[0] * fr[0] + fi[0] * fi[0])};
                lane_reduce<1, C::LB>(v, lane);
                const double pred = v[0] + (a.bias ? a.bias[0] : 0.0);
                if (a.out && valid && klow == 0) a.out[b] = pred;
                gb = a.y ? 2.0 * (pred - a.y[b]) * a.inv_bt : a.g[b];
            }
            if (!valid) gb = 0.0;
            sr[0] = gb * h * fr[0]; si[0] = gb * h * fi[0];
            basis_change<N, true>(sr, si, a.pauli, lane);
        }
        double2 (*ring)[64] = role == 0 ? psi_ring : lam_ring;
        int* prod = role == 0 ? &sync.psi_prod : &sync.lam_prod;
        int step = 0;
        auto publish = [&]() {
            if (step >= kPairRing) {
#pragma unroll
                for (int w = 0; w < kSigmaWaves; ++w)
                    pair_wait_ge(&sync.cursor[w], step - kPairRing + 1, &sync.abort, seen[w]);
            }
            ring[step & (kPairRing - 1)][lane] = make_double2(sr[0], si[0]);
            ++step;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // LDS executes a wave's instructions in order
            __hip_atomic_store(prod, step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        // reverse walk over the layers: [diagonal of record l+1]^-1, ring^-1 (ansatz), publish, [layer l]^-1
        int l = a.L - 1, col = E;
        ls.template prime<-1>(a.L);                        // record L read: dg = final diagonal
        {   // bring record L-1 in and read its coefficients (prime read the RY slots of record L: unused padding)
            ls.template begin<-1>(a.L);
            ls.rd_all_ry(a.L - 1);
        }
        double2 nxt[N];
        for (int ri = a.runs.nruns - 1; ri >= 0; --ri) {
            const int ne = a.runs.enc[ri], nld = a.runs.ld[ri];
            const int nch = (ne + N - 1) / N;
            const int m_last = ne - (nch - 1) * N;
            for (int rep = 0; rep < a.runs.count[ri]; ++rep) {
                if (ne > 0) {                               // the block's last RX chunk is used after its sub-layers
                    const int c0 = col - m_last;
                    static_for<0, N>([&](auto q) {
                        constexpr int Q = decltype(q)::value;
                        const int e = c0 + Q;
                        nxt[Q] = csrow[e < E ? e : E - 1];
                    });
                }
                for (int s = nld - 1; s >= 0; --s) {
                    apply_phase<true>(sr[0], si[0], ls.dg);
                    ls.dg = ls.rd_diag(l);
                    sr[0] = lane_gather(sr[0], ring_rev);
                    si[0] = lane_gather(si[0], ring_rev);
                    publish();                              // state after this sub-layer's RY layer
                    ls.template begin<-1>(l);
                    static_rfor<0, N>([&](auto q) {
                        constexpr int Q = decltype(q)::value;
                        apply_ry<Q, true>(sr[0], si[0], ls.ry[Q]);
                        ls.ry[Q] = ls.template rd_ry<Q>(l - 1);
                    });
                    --l;
                }
                for (int ch = nch - 1; ch >= 0; --ch) {
                    const int m = ch == nch - 1 ? m_last : N;
                    if (ch != nch - 1) {
                        const int c0 = col - ne + ch * N;
                        static_for<0, N>([&](auto q) { nxt[decltype(q)::value] = csrow[c0 + decltype(q)::value]; });
                    }
                    apply_phase<true>(sr[0], si[0], ls.dg);
                    ls.dg = ls.rd_diag(l);
                    publish();                              // state after this RX chunk
                    ls.template begin<-1>(l);
                    rfor_gates_below<N>(m, [&](auto q) {
                        constexpr int Q = decltype(q)::value;
                        apply_rx<N, Q>(sr, si, nxt[Q].x, -nxt[Q].y);
                    });
                    ls.rd_all_ry(l - 1);                    // an RX chunk uses none: read the next record's in one go
                    --l;
                }
                col -= ne;
            }
        }
    } else {
        // ------------------------------------------------------------------ sigma waves: inner products + sums
        double* __restrict__ part_w = a.partial + wave * (long)a.blk * C::KW;
        const int me = role - 2;
        int seen_p = 0, seen_l = 0;
        int col = E, sub = a.blk, step = 0;
        for (int ri = a.runs.nruns - 1; ri >= 0; --ri) {
            const int ne = a.runs.enc[ri], nld = a.runs.ld[ri];
            const int nch = (ne + N - 1) / N;
            const int m_last = ne - (nch - 1) * N;
            for (int rep = 0; rep < a.runs.count[ri]; ++rep) {
                for (int s = nld - 1; s >= 0; --s) {
                    --sub;
                    if ((step & 1) != me) { ++step; continue; }
                    pair_wait_ge(&sync.psi_prod, step + 1, &sync.abort, seen_p);
                    pair_wait_ge(&sync.lam_prod, step + 1, &sync.abort, seen_l);
                    const double2* slot = psi_ring[step & (kPairRing - 1)];
                    const double2 p = slot[lane];
                    double2 qv[N];
                    static_for<0, N>([&](auto q) { qv[decltype(q)::value] = slot[lane ^ (1 << decltype(q)::value)]; });
                    const double2 lm = lam_ring[step & (kPairRing - 1)][lane];
                    __hip_atomic_store(&sync.cursor[me], step + kSigmaWaves, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    ++step;
                    double acc3[C::KW];
#pragma unroll
                    for (int i = 0; i < C::KW; ++i) acc3[i] = 0.0;
                    static_for<0, N>([&](auto q) {
                        constexpr int Q = decltype(q)::value;
                        const double sg = ((lane >> Q) & 1) ? -1.0 : 1.0;
                        acc3[3 * Q] = lm.x * qv[Q].y - lm.y * qv[Q].x;
                        acc3[3 * Q + 1] = -sg * (lm.x * qv[Q].x + lm.y * qv[Q].y);
                        acc3[3 * Q + 2] = sg * (lm.x * p.y - lm.y * p.x);
                    });
                    lane_reduce<C::KW, 6>(acc3, lane);
                    if (lane < C::KW) part_w[(long)sub * C::KW + lane] = acc3[0];
                }
                for (int ch = nch - 1; ch >= 0; --ch) {
                    const int m = ch == nch - 1 ? m_last : N;
                    if ((step & 1) != me) { ++step; continue; }
                    pair_wait_ge(&sync.psi_prod, step + 1, &sync.abort, seen_p);
                    pair_wait_ge(&sync.lam_prod, step + 1, &sync.abort, seen_l);
                    const double2* slot = psi_ring[step & (kPairRing - 1)];
                    double2 qv[N];
                    static_for<0, N>([&](auto q) { qv[decltype(q)::value] = slot[lane ^ (1 << decltype(q)::value)]; });
                    const double2 lm = lam_ring[step & (kPairRing - 1)][lane];
                    __hip_atomic_store(&sync.cursor[me], step + kSigmaWaves, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    ++step;
                    double gx[C::KX];
#pragma unroll
                    for (int i = 0; i < C::KX; ++i) gx[i] = 0.0;
                    for_gates_below<N>(m, [&](auto q) {
                        constexpr int Q = decltype(q)::value;
                        gx[Q] = lm.x * qv[Q].y - lm.y * qv[Q].x;
                    });
                    store_grad_x<N>(gx, lane, wave, a.B, E, a.grad_x, col - ne + ch * N, m);
                }
                col -= ne;
            }
        }
    }
    report_abort(&sync.abort, a.status, lane);
}

// launch entry points (hea_inst.hip, n <= 5 only)
#ifdef QHEA_ZSUBSET     // development / test builds link a subset of the qubit counts (see the Makefile)
#define QHEA_FOR_EACH_ZN(X) QHEA_ZSUBSET(X)
#else
#define QHEA_FOR_EACH_ZN(X) X(2) X(3) X(4) X(5)
#endif
#define QHEA_ZDECLARE(NN)                                                              \
    void launch_fwd_zyz_##NN(dim3 grid, size_t dyn_lds, hipStream_t st, const ZFwdArgs& a); \
    void launch_bwd_ztri_##NN(dim3 grid, size_t dyn_lds, hipStream_t st, const ZBwdArgs& a);
QHEA_FOR_EACH_ZN(QHEA_ZDECLARE)
#undef QHEA_ZDECLARE

}  // namespace qhea
