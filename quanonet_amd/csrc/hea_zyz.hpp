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
// (reduce_xyz_block, zyz = true) before the usual 3x3 map to the three angle gradients.  RX gradients: the X (wire 4: Y)
// inner product right after the chunk in the generic walk and the one-wave kernel; in the pipeline kernel's block walk
// they are read off the NEXT sub-layer's X, Y, Z through a batch-invariant axis (bwd_ztri_kernel, sigma waves).
// Verified first in numpy against the oracle (1e-15, scripts/exp/zyz_prototype.py), then by the parity tests.
//
// Layer records (built by prep_zyz_kernel, hea_api.hip): one 1 KB record per layer in circuit order, a layer being
// one RX chunk (<= n encodings) or one ansatz sub-layer, plus a final record:
//   bytes [0, 16 * 2^n)          e^{i Phi_l(k)} as (cos, sin) for basis index k: the diagonal applied BEFORE layer l
//   bytes [512, 512 + 32 n)      ansatz layers: (cos(theta_q/2), -/+ sin(theta_q/2)) for lane-bit 0 / 1 of qubit q
//                                full RX chunks: 3 n doubles, the axes n_q of the following sub-layer's gates (see above)
// Kernels: fwd_zyz_kernel / fwd_split_kernel / fwd_zshared_kernel (forward; private ring / split layout / shared ring),
// bwd_ztri_kernel (psi / lambda / sigma pipeline, small batches), bwd_zpacked_kernel (one wave per group, large batches).
// Reference: same circuit as hea_device.hpp (core/quantum_circuits_tq.py:65-127).
#pragma once
#include "hea_device.hpp"
#include "hea_sincos.hpp"

namespace qhea {

constexpr int kRecBytes = 1024;
constexpr int kRecRy = 512;
constexpr int kZCsBytes = 20480;        // LDS bytes of (cos, sin) table per sample group: (64 >> n) * E * 16 must fit

// rows of the (cos, sin) table carry n entries of padding on both sides, so that the run-ahead reads of the
// block-unrolled walk need no clamping at the ends
__host__ __device__ inline long zyz_cs_row(int n, long E) { return E + 2 * n; }
__host__ __device__ inline bool zyz_eligible(int n, long E) {
    return n >= 2 && n <= 5 && (long)(64 >> n) * zyz_cs_row(n, E) * 16 <= kZCsBytes;
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
__device__ __forceinline__ void fill_cs(double2* cs, const AngleSrc& src, int n, int E, long b0, long B, int ns, int tid, int nthreads,
                                        int row_stride = 0 /* 0: zyz_cs_row(n, E) */) {
    const int row = row_stride ? row_stride : (int)zyz_cs_row(n, E);
    // (sample, column) pairs spread over the threads: short rows (cfg 1: E = 20) would leave most threads idle in a loop over
    // columns only, and the sweeps cannot start before the fill is done
    for (int i = tid; i < ns * E; i += nthreads) {
        const int s = i / E, e = i - s * E;
        const long b = (b0 + s < B) ? b0 + s : B - 1;
        double sn, cn;
        fast_sincos(0.5 * enc_angle(src, E, b, e), &sn, &cn);
        cs[s * row + n + e] = make_double2(cn, sn);
    }
}

// The Y and Z inner products carry the lane's sign (-1)^(bit Q of the lane).  A multiplication by +-1.0 costs an fp64
// operation plus the instructions that materialise the +-1.0 from the lane index (the compiler recomputes them inside the
// loops once registers are tight); flipping the sign bit is one 32-bit XOR with a mask that is two integer instructions away.
template <int Q>
__device__ __forceinline__ unsigned lane_sign_mask(int lane) { return ((unsigned)lane << (31 - Q)) & 0x80000000u; }
__device__ __forceinline__ double flip_sign(double v, unsigned mask) {          // -v where the lane's bit is set, v elsewhere
    return __hiloint2double(__double2hiint(v) ^ (int)mask, __double2loint(v));
}

// (re, im) <- e^{+-i Phi} (re, im), d = (cos Phi, sin Phi)
template <bool DAGGER>
__device__ __forceinline__ void apply_phase(double& re, double& im, const double2& d) {
    const double s = DAGGER ? -d.y : d.y;
    const double nr = d.x * re - s * im;
    im = d.x * im + s * re;
    re = nr;
}
// RY(theta) = [[c, -s], [s, c]] on lane qubit Q.
// Q < 4: u = (c, sv) with sv = -s for lane-bit 0 and +s for lane-bit 1 (the lane's variant of the record); the partner
//        comes through DPP.
// Q = 4: the partner is 16 lanes away, outside DPP's reach; ds_swizzle would put 52 clocks of LDS latency on the
//        chain.  v_permlane16_swap_b32 (gfx950) swaps the odd 16-lane rows of one register with the even rows of
//        another: swapping re with im leaves (re_a, re_b) in the lanes of an even row and (im_a, im_b) in those of
//        the odd row, a and b being the two partner amplitudes -- every lane then rotates the PAIR it holds with the
//        SAME (c, s), and a second swap puts the results back: 4 swaps + 4 fp64 operations, all on the vector pipe,
//        no lane-dependent coefficient.  u = (c, s) for every lane.
template <int Q, bool DAGGER>
__device__ __forceinline__ void apply_ry(double& re, double& im, const double2& u) {
    const double sv = DAGGER ? -u.y : u.y;
    if constexpr (Q == 4) {
        const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(re), (unsigned)__double2loint(im), false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(re), (unsigned)__double2hiint(im), false, false);
        const double a = __hiloint2double((int)hi[0], (int)lo[0]), b = __hiloint2double((int)hi[1], (int)lo[1]);
        const double o1 = u.x * a - sv * b, o2 = sv * a + u.x * b;
        const auto l2 = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(o1), (unsigned)__double2loint(o2), false, false);
        const auto h2 = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(o1), (unsigned)__double2hiint(o2), false, false);
        re = __hiloint2double((int)h2[0], (int)l2[0]);
        im = __hiloint2double((int)h2[1], (int)l2[1]);
    } else {
        const double qr = xchg<(1 << Q)>(re), qi = xchg<(1 << Q)>(im);
        re = u.x * re + sv * qr;
        im = u.x * im + sv * qi;
    }
}
// RX(x) of the encoding layers, cs = (cos x/2, sin x/2).  Q < 4: the native gate.  Q = 4: RX = RZ(-pi/2) RY RZ(pi/2), the
// two fixed phases folded into the diagonals before and after the layer by prep_zyz_kernel, so that this gate too is
// the swap-form RY above (its gradient is then the Y inner product, not the X one: sigma waves).
template <int N, int Q, bool DAGGER>
__device__ __forceinline__ void apply_enc(double (&re)[1], double (&im)[1], const double2& cs) {
    if constexpr (Q == 4) apply_ry<4, DAGGER>(re[0], im[0], cs);
    else apply_rx<N, Q>(re, im, cs.x, DAGGER ? -cs.y : cs.y);
}

// Layer records stream: global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`: every lane's 16 bytes land at
// M0 + 16 * lane, no VGPR staging, no ds_write) into a wave-private ring of kSlots records, kDist records ahead of the
// one being worked on; this lane's coefficients are then read one layer ahead into registers.  A layer of the ZYZ
// form is only ~300-400 clocks long, so the one-layer prefetch distance of the first-generation GateStream (two
// stage registers) left the global-load latency exposed at every layer (measured: half of the wave's cycles waiting).
// The DMA and its wait are inline assembly: behind the builtin form the compiler cannot tell which ring slot a
// ds_read touches and drains vmcnt to 0 before every LDS read.  The wave's in-order vmcnt makes the counted wait
// exact: after issuing record l + kDist*D, `vmcnt(kDist-1)` leaves only the kDist-1 youngest in flight, so record
// l + D has landed.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma_record(const u32x4& rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
// NREC consecutive records at once: the instruction offset advances the global AND the LDS address, so one M0 serves all
template <int NREC>
__device__ __forceinline__ void dma_records(const u32x4& rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
    static_assert(NREC >= 1 && NREC <= 3 && kRecBytes == 1024, "instruction offsets are 12 bits");
    if constexpr (NREC == 1)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
    else if constexpr (NREC == 2)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\t"
                     "buffer_load_dwordx4 %1, %2, %3 offen offset:1024 lds"
                     :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
    else
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\t"
                     "buffer_load_dwordx4 %1, %2, %3 offen offset:1024 lds\n\t"
                     "buffer_load_dwordx4 %1, %2, %3 offen offset:2048 lds"
                     :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
template <int CNT>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(CNT) : "memory"); }

constexpr int kSlots = 8, kDist = 6;
// sigma waves per pipeline of bwd_ztri_kernel / bwd_zquad_kernel.  Round 2 settled on three: the sigma waves bounded the reverse
// phase with two, and a fourth did not pay (B = 1024: 94.6 / 90.4 / 92.1 us per launch with 2 / 3 / 4).  That was measured with
// two chain waves sharing a SIMD (pipe_of_wave); with every chain wave on a SIMD of its own FOUR sigma waves per pipeline -- two
// per SIMD beside its one chain wave in the two-pipeline workgroup -- are the best: cfg 2, us per training step with
// 2 / 3 / 4 / 5 / 6 sigma waves: B = 1024 94.1 / 91.4 / 88.9 / 90.0 / 91.5, B = 512 (quad-chain kernel) 89.1 / 76.2 / 74.2 / 76.2 / 77.4
// (scripts/exp/small_batch_steps.py on -DQHEA_ZSIGMA=n builds).  The kernels stay under 128 VGPRs (all of their LDS is dynamic,
// so that the compiler does not cap its occupancy estimate at the LDS limit; the walks keep one layer's coefficients ahead
// instead of a block's), so a CU's twelve waves have room.
#ifndef QHEA_ZSIGMA
#define QHEA_ZSIGMA 4
#endif
constexpr int kZSigma = QHEA_ZSIGMA;
constexpr int kAxisRing = 16;                   // blocks whose RX-gradient axes (15 doubles) are kept for the sigma waves: > ring depth / LD
#ifdef QHEA_PROFILE_WAITS
struct ZSync { int psi_prod, lam_prod, ready, abort; int cursor[4]; unsigned long long waited[24]; int next; };   // abort + 20 bytes -> waited; two entries per wave of the workgroup (up to 12)
#else
struct ZSync { int psi_prod, lam_prod, ready, abort; int cursor[kZSigma]; int next; };
#endif
constexpr int kRecRingBytes = kSlots * kRecBytes;           // per streaming wave

template <int N>
struct LayerStream {
    u32x4 rsrc;
    char* ring;                 // kSlots x kRecBytes, wave-private
    unsigned lds0;              // its LDS byte address (wave-uniform)
    unsigned lane16, dgoff;
    unsigned voff[N];
    int L;                      // records 0 .. L
    double2 dg;                 // diagonal entry to apply next
    double2 ry[N];              // RY coefficients of the ansatz layer they were last read for

    __device__ __forceinline__ void init(const char* table, int bytes, char* ring_wave, int lane, int klow, int nlayers) {
        const unsigned long long base = reinterpret_cast<unsigned long long>(table);
        rsrc = u32x4{(unsigned)base, (unsigned)(base >> 32) & 0xffffu, (unsigned)bytes, 0x00020000u};
        ring = ring_wave;
        lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)ring_wave;
        lane16 = (unsigned)lane * 16u;
        dgoff = (unsigned)klow * 16u;
        L = nlayers;
        static_for<0, N>([&](auto q) {
            constexpr int Q = decltype(q)::value;
            voff[Q] = kRecRy + Q * 32 + (Q == 4 ? 1u : (((unsigned)lane >> Q) & 1u)) * 16u;     // Q = 4: (c, +s) for all lanes
        });
    }
    __device__ __forceinline__ void issue(int l) const {           // records outside [0, L] repeat an end record (never read)
        const int lc = l < 0 ? 0 : (l > L ? L : l);
        dma_record(rsrc, lds0 + (unsigned)(l & (kSlots - 1)) * kRecBytes, lane16, (unsigned)lc * kRecBytes);
    }
    __device__ __forceinline__ char* buf(int l) const { return ring + (l & (kSlots - 1)) * kRecBytes; }
    __device__ __forceinline__ double2 rd_diag(int l) const { return *reinterpret_cast<const double2*>(buf(l) + dgoff); }
    template <int Q>
    __device__ __forceinline__ double2 rd_ry(int l) const { return *reinterpret_cast<const double2*>(buf(l) + voff[Q]); }
    __device__ __forceinline__ void rd_all_ry(int l) { static_for<0, N>([&](auto q) { ry[decltype(q)::value] = rd_ry<decltype(q)::value>(l); }); }

    // walk with step D = +1 (forward) / -1 (reverse) starting at record l0: on return l0 has landed and been read,
    // l0 + D .. l0 + (kDist-1) D are in flight
    template <int D>
    __device__ __forceinline__ void prime(int l0) {
#pragma unroll
        for (int i = 0; i < kDist; ++i) issue(l0 + i * D);
        wait_vmcnt<kDist - 1>();
        dg = rd_diag(l0);
        rd_all_ry(l0);
    }
    // top of the step that works on record l: record l + D has landed when this returns
    template <int D>
    __device__ __forceinline__ void begin(int l) {
        issue(l + kDist * D);
        wait_vmcnt<kDist - 1>();
    }
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
                    apply_enc<N, Q, false>(re, im, nxt[Q]);
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

// ---------------------------------------------------------------------------------------
// Block-unrolled fast path.  Every circuit the reference builds has blocks of ONE full RX chunk (enc = n) followed
// by the same number LD of sub-layers (1 or 2: every script uses 2, the shipped Q2 checkpoint 1).  For those the
// layer loop is unrolled over a whole block: a block's 1 + LD records are fetched together (kBDist blocks ahead),
// every coefficient lives in a register of its own that is refilled with the NEXT block's value right after its last
// use, and the per-layer bookkeeping of the generic walk (slot arithmetic, clamping, kind tests: ~25 scalar and ~8
// address instructions per layer, which cost a lone wave as much issue time as arithmetic does) is paid once per block.
// Other shapes take the generic layer walk above.  The record table is padded on both sides (kPadRecs records,
// never written, never used in arithmetic), so the fetches at the ends need no clamping.
// ---------------------------------------------------------------------------------------
template <int K> using CtSlot = std::integral_constant<int, K>;
struct RtSlot { int b; };
constexpr int kBSlots = 4, kBDist = 2;         // block slots per ring; blocks in flight beyond the one being read
constexpr int kPadRecs = 9;                    // >= (kBDist + 1) * 3 records of padding before record 0 and after record L
constexpr int kBlockRingBytes = kBSlots * 3 * kRecBytes;     // per streaming wave (LD = 2)

// SHARED = false: the ring is private to the wave (chains of the pipeline kernel, forward kernel).
// SHARED = true:  one ring per workgroup, filled by its wave 0 (`loader`); every wave of the workgroup walks the same
//                 records block by block, so ONE barrier per block orders both "block b+1 has landed" and "everybody is
//                 done with the slot about to be overwritten" (bwd_zpacked_kernel: at large batches the LDS a wave
//                 needs decides how many waves a CU holds).
template <int N, int LD, bool SHARED = false>
struct BlockStream {
    static constexpr int RPB = 1 + LD;
    bool loader = true;
    u32x4 rsrc;                 // starts kPadRecs records before record 0
    char* ring;
    unsigned lds0, lane16, a_dg;
    unsigned a_ry[N];
    double2 dg[RPB];            // diagonals of the records of the block at hand
    double2 ry[LD][N];          // its RY coefficients (this lane's variants)
    double2 cs[N];              // its RX (cos, sin)

    __device__ __forceinline__ void init(const char* rec0, int nrec /* L + 1 */, char* ring_wave, int lane, int klow) {
        const unsigned long long base = reinterpret_cast<unsigned long long>(rec0) - (unsigned long long)kPadRecs * kRecBytes;
        rsrc = u32x4{(unsigned)base, (unsigned)(base >> 32) & 0xffffu, (unsigned)((nrec + 2 * kPadRecs) * kRecBytes), 0x00020000u};
        ring = ring_wave;
        lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)ring_wave;
        lane16 = (unsigned)lane * 16u;
        a_dg = (unsigned)klow * 16u;
        static_for<0, N>([&](auto q) {
            constexpr int Q = decltype(q)::value;
            a_ry[Q] = kRecRy + Q * 32 + (Q == 4 ? 1u : (((unsigned)lane >> Q) & 1u)) * 16u;     // Q = 4: (c, +s) for all lanes
        });
    }
    __device__ __forceinline__ void issue(int b) const {             // b in [-(kBDist+1), nblocks + kBDist + 1]
        const unsigned slot = lds0 + (unsigned)(b & (kBSlots - 1)) * (RPB * kRecBytes);
        const unsigned soff = (unsigned)((b * RPB + kPadRecs) * kRecBytes);
        dma_records<RPB>(rsrc, slot, lane16, soff);
    }
    __device__ __forceinline__ const char* slot(int b) const { return ring + (b & (kBSlots - 1)) * (RPB * kRecBytes); }
    // The walks are unrolled over the kBSlots ring slots (CtSlot<K>: the block at hand sits in slot K), so that every
    // ring address is this lane's offset register + an immediate: no slot arithmetic, no address adds per block.  A
    // walk's first / last few blocks, where the block index is not aligned to the unrolled body, use RtSlot.
    template <int D, int K>
    static constexpr int slot_off(CtSlot<K>) { return ((K + D) & (kBSlots - 1)) * (RPB * kRecBytes); }
    template <int D>
    static __device__ __forceinline__ int slot_off(RtSlot s) { return ((s.b + D) & (kBSlots - 1)) * (RPB * kRecBytes); }
    template <int D, class SL>
    __device__ __forceinline__ const char* slot_rel(SL s) const { return ring + slot_off<D>(s); }
    template <int D, class SL>
    __device__ __forceinline__ void ahead_rel(SL s, int b) const {       // fetch block b + D into the slot D ahead of s
        if (!SHARED || loader)
            dma_records<RPB>(rsrc, lds0 + (unsigned)slot_off<D>(s), lane16, (unsigned)(((b + D) * RPB + kPadRecs) * kRecBytes));
    }
    // Per block: landed<D>() at the top (blocks b+D .. b+kBDist*D are in flight, b+D must have landed), ahead<D>(b)
    // later in the block -- in the latency shadow of a ring gather -- fetches block b + (kBDist+1) D.
    __device__ __forceinline__ void landed() const {
        if (!SHARED || loader) wait_vmcnt<(kBDist - 1) * RPB>();
        if constexpr (SHARED) __syncthreads();
    }
    // the same check placed AFTER the block's ahead_rel (one more block in flight)
    __device__ __forceinline__ void landed_late() const {
        if (!SHARED || loader) wait_vmcnt<kBDist * RPB>();
        if constexpr (SHARED) __syncthreads();
    }
    template <int D>
    __device__ __forceinline__ void ahead(int b) const { if (!SHARED || loader) issue(b + (kBDist + 1) * D); }
    template <int D>
    __device__ __forceinline__ void step(int b) const {              // both at once (priming)
        ahead<D>(b);
        if (!SHARED || loader) wait_vmcnt<kBDist * RPB>();
        if constexpr (SHARED) __syncthreads();
    }
    // start a walk at block b0 in direction D: b0 landed, b0+D .. b0+kBDist*D in flight.  `drain`: fetches of a previous
    // walk may still be in flight towards the same slots
    template <int D>
    __device__ __forceinline__ void prime(int b0, bool drain) const {
        if (!SHARED || loader) {
            if (drain) wait_vmcnt<0>();
        }
        if constexpr (SHARED) { if (drain) __syncthreads(); }        // nobody still reads the slots of the previous walk
        if (!SHARED || loader) {
#pragma unroll
            for (int i = 0; i <= kBDist; ++i) issue(b0 + i * D);
            wait_vmcnt<kBDist * RPB>();
        }
        if constexpr (SHARED) __syncthreads();
    }
    static __device__ __forceinline__ double2 rd(const char* p, unsigned off) { return *reinterpret_cast<const double2*>(p + off); }
    __device__ __forceinline__ void load_records(const char* sl) {
#pragma unroll
        for (int i = 0; i < RPB; ++i) dg[i] = rd(sl, i * kRecBytes + a_dg);
#pragma unroll
        for (int s = 0; s < LD; ++s)
            static_for<0, N>([&](auto q) { ry[s][decltype(q)::value] = rd(sl, (1 + s) * kRecBytes + a_ry[decltype(q)::value]); });
    }
    __device__ __forceinline__ void load_cs(const double2* __restrict__ csrow, int c0) {      // c0 in [-N, E]: rows are padded
        const double2* p = csrow + c0;
        static_for<0, N>([&](auto q) { cs[decltype(q)::value] = p[decltype(q)::value]; });
    }
};

__host__ __device__ inline int zyz_fast_ld(const Runs& r, int n) {      // LD of the fast path, 0 = not applicable
    if (r.nruns < 1) return 0;
    const int ld = r.ld[0];
    if (ld != 1 && ld != 2) return 0;
    for (int i = 0; i < r.nruns; ++i)
        if (r.ld[i] != ld || r.enc[i] != n || r.count[i] < 1) return 0;
    return ld;
}

// One layer's coefficients in registers: the diagonal applied before it and its N gates' (variant of this lane).
template <int N>
struct LayerCoef { double2 dg; double2 g[N]; };

template <int N, int LD, class BS>
__device__ __forceinline__ void zyz_forward_fast(double (&re)[1], double (&im)[1], const Runs& runs, BS& bs,
                                                 const double2* __restrict__ csrow, int E, int lane, int klow, int ring_fwd) {
    re[0] = klow == 0 ? 1.0 : 0.0;
    im[0] = 0.0;
    bs.template prime<1>(0, false);
    int nblocks = 0;                                      // every block has enc = n (zyz_fast_ld)
    for (int ri = 0; ri < runs.nruns; ++ri) nblocks += runs.count[ri];
    const double2* cs_b = csrow;                          // first column of the unrolled body's first block
    // Coefficients are read ONE LAYER ahead, all of a layer's at its predecessor's start (the block walk used to keep a
    // whole block's 18 coefficients in registers: 72 VGPRs, which capped the pipeline kernel at three waves per SIMD):
    // the reads run under the predecessor's gates, well clear of the ring gather at its end.
    LayerCoef<N> ce, ca, cb;                              // RX chunk, sub-layer 1, sub-layer 2
    ce.dg = bs.rd(bs.slot(0), bs.a_dg);
    static_for<0, N>([&](auto q) { ce.g[decltype(q)::value] = cs_b[decltype(q)::value]; });
    auto read_layer = [&](LayerCoef<N>& c, const char* sl, int rec) {
        c.dg = bs.rd(sl, rec * kRecBytes + bs.a_dg);
        static_for<0, N>([&](auto q) { c.g[decltype(q)::value] = bs.rd(sl, rec * kRecBytes + bs.a_ry[decltype(q)::value]); });
        __builtin_amdgcn_sched_barrier(0);
    };
    auto read_chunk = [&](const char* nx, int kb) {       // the next block's RX chunk + the diagonal in front of it
        ce.dg = bs.rd(nx, bs.a_dg);
        const double2* cn = cs_b + (kb + 1) * N;
        static_for<0, N>([&](auto q) { ce.g[decltype(q)::value] = cn[decltype(q)::value]; });
        __builtin_amdgcn_sched_barrier(0);
    };
    auto ry_layer = [&](const LayerCoef<N>& c) {
        apply_phase<false>(re[0], im[0], c.dg);
        static_for<0, N>([&](auto q) { apply_ry<decltype(q)::value, false>(re[0], im[0], c.g[decltype(q)::value]); });
        __builtin_amdgcn_sched_barrier(0);
        re[0] = lane_gather(re[0], ring_fwd);
        im[0] = lane_gather(im[0], ring_fwd);
    };
    auto block = [&](auto sl, int b, int kb) {
        const char* cur = bs.template slot_rel<0>(sl);
        const char* nx = bs.template slot_rel<1>(sl);
        read_layer(ca, cur, 1);
        apply_phase<false>(re[0], im[0], ce.dg);
        static_for<0, N>([&](auto q) { apply_enc<N, decltype(q)::value, false>(re, im, ce.g[decltype(q)::value]); });
        if constexpr (LD == 2) {
            read_layer(cb, cur, 2);
            ry_layer(ca);
            bs.template ahead_rel<kBDist + 1>(sl, b);     // in the gather's shadow
            __builtin_amdgcn_sched_barrier(0);
            bs.landed_late();                             // block b + 1 landed
            read_chunk(nx, kb);
            ry_layer(cb);
        } else {
            bs.landed();                                  // block b + 1 landed
            read_chunk(nx, kb);
            ry_layer(ca);
            bs.template ahead_rel<kBDist + 1>(sl, b);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int b = 0;
    for (; b + kBSlots <= nblocks; b += kBSlots) {        // b is a multiple of kBSlots: block b + K sits in slot K
        block(CtSlot<0>{}, b, 0);
        block(CtSlot<1>{}, b + 1, 1);
        block(CtSlot<2>{}, b + 2, 2);
        block(CtSlot<3>{}, b + 3, 3);
        cs_b += kBSlots * N;
    }
    for (; b < nblocks; ++b) {
        block(RtSlot{b}, b, 0);
        cs_b += N;
    }
    apply_phase<false>(re[0], im[0], ce.dg);           // record L = record 0 of the slot after the last block
}

// ---------------------------------------------------------------------------------------
// Split layout, n = 5, forward sweeps of the block-unrolled shapes.  A lone wave's time is its instruction count, and
// in the all-lane layout every gate is paid twice, once for the real and once for the imaginary part of the wave's
// two samples.  An RY gate has REAL coefficients: it acts on the real parts and on the imaginary parts separately.
// So a wave may carry ONE sample with lane = k + 32 p  (k: basis index, p = 0: Re, 1: Im), one double per lane:
//   RY on qubit q < 4   partner through DPP, x' = c x -/+ s x_partner           2 moves + 2 fp64 (was 4 + 4)
//   RY on qubit 4       A = B = x; v_permlane16_swap(A, B) leaves the pair's bit-0 value in A and its bit-1 value in
//                       B for both lanes: x' = u.x A + u.y B, u = (c, -s) / (s, c)    1 copy + 2 swaps + 2 fp64
//   diagonal e^{i Phi}  the same through v_permlane32_swap (A = Re, B = Im): x' = alpha A + beta B with
//                       (alpha, beta) = (cos, -sin) for the Re lanes, (sin, cos) for the Im lanes
//   ring                one gathered double instead of two
// and EVERY encoding gate runs as RX = RZ(-pi/2) RY RZ(pi/2) with the fixed phases folded into the neighbouring
// diagonals (the all-lane kernels do that for wire 4 only).  ~30 vector instructions per layer instead of ~50, for
// half the samples per wave: that pays where waves are lone anyway -- the forward kernel at small batches, and the
// forward phase of the pipelined backward kernel, where the lambda wave would otherwise idle: psi and lambda wave
// sweep one sample each, leave psi_N in LDS in the all-lane layout, and the reverse phase runs unchanged.
//
// Split records (prep_zyz_kernel), 1 KB per layer like the all-lane ones:
//   bytes [0, 768)        per basis index k: [-sin, cos, sin] of Phi_l(k); lane (k, p) reads the 16 bytes at 24 k + 8 p
//                         as (beta, alpha)
//   bytes [768, 928)      ansatz layers: per wire 32 bytes, the lane-bit 0 / 1 variants (c, -s) / (c, +s); wire 4: (c, -s) / (s, c)
// (cos, sin) table of the encodings: 32 bytes per angle, wire = column % 5:
//   wires 0..3 [c, -s | c, +s]   (the second half is what the all-lane reverse walk's native RX wants)
//   wire 4     [s, c | c, -s]    (split: lane-bit 1 / 0 variants; all-lane reverse walk: (s, c), fields swapped)
// ---------------------------------------------------------------------------------------
constexpr int kSRecRy = 768;
__host__ __device__ inline bool zsplit_eligible(int n, long E, const Runs& r) {
    return n == 5 && zyz_fast_ld(r, n) != 0 && 2 * zyz_cs_row(n, E) * 32 <= kZCsBytes;
}
__device__ __forceinline__ void fill_cs_split(double4* cs, const AngleSrc& src, int E, long b0, long B, int ns, int tid, int nthreads) {
    const int row = (int)zyz_cs_row(5, E);
    for (int i = tid; i < ns * E; i += nthreads) {              // (sample, column) pairs spread over the threads (fill_cs)
        const int s = i / E, e = i - s * E;
        const long b = (b0 + s < B) ? b0 + s : B - 1;
        double sn, cn;
        fast_sincos(0.5 * enc_angle(src, E, b, e), &sn, &cn);
        cs[s * row + 5 + e] = (e % 5 == 4) ? make_double4(sn, cn, cn, -sn) : make_double4(cn, -sn, cn, sn);
    }
}
// what the all-lane gates (apply_enc) want from a 32-byte entry: (cos, sin)
template <int Q>
__device__ __forceinline__ double2 cs32_packed(const double2* chunk /* entry of wire 0 */) {
    if constexpr (Q == 4) { const double2 t = chunk[2 * Q]; return make_double2(t.y, t.x); }
    else return chunk[2 * Q + 1];
}

template <bool HALVES>      // A, B <- the value of the lower / upper partner (rows of 16 lanes, or the wave's halves)
__device__ __forceinline__ void swap_dup(double x, double& A, double& B) {
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    if constexpr (HALVES) {
        const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        A = __hiloint2double((int)h[0], (int)l[0]); B = __hiloint2double((int)h[1], (int)l[1]);
    } else {
        const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        A = __hiloint2double((int)h[0], (int)l[0]); B = __hiloint2double((int)h[1], (int)l[1]);
    }
}
__device__ __forceinline__ void split_phase(double& x, const double2& d /* (beta, alpha) */) {
    double A, B;
    swap_dup<true>(x, A, B);
    x = d.y * A + d.x * B;
}
template <int Q>
__device__ __forceinline__ void split_ry(double& x, const double2& u) {
    if constexpr (Q == 4) {
        double A, B;
        swap_dup<false>(x, A, B);
        x = u.x * A + u.y * B;
    } else {
        const double own = u.x * x;                 // issued while the exchange is under way: exchange -> fma on the chain
        x = fma(u.y, xchg<(1 << Q)>(x), own);
    }
}

template <int LD>
struct SplitStream : BlockStream<5, LD, false> {
    using Base = BlockStream<5, LD, false>;
    static constexpr int RPB = 1 + LD;
    unsigned a_cs[5];           // this lane's variant within a 32-byte (cos, sin) entry, + 32 * wire
    __device__ __forceinline__ void init_split(const char* srec0, int nrec, char* ring_wave, int lane) {
        Base::init(srec0, nrec, ring_wave, lane, lane & 31);
        this->a_dg = (unsigned)(lane & 31) * 24u + (unsigned)(lane >> 5) * 8u;
        static_for<0, 5>([&](auto q) {
            constexpr int Q = decltype(q)::value;
            const unsigned bit = ((unsigned)lane >> Q) & 1u;
            this->a_ry[Q] = kSRecRy + Q * 32 + bit * 16u;
            a_cs[Q] = Q * 32 + (Q == 4 ? (1u - bit) : bit) * 16u;
        });
    }
    static __device__ __forceinline__ double2 rd8(const char* p, unsigned off) {       // 8-byte aligned pair
        const double* q = reinterpret_cast<const double*>(p + off);
        return make_double2(q[0], q[1]);
    }
    __device__ __forceinline__ void load_records_split(const char* sl) {
#pragma unroll
        for (int i = 0; i < RPB; ++i) this->dg[i] = rd8(sl, i * kRecBytes + this->a_dg);
#pragma unroll
        for (int s = 0; s < LD; ++s)
            static_for<0, 5>([&](auto q) { this->ry[s][decltype(q)::value] = Base::rd(sl, (1 + s) * kRecBytes + this->a_ry[decltype(q)::value]); });
    }
    __device__ __forceinline__ void load_cs_split(const char* chunk) {
        static_for<0, 5>([&](auto q) { this->cs[decltype(q)::value] = Base::rd(chunk, a_cs[decltype(q)::value]); });
    }
};

// forward sweep of ONE sample in the split layout; csrow: the sample's table row, entry of column 0; returns this
// lane's Re (lanes 0..31) or Im (lanes 32..63) of psi_N[lane & 31]
template <int LD>
__device__ __forceinline__ double zsplit_forward(SplitStream<LD>& bs, const char* csrow, int nblocks, int lane, int ring_fwd) {
    double x = lane == 0 ? 1.0 : 0.0;
    bs.template prime<1>(0, false);
    const char* cs_b = csrow;                            // entry of block b's first column
    LayerCoef<5> ce, ca, cb;                             // coefficients one layer ahead (zyz_forward_fast)
    ce.dg = bs.rd8(bs.slot(0), bs.a_dg);
    static_for<0, 5>([&](auto q) { ce.g[decltype(q)::value] = bs.rd(cs_b, bs.a_cs[decltype(q)::value]); });
    auto read_layer = [&](LayerCoef<5>& c, const char* sl, int rec) {
        c.dg = bs.rd8(sl, rec * kRecBytes + bs.a_dg);
        static_for<0, 5>([&](auto q) { c.g[decltype(q)::value] = bs.rd(sl, rec * kRecBytes + bs.a_ry[decltype(q)::value]); });
        __builtin_amdgcn_sched_barrier(0);
    };
    auto read_chunk = [&](const char* nx, int kb) {
        ce.dg = bs.rd8(nx, bs.a_dg);
        const char* cn = cs_b + (kb + 1) * (5 * 32);
        static_for<0, 5>([&](auto q) { ce.g[decltype(q)::value] = bs.rd(cn, bs.a_cs[decltype(q)::value]); });
        __builtin_amdgcn_sched_barrier(0);
    };
    auto layer = [&](const LayerCoef<5>& c, bool ring) {
        split_phase(x, c.dg);
        static_for<0, 5>([&](auto q) { split_ry<decltype(q)::value>(x, c.g[decltype(q)::value]); });
        __builtin_amdgcn_sched_barrier(0);
        if (ring) x = lane_gather(x, ring_fwd);
    };
    auto block = [&](auto sl, int b, int kb) {           // block b = the unrolled body's block kb (its cs entries: immediates)
        const char* cur = bs.template slot_rel<0>(sl);
        const char* nx = bs.template slot_rel<1>(sl);
        read_layer(ca, cur, 1);
        layer(ce, false);
        if constexpr (LD == 2) {
            read_layer(cb, cur, 2);
            layer(ca, true);
            bs.template ahead_rel<kBDist + 1>(sl, b);    // in the gather's shadow
            __builtin_amdgcn_sched_barrier(0);
            bs.landed_late();
            read_chunk(nx, kb);
            layer(cb, true);
        } else {
            bs.landed();
            read_chunk(nx, kb);
            layer(ca, true);
            bs.template ahead_rel<kBDist + 1>(sl, b);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int b = 0;
    for (; b + kBSlots <= nblocks; b += kBSlots) {       // b is a multiple of kBSlots: block b + K sits in slot K
        block(CtSlot<0>{}, b, 0);
        block(CtSlot<1>{}, b + 1, 1);
        block(CtSlot<2>{}, b + 2, 2);
        block(CtSlot<3>{}, b + 3, 3);
        cs_b += kBSlots * (5 * 32);
    }
    for (; b < nblocks; ++b) {
        block(RtSlot{b}, b, 0);
        cs_b += 5 * 32;
    }
    split_phase(x, ce.dg);                               // record L = record 0 of the slot after the last block
    return x;
}

struct ZFwdArgs {
    Runs runs; long B; int E; const char* rec; int rec_bytes; int L; AngleSrc src; double off, co;
    const double* diag; int pauli; double* out; double* state_out; const double* bias;
    int fast_ld, nblocks;       // block-unrolled fast path: sub-layers per block (0 = generic walk), number of blocks
    const char* srec;           // split records (record 0), nullptr: shape not eligible (zsplit_eligible)
};
struct ZBwdArgs {
    Runs runs; long B; int E; int blk; const char* rec; int rec_bytes; int L; AngleSrc src; double off, co;
    const double* diag; int pauli; const double* g; const double* state_in; const double* y; const double* bias;
    double inv_bt; double* out; double* grad_x; double* partial; int* status;
    int fast_ld, nblocks;
    const char* srec;           // split records for the forward phase, nullptr: all-lane forward sweep
    int pipes;                  // bwd_ztri_kernel: sample groups per workgroup (1 or 2)
};

// kZFwdWaves sweeping waves + kFwdHelpers waves that only help to fill the (cos, sin) tables and then leave: the fill is
// 2 x 600 fp64 sincos per sweeping wave at cfg 2, ~4 us if each does its own, and the sweep cannot start before it.  Four
// sweeping waves: a workgroup's waves go to the CU's SIMDs in cyclic order, so each sweep gets a SIMD of its own; with two per
// workgroup the dispatcher put sweeps of co-resident workgroups on one SIMD while another stood empty.
constexpr int kFwdHelpers = 2, kZFwdWaves = 4;
template <int N>
__global__ __launch_bounds__((kZFwdWaves + kFwdHelpers) * 64) void fwd_zyz_kernel(ZFwdArgs a) {
    using C = Cfg<N>;
    static_assert(C::R == 1, "all-lane layout");
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];                 // kZFwdWaves x SPW x E (cos, sin)
    __shared__ __attribute__((aligned(16))) char rec_ring[kZFwdWaves * kBlockRingBytes];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    {
        const int csrow_len = (int)zyz_cs_row(N, a.E);
        fill_cs(reinterpret_cast<double2*>(dyn_lds), a.src, N, a.E, (long)blockIdx.x * kZFwdWaves * C::SPW, a.B, kZFwdWaves * C::SPW,
                (int)threadIdx.x, (kZFwdWaves + kFwdHelpers) * 64);
        (void)csrow_len;
    }
    __syncthreads();
    if (wib >= kZFwdWaves) return;
    const long wave = (long)blockIdx.x * kZFwdWaves + wib;
    const long b_raw = wave * C::SPW + (lane >> C::LB);
    const bool valid = b_raw < a.B;
    const long b = valid ? b_raw : a.B - 1;
    const int klow = lane & (C::LANES - 1);
    const int ring_fwd = ring_source<N>(lane, false);

    const int csrow_len = (int)zyz_cs_row(N, a.E);
    double2* cs = reinterpret_cast<double2*>(dyn_lds) + (long)wib * C::SPW * csrow_len;
    const double2* csrow = cs + (lane >> C::LB) * csrow_len + N;
    char* my_ring = rec_ring + wib * kBlockRingBytes;
    double re[1], im[1];
    if (a.fast_ld == 2) {
        BlockStream<N, 2> bs;
        bs.init(a.rec, a.L + 1, my_ring, lane, klow);
        zyz_forward_fast<N, 2>(re, im, a.runs, bs, csrow, a.E, lane, klow, ring_fwd);
    } else if (a.fast_ld == 1) {
        BlockStream<N, 1> bs;
        bs.init(a.rec, a.L + 1, my_ring, lane, klow);
        zyz_forward_fast<N, 1>(re, im, a.runs, bs, csrow, a.E, lane, klow, ring_fwd);
    } else {
        LayerStream<N> ls;
        ls.init(a.rec, a.rec_bytes, my_ring, lane, klow, a.L);
        zyz_forward<N>(re, im, a.runs, ls, csrow, a.E, lane, klow, ring_fwd);
    }

    if (a.state_out && valid)
        reinterpret_cast<double2*>(a.state_out)[(b << N) + klow] = make_double2(re[0], im[0]);
    basis_change<N, false>(re, im, a.pauli, lane);
    double v[1] = {ham_weight<N>(klow, a.off, a.co, a.diag) * (re[0] * re[0] + im[0] * im[0])};
    lane_reduce<1, C::LB>(v, lane);
    if (valid && klow == 0) a.out[b] = v[0] + (a.bias ? a.bias[0] : 0.0);
}

// Forward kernel in the split layout (n = 5, block-unrolled shapes, Z / diagonal read-out): one sample per sweeping
// wave, kSplitWaves of them per workgroup + helpers for the table fill.  For batches that leave SIMDs free.
#ifndef QHEA_SPLIT_WAVES
#define QHEA_SPLIT_WAVES 4
#endif
#ifndef QHEA_SPLIT_HELPERS
#define QHEA_SPLIT_HELPERS 2
#endif
constexpr int kSplitWaves = QHEA_SPLIT_WAVES, kSplitHelpers = QHEA_SPLIT_HELPERS;
template <int N>
__global__ __launch_bounds__((kSplitWaves + kSplitHelpers) * 64) void fwd_split_kernel(ZFwdArgs a) {
    static_assert(N == 5, "split layout: n = 5");
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];                 // kSplitWaves x row x 32 bytes
    __shared__ __attribute__((aligned(16))) char rec_ring[kSplitWaves * kBlockRingBytes];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long b0 = (long)blockIdx.x * kSplitWaves;
    fill_cs_split(reinterpret_cast<double4*>(dyn_lds), a.src, a.E, b0, a.B, kSplitWaves, (int)threadIdx.x,
                  (kSplitWaves + kSplitHelpers) * 64);
    __syncthreads();
    const long b = b0 + wib;
    if (wib >= kSplitWaves || b >= a.B) return;
    const int k = lane & 31, p = lane >> 5;
    const int ring_fwd = ring_source<5>(lane, false);
    const char* row = dyn_lds + (wib * (int)zyz_cs_row(5, a.E) + 5) * 32;
    double x;
    if (a.fast_ld == 2) {
        SplitStream<2> ss;
        ss.init_split(a.srec, a.L + 1, rec_ring + wib * kBlockRingBytes, lane);
        x = zsplit_forward<2>(ss, row, a.nblocks, lane, ring_fwd);
    } else {
        SplitStream<1> ss;
        ss.init_split(a.srec, a.L + 1, rec_ring + wib * kBlockRingBytes, lane);
        x = zsplit_forward<1>(ss, row, a.nblocks, lane, ring_fwd);
    }
    if (a.state_out) a.state_out[(((b << 5) + k) << 1) | p] = x;
    double v[1] = {ham_weight<5>(k, a.off, a.co, a.diag) * (x * x)};
    lane_reduce<1, 6>(v, lane);
    if (lane == 0) a.out[b] = v[0] + (a.bias ? a.bias[0] : 0.0);
}

// ---------------------------------------------------------------------------------------
// psi / lambda / sigma pipelined backward kernel, ZYZ form.  Roles, rings, counters and failure reporting as in
// bwd_tri_kernel (hea_device.hpp); a "step" is one layer (one ansatz sub-layer or one RX chunk).
// ---------------------------------------------------------------------------------------
// Before a chain overwrites hand-off slot (step mod RING) every sigma wave must be past step - RING, i.e. every cursor >= need.
// `safe` = the minimum of the cursors as last read: ONE scalar compare per publication while need <= safe (the sigma waves
// trail the chain by a few steps, so a reading is good for most of a ring's worth of publications); when it is not, all
// cursors are read in one LDS round trip, and only a sigma wave that really is behind is waited for (pair_wait_ge).  (Four
// compare-and-branch pairs per publication before: a lone chain wave pays ~5 clocks for every instruction, scalar ones too.)
template <int NSIG>
__device__ __forceinline__ void wait_slot_free(int* cursor, int need, int* abort_flag, int (&seen)[NSIG], int& safe) {
    if (__builtin_expect(need <= safe, 1)) return;           // (the common case falls through: no taken branch on the chain)
    int c[NSIG];
#pragma unroll
    for (int w = 0; w < NSIG; ++w) c[w] = __hip_atomic_load(&cursor[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    int lo = 0x7fffffff;
#pragma unroll
    for (int w = 0; w < NSIG; ++w) { seen[w] = __builtin_amdgcn_readfirstlane(c[w]); lo = seen[w] < lo ? seen[w] : lo; }
    if (lo < need) {
        lo = 0x7fffffff;
#pragma unroll
        for (int w = 0; w < NSIG; ++w) { pair_wait_ge(&cursor[w], need, abort_flag, seen[w]); lo = seen[w] < lo ? seen[w] : lo; }
    }
    safe = lo;
}

// One chain wave of the pipelined backward kernel: role 0 = psi (forward sweep or state_in, then psi walked back),
// role 1 = lambda (lambda_N = g H psi_N, walked back).  MODE 0: generic layer walk; 1 / 2: block-unrolled fast path
// with that many sub-layers per block.  Publishing protocol as in bwd_tri_kernel (hea_device.hpp).
// SPLIT (n = 5, MODE != 0): the (cos, sin) table has the 32-byte entries of the split layout, and the forward phase is
// swept in that layout by BOTH chain waves, one sample each (zsplit_forward).
template <int N, int MODE, bool SPLIT, int RING>
__device__ __forceinline__ void ztri_chain(const ZBwdArgs& a, int role, int lane, int klow, bool valid, long b,
                                           const double2* cs, char* my_ring, double2 (*psi_ring)[64],
                                           double2 (*lam_ring)[64], double2* psi_final, ZSync* sync, double* axis_ring) {
    using C = Cfg<N>;
    static_assert(!SPLIT || (N == 5 && MODE != 0), "split layout: n = 5, block-unrolled shapes");
    __builtin_amdgcn_s_setprio(3);                          // the chains are the critical path (hea_device.hpp)
    const int E = a.E;
    const int ring_fwd = ring_source<N>(lane, false);
    const int ring_rev = ring_source<N>(lane, true);
    // all-lane walks: this lane's sample row; SPLIT: two double2 per column
    const double2* csrow = cs + ((lane >> C::LB) * (int)zyz_cs_row(N, E) + N) * (SPLIT ? 2 : 1);
    LayerStream<N> ls;
    BlockStream<N, MODE == 0 ? 1 : MODE> bs;
    if constexpr (MODE == 0) ls.init(a.rec, a.rec_bytes, my_ring, lane, klow, a.L);
    else bs.init(a.rec, a.L + 1, my_ring, lane, klow);
    double sr[1], si[1];
    int seen[kZSigma], safe = 0;
#pragma unroll
    for (int w = 0; w < kZSigma; ++w) seen[w] = 0;
    bool swept = false;                                     // psi_N already in psi_final, `ready` counted up by both chains
    if constexpr (SPLIT) {
        if (!a.state_in) {
            SplitStream<MODE> ss;
            ss.init_split(a.srec, a.L + 1, my_ring, lane);
            const char* row = reinterpret_cast<const char*>(cs) + (role * (int)zyz_cs_row(5, E) + 5) * 32;
#ifdef QHEA_PROFILE_WAITS
            const unsigned long long tf0 = __builtin_amdgcn_s_memtime();
#endif
            const double x = zsplit_forward<MODE>(ss, row, a.nblocks, lane, ring_fwd);
#ifdef QHEA_PROFILE_WAITS
            if (blockIdx.x == 100 && lane == 0) printf("wave %d: forward sweep %llu ticks\n", (int)(threadIdx.x >> 6), __builtin_amdgcn_s_memtime() - tf0);
#endif
            reinterpret_cast<double*>(psi_final)[(((role << 5) | (lane & 31)) << 1) | (lane >> 5)] = x;   // all-lane layout
            handoff_release();
            if (lane == 0) __hip_atomic_fetch_add(&sync->ready, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            swept = true;
        }
    }
    if (role == 0) {
        if (swept) {
            int seen_ready = 0;
            pair_wait_ge(&sync->ready, 2, &sync->abort, seen_ready);
            sr[0] = psi_final[lane].x; si[0] = psi_final[lane].y;
        } else {
            if (a.state_in) {
                const double2 s0 = reinterpret_cast<const double2*>(a.state_in)[(b << N) + klow];
                sr[0] = s0.x; si[0] = s0.y;
            } else {
                if constexpr (MODE == 0) zyz_forward<N>(sr, si, a.runs, ls, csrow, E, lane, klow, ring_fwd);
                else if constexpr (!SPLIT) zyz_forward_fast<N, MODE>(sr, si, a.runs, bs, csrow, E, lane, klow, ring_fwd);
            }
            psi_final[lane] = make_double2(sr[0], si[0]);
            __hip_atomic_store(&sync->ready, 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else {
        int seen_ready = 0;
        pair_wait_ge(&sync->ready, 2, &sync->abort, seen_ready);
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
    int* prod = role == 0 ? &sync->psi_prod : &sync->lam_prod;
    int step = 0;
    // A publication is two stores: the state, then the step counter the sigma waves poll, ordered by handoff_release()
    // (s_waitcnt lgkmcnt(0): the state's store has been executed).  Issued back to back the wait stalls the chain for the
    // store's whole LDS round trip; the block-unrolled walks therefore store the state at the publication point, go on with
    // the layer's gates (registers only) and raise the counter AFTER them, when the store has long completed and the wait
    // costs nothing.  The sigma waves see a step ~a layer later, which is off the critical path.
    auto publish_data = [&]() {
        wait_slot_free<kZSigma>(sync->cursor, step - RING + 1, &sync->abort, seen, safe);     // (safe starts at 0: nothing to wait for below RING)
        ring[step & (RING - 1)][lane] = make_double2(sr[0], si[0]);
        ++step;
    };
    auto publish_flag = [&]() {
        handoff_release();                                   // hea_device.hpp: data before counter, s_waitcnt lgkmcnt(0)
        __hip_atomic_store(prod, step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto publish = [&]() { publish_data(); publish_flag(); };

    if constexpr (MODE != 0) {
        // ---- block-unrolled reverse walk: per block  ring^-1 publish RY(b,LD)^-1 [dg(b,LD)]^-1 ... [dg(b,1)]^-1 publish RX(b)^-1 [dg(b,0)]^-1
        constexpr int LD = MODE;
        const int nb = a.nblocks;
        bs.template prime<-1>(nb, true);                      // (drain: the forward sweep's run-ahead fetches target the same slots)
        apply_phase<true>(sr[0], si[0], bs.rd(bs.slot(nb), bs.a_dg));     // block nb's slot: its record 0 is the final diagonal
        bs.template step<-1>(nb);                             // block nb-1 landed
        constexpr int CW = SPLIT ? 2 : 1;                     // double2 per table column
        const double2* cs_b = csrow + (long)(nb - 1) * N * CW;    // chunk of the block at hand (every block has enc = n)
        // coefficients one layer ahead (zyz_forward_fast): ct = the block's last sub-layer (record LD), cm = its first one
        // when LD = 2 (record 1), c0 = its RX chunk with the diagonal in front of it (record 0)
        LayerCoef<N> ct, cm, c0;
        auto read_layer = [&](LayerCoef<N>& c, const char* sl, int rec) {
            c.dg = bs.rd(sl, rec * kRecBytes + bs.a_dg);
            static_for<0, N>([&](auto q) { c.g[decltype(q)::value] = bs.rd(sl, rec * kRecBytes + bs.a_ry[decltype(q)::value]); });
            __builtin_amdgcn_sched_barrier(0);
        };
        auto read_chunk = [&](const char* sl, const double2* chunk) {
            c0.dg = bs.rd(sl, bs.a_dg);
            if constexpr (SPLIT) static_for<0, N>([&](auto q) { c0.g[decltype(q)::value] = cs32_packed<decltype(q)::value>(chunk); });
            else static_for<0, N>([&](auto q) { c0.g[decltype(q)::value] = chunk[decltype(q)::value]; });
            __builtin_amdgcn_sched_barrier(0);
        };
        auto undo_ry = [&](const LayerCoef<N>& c) {            // ring^-1, publish, RY^-1 of a sub-layer
            sr[0] = lane_gather(sr[0], ring_rev);
            si[0] = lane_gather(si[0], ring_rev);
        };
        read_layer(ct, bs.slot(nb - 1), LD);
        // one block, sitting in ring slot `sl`; kb: its position in the unrolled body (cs_b stays on the body's first block)
        auto block = [&](auto sl, int bl, int kb) {
            const char* cur = bs.template slot_rel<0>(sl);
            const char* nx = bs.template slot_rel<-1>(sl);
            // the axes that turn this block's first sub-layer's (X, Y, Z) into its RX chunk's gradients ride in the chunk's
            // record (prep_zyz_kernel); the lambda wave leaves them where the sigma waves find them, before it publishes
            // the block's first step
            if (role == 1 && lane < 3 * N) axis_ring[(bl & (kAxisRing - 1)) * (3 * N) + lane] = reinterpret_cast<const double*>(cur + kRecRy)[lane];
            // ---- last sub-layer: ring^-1, publish, RY^-1, then the diagonal in front of it
            if constexpr (LD == 2) read_layer(cm, cur, 1);
            else read_chunk(cur, cs_b - kb * (N * CW));
            undo_ry(ct);
            bs.template ahead_rel<-(kBDist + 1)>(sl, bl);     // in the gather's shadow
            __builtin_amdgcn_sched_barrier(0);
            publish_data();
            static_rfor<0, N>([&](auto q) { apply_ry<decltype(q)::value, true>(sr[0], si[0], ct.g[decltype(q)::value]); });
            __builtin_amdgcn_sched_barrier(0);
            publish_flag();
            if constexpr (LD == 2) {
                read_chunk(cur, cs_b - kb * (N * CW));
                apply_phase<true>(sr[0], si[0], ct.dg);
                undo_ry(cm);
                __builtin_amdgcn_sched_barrier(0);
                publish_data();
                static_rfor<0, N>([&](auto q) { apply_ry<decltype(q)::value, true>(sr[0], si[0], cm.g[decltype(q)::value]); });
                __builtin_amdgcn_sched_barrier(0);
                publish_flag();
                apply_phase<true>(sr[0], si[0], cm.dg);
            } else {
                apply_phase<true>(sr[0], si[0], ct.dg);
            }
            // ---- RX chunk (no publication: its gradients are read off the sub-layer just undone, see the sigma waves);
            //      meanwhile the next block's last sub-layer comes in
            bs.landed_late();                                 // block bl - 1 landed
            read_layer(ct, nx, LD);
            static_rfor<0, N>([&](auto q) { apply_enc<N, decltype(q)::value, true>(sr, si, c0.g[decltype(q)::value]); });
            __builtin_amdgcn_sched_barrier(0);
            apply_phase<true>(sr[0], si[0], c0.dg);           // the diagonal in front of this block's RX chunk
        };
        int bl = nb - 1;
        for (; bl >= 0 && (bl & (kBSlots - 1)) != kBSlots - 1; --bl) {     // down to a block in the last slot
            block(RtSlot{bl}, bl, 0);
            cs_b -= N * CW;
        }
        for (; bl >= kBSlots - 1; bl -= kBSlots) {            // bl = 3 (mod 4): block bl - K sits in slot 3 - K
            block(CtSlot<3>{}, bl, 0);
            block(CtSlot<2>{}, bl - 1, 1);
            block(CtSlot<1>{}, bl - 2, 2);
            block(CtSlot<0>{}, bl - 3, 3);
            cs_b -= kBSlots * (N * CW);
        }
    } else {
        // ---- generic reverse walk over the layers: [diagonal of record l+1]^-1, ring^-1 (ansatz), publish, [layer l]^-1
        int l = a.L - 1, col = E;
        wait_vmcnt<0>();                                   // the forward sweep's run-ahead fetches target the same slots
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
                        apply_enc<N, Q, true>(sr, si, nxt[Q]);
                    });
                    ls.rd_all_ry(l - 1);                    // an RX chunk uses none: read the next record's in one go
                    --l;
                }
                col -= ne;
            }
        }
    }
}

// PIPES = 2: two such pipelines (two sample groups) per workgroup; their sigma waves add the gradient sums into ONE
// row held in LDS (two addends per element: the order cannot matter), written out once at the end -- half the
// partial rows for the reduce kernel to read.  Chosen when two workgroups would share a CU anyway (hea_api.hip).
constexpr int kZPipeWaves = 2 + kZSigma;
// LDS of one pipeline apart from its (cos, sin) table: two record rings, psi and lambda hand-off rings, psi_N, counters
__host__ __device__ constexpr size_t ztri_fixed_lds(int ring) {
    return 2 * (size_t)kBlockRingBytes + 2 * (size_t)ring * 1024 + 1024 + 256 + (size_t)kAxisRing * 15 * sizeof(double);
}
// Wave -> (pipeline, role) of bwd_ztri_kernel.  PIPES = 2 (QHEA_ZTRI2_MAP = 1): the four chain waves are the workgroup's
// waves 0..3 (psi A, lambda A, psi B, lambda B) and the sigma waves follow, three per pipeline: a workgroup's waves go to
// the SIMDs in cyclic order, so every SIMD gets exactly one chain wave (with roles laid out pipeline by pipeline, waves 0, 1,
// 5, 6 were the chains: two of them on one SIMD).
#ifndef QHEA_ZTRI2_MAP
#define QHEA_ZTRI2_MAP 1
#endif
template <int PIPES>
__device__ __forceinline__ int pipe_of_wave() {
    if constexpr (PIPES == 1) return 0;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (QHEA_ZTRI2_MAP) return wv < 4 ? (wv >> 1) : (wv - 4) / kZSigma;
    return wv / kZPipeWaves;
}
template <int PIPES>
__device__ __forceinline__ int role_of_wave() {          // 0: psi, 1: lambda, 2..: sigma
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if constexpr (PIPES == 1) return wv;
    if (QHEA_ZTRI2_MAP) return wv < 4 ? (wv & 1) : 2 + (wv - 4) % kZSigma;
    return wv % kZPipeWaves;
}
template <int PIPES> constexpr int kZRingDepth = PIPES == 1 ? kPairRing : 8;      // LDS: 2 x (24 + 16 + 20) KB + the row
template <int N, int PIPES>
__global__ __launch_bounds__(64 * kZPipeWaves * PIPES) __attribute__((amdgpu_waves_per_eu(4, 4))) void bwd_ztri_kernel(ZBwdArgs a) {
    using C = Cfg<N>;
    static_assert(C::R == 1, "all-lane layout");
    constexpr int RING = kZRingDepth<PIPES>;
    // All of the kernel's LDS is dynamic -- per pipeline [record rings | psi ring | lambda ring | psi_N | sync], then the
    // (cos, sin) tables [+ the shared row]: with a static part the compiler concludes that LDS caps the occupancy at two
    // waves per SIMD and gives the register allocation 256 VGPRs to play with.
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
    char* fixed = dyn_lds + pipe_of_wave<PIPES>() * (int)ztri_fixed_lds(RING);
    char* lds_tables = dyn_lds + PIPES * (int)ztri_fixed_lds(RING);

    const int lane = threadIdx.x & 63;
    const int pipe = pipe_of_wave<PIPES>();
    const int role = role_of_wave<PIPES>();                                        // 0: psi, 1: lambda, 2..: sigma
    const int tid = role * 64 + lane;                                              // within the pipeline
    char* rec_ring = fixed;
    double2 (*psi_ring)[64] = reinterpret_cast<double2 (*)[64]>(fixed + 2 * kBlockRingBytes);
    double2 (*lam_ring)[64] = reinterpret_cast<double2 (*)[64]>(fixed + 2 * kBlockRingBytes + RING * 1024);
    double2* psi_final = reinterpret_cast<double2*>(fixed + 2 * kBlockRingBytes + 2 * RING * 1024);
    ZSync& sync = *reinterpret_cast<ZSync*>(fixed + 2 * kBlockRingBytes + 2 * RING * 1024 + 1024);
    static_assert(sizeof(ZSync) <= 256, "reserved");
    double* axis_ring = reinterpret_cast<double*>(fixed + 2 * kBlockRingBytes + 2 * RING * 1024 + 1024 + 256);
    const long wave = (long)blockIdx.x * PIPES + pipe;                            // one sample group per pipeline; a group past
    const long b_raw = wave * C::SPW + (lane >> C::LB);                           // the batch runs on copies of the last sample
    const bool valid = b_raw < a.B;                                               // with lambda = 0
    const long b = valid ? b_raw : a.B - 1;
    const int klow = lane & (C::LANES - 1);
    const int E = a.E;

#ifdef QHEA_PROFILE_WAITS
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
    if (tid < 24) sync.waited[tid] = 0;
#endif
    const bool split_steps = a.fast_ld != 0;              // sigma waves draw their steps from a counter (below)
    if (tid == 0) {
        sync.psi_prod = 0; sync.lam_prod = 0; sync.ready = 0; sync.abort = 0;
        for (int w = 0; w < kZSigma; ++w) sync.cursor[w] = split_steps ? 0 : w;
        sync.next = 0;
    }
    const bool split = N == 5 && a.srec != nullptr && a.fast_ld != 0;
    const int cs_bytes = (int)(C::SPW * zyz_cs_row(N, E) * (split ? 32 : 16));
    double2* cs = reinterpret_cast<double2*>(lds_tables + pipe * cs_bytes);
    double* row_lds = reinterpret_cast<double*>(lds_tables + PIPES * cs_bytes);    // PIPES = 2: blk x KW sums of both groups
    if constexpr (PIPES > 1) {
        for (int i = (int)threadIdx.x; i < a.blk * C::KW; i += 64 * kZPipeWaves * PIPES) row_lds[i] = 0.0;
    }
    if constexpr (N == 5) {
        if (split) fill_cs_split(reinterpret_cast<double4*>(cs), a.src, E, wave * C::SPW, a.B, C::SPW, tid, 64 * kZPipeWaves);
    }
    if (!split) fill_cs(cs, a.src, N, E, wave * C::SPW, a.B, C::SPW, tid, 64 * kZPipeWaves);   // all waves of the pipeline
    __syncthreads();

    if (role < 2) {
        // ------------------------------------------------------------------ psi / lambda chains
        char* my_ring = rec_ring + role * kBlockRingBytes;
        bool done = false;
        if constexpr (N == 5) {
            if (split) {
                if (a.fast_ld == 2) ztri_chain<N, 2, true, RING>(a, role, lane, klow, valid, b, cs, my_ring, psi_ring, lam_ring, psi_final, &sync, axis_ring);
                else ztri_chain<N, 1, true, RING>(a, role, lane, klow, valid, b, cs, my_ring, psi_ring, lam_ring, psi_final, &sync, axis_ring);
                done = true;
            }
        }
        if (done) {}
        else if (a.fast_ld == 2) ztri_chain<N, 2, false, RING>(a, role, lane, klow, valid, b, cs, my_ring, psi_ring, lam_ring, psi_final, &sync, axis_ring);
        else if (a.fast_ld == 1) ztri_chain<N, 1, false, RING>(a, role, lane, klow, valid, b, cs, my_ring, psi_ring, lam_ring, psi_final, &sync, axis_ring);
        else ztri_chain<N, 0, false, RING>(a, role, lane, klow, valid, b, cs, my_ring, psi_ring, lam_ring, psi_final, &sync, axis_ring);
    } else {
        // ------------------------------------------------------------------ sigma waves: inner products + sums
        double* __restrict__ part_w = a.partial + wave * (long)a.blk * C::KW;       // PIPES = 1: this group's row
        const int me = role - 2;
        int seen_p = 0, seen_l = 0;
        // the X, Y, Z terms of this lane for every qubit, from psi (own p, partners qv) and lambda of one published step
        auto products = [&](double (&acc3)[C::KW], const double2& p, const double2 (&qv)[N], const double2& lm) {
#pragma unroll
            for (int i = 0; i < C::KW; ++i) acc3[i] = 0.0;
            static_for<0, N>([&](auto q) {
                constexpr int Q = decltype(q)::value;
                const unsigned m = lane_sign_mask<Q>(lane);
                acc3[3 * Q] = lm.x * qv[Q].y - lm.y * qv[Q].x;
                acc3[3 * Q + 1] = flip_sign(-(lm.x * qv[Q].x) - lm.y * qv[Q].y, m);     // -sg (...), sg = (-1)^bit
                acc3[3 * Q + 2] = flip_sign(lm.x * p.y - lm.y * p.x, m);                 //  sg (...)
            });
        };
        auto store_sums = [&](double (&acc3)[C::KW], int sub) {
            const int vi = butterfly_sum<C::KW>(acc3, lane);
            if (butterfly_owner<C::KW>(lane)) {
                if constexpr (PIPES == 1) store_through(&part_w[(long)sub * C::KW + vi], acc3[0]);
                else __hip_atomic_fetch_add(&row_lds[sub * C::KW + vi], acc3[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        };
        if (split_steps) {
            // Block-unrolled shapes (every block: one full RX chunk + LD sub-layers).
            //  * The chains do not publish at the RX chunk.  With psi_C = W psi_E, lambda_C = W lambda_E, W = prod_q RY(theta_q)
            //    RZ(beta_q) the layer between the chunk (E) and the block's first sub-layer's publication point (C):
            //    Im<lam_E|X_q|psi_E> = n_q . (X, Y, Z)_q at C with n_q batch invariant (prep_zyz_kernel; handed over by the
            //    lambda wave in `axis_ring`) -- the chunk's gradients are a per-lane combination of the products that sub-layer's
            //    step forms anyway, summed per sample: one pipeline step and two publications less per block.
            //  * The sigma waves DRAW their steps from a counter instead of owning every kZSigma-th one: the waves of a
            //    workgroup do not run equally fast (the fifth shares its SIMD with a chain wave, steps with a chunk are heavier),
            //    and the slowest sigma wave is the kernel's tail.  A wave holds its current step and the next one (`after`, drawn
            //    early so that the counter's latency is off its path); cursor[me] = the step it will read next, as before.
            const int LDr = a.fast_ld, nsteps = a.nblocks * LDr;
            auto draw = [&]() {                                // valid in lane 0
                int v = 0;
                if (lane == 0) v = __hip_atomic_fetch_add(&sync.next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                return v;
            };
            int mine = __builtin_amdgcn_readfirstlane(draw());
            // A wave that draws late may get a first step beyond the ring depth, and the chains would wait forever for a cursor
            // that still says 0 (seen with six waves on an 8-entry ring: caught by the overrun report): it reads nothing below
            // `mine`, so say so at once.
            __hip_atomic_store(&sync.cursor[me], mine, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            int after_v = draw();
            // one step; CHUNK: it also carries the block's RX-chunk gradients.  Two instantiations rather than a run-time test
            // inside: with the axes loaded conditionally the compiler copies all fifteen at the join of the two paths.
            auto sigma_step = [&](auto chunk_c) {
                constexpr bool CHUNK = decltype(chunk_c)::value;
                const int t = mine;
                const int j = LDr == 2 ? (t >> 1) : t;         // block, in walking order
                const int sub = a.blk - 1 - t, bl = a.nblocks - 1 - j;
                pair_wait_ge(&sync.psi_prod, t + 1, &sync.abort, seen_p);
                pair_wait_ge(&sync.lam_prod, t + 1, &sync.abort, seen_l);
                const double2* slot = psi_ring[t & (RING - 1)];
                const double2 p = slot[lane];
                double2 qv[N];
                static_for<0, N>([&](auto q) { qv[decltype(q)::value] = slot[lane ^ (1 << decltype(q)::value)]; });
                const double2 lm = lam_ring[t & (RING - 1)][lane];
                double em[CHUNK ? 3 * N : 1];
                if constexpr (CHUNK) {
                    const double* __restrict__ e = axis_ring + (bl & (kAxisRing - 1)) * (3 * N);
#pragma unroll
                    for (int i = 0; i < 3 * N; ++i) em[i] = e[i];
                }
                if (lane == 0) __hip_atomic_store(&sync.cursor[me], after_v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                mine = __builtin_amdgcn_readfirstlane(after_v);
                after_v = draw();
                double acc3[C::KW];
                products(acc3, p, qv, lm);
                if constexpr (CHUNK) {
                    double gx[C::KX];
#pragma unroll
                    for (int i = 0; i < C::KX; ++i) gx[i] = 0.0;
                    static_for<0, N>([&](auto q) {
                        constexpr int Q = decltype(q)::value;
                        gx[Q] = em[3 * Q] * acc3[3 * Q] + em[3 * Q + 1] * acc3[3 * Q + 1] + em[3 * Q + 2] * acc3[3 * Q + 2];
                    });
                    store_sums(acc3, sub);
                    if constexpr (N == 5) store_grad_x5(gx, lane, wave, a.B, E, a.grad_x, bl * N, N);
                    else store_grad_x<N>(gx, lane, wave, a.B, E, a.grad_x, bl * N, N);
                } else {
                    store_sums(acc3, sub);
                }
            };
            while (mine < nsteps) {
                if (LDr == 1 || (mine & 1)) sigma_step(std::true_type{});     // the block's first sub-layer is its last step
                else sigma_step(std::false_type{});
            }
        } else {
            // other shapes: the chains publish at every layer, sigma wave w takes the steps t = w (mod kZSigma)
            int col = E, sub = a.blk, step = 0;
            for (int ri = a.runs.nruns - 1; ri >= 0; --ri) {
                const int ne = a.runs.enc[ri], nld = a.runs.ld[ri];
                const int nch = (ne + N - 1) / N;
                const int m_last = ne - (nch - 1) * N;
                for (int rep = 0; rep < a.runs.count[ri]; ++rep) {
                    for (int s = nld - 1; s >= 0; --s) {
                        --sub;
                        if (step % kZSigma != me) { ++step; continue; }
                        pair_wait_ge(&sync.psi_prod, step + 1, &sync.abort, seen_p);
                        pair_wait_ge(&sync.lam_prod, step + 1, &sync.abort, seen_l);
                        const double2* slot = psi_ring[step & (RING - 1)];
                        const double2 p = slot[lane];
                        double2 qv[N];
                        static_for<0, N>([&](auto q) { qv[decltype(q)::value] = slot[lane ^ (1 << decltype(q)::value)]; });
                        const double2 lm = lam_ring[step & (RING - 1)][lane];
                        __hip_atomic_store(&sync.cursor[me], step + kZSigma, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        ++step;
                        double acc3[C::KW];
                        products(acc3, p, qv, lm);
                        store_sums(acc3, sub);
                    }
                    for (int ch = nch - 1; ch >= 0; --ch) {
                        const int m = ch == nch - 1 ? m_last : N;
                        if (step % kZSigma != me) { ++step; continue; }
                        pair_wait_ge(&sync.psi_prod, step + 1, &sync.abort, seen_p);
                        pair_wait_ge(&sync.lam_prod, step + 1, &sync.abort, seen_l);
                        const double2* slot = psi_ring[step & (RING - 1)];
                        double2 qv[N];
                        static_for<0, N>([&](auto q) { qv[decltype(q)::value] = slot[lane ^ (1 << decltype(q)::value)]; });
                        const double2 lm = lam_ring[step & (RING - 1)][lane];
                        __hip_atomic_store(&sync.cursor[me], step + kZSigma, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        ++step;
                        double gx[C::KX];
#pragma unroll
                        for (int i = 0; i < C::KX; ++i) gx[i] = 0.0;
                        for_gates_below<N>(m, [&](auto q) {
                            constexpr int Q = decltype(q)::value;
                            if constexpr (Q == 4) {           // this wire's encoding gate runs as RY (apply_enc): Y inner product
                                gx[Q] = flip_sign(-(lm.x * qv[Q].x) - lm.y * qv[Q].y, lane_sign_mask<Q>(lane));
                            } else {
                                gx[Q] = lm.x * qv[Q].y - lm.y * qv[Q].x;
                            }
                        });
                        if constexpr (N == 5) store_grad_x5(gx, lane, wave, a.B, E, a.grad_x, col - ne + ch * N, m);
                        else store_grad_x<N>(gx, lane, wave, a.B, E, a.grad_x, col - ne + ch * N, m);
                    }
                    col -= ne;
                }
            }
        }
    }
#ifdef QHEA_PROFILE_WAITS
    if (blockIdx.x == 100 && lane == 0)
        printf("role %d: total %llu ticks, in waits %llu ticks, %llu waits that read the counter (%llu of them spun)\n", role,
               __builtin_amdgcn_s_memtime() - t_begin, sync.waited[2 * (threadIdx.x >> 6)], sync.waited[2 * (threadIdx.x >> 6) + 1] & 0xffffffffull, sync.waited[2 * (threadIdx.x >> 6) + 1] >> 32);
#endif
    report_abort(&sync.abort, a.status, lane);
    if constexpr (PIPES > 1) {
        __syncthreads();                                   // every wave gets here, also after an overrun
        double* __restrict__ row = a.partial + (long)blockIdx.x * a.blk * C::KW;
        for (int i = (int)threadIdx.x; i < a.blk * C::KW; i += 64 * kZPipeWaves * PIPES) store_through(&row[i], row_lds[i]);
    }
}

// ---------------------------------------------------------------------------------------
// Quad-chain pipeline (n = 5, block-unrolled shapes, Z / diagonal read-out): the pipelined backward kernel for batches that
// leave whole CUs' worth of SIMDs free (at most one sample group per CU: B <= 512 on 256 CUs -- cfg 3's per-GPU shard, the
// reference's own training batch of 100).  There a chain wave is alone on its SIMD and the step time is the length of the
// dependent chain, so the chains are shortened the way the forward sweeps were: EVERY chain runs in the split layout
// (lane = k + 32 p, one double per lane, one sample per wave: an RY gate is 2 moves + 2 fp64 instead of 4 + 4), the
// REVERSE walks too, which takes two chain waves per state where the all-lane walk has one:
//     wave 0, 1   psi chains of the group's two samples: forward sweep, then psi walked back, published per sub-layer
//     wave 2, 3   lambda chains of the two samples: lambda_N = g H psi_N, walked back, published likewise
//     wave 4..    sigma waves, unchanged: they read the hand-off rings in the all-lane order (lane = k + 32 sample), which
//                 is how the chains store their halves of a slot (8-byte stores at [sample][k].{re, im}).
// The published states are those of the all-lane walk (the split records differ from the all-lane ones only by fixed phases
// between an RX chunk's gates and the next diagonal, never at a publication point), so the sums, the chunk-gradient axes and
// the reduce kernel's gradient map are unchanged and the partial rows have the layout of bwd_ztri_kernel<N, 1>.
// Daggered split gates: RY(theta)^-1 flips the sign of the partner's coefficient (wire 4's swap form: x' = +-(u.x A - u.y B),
// - for the lanes with bit 4 set); a diagonal's inverse reads the other two of its record's three entries [-sin, cos, sin].
// ---------------------------------------------------------------------------------------
struct ZQSync { int psi_prod[2], lam_prod[2], ready[2], abort; int cursor[8]; int next; };
__host__ __device__ constexpr size_t zquad_fixed_lds(int ring) {
    return 4 * (size_t)kBlockRingBytes + 2 * (size_t)ring * 1024 + 1024 + 256 + (size_t)kAxisRing * 15 * sizeof(double);
}
__device__ __forceinline__ void split_phase_dag(double& x, const double2& d /* p = 0: (cos, sin); p = 1: (-sin, cos) */) {
    double A, B;
    swap_dup<true>(x, A, B);
    x = d.x * A + d.y * B;
}
template <int Q>
__device__ __forceinline__ void split_ry_dag(double& x, const double2& u, unsigned mask4) {
    if constexpr (Q == 4) {
        double A, B;
        swap_dup<false>(x, A, B);
        x = flip_sign(u.x * A - u.y * B, mask4);
    } else {
        const double own = u.x * x;
        x = fma(-u.y, xchg<(1 << Q)>(x), own);
    }
}

// The sigma waves' walk over a block-unrolled shape (one full RX chunk + LD sub-layers per block), shared by the kernels whose
// chains run in the split layout: steps drawn from a counter, the RX-chunk gradients read off the block's first sub-layer's
// products through the batch-invariant axes (see bwd_ztri_kernel, which has the same walk inline).  `prod`: NPROD consecutive
// producer counters, all of which must have reached a step before its slot is read.
template <int RING, int NPROD>
__device__ __forceinline__ void zsigma_walk(const ZBwdArgs& a, int lane, int me, long wave, double2 (*psi_ring)[64], double2 (*lam_ring)[64],
                                            const double* axis_ring, int* prod, int* abort_flag, int* cursor, int* next) {
    constexpr int N = 5;
    using C = Cfg<N>;
    const int E = a.E;
    double* __restrict__ part_w = a.partial + wave * (long)a.blk * C::KW;
    int seen_c[NPROD];
#pragma unroll
    for (int i = 0; i < NPROD; ++i) seen_c[i] = 0;
    auto products = [&](double (&acc3)[C::KW], const double2& pv, const double2 (&qv)[N], const double2& lm) {
#pragma unroll
        for (int i = 0; i < C::KW; ++i) acc3[i] = 0.0;
        static_for<0, N>([&](auto q) {
            constexpr int Q = decltype(q)::value;
            const unsigned m = lane_sign_mask<Q>(lane);
            acc3[3 * Q] = lm.x * qv[Q].y - lm.y * qv[Q].x;
            acc3[3 * Q + 1] = flip_sign(-(lm.x * qv[Q].x) - lm.y * qv[Q].y, m);
            acc3[3 * Q + 2] = flip_sign(lm.x * pv.y - lm.y * pv.x, m);
        });
    };
    auto store_sums = [&](double (&acc3)[C::KW], int sub) {
        const int vi = butterfly_sum<C::KW>(acc3, lane);
        if (butterfly_owner<C::KW>(lane)) store_through(&part_w[(long)sub * C::KW + vi], acc3[0]);
    };
    const int LDr = a.fast_ld, nsteps = a.nblocks * LDr;
    auto draw = [&]() {
        int v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return v;
    };
    int mine = __builtin_amdgcn_readfirstlane(draw());
    __hip_atomic_store(&cursor[me], mine, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    int after_v = draw();
    auto sigma_step = [&](auto chunk_c) {
        constexpr bool CHUNK = decltype(chunk_c)::value;
        const int t = mine;
        const int j = LDr == 2 ? (t >> 1) : t;
        const int sub = a.blk - 1 - t, bl = a.nblocks - 1 - j;
#pragma unroll
        for (int i = 0; i < NPROD; ++i) pair_wait_ge(&prod[i], t + 1, abort_flag, seen_c[i]);
        const double2* slot = psi_ring[t & (RING - 1)];
        const double2 pv = slot[lane];
        double2 qv[N];
        static_for<0, N>([&](auto q) { qv[decltype(q)::value] = slot[lane ^ (1 << decltype(q)::value)]; });
        const double2 lm = lam_ring[t & (RING - 1)][lane];
        double em[CHUNK ? 3 * N : 1];
        if constexpr (CHUNK) {
            const double* __restrict__ e = axis_ring + (bl & (kAxisRing - 1)) * (3 * N);
#pragma unroll
            for (int i = 0; i < 3 * N; ++i) em[i] = e[i];
        }
        if (lane == 0) __hip_atomic_store(&cursor[me], after_v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        mine = __builtin_amdgcn_readfirstlane(after_v);
        after_v = draw();
        double acc3[C::KW];
        products(acc3, pv, qv, lm);
        if constexpr (CHUNK) {
            double gx[C::KX];
#pragma unroll
            for (int i = 0; i < C::KX; ++i) gx[i] = 0.0;
            static_for<0, N>([&](auto q) {
                constexpr int Q = decltype(q)::value;
                gx[Q] = em[3 * Q] * acc3[3 * Q] + em[3 * Q + 1] * acc3[3 * Q + 1] + em[3 * Q + 2] * acc3[3 * Q + 2];
            });
            store_sums(acc3, sub);
            store_grad_x5(gx, lane, wave, a.B, E, a.grad_x, bl * N, N);
        } else {
            store_sums(acc3, sub);
        }
    };
    while (mine < nsteps) {
        if (LDr == 1 || (mine & 1)) sigma_step(std::true_type{});     // the block's first sub-layer is its last step
        else sigma_step(std::false_type{});
    }
}

template <int LD, int RING, int NSIG>
__device__ __forceinline__ void zquad_chain(const ZBwdArgs& a, int role /* 0: psi, 1: lambda */, int smp, int lane, bool valid, long b,
                                            const char* cs_tables, char* my_ring, double2 (*psi_ring)[64], double2 (*lam_ring)[64],
                                            double2* psi_final, ZQSync* sync, double* axis_ring) {
    constexpr int N = 5;
    __builtin_amdgcn_s_setprio(3);
    const int E = a.E, k = lane & 31, p = lane >> 5;
    const int ring_fwd = ring_source<N>(lane, false);
    const int ring_rev = ring_source<N>(lane, true);
    const unsigned mask4 = lane_sign_mask<4>(lane);
    SplitStream<LD> ss;
    ss.init_split(a.srec, a.L + 1, my_ring, lane);
    const char* row = cs_tables + (smp * (int)zyz_cs_row(N, E) + N) * 32;          // entry of column 0 of this sample's row
    const int slot_idx = (((smp << 5) | k) << 1) | p;                              // this lane's double in an all-lane slot
    int seen[NSIG], safe = 0;
#pragma unroll
    for (int w = 0; w < NSIG; ++w) seen[w] = 0;
    double x;
    if (role == 0) {
        if (a.state_in) x = a.state_in[(((b << N) + k) << 1) | p];
        else x = zsplit_forward<LD>(ss, row, a.nblocks, lane, ring_fwd);
        reinterpret_cast<double*>(psi_final)[slot_idx] = x;
        handoff_release();
        if (lane == 0) __hip_atomic_store(&sync->ready[smp], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        int seen_ready = 0;
        pair_wait_ge(&sync->ready[smp], 1, &sync->abort, seen_ready);
        const double2 f = psi_final[(smp << 5) | k];
        const double h = ham_weight<N>(k, a.off, a.co, a.diag);
        double v[1] = {h * (f.x * f.x + f.y * f.y)};
        lane_reduce<1, 5>(v, lane);                            // over the sample's 32 basis states (both halves hold the sum)
        const double pred = v[0] + (a.bias ? a.bias[0] : 0.0);
        if (a.out && valid && lane == 0) a.out[b] = pred;
        double gb = a.y ? 2.0 * (pred - a.y[b]) * a.inv_bt : a.g[b];
        if (!valid) gb = 0.0;
        x = gb * h * (p ? f.y : f.x);
    }
    double* ring = reinterpret_cast<double*>(role == 0 ? psi_ring : lam_ring);
    int* prod = role == 0 ? &sync->psi_prod[smp] : &sync->lam_prod[smp];
    int step = 0;
    auto publish_data = [&]() {                                // the state now, the counter after the layer's gates (ztri_chain)
        wait_slot_free<NSIG>(sync->cursor, step - RING + 1, &sync->abort, seen, safe);
        ring[(step & (RING - 1)) * 128 + slot_idx] = x;
        ++step;
    };
    auto publish_flag = [&]() {
        handoff_release();
        __hip_atomic_store(prod, step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // ---- block-unrolled reverse walk in the split layout (the all-lane one: ztri_chain)
    const int nb = a.nblocks;
    const unsigned a_dgd = (unsigned)k * 24u + (unsigned)(1 - p) * 8u;             // the inverse diagonal's two entries
    ss.template prime<-1>(nb, true);
    split_phase_dag(x, ss.rd8(ss.slot(nb), a_dgd));                                // block nb's slot: its record 0 is the final diagonal
    ss.template step<-1>(nb);
    const char* cs_b = row + (long)(nb - 1) * (N * 32);                            // chunk of the block at hand
    LayerCoef<N> ct, cm, c0;
    auto read_layer = [&](LayerCoef<N>& c, const char* sl, int rec) {
        c.dg = ss.rd8(sl, rec * kRecBytes + a_dgd);
        static_for<0, N>([&](auto q) { c.g[decltype(q)::value] = ss.rd(sl, rec * kRecBytes + ss.a_ry[decltype(q)::value]); });
        __builtin_amdgcn_sched_barrier(0);
    };
    auto read_chunk = [&](const char* sl, const char* chunk) {
        c0.dg = ss.rd8(sl, a_dgd);
        static_for<0, N>([&](auto q) { c0.g[decltype(q)::value] = ss.rd(chunk, ss.a_cs[decltype(q)::value]); });
        __builtin_amdgcn_sched_barrier(0);
    };
    auto undo_gates = [&](const LayerCoef<N>& c) {
        static_rfor<0, N>([&](auto q) { split_ry_dag<decltype(q)::value>(x, c.g[decltype(q)::value], mask4); });
        __builtin_amdgcn_sched_barrier(0);
    };
    read_layer(ct, ss.slot(nb - 1), LD);
    auto block = [&](auto sl, int bl, int kb) {
        const char* cur = ss.template slot_rel<0>(sl);
        const char* nx = ss.template slot_rel<-1>(sl);
        // the chunk-gradient axes of this block (prep_zyz_kernel leaves them in the chunk's split record too): handed to the
        // sigma waves by the first sample's lambda wave before it publishes the block's first step
        if (role == 1 && smp == 0 && lane < 3 * N)
            axis_ring[(bl & (kAxisRing - 1)) * (3 * N) + lane] = reinterpret_cast<const double*>(cur + kSRecRy)[lane];
        if constexpr (LD == 2) read_layer(cm, cur, 1);
        else read_chunk(cur, cs_b - kb * (N * 32));
        x = lane_gather(x, ring_rev);
        ss.template ahead_rel<-(kBDist + 1)>(sl, bl);          // in the gather's shadow
        __builtin_amdgcn_sched_barrier(0);
        publish_data();
        undo_gates(ct);
        publish_flag();
        if constexpr (LD == 2) {
            read_chunk(cur, cs_b - kb * (N * 32));
            split_phase_dag(x, ct.dg);
            x = lane_gather(x, ring_rev);
            __builtin_amdgcn_sched_barrier(0);
            publish_data();
            undo_gates(cm);
            publish_flag();
            split_phase_dag(x, cm.dg);
        } else {
            split_phase_dag(x, ct.dg);
        }
        ss.landed_late();                                     // block bl - 1 landed
        read_layer(ct, nx, LD);
        undo_gates(c0);                                       // the RX chunk (every gate in the RY form of the split records)
        split_phase_dag(x, c0.dg);
    };
    int bl = nb - 1;
    for (; bl >= 0 && (bl & (kBSlots - 1)) != kBSlots - 1; --bl) {
        block(RtSlot{bl}, bl, 0);
        cs_b -= N * 32;
    }
    for (; bl >= kBSlots - 1; bl -= kBSlots) {
        block(CtSlot<3>{}, bl, 0);
        block(CtSlot<2>{}, bl - 1, 1);
        block(CtSlot<1>{}, bl - 2, 2);
        block(CtSlot<0>{}, bl - 3, 3);
        cs_b -= kBSlots * (N * 32);
    }
}

template <int RING, int NSIG>
__global__ __launch_bounds__(64 * (4 + NSIG)) void bwd_zquad_kernel(ZBwdArgs a) {
    constexpr int N = 5;
    using C = Cfg<N>;
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
    char* fixed = dyn_lds;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tid = (int)threadIdx.x;
    char* rec_ring = fixed;
    double2 (*psi_ring)[64] = reinterpret_cast<double2 (*)[64]>(fixed + 4 * kBlockRingBytes);
    double2 (*lam_ring)[64] = reinterpret_cast<double2 (*)[64]>(fixed + 4 * kBlockRingBytes + RING * 1024);
    double2* psi_final = reinterpret_cast<double2*>(fixed + 4 * kBlockRingBytes + 2 * RING * 1024);
    ZQSync& sync = *reinterpret_cast<ZQSync*>(fixed + 4 * kBlockRingBytes + 2 * RING * 1024 + 1024);
    static_assert(sizeof(ZQSync) <= 256, "reserved");
    double* axis_ring = reinterpret_cast<double*>(fixed + 4 * kBlockRingBytes + 2 * RING * 1024 + 1024 + 256);
    char* cs_tables = dyn_lds + zquad_fixed_lds(RING);
    const long wave = blockIdx.x;                              // one sample group (two samples) per workgroup
    const int E = a.E;
    if (tid == 0) {
        sync.psi_prod[0] = sync.psi_prod[1] = sync.lam_prod[0] = sync.lam_prod[1] = 0;
        sync.ready[0] = sync.ready[1] = 0; sync.abort = 0;
        for (int w = 0; w < NSIG; ++w) sync.cursor[w] = 0;
        sync.next = 0;
    }
    fill_cs_split(reinterpret_cast<double4*>(cs_tables), a.src, E, wave * C::SPW, a.B, C::SPW, tid, 64 * (4 + NSIG));
    __syncthreads();

    if (wv < 4) {
        const int role = wv >> 1, smp = wv & 1;
        const long b_raw = wave * C::SPW + smp;
        const bool valid = b_raw < a.B;
        const long b = valid ? b_raw : a.B - 1;
        char* my_ring = rec_ring + wv * kBlockRingBytes;
        if (a.fast_ld == 2) zquad_chain<2, RING, NSIG>(a, role, smp, lane, valid, b, cs_tables, my_ring, psi_ring, lam_ring, psi_final, &sync, axis_ring);
        else zquad_chain<1, RING, NSIG>(a, role, smp, lane, valid, b, cs_tables, my_ring, psi_ring, lam_ring, psi_final, &sync, axis_ring);
    } else {
        zsigma_walk<RING, 4>(a, lane, wv - 4, wave, psi_ring, lam_ring, axis_ring, &sync.psi_prod[0], &sync.abort, sync.cursor, &sync.next);
    }
    report_abort(&sync.abort, a.status, lane);
}

// ---------------------------------------------------------------------------------------
// One-wave-per-sample-group backward kernel in the ZYZ form, for batches that fill the SIMDs (AUTO: more sample
// groups than SIMDs; block-unrolled shapes only, others keep bwd_kernel of hea_device.hpp).  At those batches the
// first-generation packed kernel is bound by the CU's single LDS pipe (per sample group and block ~170 LDS
// instructions: gate-table reads for three sweeps, sums staged through LDS, ds_swizzle exchanges), so this kernel
// spends vector instructions instead, which have headroom there:
//   * psi and lambda are walked back together and share every coefficient read; the DPP exchange of psi that a gate's
//     inverse needs also feeds that qubit's inner products (gates of a layer commute, so X,Y,Z of qubit q may be
//     taken when the other qubits' gates are already undone on both states);
//   * gradient sums by the register butterfly (pair sums over lane bits 4, 5 through permlane swaps), not through LDS;
//   * one record ring per WORKGROUP (BlockStream<SHARED>): 12 KB + 4 x 10 KB of (cos, sin) tables = 52 KB per four
//     waves, so three workgroups (12 waves) fit a CU.
// Only wire 4's partner fetch for the inner products (4 ds_swizzle per layer) and the ring gathers use the LDS pipe.
// ---------------------------------------------------------------------------------------
constexpr int kZPWaves = 4;
// row stride of the one-wave kernel's (cos, sin) tables: neighbouring rows SHARE their n entries of padding (padding is only
// ever read ahead, never used), n more entries follow the last row -- 640 bytes less per workgroup, which is what keeps three
// workgroups (54 KB each with the sums' staging area, in 1280-byte LDS granules) on a CU
__host__ __device__ inline long zp_cs_row(int n, long E) { return E + n; }
__host__ __device__ inline size_t zp_cs_bytes(int n, long E, int rows) { return ((size_t)rows * zp_cs_row(n, E) + n) * sizeof(double2); }

// psi <- RY_Q^-1 psi, lambda <- RY_Q^-1 lambda, and this lane's terms of X,Y,Z = Im<lambda|sigma_Q|psi> taken before
template <int Q>
__device__ __forceinline__ void ry_inv_with_inner(double& pr, double& pi, double& lr, double& li, const double2& u, int lane,
                                                  double& X, double& Y, double& Z) {
    const unsigned m = lane_sign_mask<Q>(lane);
    if constexpr (Q == 4) {
        const double qr = xchg<16>(pr), qi = xchg<16>(pi);
        X = lr * qi - li * qr;
        Y = flip_sign(-(lr * qr) - li * qi, m);
        Z = flip_sign(lr * pi - li * pr, m);
        apply_ry<4, true>(pr, pi, u);
        apply_ry<4, true>(lr, li, u);
    } else {
        const double qr = xchg<(1 << Q)>(pr), qi = xchg<(1 << Q)>(pi);
        X = lr * qi - li * qr;
        Y = flip_sign(-(lr * qr) - li * qi, m);
        Z = flip_sign(lr * pi - li * pr, m);
        pr = u.x * pr - u.y * qr;                 // RY^-1: sv -> -sv
        pi = u.x * pi - u.y * qi;
        apply_ry<Q, true>(lr, li, u);
    }
}
// the same for an encoding gate (apply_enc): its gradient is the X inner product (wires 0..3) or the Y one (wire 4)
template <int N, int Q>
__device__ __forceinline__ void enc_inv_with_inner(double (&pr)[1], double (&pi)[1], double (&lr)[1], double (&li)[1],
                                                   const double2& cs, int lane, double& G) {
    if constexpr (Q == 4) {
        const double qr = xchg<16>(pr[0]), qi = xchg<16>(pi[0]);
        G = flip_sign(-(lr[0] * qr) - li[0] * qi, lane_sign_mask<Q>(lane));
        apply_ry<4, true>(pr[0], pi[0], cs);
        apply_ry<4, true>(lr[0], li[0], cs);
    } else {
        const double qr = xchg<(1 << Q)>(pr[0]), qi = xchg<(1 << Q)>(pi[0]);
        G = lr[0] * qi - li[0] * qr;
        const double nr = cs.x * pr[0] - cs.y * qi;       // RX^-1: s -> -s in (c p + s q_i, c p_i - s q_r)
        pi[0] = cs.x * pi[0] + cs.y * qr;
        pr[0] = nr;
        apply_rx<N, Q>(lr, li, cs.x, -cs.y);
    }
}

template <int N, int LD>
__device__ __forceinline__ void zpacked_body(const ZBwdArgs& a, int wib, int lane, int klow, bool valid, long b, long wave,
                                             const double2* cs, char* wg_ring, double* comb /* [2][kZPWaves][2 * KW] */) {
    using C = Cfg<N>;
    const int E = a.E;
    const int ring_fwd = ring_source<N>(lane, false);
    const int ring_rev = ring_source<N>(lane, true);
    const double2* csrow = cs + (wib * C::SPW + (lane >> C::LB)) * (int)zp_cs_row(N, E) + N;
    BlockStream<N, LD, true> bs;
    bs.init(a.rec, a.L + 1, wg_ring, lane, klow);
    bs.loader = wib == 0;

    double pr[1], pi[1], lr[1], li[1];
    if (a.state_in) {
        const double2 s0 = reinterpret_cast<const double2*>(a.state_in)[(b << N) + klow];
        pr[0] = s0.x; pi[0] = s0.y;
    } else {
        zyz_forward_fast<N, LD>(pr, pi, a.runs, bs, csrow, E, lane, klow, ring_fwd);
    }
    {   // upstream weight and lambda_N = g H psi_N (in the read-out basis, then rotated back)
        double fr[1] = {pr[0]}, fi[1] = {pi[0]};
        basis_change<N, false>(fr, fi, a.pauli, lane);
        const double h = ham_weight<N>(klow, a.off, a.co, a.diag);
        double v[1] = {h * (fr[0] * fr[0] + fi[0] * fi[0])};
        lane_reduce<1, C::LB>(v, lane);
        const double pred = v[0] + (a.bias ? a.bias[0] : 0.0);
        if (a.out && valid && klow == 0) a.out[b] = pred;
        double gb = a.y ? 2.0 * (pred - a.y[b]) * a.inv_bt : a.g[b];
        if (!valid) gb = 0.0;
        lr[0] = gb * h * fr[0]; li[0] = gb * h * fi[0];
        basis_change<N, true>(lr, li, a.pauli, lane);
    }

    const int nb = a.nblocks;
    bs.template prime<-1>(nb, true);
    {
        const double2 d = bs.rd(bs.slot(nb), bs.a_dg);         // record L: the final diagonal
        apply_phase<true>(pr[0], pi[0], d);
        apply_phase<true>(lr[0], li[0], d);
    }
    bs.template step<-1>(nb);
    bs.load_records(bs.slot(nb - 1));
    const double2* cs_b = csrow + (long)(nb - 1) * N;             // chunk of the block at hand (every block has enc = n)
    bs.load_cs(cs_b, 0);
    // The workgroup's four waves walk the blocks in step (one barrier per block, BlockStream<SHARED>), so their gradient sums
    // are combined before they leave the chip: every wave parks a block's sums in LDS, and after the next block's barrier one
    // wave adds the four in a fixed order and writes ONE row per workgroup -- a quarter of the partial rows (B = 16384 at cfg 2:
    // 63 MB instead of 252 MB written and read back per call).
    double* __restrict__ row_w = a.partial + (long)blockIdx.x * a.blk * C::KW;
    constexpr int kSlotVals = 2 * C::KW;                          // room for LD <= 2 sub-layers of KW values
    auto combine = [&](int blk_done) {                            // the sums of block `blk_done` (parked before the last barrier)
        if (wib == (blk_done & (kZPWaves - 1)) && lane < LD * C::KW) {
            const double* c = comb + (blk_done & 1) * (kZPWaves * kSlotVals) + lane;
            double t = c[0];
#pragma unroll
            for (int w = 1; w < kZPWaves; ++w) t += c[w * kSlotVals];
            row_w[((long)blk_done * LD + lane / C::KW) * C::KW + lane % C::KW] = t;
        }
    };
    // one block, sitting in ring slot `sl` (walk unrolled over the ring slots like the other block walks)
    auto block = [&](auto sl, int bl, int kb) {
        bs.landed();
        if (bl + 1 < nb) combine(bl + 1);
        const char* nx = bs.template slot_rel<-1>(sl);
        double* park = comb + (bl & 1) * (kZPWaves * kSlotVals) + wib * kSlotVals;
#pragma unroll
        for (int s = LD - 1; s >= 0; --s) {
            if (s != LD - 1) {
                apply_phase<true>(pr[0], pi[0], bs.dg[s + 2]);
                apply_phase<true>(lr[0], li[0], bs.dg[s + 2]);
                bs.dg[s + 2] = bs.rd(nx, (s + 2) * kRecBytes + bs.a_dg);
            }
            pr[0] = lane_gather(pr[0], ring_rev); pi[0] = lane_gather(pi[0], ring_rev);
            lr[0] = lane_gather(lr[0], ring_rev); li[0] = lane_gather(li[0], ring_rev);
            if (s == LD - 1) bs.template ahead_rel<-(kBDist + 1)>(sl, bl);
            double acc3[C::KW];
#pragma unroll
            for (int i = 0; i < C::KW; ++i) acc3[i] = 0.0;
            static_rfor<0, N>([&](auto q) {
                constexpr int Q = decltype(q)::value;
                ry_inv_with_inner<Q>(pr[0], pi[0], lr[0], li[0], bs.ry[s][Q], lane, acc3[3 * Q], acc3[3 * Q + 1], acc3[3 * Q + 2]);
                bs.ry[s][Q] = bs.rd(nx, (1 + s) * kRecBytes + bs.a_ry[Q]);
            });
            const int vi = butterfly_sum<C::KW>(acc3, lane);
            if (butterfly_owner<C::KW>(lane)) park[s * C::KW + vi] = acc3[0];
        }
        apply_phase<true>(pr[0], pi[0], bs.dg[1]);
        apply_phase<true>(lr[0], li[0], bs.dg[1]);
        bs.dg[1] = bs.rd(nx, kRecBytes + bs.a_dg);
        const double2* cn = cs_b - (kb + 1) * N;                  // the previous block's chunk
        double gx[C::KX];
#pragma unroll
        for (int i = 0; i < C::KX; ++i) gx[i] = 0.0;
        static_rfor<0, N>([&](auto q) {
            constexpr int Q = decltype(q)::value;
            enc_inv_with_inner<N, Q>(pr, pi, lr, li, bs.cs[Q], lane, gx[Q]);
            bs.cs[Q] = cn[Q];
        });
        if constexpr (N == 5) store_grad_x5(gx, lane, wave, a.B, E, a.grad_x, bl * N, N);
        else store_grad_x<N>(gx, lane, wave, a.B, E, a.grad_x, bl * N, N);
        apply_phase<true>(pr[0], pi[0], bs.dg[0]);
        apply_phase<true>(lr[0], li[0], bs.dg[0]);
        bs.dg[0] = bs.rd(nx, bs.a_dg);
    };
    int bl = nb - 1;
    for (; bl >= 0 && (bl & (kBSlots - 1)) != kBSlots - 1; --bl) {     // down to a block in the last slot
        block(RtSlot{bl}, bl, 0);
        cs_b -= N;
    }
    for (; bl >= kBSlots - 1; bl -= kBSlots) {                    // bl = 3 (mod 4): block bl - K sits in slot 3 - K
        block(CtSlot<3>{}, bl, 0);
        block(CtSlot<2>{}, bl - 1, 1);
        block(CtSlot<1>{}, bl - 2, 2);
        block(CtSlot<0>{}, bl - 3, 3);
        cs_b -= kBSlots * N;
    }
    __syncthreads();
    combine(0);
}

template <int N>
__global__ __launch_bounds__(kZPWaves * 64) void bwd_zpacked_kernel(ZBwdArgs a) {
    using C = Cfg<N>;
    static_assert(C::R == 1, "all-lane layout");
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];                 // kZPWaves x SPW x (E + 2N) (cos, sin)
    __shared__ __attribute__((aligned(16))) char wg_ring[kBlockRingBytes];          // ONE record ring per workgroup
    __shared__ double comb[2 * kZPWaves * 2 * C::KW];                               // the waves' sums of two blocks (zpacked_body)
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long wave = (long)blockIdx.x * kZPWaves + wib;
    const long b_raw = wave * C::SPW + (lane >> C::LB);
    const bool valid = b_raw < a.B;
    const long b = valid ? b_raw : a.B - 1;
    const int klow = lane & (C::LANES - 1);
    double2* cs = reinterpret_cast<double2*>(dyn_lds);
    fill_cs(cs, a.src, N, a.E, (long)blockIdx.x * kZPWaves * C::SPW, a.B, kZPWaves * C::SPW, (int)threadIdx.x, kZPWaves * 64,
            (int)zp_cs_row(N, a.E));
    __syncthreads();
    if (a.fast_ld == 2) zpacked_body<N, 2>(a, wib, lane, klow, valid, b, wave, cs, wg_ring, comb);
    else zpacked_body<N, 1>(a, wib, lane, klow, valid, b, wave, cs, wg_ring, comb);
}

// Forward-only counterpart for batches that fill the SIMDs: four sweeping waves per workgroup on ONE shared record ring
// (52 KB per workgroup at cfg 2 instead of 44 KB per two sweeping waves: twice the resident sweeps per CU).
template <int N>
__global__ __launch_bounds__(kZPWaves * 64) void fwd_zshared_kernel(ZFwdArgs a) {
    using C = Cfg<N>;
    static_assert(C::R == 1, "all-lane layout");
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
    __shared__ __attribute__((aligned(16))) char wg_ring[kBlockRingBytes];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long wave = (long)blockIdx.x * kZPWaves + wib;
    const long b_raw = wave * C::SPW + (lane >> C::LB);
    const bool valid = b_raw < a.B;
    const long b = valid ? b_raw : a.B - 1;
    const int klow = lane & (C::LANES - 1);
    const int ring_fwd = ring_source<N>(lane, false);
    double2* cs = reinterpret_cast<double2*>(dyn_lds);
    fill_cs(cs, a.src, N, a.E, (long)blockIdx.x * kZPWaves * C::SPW, a.B, kZPWaves * C::SPW, (int)threadIdx.x, kZPWaves * 64);
    __syncthreads();
    const double2* csrow = cs + (wib * C::SPW + (lane >> C::LB)) * (int)zyz_cs_row(N, a.E) + N;
    double re[1], im[1];
    if (a.fast_ld == 2) {
        BlockStream<N, 2, true> bs;
        bs.init(a.rec, a.L + 1, wg_ring, lane, klow);
        bs.loader = wib == 0;
        zyz_forward_fast<N, 2>(re, im, a.runs, bs, csrow, a.E, lane, klow, ring_fwd);
    } else {
        BlockStream<N, 1, true> bs;
        bs.init(a.rec, a.L + 1, wg_ring, lane, klow);
        bs.loader = wib == 0;
        zyz_forward_fast<N, 1>(re, im, a.runs, bs, csrow, a.E, lane, klow, ring_fwd);
    }
    if (a.state_out && valid)
        reinterpret_cast<double2*>(a.state_out)[(b << N) + klow] = make_double2(re[0], im[0]);
    basis_change<N, false>(re, im, a.pauli, lane);
    double v[1] = {ham_weight<N>(klow, a.off, a.co, a.diag) * (re[0] * re[0] + im[0] * im[0])};
    lane_reduce<1, C::LB>(v, lane);
    if (valid && klow == 0) a.out[b] = v[0] + (a.bias ? a.bias[0] : 0.0);
}

// launch entry points (hea_inst.hip, n <= 5 only)
#ifdef QHEA_ZSUBSET     // development / test builds link a subset of the qubit counts (see the Makefile)
#define QHEA_FOR_EACH_ZN(X) QHEA_ZSUBSET(X)
#else
#define QHEA_FOR_EACH_ZN(X) X(2) X(3) X(4) X(5)
#endif
#define QHEA_ZDECLARE(NN)                                                              \
    void launch_fwd_zyz_##NN(dim3 grid, size_t dyn_lds, hipStream_t st, const ZFwdArgs& a); \
    void launch_fwd_zshared_##NN(dim3 grid, size_t dyn_lds, hipStream_t st, const ZFwdArgs& a); \
    void launch_bwd_ztri_##NN(dim3 grid, size_t dyn_lds, hipStream_t st, const ZBwdArgs& a);   \
    void launch_bwd_zpacked_##NN(dim3 grid, size_t dyn_lds, hipStream_t st, const ZBwdArgs& a);
QHEA_FOR_EACH_ZN(QHEA_ZDECLARE)
void launch_fwd_split_5(dim3 grid, size_t dyn_lds, hipStream_t st, const ZFwdArgs& a);
void launch_bwd_zquad_5(dim3 grid, size_t dyn_lds, hipStream_t st, const ZBwdArgs& a);
#undef QHEA_ZDECLARE

}  // namespace qhea
