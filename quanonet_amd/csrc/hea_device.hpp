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

constexpr int kWaves = 2;              // waves per workgroup: small, so that a small batch spreads over all CUs
                                       // (LDS and texture paths are per CU); measured at B=1024: 4 -> 218, 2 -> 204, 1 -> 205 us/step
constexpr int kMaxRuns = 16;           // run-length-encoded (count, enc, ld) block list
constexpr int kCsPerWave = 512;        // wave-private LDS staging of (cos,sin) pairs: 8 KB per wave
constexpr int kRedStride = 66;         // doubles per row of the wave-private reduction scratch (64 lanes + pad)
constexpr int kRedPerWave = 16 * kRedStride;   // n <= 5 (wave-pair kernel); the packed kernel sizes it by Cfg<N>::REDW
constexpr int kGateBytes = 64;         // one gate-table entry: 2 lane variants x (ar, s*ai, s*br, bi)

struct Runs {
    int nruns;
    int count[kMaxRuns];
    int enc[kMaxRuns];
    int ld[kMaxRuns];
};

// Frequency layers of the model-level calls (core/models_pt.py:14-68): encoding column e of segment `seg` is
//   x[b, e] = in[b, e % width] * w[e] + b[e]      (_TiledElementWise)     or     in[b, e % width] * scale   (w == NULL: _ScaleRepeat)
// segment 0 = trunk (QuanONet) / the only input (HEAQNN), segment 1 = branch; columns trunk first.
struct EncSeg {
    const double* in; const double* w; const double* b;
    double scale; int width; int ncols;
};
struct EncDesc { EncSeg seg[2]; };

// lane bits per sample: qubits 0 .. LB-1 live in the lane index, the rest in a per-lane register index.  (Fewer lane bits
// for n = 5 were measured in round 2 and lose at every BASELINE batch: profiles/r02_layout_sweep.txt)
__host__ __device__ constexpr int lane_bits(int n) { return n < 6 ? n : 6; }

template <int N>
struct Cfg {
    static constexpr int LB = lane_bits(N);         // lane bits per sample
    static constexpr int RB = N - LB;               // register bits
    static constexpr int R = 1 << RB;               // amplitudes per lane
    static constexpr int SPW = 64 >> LB;            // samples per wave
    static constexpr int LANES = 1 << LB;           // lanes per sample
    static constexpr int KW = (3 * N <= 8) ? 8 : (3 * N <= 16) ? 16 : (3 * N <= 32) ? 32 : 64;  // padded 3N
    static constexpr int KX = (N <= 2) ? 2 : (N <= 4) ? 4 : (N <= 8) ? 8 : 16;                   // padded N
    static constexpr int CAP = kCsPerWave / SPW;    // staged encoding columns per sample
    static constexpr int UD = (N <= 6) ? N : 1;     // gate-coefficient prefetch distance (gates)
    static constexpr bool LDSRED = N <= 5;          // gradient sums through LDS (few instructions) instead of the
                                                    // register butterfly.  Measured for n = 8 (K = 32, 17 KB of scratch
                                                    // per wave): the LDS footprint halves the resident waves at
                                                    // B = 2048 and the backward kernel gets 2x slower -> butterfly there.
    static constexpr int REDW = KW * 66;            // doubles of wave-private reduction scratch (rows of kRedStride)
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
// body(q) for the gate qubits q < m of an RX chunk: ONE wave-uniform test for the usual full chunk (m == n) instead
// of a compare-and-branch per gate
template <int N, class F>
__device__ __forceinline__ void for_gates_below(int m, F body) {
    if (m >= N) static_for<0, N>([&](auto q) { body(q); });
    else static_for<0, N>([&](auto q) { if (decltype(q)::value < m) body(q); });
}
template <int N, class F>
__device__ __forceinline__ void rfor_gates_below(int m, F body) {
    if (m >= N) static_rfor<0, N>([&](auto q) { body(q); });
    else static_rfor<0, N>([&](auto q) { if (decltype(q)::value < m) body(q); });
}

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

// own + partner(lane ^ MASK), the same value in both lanes of the pair.  For the two lane bits DPP cannot reach,
// gfx950's v_permlane16_swap / v_permlane32_swap applied to (x, copy of x) leave one register holding the own value
// and the other the partner's in EVERY lane (which is which differs by lane; a sum does not care): 4 vector
// instructions + the add and no LDS round trip, where ds_swizzle / ds_bpermute put 52 / 60 clocks of latency in
// front of the add.
template <int MASK>
__device__ __forceinline__ double pair_sum(double x) {
    if constexpr (MASK == 16 || MASK == 32) {
        const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
        if constexpr (MASK == 16) {
            const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
            const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
            return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
        } else {
            const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
            const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
            return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
        }
    } else {
        return x + xchg<MASK>(x);
    }
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
            v[0] = pair_sum<(1 << T)>(v[0]);
        }
        lane_reduce<K, BITS, T + 1>(v, lane);
    }
}

// Whole-wave sums of K = 8 or 16 values per lane with the transposing butterfly taken in the CHEAPEST order of lane
// bits.  A transposing step on lane bit t halves the live values: lanes with bit t = 0 end with own + partner of the
// even value of each pair, lanes with bit t = 1 of the odd one.  On bits 0..3 that costs 7 instructions per pair
// (4 selects, 2 DPP moves, 1 add: lane_reduce above); but
//   bit 4: v_permlane16_swap_b32 (A, B) swaps the odd 16-lane rows of A with the even rows of B -- afterwards A + B IS
//          the transposed pair sum: 2 swaps + 1 add per pair, no select;
//   bit 5: v_permlane32_swap_b32 likewise on the wave's halves;
//   bits 3, 2: two bank-masked DPP moves per 32-bit word (row_shr / row_shl by 8 or 4 into the lanes whose bit is 1 / 0,
//          the other lanes keep the old value) build (own A | partner's B) and (partner's A | own B): 4 moves + 1 add.
// So the steps with many pairs go to bits 4 and 5: 16 values cost 8*3 + 4*3 + 2*5 + 1*5 + 2*3 = 57 instructions
// instead of 111, 32 values (n = 6...9) 16*3 + 8*3 + 4*5 + 2*5 + 7 + 3 = 112 instead of 220.  Returns the index of the value whose 64-lane total this lane holds in v[0]; lanes whose remaining
// low bits are zero are the ones that should store it (`butterfly_owner`).
template <int K>
__device__ __forceinline__ int butterfly_sum(double (&v)[K], int lane) {
    static_assert(K == 8 || K == 16 || K == 32, "K");
    auto swap_level = [&](int c, auto which) {              // c live values -> c/2; bit 4 (which = 16) or bit 5 (32)
#pragma unroll
        for (int i = 0; i < K / 2; ++i) {
            if (i < c / 2) {
                const unsigned alo = (unsigned)__double2loint(v[2 * i]), ahi = (unsigned)__double2hiint(v[2 * i]);
                const unsigned blo = (unsigned)__double2loint(v[2 * i + 1]), bhi = (unsigned)__double2hiint(v[2 * i + 1]);
                if constexpr (decltype(which)::value == 16) {
                    const auto lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
                    const auto hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
                    v[i] = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
                } else {
                    const auto lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
                    const auto hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
                    v[i] = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
                }
            }
        }
    };
    swap_level(K, std::integral_constant<int, 16>{});
    swap_level(K / 2, std::integral_constant<int, 32>{});
    // bit 3 (xor 8): row_shr:8 = 0x118 into banks 2,3; row_shl:8 = 0x108 into banks 0,1
#pragma unroll
    for (int i = 0; i < K / 8; ++i) {
        const int alo = __double2loint(v[2 * i]), ahi = __double2hiint(v[2 * i]);
        const int blo = __double2loint(v[2 * i + 1]), bhi = __double2hiint(v[2 * i + 1]);
        const int a2lo = __builtin_amdgcn_update_dpp(alo, blo, 0x118, 0xF, 0xC, false), a2hi = __builtin_amdgcn_update_dpp(ahi, bhi, 0x118, 0xF, 0xC, false);
        const int b2lo = __builtin_amdgcn_update_dpp(blo, alo, 0x108, 0xF, 0x3, false), b2hi = __builtin_amdgcn_update_dpp(bhi, ahi, 0x108, 0xF, 0x3, false);
        v[i] = __hiloint2double(a2hi, a2lo) + __hiloint2double(b2hi, b2lo);
    }
    int idx = ((lane >> 4) & 1) | (((lane >> 5) & 1) << 1) | (((lane >> 3) & 1) << 2);
    if constexpr (K >= 16) {
        // bit 2 (xor 4): row_shr:4 = 0x114 into banks 1,3; row_shl:4 = 0x104 into banks 0,2
#pragma unroll
        for (int i = 0; i < K / 16; ++i) {
            const int alo = __double2loint(v[2 * i]), ahi = __double2hiint(v[2 * i]);
            const int blo = __double2loint(v[2 * i + 1]), bhi = __double2hiint(v[2 * i + 1]);
            const int a2lo = __builtin_amdgcn_update_dpp(alo, blo, 0x114, 0xF, 0xA, false), a2hi = __builtin_amdgcn_update_dpp(ahi, bhi, 0x114, 0xF, 0xA, false);
            const int b2lo = __builtin_amdgcn_update_dpp(blo, alo, 0x104, 0xF, 0x5, false), b2hi = __builtin_amdgcn_update_dpp(bhi, ahi, 0x104, 0xF, 0x5, false);
            v[i] = __hiloint2double(a2hi, a2lo) + __hiloint2double(b2hi, b2lo);
        }
        idx |= ((lane >> 2) & 1) << 3;
    } else {
        v[0] = pair_sum<4>(v[0]);
    }
    if constexpr (K >= 32) {                                  // bit 1: one pair left, select + quad exchange (7 instructions)
        const bool up = (lane >> 1) & 1;
        const double keep = up ? v[1] : v[0], send = up ? v[0] : v[1];
        v[0] = keep + xchg<2>(send);
        idx |= ((lane >> 1) & 1) << 4;
    } else {
        v[0] = pair_sum<2>(v[0]);
    }
    v[0] = pair_sum<1>(v[0]);
    return idx;
}
template <int K>
__device__ __forceinline__ bool butterfly_owner(int lane) { return (lane & (K == 32 ? 1 : (K == 16 ? 3 : 7))) == 0; }
// lane that holds value v after butterfly_sum<K> (the owner among those that do)
template <int K>
__device__ __forceinline__ int butterfly_lane_of(int v) {
    int l = ((v & 1) << 4) | (((v >> 1) & 1) << 5) | (((v >> 2) & 1) << 3);
    if constexpr (K >= 16) l |= ((v >> 3) & 1) << 2;
    if constexpr (K >= 32) l |= ((v >> 4) & 1) << 1;
    return l;
}

// Sums through a wave-private LDS scratch (rows of kRedStride doubles, one row per value, one column
// per lane).  Far fewer vector-ALU instructions than the register butterfly: K stores, K/2 wide loads,
// K adds and log2(64/K) exchange-adds.  The wave reads only what it wrote itself (LDS is in-order per
// wave), so no barrier is needed.  Summation order is fixed -> bitwise reproducible.
//
// The sums are split in two halves so that the LDS round trip overlaps the next sub-layer's gates:
// *_put() stores the lane values and issues the loads (no wait), *_finish() adds them up one
// sub-layer (or block) later.
//
// wave_sum: every lane contributes v[0..K); lane l ends with the 64-lane total of value l / (64/K).
template <int K>
__device__ __forceinline__ void wave_sum_put(const double (&v)[K], double* red, int lane, double (&t)[K]) {
    static_assert(K == 8 || K == 16 || K == 32, "K");
    constexpr int LPV = 64 / K;                       // lanes sharing one value
#pragma unroll
    for (int j = 0; j < K; ++j) red[j * kRedStride + lane] = v[j];
    const double* row = red + (lane / LPV) * kRedStride + (lane % LPV) * K;
#pragma unroll
    for (int i = 0; i < K; ++i) t[i] = row[i];
}
template <int K>
__device__ __forceinline__ double tree_sum(double (&t)[K]) {
#pragma unroll
    for (int w = K / 2; w > 0; w >>= 1)
#pragma unroll
        for (int i = 0; i < w; ++i) t[i] += t[i + w];
    return t[0];
}
template <int K>
__device__ __forceinline__ double wave_sum_finish(double (&t)[K]) {
    double s = tree_sum<K>(t);
    s += xchg<1>(s);
    if constexpr (64 / K >= 4) s += xchg<2>(s);
    if constexpr (64 / K >= 8) s += xchg<4>(s);
    return s;
}
// sample_sum: every lane contributes v[0..K); a wave holds 64>>LB samples of 2^LB lanes.  Lane l ends
// with the per-sample total of value (l / LPP) % K for sample (l / LPP) / K, LPP = 2^LB / K.
template <int K, int LB>
__device__ __forceinline__ void sample_sum_put(const double (&v)[K], double* red, int lane, double (&t)[K]) {
    constexpr int LANES = 1 << LB;
    constexpr int LPP = LANES / K;                    // lanes sharing one (sample, value) pair; each sums K entries
    static_assert(LPP >= 1 && LPP <= 8, "LPP");
#pragma unroll
    for (int j = 0; j < K; ++j) red[j * kRedStride + lane] = v[j];
    const int pidx = lane / LPP;
    const double* row = red + (pidx % K) * kRedStride + (pidx / K) * LANES + (lane % LPP) * K;
#pragma unroll
    for (int i = 0; i < K; ++i) t[i] = row[i];
}
template <int K, int LB>
__device__ __forceinline__ double sample_sum_finish(double (&t)[K]) {
    constexpr int LPP = (1 << LB) / K;
    double s = tree_sum<K>(t);
    if constexpr (LPP >= 2) s += xchg<1>(s);
    if constexpr (LPP >= 4) s += xchg<2>(s);
    if constexpr (LPP >= 8) s += xchg<4>(s);
    return s;
}

// ---------------------------------------------------------------------------------------
// gate application on the wave-resident state
// ---------------------------------------------------------------------------------------
// SU(2) gate [[a,b],[-conj b, conj a]] on qubit Q.  Coefficients come from the gate table already
// specialised for this lane: u = (ar, s*ai, s*br, bi) with s = +1 when the lane's bit Q is 0 and -1
// when it is 1 (for a register qubit s = +1).  The adjoint is the same call with (x,-y,-z,-w).
// Lane qubits 4 and 5 are out of DPP's reach (ds_swizzle / ds_bpermute: the CU's one LDS pipe and 50-60 clocks on the
// chain).  With two or more amplitudes per lane (n >= 7) the pair a gate on such a qubit mixes can be brought INTO one lane
// instead: v_permlane16/32_swap(re[r], re[r|1]) leaves (amplitude r of the bit-0 lane, amplitude r of the bit-1 lane) in
// the even rows / lower half and the same for amplitude r|1 in the odd rows / upper half -- every lane then holds complete
// pairs in (v[r], v[r|1]), the gate is the in-lane update of a register qubit with the UNSIGNED coefficients (gate-table
// variant 0 for every lane), and the same swaps put the results back.  The inner products of the backward pass are taken
// in that frame too (su2_inverse_with_inner), so they need no exchange of their own.
template <int N, int Q>
constexpr bool kSwapQubit = (Q >= 4) && (Q < Cfg<N>::LB) && (Cfg<N>::R >= 2);
template <int N, int Q>
__device__ __forceinline__ void swap_frame(double (&re)[Cfg<N>::R], double (&im)[Cfg<N>::R]) {
    auto sw = [](double& a, double& b) {
        const unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
        const unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
        if constexpr (Q == 4) {
            const auto lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
            const auto hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
            a = __hiloint2double((int)hi[0], (int)lo[0]); b = __hiloint2double((int)hi[1], (int)lo[1]);
        } else {
            const auto lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
            const auto hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
            a = __hiloint2double((int)hi[0], (int)lo[0]); b = __hiloint2double((int)hi[1], (int)lo[1]);
        }
    };
#pragma unroll
    for (int r0 = 0; r0 < Cfg<N>::R; r0 += 2) { sw(re[r0], re[r0 + 1]); sw(im[r0], im[r0 + 1]); }
}
// [[a, b], [-conj b, conj a]] on the in-lane pairs (r0, r0 | J)
template <int N, int J>
__device__ __forceinline__ void su2_in_lane(double (&re)[Cfg<N>::R], double (&im)[Cfg<N>::R], double ar, double ai, double br, double bi) {
#pragma unroll
    for (int r0 = 0; r0 < Cfg<N>::R; ++r0) {
        if (r0 & J) continue;
        const int r1 = r0 | J;
        const double p0r = re[r0], p0i = im[r0], p1r = re[r1], p1i = im[r1];
        re[r0] = ar * p0r - ai * p0i + br * p1r - bi * p1i;
        im[r0] = ar * p0i + ai * p0r + br * p1i + bi * p1r;
        re[r1] = ar * p1r + ai * p1i - br * p0r - bi * p0i;     // conj(a) p1 - conj(b) p0
        im[r1] = ar * p1i - ai * p1r - br * p0i + bi * p0r;
    }
}

template <int N, int Q>
__device__ __forceinline__ void apply_su2(double (&re)[Cfg<N>::R], double (&im)[Cfg<N>::R],
                                          double ar, double sai, double sbr, double bi) {
    using C = Cfg<N>;
    if constexpr (kSwapQubit<N, Q>) {
        swap_frame<N, Q>(re, im);
        su2_in_lane<N, 1>(re, im, ar, sai, sbr, bi);      // variant 0 of the gate table: unsigned
        swap_frame<N, Q>(re, im);
    } else if constexpr (Q < C::LB) {
#pragma unroll
        for (int r = 0; r < C::R; ++r) {
            const double pr = re[r], pi = im[r];
            const double qr = xchg<(1 << Q)>(pr), qi = xchg<(1 << Q)>(pi);
            re[r] = ar * pr - sai * pi + sbr * qr - bi * qi;
            im[r] = ar * pi + sai * pr + sbr * qi + bi * qr;
        }
    } else {
        constexpr int J = 1 << (Q - C::LB);
        const double ai = sai, br = sbr;
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
// n <= 6: the whole ring is one lane gather.  n > 6: CNOT_0..CNOT_4 (control and target both on lane bits) are
// one lane gather per register, the remaining CNOTs touch register bits and are applied one by one.
template <int N, bool REVERSE>
__device__ __forceinline__ void apply_ring(double (&re)[Cfg<N>::R], double (&im)[Cfg<N>::R],
                                           int lane, int ring_src_x4) {
    using C = Cfg<N>;
    if constexpr (C::RB == 0) {                        // whole state on lanes: one gather
        re[0] = lane_gather(re[0], ring_src_x4);
        im[0] = lane_gather(im[0], ring_src_x4);
    } else if constexpr (!REVERSE) {
#pragma unroll
        for (int r = 0; r < C::R; ++r) { re[r] = lane_gather(re[r], ring_src_x4); im[r] = lane_gather(im[r], ring_src_x4); }
        static_for<C::LB - 1, N>([&](auto i) { apply_cnot<N, (decltype(i)::value + 1) % N, decltype(i)::value>(re, im, lane); });
    } else {
        static_rfor<C::LB - 1, N>([&](auto i) { apply_cnot<N, (decltype(i)::value + 1) % N, decltype(i)::value>(re, im, lane); });
#pragma unroll
        for (int r = 0; r < C::R; ++r) { re[r] = lane_gather(re[r], ring_src_x4); im[r] = lane_gather(im[r], ring_src_x4); }
    }
}

// source lane (x4) of the lane-gather part of the ring: all N CNOTs for n <= 6, else CNOT_0..CNOT_{LB-2}
template <int N>
__device__ __forceinline__ int ring_source(int lane, bool reverse) {
    using C = Cfg<N>;
    constexpr int NC = C::RB == 0 ? N : C::LB - 1;      // CNOTs folded into the gather
    constexpr int M = C::RB == 0 ? N : C::LB;           // index bits they act on
    int k = lane & ((1 << M) - 1);
    if (!reverse) {     // final[k] = old[f0(f1(...f_{NC-1}(k)))]
        for (int i = NC - 1; i >= 0; --i) k ^= ((k >> ((i + 1) % M)) & 1) << i;
    } else {            // inverse: f_{NC-1}(...f0(k))
        for (int i = 0; i < NC; ++i) k ^= ((k >> ((i + 1) % M)) & 1) << i;
    }
    return ((lane & ~((1 << M) - 1)) | k) << 2;
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
            x = fma(-li[r], qr, fma(lr[r], qi, x));          // FMA chains: 2 instead of 3 fp64 instructions each
            y = fma(li[r], qi, fma(lr[r], qr, y));
            z = fma(-li[r], pr[r], fma(lr[r], pi[r], z));
        }
        const double s = ((lane >> Q) & 1) ? -1.0 : 1.0;
        X = x; Y = -s * y; Z = s * z;
    } else {
        constexpr int J = 1 << (Q - C::LB);
#pragma unroll
        for (int r0 = 0; r0 < C::R; ++r0) {
            if (r0 & J) continue;
            const int r1 = r0 | J;
            x = fma(-li[r1], pr[r0], fma(lr[r1], pi[r0], fma(-li[r0], pr[r1], fma(lr[r0], pi[r1], x))));
            y = fma(li[r1], pi[r0], fma(lr[r1], pr[r0], fma(-li[r0], pi[r1], fma(-lr[r0], pr[r1], y))));
            z = fma(li[r1], pr[r1], fma(-lr[r1], pi[r1], fma(-li[r0], pr[r0], fma(lr[r0], pi[r0], z))));
        }
        X = x; Y = y; Z = z;
    }
}
// X, Y, Z terms of gate (sub, Q) and the gate's inverse on psi and lambda (backward sweep); u = the lane's variant of the gate.
// Swap-form qubits: everything in the swapped frame, where the pair is in-lane.
template <int N, int Q>
__device__ __forceinline__ void su2_inverse_with_inner(double (&pr)[Cfg<N>::R], double (&pi)[Cfg<N>::R],
                                                       double (&lr)[Cfg<N>::R], double (&li)[Cfg<N>::R],
                                                       const double4& u, int lane, double& X, double& Y, double& Z) {
    if constexpr (kSwapQubit<N, Q>) {
        swap_frame<N, Q>(pr, pi);
        swap_frame<N, Q>(lr, li);
        double x = 0.0, y = 0.0, z = 0.0;
#pragma unroll
        for (int r0 = 0; r0 < Cfg<N>::R; r0 += 2) {
            const int r1 = r0 + 1;
            x = fma(-li[r1], pr[r0], fma(lr[r1], pi[r0], fma(-li[r0], pr[r1], fma(lr[r0], pi[r1], x))));
            y = fma(li[r1], pi[r0], fma(lr[r1], pr[r0], fma(-li[r0], pi[r1], fma(-lr[r0], pr[r1], y))));
            z = fma(li[r1], pr[r1], fma(-lr[r1], pi[r1], fma(-li[r0], pr[r0], fma(lr[r0], pi[r0], z))));
        }
        X = x; Y = y; Z = z;
        su2_in_lane<N, 1>(pr, pi, u.x, -u.y, -u.z, -u.w);
        su2_in_lane<N, 1>(lr, li, u.x, -u.y, -u.z, -u.w);
        swap_frame<N, Q>(pr, pi);
        swap_frame<N, Q>(lr, li);
    } else {
        pauli_inner<N, Q>(pr, pi, lr, li, lane, X, Y, Z);
        apply_su2<N, Q>(pr, pi, u.x, -u.y, -u.z, -u.w);
        apply_su2<N, Q>(lr, li, u.x, -u.y, -u.z, -u.w);
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
            x = fma(-li[r], qr, fma(lr[r], qi, x));
        }
    } else {
        constexpr int J = 1 << (Q - C::LB);
#pragma unroll
        for (int r0 = 0; r0 < C::R; ++r0) {
            if (r0 & J) continue;
            const int r1 = r0 | J;
            x = fma(-li[r1], pr[r0], fma(lr[r1], pi[r0], fma(-li[r0], pr[r1], fma(lr[r0], pi[r1], x))));
        }
    }
    return x;
}

template <int N>
__device__ __forceinline__ double ham_weight(int k, double off, double co, const double* __restrict__ diag) {
    if (diag) return diag[k];
    return off + co * (double)(N - 2 * (int)__popc((unsigned)k));
}

// Readout in another Pauli basis (reference ham_pauli, core/quantum_circuits_ms.py:28-39): sum_i <P_i> is the
// Z readout of V psi with V = RY(-pi/2) on every qubit for P = X and RX(+pi/2) for P = Y.  DAGGER undoes it.
// pauli: 0 = Z (nothing to do), 1 = X, 2 = Y.
template <int N, bool DAGGER>
__device__ __forceinline__ void basis_change(double (&re)[Cfg<N>::R], double (&im)[Cfg<N>::R], int pauli, int lane) {
    using C = Cfg<N>;
    constexpr double kR = 0.70710678118654752440;
    if (pauli == 1) {
        static_for<0, N>([&](auto q) {
            constexpr int Q = decltype(q)::value;
            double sbr = DAGGER ? -kR : kR;
            if constexpr (Q < C::LB && !kSwapQubit<N, Q>) sbr = ((lane >> Q) & 1) ? -sbr : sbr;   // (swap form: unsigned)
            apply_su2<N, Q>(re, im, kR, 0.0, sbr, 0.0);
        });
    } else if (pauli == 2) {
        static_for<0, N>([&](auto q) { apply_rx<N, decltype(q)::value>(re, im, kR, DAGGER ? -kR : kR); });
    }
}

// RX fold: a block's first RX chunk is folded, per sample, into the fused gates of the block's first sub-layer.
// U RX(theta) is again [[a,b],[-conj b, conj a]], and the formula holds unchanged for the lane-specialised variant
// (ar, s*ai, s*br, bi).  The merge is done ONCE per block on the gate table's copy in the wave's LDS ring
// (GateStream::fold: ~15 instructions, every lane merges 16 bytes) instead of per lane and gate -- the first
// attempt, merging in registers inside a second copy of the sub-layer body, cost 8 VALU per gate and ~200 VGPRs.
// Only with one sample per wave (n >= 6).  The encoding gradient Im<lam|X|psi> taken between RX and U equals
// n . (X,Y,Z) of the inner products taken after the merged gate, n = axis of U X U^dagger, so the reverse sweep needs
// no RX phase and no extra inner products for the folded chunk.  For n <= 5 a wave carries several samples: the ring
// would need a table copy per sample, the fold's three dependent LDS round trips cost more than the five cheap
// one-register RX gates they replace (measured at n = 5: forward 58.8 vs 54 us), and the per-sample X,Y,Z sums the
// reverse trick needs are the expensive ones -- so those kernels keep the separate RX phase.  (Merging per lane in
// registers instead, 8 multiply-adds per gate off the dependency chain, was also tried at n = 5: forward 62.0 vs
// 61.9 us -- the RX applications it replaces cost as many instructions, and the sweep is mostly issue-bound.)
template <int N>
constexpr bool kFold = Cfg<N>::SPW == 1;
__device__ __forceinline__ double4 merge_rx(const double4& u, const double2& cs) {
    return make_double4(u.x * cs.x + u.w * cs.y, u.y * cs.x - u.z * cs.y, u.z * cs.x + u.y * cs.y, u.w * cs.x - u.x * cs.y);
}
__device__ __forceinline__ void rotated_x_axis(const double4& u, double& nx, double& ny, double& nz) {
    nx = u.x * u.x - u.y * u.y - u.z * u.z + u.w * u.w;
    ny = -2.0 * (u.x * u.y - u.z * u.w);
    nz = 2.0 * (u.x * u.z + u.y * u.w);
}

// ---------------------------------------------------------------------------------------
// operand streams: gate coefficients (global, per-lane variant, rolling prefetch) and
// per-sample RX (cos,sin) pairs (wave-private LDS window + one-block register prefetch)
// ---------------------------------------------------------------------------------------
// Gate table (built by prep_kernel): entry g = sub*N + q is 64 bytes = two double4 variants,
// variant v for lanes whose bit q is v; one sub-layer = 64N contiguous bytes; N identity entries of
// padding on both sides.  Every lane needs one 32-byte variant per gate, but a wave only ever needs 2N
// distinct ones per sub-layer, and per-lane vector loads of them would cost the CU's single texture path
// 2 KB per gate per wave (measured: 36 % of the forward kernel).  So each wave streams the table through
// a private two-slot LDS ring instead: ONE buffer_load_dwordx4 (lanes 0..4N-1, 16 B each) fetches a whole
// sub-layer two sub-layers ahead, ONE ds_write_b128 parks it, and the lanes pick their variant with
// broadcast ds_read_b128 one sub-layer ahead (N <= 8: one register slot per qubit refilled in place right
// after use; N >= 9: the next gate's coefficients are read while the current, long gate runs).
using rsrc_t = __amdgpu_buffer_rsrc_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kRingBytesPerWave = 2 * QHEA_MAX_QUBITS * kGateBytes;      // two sub-layers

template <int N>
struct CsStream;

template <int N>
struct GateStream {
    using C = Cfg<N>;
    static constexpr bool SLOTS = N <= 8;
    static constexpr int SUBBYTES = N * kGateBytes;
    static_assert(2 * SUBBYTES <= kRingBytesPerWave && 4 * N <= 64, "gate ring");
    double4 slot[SLOTS ? N : 1];
    u32x4 stage;                    // this lane's 16 B of the sub-layer in flight from global memory
    rsrc_t rsrc;
    char* ring;                     // wave-private LDS: 2 x SUBBYTES
    int sub;                        // current sub-layer (wave-uniform)
    unsigned lane16;                // lane * 16 (clamped): byte offset of this lane's 16 B within a sub-layer
    bool loader;                    // lane < 4N
    unsigned voff[N];               // per-lane variant offset (0 or 32) for each qubit

    __device__ __forceinline__ void init(const char* table, int table_bytes, char* ring_wave, int lane) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(table), 0, table_bytes, 0x00020000);
        ring = ring_wave;
        loader = lane < 4 * N;
        lane16 = loader ? (unsigned)lane * 16u : 0u;
        static_for<0, N>([&](auto q) {
            constexpr int Q = decltype(q)::value;
            voff[Q] = (Q < C::LB && !kSwapQubit<N, Q>) ? (((unsigned)lane >> Q) & 1u) * 32u : 0u;   // swap-form qubits: variant 0
        });
    }
    __device__ __forceinline__ char* buf(int s) const { return ring + ((s & 1) ? SUBBYTES : 0); }
    __device__ __forceinline__ u32x4 gload(int s) const {          // sub-layer s (-1 and blk are the padding)
        return __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane16, (s + 1) * SUBBYTES, 0);
    }
    __device__ __forceinline__ void park(int s) {                   // stage -> LDS slot of sub-layer s
        // every lane stores: lanes >= 4N hold a copy of lane 0's 16 bytes (lane16 == 0 in gload and here), and writing
        // that again is cheaper than an exec-mask branch per sub-layer
        *reinterpret_cast<u32x4*>(buf(s) + lane16) = stage;
    }
    template <int Q>
    __device__ __forceinline__ double4 pick(int s) const {          // this lane's variant of gate (s, Q)
        return *reinterpret_cast<const double4*>(buf(s) + Q * kGateBytes + voff[Q]);
    }
    template <bool FWD>
    __device__ __forceinline__ void repick() {                      // (re)load the register slots for sub-layer `sub`
        if constexpr (SLOTS) {
            static_for<0, N>([&](auto q) { slot[decltype(q)::value] = pick<decltype(q)::value>(sub); });
        } else {
            slot[0] = FWD ? pick<0>(sub) : pick<N - 1>(sub);
        }
    }
    // position at sub-layer s0 for a forward (dir +1) or reverse (dir -1) walk
    template <bool FWD>
    __device__ __forceinline__ void prime(int s0) {
        constexpr int D = FWD ? 1 : -1;
        sub = s0;
        stage = gload(s0);
        park(s0);
        stage = gload(s0 + D);
        repick<FWD>();
    }
    // top of a sub-layer: park the next sub-layer's table (fetched one sub-layer ago), fetch the one after
    template <bool FWD>
    __device__ __forceinline__ void begin() {
        constexpr int D = FWD ? 1 : -1;
        park(sub + D);
        stage = gload(sub + 2 * D);
    }
    // RX fold (one sample per wave): gate g < m0 of the CURRENT sub-layer becomes U_g RX(x[col0 + g]) in the ring
    // (both lane variants), then the register slots are reloaded.  Each loader lane owns half a variant, (ar, s ai)
    // or (s br, bi), and its neighbour the other half; with own = (p0,p1), other = (q0,q1) both halves obey
    // (p0 c + q1 s, p1 c - q0 s).  Call before begin() of that sub-layer; the CsStream window must cover the columns.
    template <bool FWD>
    __device__ __forceinline__ void fold(const CsStream<N>& csx, int col0, int m0, int lane);
    // use: u = gs.cur<FWD,Q>(); ...apply...; gs.done<FWD,Q>();
    template <bool FWD, int Q>
    __device__ __forceinline__ double4 cur() {
        if constexpr (SLOTS) {
            return slot[Q];
        } else {
            const double4 u = slot[0];
            if constexpr (FWD) slot[0] = (Q + 1 < N) ? pick<(Q + 1) % N>(sub) : pick<0>(sub + 1);
            else               slot[0] = (Q > 0) ? pick<(Q + N - 1) % N>(sub) : pick<N - 1>(sub - 1);
            return u;
        }
    }
    template <bool FWD, int Q>
    __device__ __forceinline__ void done() {
        if constexpr (SLOTS) slot[Q] = pick<Q>(sub + (FWD ? 1 : -1));
    }
    template <bool FWD>
    __device__ __forceinline__ void advance() { sub += FWD ? 1 : -1; }
};

template <int N>
struct CsStream {
    using C = Cfg<N>;
    double2* lds;                   // wave-private region of kCsPerWave entries
    const double2* src;             // global row of this lane's sample: cs + b*E
    int E, klow, srow, lo, hi;      // staged column window [lo,hi) (wave-uniform)
    double2 nxt[N];                 // register prefetch of the next block's first RX chunk

    __device__ __forceinline__ void init(double2* lds_wave, const double2* cs, long b, int E_, int lane,
                                         int sample_in_wave) {
        lds = lds_wave; src = cs + b * E_; E = E_;
        klow = lane & (C::LANES - 1);
        srow = sample_in_wave * C::CAP;
        lo = hi = 0;
    }
    // every sample's LANES lanes stage that sample's columns [c0, c0+len): 8 loads in flight per lane
    __device__ __forceinline__ void refill(int c0) {
        const int len = (E - c0) < C::CAP ? (E - c0) : C::CAP;
        constexpr int BATCH = 8;
        for (int e0 = 0; e0 < len; e0 += BATCH * C::LANES) {
            double2 t[BATCH];
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int e = e0 + i * C::LANES + klow;
                t[i] = src[c0 + (e < len ? e : len - 1)];
            }
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int e = e0 + i * C::LANES + klow;
                if (e < len) lds[srow + e] = t[i];
            }
        }
        lo = c0; hi = c0 + len;
    }
    // make columns [c, c+m) resident; FWD windows start at c, reverse windows end at c+m
    template <bool FWD>
    __device__ __forceinline__ void need(int c, int m) {
        c = __builtin_amdgcn_readfirstlane(c);
        m = __builtin_amdgcn_readfirstlane(m);
        if (c < __builtin_amdgcn_readfirstlane(lo) || c + m > __builtin_amdgcn_readfirstlane(hi)) {
            int c0 = c;
            if (!FWD) { c0 = c + m - C::CAP; if (c0 < 0) c0 = 0; }
            refill(c0);
        }
    }
    __device__ __forceinline__ double2 at(int c) const { return lds[srow + (c - lo)]; }
    // nxt[q] = (cos,sin) of column c+q; entries past the block (or past E) are staged garbage, never used
    template <bool FWD>
    __device__ __forceinline__ void prefetch(int c) {
        if (c >= E) return;
        const int m = (E - c) < N ? (E - c) : N;
        need<FWD>(c, m);
        const double2* p = lds + srow + (c - lo);
        static_for<0, N>([&](auto q) { nxt[decltype(q)::value] = p[decltype(q)::value]; });
    }
};

template <int N>
template <bool FWD>
__device__ __forceinline__ void GateStream<N>::fold(const CsStream<N>& csx, int col0, int m0, int lane) {
    static_assert(kFold<N>, "one sample per wave");
    const int g = lane >> 2;                                  // gate this lane's 16 B belong to
    double2* mine = reinterpret_cast<double2*>(buf(sub) + lane16);
    const double2 p = *mine;
    const double q0 = xchg<1>(p.x), q1 = xchg<1>(p.y);
    double2 cs = make_double2(1.0, 0.0);
    if (loader && g < m0) cs = csx.lds[col0 + g - csx.lo];
    if (loader) *mine = make_double2(p.x * cs.x + q1 * cs.y, p.y * cs.x - q0 * cs.y);
    repick<FWD>();
}

// Deferred gradient sums of the backward kernel (N <= 5): values go through the wave's LDS scratch; the
// additions and the store of sub-layer s / block b run one sub-layer / block later.
template <int N>
struct GradSums {
    using C = Cfg<N>;
    double tw[C::KW];       // loaded rows of the pending sub-layer (X,Y,Z sums)
    double tx[C::KX];       // loaded rows of the pending RX chunk
    int sub_w, col_x, m_x;  // what is pending (-1: nothing)
    double* red;
    int lane;
    long wave, B;
    int E;
    double* __restrict__ part_w;
    double* __restrict__ grad_x;

    __device__ __forceinline__ void flush_w() {
        if (sub_w >= 0) {
            const double tot = wave_sum_finish<C::KW>(tw);
            constexpr int LPV = 64 / C::KW;
            if (lane % LPV == 0) part_w[(long)sub_w * C::KW + lane / LPV] = tot;
            sub_w = -1;
        }
    }
    __device__ __forceinline__ void flush_x() {
        if (col_x >= 0) {
            const double tot = sample_sum_finish<C::KX, C::LB>(tx);
            constexpr int LPP = C::LANES / C::KX;
            const int pidx = lane / LPP, j = pidx % C::KX;
            const long bs = wave * C::SPW + pidx / C::KX;
            if (lane % LPP == 0 && j < m_x && bs < B) grad_x[bs * E + col_x + j] = tot;
            col_x = -1;
        }
    }
    __device__ __forceinline__ void put_w(const double (&acc3)[C::KW], int sub) {
        flush_w();
        wave_sum_put<C::KW>(acc3, red, lane, tw);
        sub_w = sub;
    }
    __device__ __forceinline__ void put_x(const double (&gx)[C::KX], int col, int m) {
        flush_x();
        sample_sum_put<C::KX, C::LB>(gx, red, lane, tx);
        col_x = col; m_x = m;
    }
};

// register-butterfly versions (N >= 6)
template <int N>
__device__ __forceinline__ void store_grad_x(double (&gx)[Cfg<N>::KX], int lane, long wave, long B,
                                             int E, double* __restrict__ grad_x, int col, int m) {
    using C = Cfg<N>;
    lane_reduce<C::KX, C::LB>(gx, lane);          // lane with klow == j holds value j (KX <= 2^LB for every N)
    const int klow = lane & (C::LANES - 1);
    const long bs = wave * C::SPW + (lane >> C::LB);
    if (bs < B && klow < m) grad_x[bs * E + col + klow] = gx[0];
}

// A result the NEXT launch reads (partial gradient rows, grad_x): stored write-through (`global_store ... sc1`, what a relaxed
// agent-scope atomic store lowers to) instead of left dirty in this XCD's L2 for the end-of-kernel write-back -- the 6 MB a
// cfg-2 launch produces then leave the chip while the kernel still runs (-1 % per step at B = 512 and 1024).
__device__ __forceinline__ void store_through(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (16 bytes: one `global_store_dwordx4 ... sc1`; the atomic builtins stop at 8)
__device__ __forceinline__ void store_through(double2* p, const double2& v) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    const v2d x = {v.x, v.y};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(x) : "memory");
}

// n = 5 (two samples of 32 lanes per wave, KX = 8): the same per-sample sums with the transposing steps in the cheap
// order of butterfly_sum -- bit 4 by v_permlane16_swap (4 pairs x 3), bits 3 and 2 by bank-masked DPP (2 x 5 + 5), bits
// 1, 0 as plain pair sums: 33 instructions instead of 55.  The lane with bits (4, 3, 2) = (j0, j1, j2) and bits 1, 0
// clear ends with value j0 + 2 j1 + 4 j2 of its sample.
__device__ __forceinline__ void store_grad_x5(double (&v)[8], int lane, long wave, long B, int E,
                                              double* __restrict__ grad_x, int col, int m) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v[2 * i]), (unsigned)__double2loint(v[2 * i + 1]), false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v[2 * i]), (unsigned)__double2hiint(v[2 * i + 1]), false, false);
        v[i] = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {       // bit 3: row_shr:8 into banks 2,3 / row_shl:8 into banks 0,1
        const int alo = __double2loint(v[2 * i]), ahi = __double2hiint(v[2 * i]);
        const int blo = __double2loint(v[2 * i + 1]), bhi = __double2hiint(v[2 * i + 1]);
        const int a2lo = __builtin_amdgcn_update_dpp(alo, blo, 0x118, 0xF, 0xC, false), a2hi = __builtin_amdgcn_update_dpp(ahi, bhi, 0x118, 0xF, 0xC, false);
        const int b2lo = __builtin_amdgcn_update_dpp(blo, alo, 0x108, 0xF, 0x3, false), b2hi = __builtin_amdgcn_update_dpp(bhi, ahi, 0x108, 0xF, 0x3, false);
        v[i] = __hiloint2double(a2hi, a2lo) + __hiloint2double(b2hi, b2lo);
    }
    {                                   // bit 2: row_shr:4 into banks 1,3 / row_shl:4 into banks 0,2
        const int alo = __double2loint(v[0]), ahi = __double2hiint(v[0]);
        const int blo = __double2loint(v[1]), bhi = __double2hiint(v[1]);
        const int a2lo = __builtin_amdgcn_update_dpp(alo, blo, 0x114, 0xF, 0xA, false), a2hi = __builtin_amdgcn_update_dpp(ahi, bhi, 0x114, 0xF, 0xA, false);
        const int b2lo = __builtin_amdgcn_update_dpp(blo, alo, 0x104, 0xF, 0x5, false), b2hi = __builtin_amdgcn_update_dpp(bhi, ahi, 0x104, 0xF, 0x5, false);
        v[0] = __hiloint2double(a2hi, a2lo) + __hiloint2double(b2hi, b2lo);
    }
    v[0] = pair_sum<2>(v[0]);
    v[0] = pair_sum<1>(v[0]);
    const int j = ((lane >> 4) & 1) | (((lane >> 3) & 1) << 1) | (((lane >> 2) & 1) << 2);
    const long bs = wave * 2 + (lane >> 5);
    if ((lane & 3) == 0 && bs < B && j < m) store_through(&grad_x[bs * E + col + j], v[0]);
}

// ---------------------------------------------------------------------------------------
// forward sweep (shared by the forward and backward kernels)
// ---------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void forward_sweep(double (&re)[Cfg<N>::R], double (&im)[Cfg<N>::R],
                                              const Runs& runs, CsStream<N>& csx, GateStream<N>& gs,
                                              int lane, int ring_fwd) {
    using C = Cfg<N>;
#pragma unroll
    for (int r = 0; r < C::R; ++r) { re[r] = 0.0; im[r] = 0.0; }
    if ((lane & (C::LANES - 1)) == 0) re[0] = 1.0;
    int col = 0;
    gs.template prime<true>(0);
    csx.template prefetch<true>(0);
    for (int ri = 0; ri < runs.nruns; ++ri) {
        const int ne = runs.enc[ri], nld = runs.ld[ri];
        const bool fold = kFold<N> && nld > 0 && ne > 0;   // first RX chunk rides on the first sub-layer's gates
        const int m0 = ne < N ? ne : N;
        for (int rep = 0; rep < runs.count[ri]; ++rep) {
            if (!fold) {
                for_gates_below<N>(ne, [&](auto q) {
                    constexpr int Q = decltype(q)::value;
                    apply_rx<N, Q>(re, im, csx.nxt[Q].x, csx.nxt[Q].y);
                });
            }
            for (int j0 = N; j0 < ne; j0 += N) {          // more encodings than wires (not used by the reference)
                const int m = (ne - j0) < N ? (ne - j0) : N;
                csx.template need<true>(col + j0, m);
                static_for<0, N>([&](auto q) {
                    constexpr int Q = decltype(q)::value;
                    if (Q < m) { const double2 c = csx.at(col + j0 + Q); apply_rx<N, Q>(re, im, c.x, c.y); }
                });
            }
            if constexpr (kFold<N>) {
                if (fold) {
                    csx.template need<true>(col, m0);
                    gs.template fold<true>(csx, col, m0, lane);
                }
            }
            col += ne;
            csx.template prefetch<true>(col);              // next block's angles: two sub-layers of cover
            for (int l = 0; l < nld; ++l) {
                gs.template begin<true>();
                static_for<0, N>([&](auto q) {
                    constexpr int Q = decltype(q)::value;
                    const double4 u = gs.template cur<true, Q>();
                    apply_su2<N, Q>(re, im, u.x, u.y, u.z, u.w);
                    gs.template done<true, Q>();
                });
                gs.template advance<true>();
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
                                                          const char* __restrict__ gates, int gates_bytes,
                                                          double off, double co,
                                                          const double* __restrict__ diag, int pauli,
                                                          double* __restrict__ out,
                                                          double* __restrict__ state_out,
                                                          const double* __restrict__ bias) {
    using C = Cfg<N>;
    __shared__ double2 cs_lds[kWaves * kCsPerWave + 16];   // +16: slack for the unclamped prefetch
    __shared__ __attribute__((aligned(16))) char gate_ring[kWaves * kRingBytesPerWave];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long wave = (long)blockIdx.x * kWaves + wib;
    const long b_raw = wave * C::SPW + (lane >> C::LB);
    const bool valid = b_raw < B;
    const long b = valid ? b_raw : B - 1;
    const int klow = lane & (C::LANES - 1);
    const int ring_fwd = ring_source<N>(lane, false);

    CsStream<N> csx;
    csx.init(cs_lds + wib * kCsPerWave, cs, b, E, lane, lane >> C::LB);
    GateStream<N> gs;
    gs.init(gates, gates_bytes, gate_ring + wib * kRingBytesPerWave, lane);

    double re[C::R], im[C::R];
    forward_sweep<N>(re, im, runs, csx, gs, lane, ring_fwd);

    if (state_out && valid) {                                  // psi_N, before any readout-basis change
#pragma unroll
        for (int r = 0; r < C::R; ++r) {
            double2* dst = reinterpret_cast<double2*>(state_out) + (b << N) + ((r << C::LB) | klow);
            *dst = make_double2(re[r], im[r]);
        }
    }
    basis_change<N, false>(re, im, pauli, lane);
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < C::R; ++r)
        acc += ham_weight<N>((r << C::LB) | klow, off, co, diag) * (re[r] * re[r] + im[r] * im[r]);
    double v[1] = {acc};
    lane_reduce<1, C::LB>(v, lane);
    if (valid && klow == 0) out[b] = v[0] + (bias ? bias[0] : 0.0);
}

// MINW = waves per SIMD the register allocation must leave room for.  At n = 8 / 9 the kernel wants 278 / 308 registers:
// fine while every wave has a SIMD to itself, but once the batch puts two waves on a SIMD it pays to cap the
// allocation at 256 (20 / 116 spilled) so that both are resident -- measured, forward + backward of 24 sub-layers,
// uncapped / capped: n = 8 B = 1024 136 / 179 us, B = 2048 249 / 206 us, B = 4096 484 / 378 us; n = 9 B = 1024
// 234 / 329 us, B = 2048 446 / 379 us, B = 4096 875 / 711 us.  The host picks (BwdArgs::dense).
template <int N, int MINW = 1>
__global__ __launch_bounds__(kWaves * 64, MINW) void bwd_kernel(Runs runs, long B, int E, int blk,
                                                          const double2* __restrict__ cs,
                                                          const char* __restrict__ gates, int gates_bytes,
                                                          double off, double co,
                                                          const double* __restrict__ diag, int pauli,
                                                          const double* __restrict__ g,
                                                          const double* __restrict__ state_in,
                                                          const double* __restrict__ y,
                                                          const double* __restrict__ bias,
                                                          double inv_bt,
                                                          double* __restrict__ out,
                                                          double* __restrict__ grad_x,
                                                          double* __restrict__ partial) {
    using C = Cfg<N>;
    __shared__ double2 cs_lds[kWaves * kCsPerWave + 16];   // +16: slack for the unclamped prefetch
    __shared__ double red_lds[C::LDSRED ? kWaves * C::REDW : 1];
    __shared__ __attribute__((aligned(16))) char gate_ring[kWaves * kRingBytesPerWave];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;         // (scalarising it costs the n = 8 kernel 3 %: measured)
    double* red = red_lds + (C::LDSRED ? wib * C::REDW : 0);
    const long wave = (long)blockIdx.x * kWaves + wib;
    const long b_raw = wave * C::SPW + (lane >> C::LB);
    const bool valid = b_raw < B;
    const long b = valid ? b_raw : B - 1;
    const int klow = lane & (C::LANES - 1);
    const int ring_fwd = ring_source<N>(lane, false);
    const int ring_rev = ring_source<N>(lane, true);

    CsStream<N> csx;
    csx.init(cs_lds + wib * kCsPerWave, cs, b, E, lane, lane >> C::LB);
    GateStream<N> gs;
    gs.init(gates, gates_bytes, gate_ring + wib * kRingBytesPerWave, lane);

    double pr[C::R], pi[C::R], lr[C::R], li[C::R];
    if (state_in) {
#pragma unroll
        for (int r = 0; r < C::R; ++r) {
            const double2 a = reinterpret_cast<const double2*>(state_in)[(b << N) + ((r << C::LB) | klow)];
            pr[r] = a.x; pi[r] = a.y;
        }
    } else {
        forward_sweep<N>(pr, pi, runs, csx, gs, lane, ring_fwd);
    }

    basis_change<N, false>(pr, pi, pauli, lane);
    // upstream weight: given (g), or the fused MSE residual 2 (out + bias - y) / batch_total when y != NULL
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < C::R; ++r)
        acc += ham_weight<N>((r << C::LB) | klow, off, co, diag) * (pr[r] * pr[r] + pi[r] * pi[r]);
    double gb;
    if (y || out) {
        double v[1] = {acc};
        lane_reduce<1, C::LB>(v, lane);                      // butterfly: every lane of the sample gets the sum
        const double pred = v[0] + (bias ? bias[0] : 0.0);
        if (out && valid && klow == 0) out[b] = pred;
        gb = y ? 2.0 * (pred - y[b]) * inv_bt : g[b];
    } else {
        gb = g[b];
    }
    if (!valid) gb = 0.0;                                    // padding lanes carry lambda = 0: no gradient contribution
#pragma unroll
    for (int r = 0; r < C::R; ++r) {
        const double h = ham_weight<N>((r << C::LB) | klow, off, co, diag);
        lr[r] = gb * h * pr[r];
        li[r] = gb * h * pi[r];
    }
    if (pauli) {
        basis_change<N, true>(pr, pi, pauli, lane);
        basis_change<N, true>(lr, li, pauli, lane);
    }

    double* __restrict__ part_w = partial + wave * (long)blk * C::KW;
    GradSums<N> sums;
    sums.sub_w = -1; sums.col_x = -1; sums.m_x = 0; sums.red = red; sums.lane = lane; sums.wave = wave; sums.B = B;
    sums.E = E; sums.part_w = part_w; sums.grad_x = grad_x;
    int col = E, sub = blk;
    gs.template prime<false>(blk - 1);
    for (int ri = runs.nruns - 1; ri >= 0; --ri) {
        const int ne = runs.enc[ri], nld = runs.ld[ri];
        const bool one_chunk = ne <= N;
        const bool fold = kFold<N> && nld > 0 && ne > 0;   // the block's first RX chunk rides on sub-layer 0's gates
        const int m0 = ne < N ? ne : N;
        for (int rep = 0; rep < runs.count[ri]; ++rep) {
            if (one_chunk && !fold && ne > 0) csx.template prefetch<false>(col - ne);   // this block's angles, used after its sub-layers
            for (int l = nld - 1; l >= 0; --l) {
                --sub;
                const bool folded = fold && l == 0;
                if constexpr (kFold<N>) {
                    if (folded) {
                        csx.template need<false>(col - ne, m0);
                        gs.template fold<false>(csx, col - ne, m0, lane);
                    }
                }
                apply_ring<N, true>(pr, pi, lane, ring_rev);
                apply_ring<N, true>(lr, li, lane, ring_rev);
                double acc3[C::KW];
#pragma unroll
                for (int i = 0; i < C::KW; ++i) acc3[i] = 0.0;
                gs.template begin<false>();
                static_rfor<0, N>([&](auto q) {
                    constexpr int Q = decltype(q)::value;
                    const double4 u = gs.template cur<false, Q>();
                    su2_inverse_with_inner<N, Q>(pr, pi, lr, li, u, lane, acc3[3 * Q], acc3[3 * Q + 1], acc3[3 * Q + 2]);
                    gs.template done<false, Q>();
                });
                gs.template advance<false>();
                if constexpr (C::LDSRED) {
                    sums.put_w(acc3, sub);
                } else {
                    const int vi = butterfly_sum<C::KW>(acc3, lane);   // the sample's total of value vi (cheapest lane-bit order)
                    if (butterfly_owner<C::KW>(lane)) part_w[(long)sub * C::KW + vi] = acc3[0];
                    if constexpr (kFold<N>) {
                        if (folded) {                                // encoding gradients of the folded chunk: n . (X,Y,Z)
                            const int q = vi / 3;                    // the lane holding X_q fetches Y_q and Z_q
                            const double Y = __shfl(acc3[0], butterfly_lane_of<C::KW>(vi + 1 < C::KW ? vi + 1 : vi));
                            const double Z = __shfl(acc3[0], butterfly_lane_of<C::KW>(vi + 2 < C::KW ? vi + 2 : vi));
                            if (butterfly_owner<C::KW>(lane) && vi % 3 == 0 && q < m0 && valid) {
                                const double4 ub = *reinterpret_cast<const double4*>(gates + ((long)(sub + 1) * N + q) * kGateBytes);
                                double nx, ny, nz;
                                rotated_x_axis(ub, nx, ny, nz);
                                grad_x[b * E + (col - ne) + q] = nx * acc3[0] + ny * Y + nz * Z;
                            }
                        }
                    }
                }
            }
            col -= ne;
            if (fold) {
                const int nchunks = (ne + N - 1) / N;
                for (int ch = nchunks - 1; ch >= 1; --ch) {          // chunk 0 was folded
                    const int j0 = ch * N;
                    const int m = (ne - j0) < N ? (ne - j0) : N;
                    csx.template need<false>(col + j0, m);
                    double gx[C::KX];
#pragma unroll
                    for (int i = 0; i < C::KX; ++i) gx[i] = 0.0;
                    static_rfor<0, N>([&](auto q) {
                        constexpr int Q = decltype(q)::value;
                        if (Q < m) {
                            const double2 c = csx.at(col + j0 + Q);
                            gx[Q] = pauli_x_inner<N, Q>(pr, pi, lr, li);
                            apply_rx<N, Q>(pr, pi, c.x, -c.y);
                            apply_rx<N, Q>(lr, li, c.x, -c.y);
                        }
                    });
                    store_grad_x<N>(gx, lane, wave, B, E, grad_x, col + j0, m);
                }
            } else if (one_chunk) {
                if (ne > 0) {
                    double gx[C::KX];
#pragma unroll
                    for (int i = 0; i < C::KX; ++i) gx[i] = 0.0;
                    rfor_gates_below<N>(ne, [&](auto q) {
                        constexpr int Q = decltype(q)::value;
                        gx[Q] = pauli_x_inner<N, Q>(pr, pi, lr, li);
                        apply_rx<N, Q>(pr, pi, csx.nxt[Q].x, -csx.nxt[Q].y);
                        apply_rx<N, Q>(lr, li, csx.nxt[Q].x, -csx.nxt[Q].y);
                    });
                    if constexpr (C::LDSRED) sums.put_x(gx, col, ne);
                    else store_grad_x<N>(gx, lane, wave, B, E, grad_x, col, ne);
                }
            } else {
                const int nchunks = (ne + N - 1) / N;
                for (int ch = nchunks - 1; ch >= 0; --ch) {
                    const int j0 = ch * N;
                    const int m = (ne - j0) < N ? (ne - j0) : N;
                    csx.template need<false>(col + j0, m);
                    double gx[C::KX];
#pragma unroll
                    for (int i = 0; i < C::KX; ++i) gx[i] = 0.0;
                    static_rfor<0, N>([&](auto q) {
                        constexpr int Q = decltype(q)::value;
                        if (Q < m) {
                            const double2 c = csx.at(col + j0 + Q);
                            gx[Q] = pauli_x_inner<N, Q>(pr, pi, lr, li);
                            apply_rx<N, Q>(pr, pi, c.x, -c.y);
                            apply_rx<N, Q>(lr, li, c.x, -c.y);
                        }
                    });
                    if constexpr (C::LDSRED) sums.put_x(gx, col + j0, m);
                    else store_grad_x<N>(gx, lane, wave, B, E, grad_x, col + j0, m);
                }
            }
        }
    }
    if constexpr (C::LDSRED) { sums.flush_w(); sums.flush_x(); }
}

// ---------------------------------------------------------------------------------------
// wave-pair pipelined backward kernel (n <= 5, small batches)
// ---------------------------------------------------------------------------------------
// Facts it is built on (profiles/): one wave issues roughly one instruction per 4-5 cycles, a
// SIMD is saturated by a single wave, and at B = 1024 (Q5) the packed kernel only occupies half of the
// chip's SIMDs.  The reverse sweep is two dependent chains that share no arithmetic: psi is un-computed
// (it never needs lambda), lambda is pulled back and meets psi only in the inner products.  So a
// workgroup is TWO waves working on the same samples:
//   wave 0 ("psi wave"):    forward sweep, publishes psi_N, then walks psi backwards and publishes the
//                           state after every sub-layer's gates / every RX phase into a small LDS ring;
//   wave 1 ("lambda wave"): waits for psi_N, forms lambda_N = g H psi_N, walks lambda backwards; per
//                           sub-layer it reads ONE published psi (own amplitude + the n partners straight
//                           from LDS, no DPP), takes all 3n inner products against the not-yet-evolved
//                           lambda (gates of a sub-layer commute), then applies the n U-daggers to lambda.
// The psi wave runs ahead (it has ~1/3 of the work), the lambda wave is the critical path with ~70 % of the
// packed kernel's reverse-sweep instructions.  Hand-off = monotonic counters in LDS with workgroup-scope
// release/acquire; every spin is bounded (an overrun raises `abort` in LDS and both waves run out).
// Wave-to-wave hand-off through LDS: a producer stores its data, then a step counter the consumers poll (workgroup-scope
// acquire loads).  Between the two stores stands handoff_release(): `s_waitcnt lgkmcnt(0)` -- the data store has been
// EXECUTED by the LDS before the counter store is issued, which orders the two without relying on anything but the
// counter semantics of lgkmcnt (LDS operations of a wave return in order and decrement it on completion).  It waits for
// LDS / scalar-memory operations only: the layer-record prefetches in flight (vmcnt) are not drained, which a generic
// workgroup-scope release fence would do.  -DQHEA_RELAXED_HANDOFF builds the round-2 form instead (compiler-only fence,
// relying on the LDS pipe executing one wave's instructions in issue order; measured 1.1 % faster per step, not shipped).
__device__ __forceinline__ void handoff_release() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#ifndef QHEA_RELAXED_HANDOFF
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}
constexpr int kPairRing = 16;                // published psi snapshots in flight (16 KB)
static_assert((kPairRing & (kPairRing - 1)) == 0, "ring slots are indexed with step & (kPairRing - 1)");
#ifndef QHEA_SPIN_LIMIT
#define QHEA_SPIN_LIMIT (1 << 24)      // `make spin1` builds a test library with 1: every hand-off wait that is not already
#endif                                 // satisfied overruns, which is how the failure-reporting path is exercised on a healthy GPU
constexpr int kSpinLimit = QHEA_SPIN_LIMIT;

// Device status word in the header of the caller's workspace (hea_api.hip: WorkspaceHeader).  A pipelined backward
// kernel whose hand-off overran ORs kStatusHandoff into it before it exits; the reduce kernel of the same call then
// poisons every gradient and the loss scalars with NaN and skips the fused Adam update, and qhea_check_status()
// returns QHEA_EPIPELINE -- an overrun can no longer turn into silently wrong training.
constexpr int kStatusHandoff = 1;
__device__ __forceinline__ void report_abort(const int* abort_flag, int* status, int lane) {
    if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) && lane == 0 && status)
        atomicOr(status, kStatusHandoff);
}

struct PairSync {
    int produced, consumed, ready, abort;
};

// Wait until *counter >= target.  `seen` caches the last value read, so the common case (the other wave is
// already far enough) costs no LDS access at all.
__device__ __forceinline__ bool pair_wait_ge(int* counter, int target, int* abort_flag, int& seen) {
    if (seen >= target) return true;
#ifdef QHEA_PROFILE_WAITS   // measurement build (make profile_waits): time spent in hand-off waits, per wave, in ZSync::waited
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long* wt = reinterpret_cast<unsigned long long*>(abort_flag + 5) + 2 * (threadIdx.x >> 6);   // ZSync::waited
#endif
#pragma clang loop unroll(disable)              // (the compiler unrolls the spin 8x otherwise: 26 KB of reverse-walk code)
    for (int it = 0; it < kSpinLimit; ++it) {
        seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
#ifdef QHEA_PROFILE_WAITS
        if (seen >= target && (threadIdx.x & 63) == 0) { wt[0] += __builtin_amdgcn_s_memtime() - t0; wt[1] += 1 + ((unsigned long long)(it > 0) << 32); }
#endif
        if (seen >= target) return true;
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return false;
        __builtin_amdgcn_s_sleep(2);
    }
#ifdef QHEA_DEBUG_ABORT     // debugging build: say which wait overran (LDS address of the counter, target, last value seen)
    if ((threadIdx.x & 63) == 0)
        printf("overrun: block %d wave %d waited for counter@%u >= %d, saw %d\n", (int)blockIdx.x, (int)(threadIdx.x >> 6),
               (unsigned)(unsigned long long)(__attribute__((address_space(3))) int*)counter, target, seen);
#endif
    __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return false;
}

template <int N>
__global__ __launch_bounds__(128) void bwd_pair_kernel(Runs runs, long B, int E, int blk,
                                                       const double2* __restrict__ cs,
                                                       const char* __restrict__ gates, int gates_bytes,
                                                       double off, double co,
                                                       const double* __restrict__ diag, int pauli,
                                                       const double* __restrict__ g,
                                                       const double* __restrict__ state_in,
                                                       const double* __restrict__ y,
                                                       const double* __restrict__ bias,
                                                       double inv_bt,
                                                       double* __restrict__ out,
                                                       double* __restrict__ grad_x,
                                                       double* __restrict__ partial,
                                                       int* __restrict__ status) {
    using C = Cfg<N>;
    static_assert(C::R == 1 && C::LDSRED, "wave-pair kernel: all-lane layout, n <= 5");
    __shared__ double2 cs_lds[2 * kCsPerWave + 16];
    __shared__ double red_lds[kRedPerWave];
    __shared__ __attribute__((aligned(16))) char gate_ring[2 * kRingBytesPerWave];
    __shared__ double2 psi_ring[kPairRing][64];
    __shared__ double2 psi_final[64];
    __shared__ PairSync sync;

    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // 0: psi wave, 1: lambda wave (scalar: role branches stay scalar)
    const long wave = blockIdx.x;                       // one sample group per workgroup
    const long b_raw = wave * C::SPW + (lane >> C::LB);
    const bool valid = b_raw < B;
    const long b = valid ? b_raw : B - 1;
    const int klow = lane & (C::LANES - 1);
    const int ring_fwd = ring_source<N>(lane, false);
    const int ring_rev = ring_source<N>(lane, true);

    if (threadIdx.x == 0) { sync.produced = 0; sync.consumed = 0; sync.ready = 0; sync.abort = 0; }
    __syncthreads();

    CsStream<N> csx;
    csx.init(cs_lds + role * kCsPerWave, cs, b, E, lane, lane >> C::LB);
    GateStream<N> gs;
    gs.init(gates, gates_bytes, gate_ring + role * kRingBytesPerWave, lane);

    if (role == 0) {
        // ------------------------------------------------------------------ psi wave
        double pr[1], pi[1];
        if (state_in) {
            const double2 a = reinterpret_cast<const double2*>(state_in)[(b << N) + klow];
            pr[0] = a.x; pi[0] = a.y;
        } else {
            forward_sweep<N>(pr, pi, runs, csx, gs, lane, ring_fwd);
        }
        psi_final[lane] = make_double2(pr[0], pi[0]);
        __hip_atomic_store(&sync.ready, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);

        int col = E, step = 0, seen = 0;
        bool ok = true;
        auto publish = [&]() {
            if (step >= kPairRing) ok = ok && pair_wait_ge(&sync.consumed, step - kPairRing + 1, &sync.abort, seen);
            psi_ring[step & (kPairRing - 1)][lane] = make_double2(pr[0], pi[0]);
            ++step;
            __hip_atomic_store(&sync.produced, step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        gs.template prime<false>(blk - 1);
        for (int ri = runs.nruns - 1; ri >= 0 && ok; --ri) {
            const int ne = runs.enc[ri], nld = runs.ld[ri];
            const int nchunks = (ne + N - 1) / N;
            for (int rep = 0; rep < runs.count[ri] && ok; ++rep) {
                if (ne > 0) csx.template prefetch<false>(col - (ne - (nchunks - 1) * N));
                for (int l = nld - 1; l >= 0; --l) {
                    apply_ring<N, true>(pr, pi, lane, ring_rev);
                    publish();                                           // psi after this sub-layer's gates
                    gs.template begin<false>();
                    static_rfor<0, N>([&](auto q) {
                        constexpr int Q = decltype(q)::value;
                        const double4 u = gs.template cur<false, Q>();
                        apply_su2<N, Q>(pr, pi, u.x, -u.y, -u.z, -u.w);
                        gs.template done<false, Q>();
                    });
                    gs.template advance<false>();
                }
                col -= ne;
                if (ne > 0) {
                    publish();                                           // psi after this block's RX phase
                    for (int ch = nchunks - 1; ch >= 0; --ch) {
                        const int j0 = ch * N;
                        const int m = (ne - j0) < N ? (ne - j0) : N;
                        if (ch != nchunks - 1) csx.template prefetch<false>(col + j0);
                        rfor_gates_below<N>(m, [&](auto q) {
                            constexpr int Q = decltype(q)::value;
                            apply_rx<N, Q>(pr, pi, csx.nxt[Q].x, -csx.nxt[Q].y);
                        });
                    }
                }
            }
        }
    } else {
        // ------------------------------------------------------------------ lambda wave
        int seen_ready = 0, seen = 0;
        bool ok = pair_wait_ge(&sync.ready, 1, &sync.abort, seen_ready);
        double fr[1] = {psi_final[lane].x}, fi[1] = {psi_final[lane].y};
        basis_change<N, false>(fr, fi, pauli, lane);
        const double2 pN = make_double2(fr[0], fi[0]);
        const double h = ham_weight<N>(klow, off, co, diag);
        double gb;
        {
            double v[1] = {h * (pN.x * pN.x + pN.y * pN.y)};
            lane_reduce<1, C::LB>(v, lane);
            const double pred = v[0] + (bias ? bias[0] : 0.0);
            if (out && valid && klow == 0) out[b] = pred;
            gb = y ? 2.0 * (pred - y[b]) * inv_bt : g[b];
        }
        if (!valid) gb = 0.0;
        double lr[1] = {gb * h * pN.x}, li[1] = {gb * h * pN.y};
        basis_change<N, true>(lr, li, pauli, lane);

        GradSums<N> sums;
        sums.sub_w = -1; sums.col_x = -1; sums.m_x = 0; sums.red = red_lds; sums.lane = lane; sums.wave = wave;
        sums.B = B; sums.E = E; sums.part_w = partial + wave * (long)blk * C::KW; sums.grad_x = grad_x;

        int col = E, sub = blk, step = 0;
        gs.template prime<false>(blk - 1);
        for (int ri = runs.nruns - 1; ri >= 0 && ok; --ri) {
            const int ne = runs.enc[ri], nld = runs.ld[ri];
            const int nchunks = (ne + N - 1) / N;
            for (int rep = 0; rep < runs.count[ri] && ok; ++rep) {
                if (ne > 0) csx.template prefetch<false>(col - (ne - (nchunks - 1) * N));
                for (int l = nld - 1; l >= 0; --l) {
                    --sub;
                    apply_ring<N, true>(lr, li, lane, ring_rev);
                    ok = ok && pair_wait_ge(&sync.produced, step + 1, &sync.abort, seen);
                    const double2* slot = psi_ring[step & (kPairRing - 1)];
                    const double2 p = slot[lane];
                    double2 qv[N];
                    static_for<0, N>([&](auto q) { qv[decltype(q)::value] = slot[lane ^ (1 << decltype(q)::value)]; });
                    ++step;
                    __hip_atomic_store(&sync.consumed, step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    double acc3[C::KW];
#pragma unroll
                    for (int i = 0; i < C::KW; ++i) acc3[i] = 0.0;
                    static_for<0, N>([&](auto q) {                       // all against the not-yet-evolved lambda
                        constexpr int Q = decltype(q)::value;
                        const double s = ((lane >> Q) & 1) ? -1.0 : 1.0;
                        acc3[3 * Q] = lr[0] * qv[Q].y - li[0] * qv[Q].x;
                        acc3[3 * Q + 1] = -s * (lr[0] * qv[Q].x + li[0] * qv[Q].y);
                        acc3[3 * Q + 2] = s * (lr[0] * p.y - li[0] * p.x);
                    });
                    gs.template begin<false>();
                    static_rfor<0, N>([&](auto q) {
                        constexpr int Q = decltype(q)::value;
                        const double4 u = gs.template cur<false, Q>();
                        apply_su2<N, Q>(lr, li, u.x, -u.y, -u.z, -u.w);
                        gs.template done<false, Q>();
                    });
                    gs.template advance<false>();
                    sums.put_w(acc3, sub);
                }
                col -= ne;
                if (ne > 0) {
                    ok = ok && pair_wait_ge(&sync.produced, step + 1, &sync.abort, seen);
                    const double2* slot = psi_ring[step & (kPairRing - 1)];
                    double2 qv[N];
                    static_for<0, N>([&](auto q) { qv[decltype(q)::value] = slot[lane ^ (1 << decltype(q)::value)]; });
                    ++step;
                    __hip_atomic_store(&sync.consumed, step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    double gx[C::KX];
#pragma unroll
                    for (int i = 0; i < C::KX; ++i) gx[i] = 0.0;
                    for_gates_below<N>(ne, [&](auto q) {                 // every RX of the phase commutes with X_q
                        constexpr int Q = decltype(q)::value;
                        gx[Q] = lr[0] * qv[Q].y - li[0] * qv[Q].x;
                    });
                    for (int ch = nchunks - 1; ch >= 0; --ch) {
                        const int j0 = ch * N;
                        const int m = (ne - j0) < N ? (ne - j0) : N;
                        if (ch != nchunks - 1) csx.template prefetch<false>(col + j0);
                        rfor_gates_below<N>(m, [&](auto q) {
                            constexpr int Q = decltype(q)::value;
                            apply_rx<N, Q>(lr, li, csx.nxt[Q].x, -csx.nxt[Q].y);
                        });
                        sums.put_x(gx, col + j0, m);                     // wires repeat across chunks with equal gradients
                    }
                }
            }
        }
        sums.flush_w();
        sums.flush_x();
    }
    report_abort(&sync.abort, status, lane);
}

// ---------------------------------------------------------------------------------------
// psi / lambda / sigma pipelined backward kernel (n <= 5, small batches) -- the default for them
// ---------------------------------------------------------------------------------------
// Cycle counters in the wave-pair kernel showed that its lambda wave is the critical path and that a chain step
// (ring^-1, n adjoint gates) is LATENCY-bound: ~85 dependent instructions in ~1000-1400 cycles, the wave issues
// less than a quarter of the time.  What the lambda wave does besides its chain -- the inner products against psi
// and the cross-lane gradient sums -- neither belongs to that dependency chain nor needs to wait for it.  So here a
// workgroup is 2 + kSigmaWaves waves:
//   wave 0 (psi):    as in the pair kernel -- forward sweep, then walks psi backwards and publishes it after every
//                    ring^-1 / at every RX phase;
//   wave 1 (lambda): forms lambda_N, walks lambda backwards and publishes it at the same points -- nothing else;
//   waves 2.. (sigma): step t belongs to sigma wave t % kSigmaWaves: it reads psi (own amplitude + n partners) and
//                    lambda (own amplitude) of that step from the two LDS rings, takes the inner products and sums
//                    them over the wave with the REGISTER butterfly (no LDS: the LDS pipe is what the two chains'
//                    exchanges, gate reads and publishes wait on; LDS-staged sums here cost +5 us).
// With 2 sigma waves a 1024-sample batch puts two waves on every SIMD (one chain-type, one sigma-type on average);
// 1 sigma wave: 157 us, 2: 148 us, 3: 155 us, 4: 167 us per launch (pair kernel: 170 us).  A variant in which the two
// chain waves share the sigma work by step parity (no extra waves) ran at 225 us: sigma work inside a chain wave
// delays every later chain step.  Hand-off as in the pair kernel: monotonic LDS counters, cached reads, bounded spins
// (an overrun raises `abort`, after which every wait returns at once and the waves run out; keeping an `ok` flag through
// the loops cost ~10 scalar instructions per step.  The numbers of such a launch are wrong, so every wave that sees the
// flag at its end reports it in the workspace's status word: report_abort above).
constexpr int kSigmaWaves = 2;        // sigma waves per workgroup: step t belongs to sigma wave t % kSigmaWaves
struct TriSync {
    int psi_prod, lam_prod, ready, abort;
    int cursor[kSigmaWaves];          // every step below cursor[w] has been read by sigma wave w or is not its
};

template <int N>
__global__ __launch_bounds__(128 + 64 * kSigmaWaves) void bwd_tri_kernel(Runs runs, long B, int E, int blk,
                                                      const double2* __restrict__ cs,
                                                      const char* __restrict__ gates, int gates_bytes,
                                                      double off, double co,
                                                      const double* __restrict__ diag, int pauli,
                                                      const double* __restrict__ g,
                                                      const double* __restrict__ state_in,
                                                      const double* __restrict__ y,
                                                      const double* __restrict__ bias,
                                                      double inv_bt,
                                                      double* __restrict__ out,
                                                      double* __restrict__ grad_x,
                                                      double* __restrict__ partial,
                                                      int* __restrict__ status) {
    using C = Cfg<N>;
    static_assert(C::R == 1 && C::LDSRED, "three-wave kernel: all-lane layout, n <= 5");
    __shared__ double2 cs_lds[2 * kCsPerWave + 16];
    __shared__ __attribute__((aligned(16))) char gate_ring[2 * kRingBytesPerWave];
    __shared__ double2 psi_ring[kPairRing][64];
    __shared__ double2 lam_ring[kPairRing][64];
    __shared__ double2 psi_final[64];
    __shared__ TriSync sync;

    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // 0: psi, 1: lambda, 2..: sigma; made scalar so that
                                                        // every role-dependent loop and wait compiles to scalar control flow
    const long wave = blockIdx.x;                       // one sample group per workgroup
    const long b_raw = wave * C::SPW + (lane >> C::LB);
    const bool valid = b_raw < B;
    const long b = valid ? b_raw : B - 1;
    const int klow = lane & (C::LANES - 1);

    if (threadIdx.x == 0) {
        sync.psi_prod = 0; sync.lam_prod = 0; sync.ready = 0; sync.abort = 0;
        for (int w = 0; w < kSigmaWaves; ++w) sync.cursor[w] = w;
    }
    __syncthreads();

    if (role < 2) {
        // ------------------------------------------------------------------ psi / lambda chains
        __builtin_amdgcn_s_setprio(3);                  // the chains are the critical path, the sigma wave sharing the
                                                        // SIMD fills their gaps (measured: 148.0 vs 155.1 us per call)
        const int ring_fwd = ring_source<N>(lane, false);
        const int ring_rev = ring_source<N>(lane, true);
        CsStream<N> csx;
        csx.init(cs_lds + role * kCsPerWave, cs, b, E, lane, lane >> C::LB);
        GateStream<N> gs;
        gs.init(gates, gates_bytes, gate_ring + role * kRingBytesPerWave, lane);
        double sr[1], si[1];                            // this wave's state: psi or lambda
        int seen[kSigmaWaves];
#pragma unroll
        for (int w = 0; w < kSigmaWaves; ++w) seen[w] = 0;
        if (role == 0) {
            if (state_in) {
                const double2 a = reinterpret_cast<const double2*>(state_in)[(b << N) + klow];
                sr[0] = a.x; si[0] = a.y;
            } else {
                forward_sweep<N>(sr, si, runs, csx, gs, lane, ring_fwd);
            }
            psi_final[lane] = make_double2(sr[0], si[0]);
            __hip_atomic_store(&sync.ready, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            int seen_ready = 0;
            pair_wait_ge(&sync.ready, 1, &sync.abort, seen_ready);
            double fr[1] = {psi_final[lane].x}, fi[1] = {psi_final[lane].y};
            basis_change<N, false>(fr, fi, pauli, lane);
            const double h = ham_weight<N>(klow, off, co, diag);
            double gb;
            {
                double v[1] = {h * (fr[0] * fr[0] + fi[0] * fi[0])};
                lane_reduce<1, C::LB>(v, lane);
                const double pred = v[0] + (bias ? bias[0] : 0.0);
                if (out && valid && klow == 0) out[b] = pred;
                gb = y ? 2.0 * (pred - y[b]) * inv_bt : g[b];
            }
            if (!valid) gb = 0.0;
            sr[0] = gb * h * fr[0]; si[0] = gb * h * fi[0];
            basis_change<N, true>(sr, si, pauli, lane);
        }
        double2 (*ring)[64] = role == 0 ? psi_ring : lam_ring;
        int* prod = role == 0 ? &sync.psi_prod : &sync.lam_prod;
        int col = E, step = 0;
        auto publish = [&]() {
            if (step >= kPairRing) {
#pragma unroll
                for (int w = 0; w < kSigmaWaves; ++w)
                    pair_wait_ge(&sync.cursor[w], step - kPairRing + 1, &sync.abort, seen[w]);
            }
            ring[step & (kPairRing - 1)][lane] = make_double2(sr[0], si[0]);
            ++step;
            handoff_release();                               // the data store has completed before the counter store issues
            __hip_atomic_store(prod, step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        gs.template prime<false>(blk - 1);
        for (int ri = runs.nruns - 1; ri >= 0; --ri) {
            const int ne = runs.enc[ri], nld = runs.ld[ri];
            const int nchunks = (ne + N - 1) / N;
            for (int rep = 0; rep < runs.count[ri]; ++rep) {
                if (ne > 0) csx.template prefetch<false>(col - (ne - (nchunks - 1) * N));
                for (int l = nld - 1; l >= 0; --l) {
                    apply_ring<N, true>(sr, si, lane, ring_rev);
                    publish();                                           // state after this sub-layer's gates
                    gs.template begin<false>();
                    static_rfor<0, N>([&](auto q) {
                        constexpr int Q = decltype(q)::value;
                        const double4 u = gs.template cur<false, Q>();
                        apply_su2<N, Q>(sr, si, u.x, -u.y, -u.z, -u.w);
                        gs.template done<false, Q>();
                    });
                    gs.template advance<false>();
                }
                col -= ne;
                if (ne > 0) {
                    publish();                                           // state after this block's RX phase
                    for (int ch = nchunks - 1; ch >= 0; --ch) {
                        const int j0 = ch * N;
                        const int m = (ne - j0) < N ? (ne - j0) : N;
                        if (ch != nchunks - 1) csx.template prefetch<false>(col + j0);
                        rfor_gates_below<N>(m, [&](auto q) {
                            constexpr int Q = decltype(q)::value;
                            apply_rx<N, Q>(sr, si, csx.nxt[Q].x, -csx.nxt[Q].y);
                        });
                    }
                }
            }
        }
    } else {
        // ------------------------------------------------------------------ sigma wave: inner products + sums
        double* __restrict__ part_w = partial + wave * (long)blk * C::KW;
        const int me = role - 2;                        // this sigma wave takes the steps with step % kSigmaWaves == me
        int seen_p = 0, seen_l = 0;
        int col = E, sub = blk, step = 0;
        for (int ri = runs.nruns - 1; ri >= 0; --ri) {
            const int ne = runs.enc[ri], nld = runs.ld[ri];
            const int nchunks = (ne + N - 1) / N;
            for (int rep = 0; rep < runs.count[ri]; ++rep) {
                for (int l = nld - 1; l >= 0; --l) {
                    --sub;
                    if (step % kSigmaWaves != me) { ++step; continue; }
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
                        const double s = ((lane >> Q) & 1) ? -1.0 : 1.0;
                        acc3[3 * Q] = lm.x * qv[Q].y - lm.y * qv[Q].x;
                        acc3[3 * Q + 1] = -s * (lm.x * qv[Q].x + lm.y * qv[Q].y);
                        acc3[3 * Q + 2] = s * (lm.x * p.y - lm.y * p.x);
                    });
                    const int vi = butterfly_sum<C::KW>(acc3, lane);     // registers only: the LDS pipe stays with the chains
                    if (butterfly_owner<C::KW>(lane)) part_w[(long)sub * C::KW + vi] = acc3[0];
                }
                col -= ne;
                if (ne > 0 && step % kSigmaWaves != me) {
                    ++step;
                } else if (ne > 0) {
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
                    for_gates_below<N>(ne, [&](auto q) {                 // every RX of the phase commutes with X_q
                        constexpr int Q = decltype(q)::value;
                        gx[Q] = lm.x * qv[Q].y - lm.y * qv[Q].x;
                    });
                    for (int ch = nchunks - 1; ch >= 0; --ch) {
                        const int j0 = ch * N;
                        const int m = (ne - j0) < N ? (ne - j0) : N;
                        double gc[C::KX];                                // wires repeat across chunks with equal gradients
#pragma unroll
                        for (int i = 0; i < C::KX; ++i) gc[i] = gx[i];
                        store_grad_x<N>(gc, lane, wave, B, E, grad_x, col + j0, m);
                    }
                }
            }
        }
    }
    report_abort(&sync.abort, status, lane);
}

// ---------------------------------------------------------------------------------------
// launch entry points; one translation unit per qubit count (hea_inst.hip, -DQHEA_N=n)
// ---------------------------------------------------------------------------------------
struct FwdArgs {
    Runs runs; long B; int E; const double2* cs; const char* gates; int gates_bytes; double off, co;
    const double* diag; double* out; double* state_out; const double* bias; int pauli;
};
struct BwdArgs {
    Runs runs; long B; int E; int blk; const double2* cs; const char* gates; int gates_bytes; double off, co;
    const double* diag; const double* g; const double* state_in; const double* y; const double* bias; double inv_bt;
    double* out; double* grad_x; double* partial; int pauli;
    int tri;                  // n <= 5 small-batch backward: 0 = psi/lambda pair, 1 = psi / lambda / sigma waves
    int dense;                // more sample groups than SIMDs: n = 8, 9 use the 256-register build of bwd_kernel
    int* status;              // workspace status word (pipelined kernels report hand-off overruns there)
};

#ifdef QHEA_SUBSET      // development builds: -D'QHEA_SUBSET(X)=X(2) X(5)' links only those qubit counts
#define QHEA_FOR_EACH_N(X) QHEA_SUBSET(X)
#else
#define QHEA_FOR_EACH_N(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9)      /* n = 10..12: hea_lds.hip */
#endif
#define QHEA_DECLARE(NN)                                              \
    void launch_fwd_##NN(dim3 grid, hipStream_t st, const FwdArgs& a); \
    void launch_bwd_##NN(dim3 grid, hipStream_t st, const BwdArgs& a); \
    void launch_bwd_pair_##NN(dim3 grid, hipStream_t st, const BwdArgs& a);   /* n <= 5 only, else a stub */
QHEA_FOR_EACH_N(QHEA_DECLARE)
#undef QHEA_DECLARE

// workgroup-resident variant for the largest qubit counts (hea_lds.hip)
bool lds_supported(int n);
int launch_lds_fwd(int n, long B, hipStream_t st, const FwdArgs& a);
int launch_lds_bwd(int n, long B, hipStream_t st, const BwdArgs& a);

}  // namespace qhea
