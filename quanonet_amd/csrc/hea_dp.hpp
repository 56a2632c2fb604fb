// hea_dp.hpp -- layout of a data-parallel exchange buffer and the device side of the exchange, shared by the one-workgroup
// exchange kernel (hea_dp.hip) and by the reduce kernel of the data-parallel training step (hea_api.hip), whose blocks
// exchange their own columns.
//
//   [ DpHeader (256 B) | slots [2 parities][world][padded n] of 16 bytes ]
//
// A value travels as TWO self-validating 8-byte words, (low half | tag) and (high half | tag), tag = the exchange's sequence
// number: a rank publishes a value by storing the two words into slot [parity of seq][its rank] of every OTHER rank's buffer
// (peer buffers are mapped through hipIpc; 8-byte system-scope stores, each atomic by itself), and a collector polls the
// slots of its OWN buffer until both words of a slot carry the tag.  No flag, no drain of the stores before a flag, no second
// round trip for the data after the flag: an exchange is ONE store propagation (the "LL" idea of the collective libraries).
// The first version -- data stores, `s_waitcnt vmcnt(0)`, a flag per block, poll, then the data loads: three dependent trips
// through uncached memory -- cost 11 us inside the reduce kernel (profiles/r03_dp_loopback.json).
// The ranks' values are added in rank order -- every rank adds the same numbers in the same order: replicas stay bitwise
// identical, runs are reproducible.  Two slot sets alternate with the parity of seq; a rank is at most one exchange ahead of
// the slowest (its next collect needs everybody's next words, which a rank stores only after it has read the previous slots).
//
// Failure is fatal on EVERY rank: a wait that overruns its wall-clock bound sets the sticky `poison` word in
// every rank's header.  A collector that finds `poison` set -- while it waits or after its words have arrived -- fails too:
// NaN results, no Adam update, error bit (qhea_dp_status -> QHEA_EEXCHANGE).  The word is never cleared: after one timeout
// every later exchange on these buffers fails on every rank, so a late rank cannot complete the step the others gave up on
// and no replica trains on.
#pragma once
#include <hip/hip_runtime.h>

#include <limits>

#include "quanonet_hea.h"

namespace qhea {

constexpr size_t kDpHeaderBytes = 256;
constexpr int kDpMaxBlocks = 512;                    // reduce blocks that may exchange per launch (they must all be resident)
constexpr int kDpBlockValues = 64;                   // values one workgroup exchanges per call of dpx_exchange_block

struct DpHeader {
    unsigned long long reserved[QHEA_DP_MAX_RANKS];
    unsigned int error;                              // bit 0: a collect on this rank failed since the last qhea_dp_status
    unsigned int poison;                             // sticky: some rank's wait overran -- every later exchange fails
};
static_assert(sizeof(DpHeader) <= kDpHeaderBytes, "header");

__host__ __device__ inline long dp_padded(long n) { return (n + 1) & ~1L; }
__host__ __device__ inline size_t dp_slots_offset() { return kDpHeaderBytes; }
__host__ __device__ inline size_t dp_bytes(long n, int world) {
    return dp_slots_offset() + (size_t)2 * world * dp_padded(n) * 2 * sizeof(unsigned long long);
}
// the two tagged words of element idx in slot [parity][r]
__device__ __forceinline__ unsigned long long* dp_words(char* buf, int parity, int world, int r, long npad, long idx) {
    return reinterpret_cast<unsigned long long*>(buf + dp_slots_offset()) + (((long)parity * world + r) * npad + idx) * 2;
}

// one exchange, as seen by one launch
struct DpX {
    char* bufs[QHEA_DP_MAX_RANKS];                   // every rank's exchange buffer as mapped in THIS process
    int rank, world;                                 // world == 0: no exchange
    long npad;                                       // dp_padded(values per rank the buffers were allocated for)
    unsigned long long seq;
    long long timeout_ticks;                         // of wall_clock64() (100 MHz)
};
// never 0 (what a fresh buffer holds)
__device__ __forceinline__ unsigned long long dpx_tag(const DpX& x) { return x.seq % 0xffffffffull + 1ull; }

// this rank's value `v` of element `idx` into rank p's buffer
__device__ __forceinline__ void dpx_store(const DpX& x, int p, long idx, double v) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v), tag = dpx_tag(x) << 32;
#ifdef QHEA_DP_LOOPBACK     // measurement build (scripts/exp/dp_loopback.py): one process plays every rank
    for (int r = 0; r < x.world; ++r) {
        unsigned long long* w = dp_words(x.bufs[p], (int)(x.seq & 1), x.world, r, x.npad, idx);
        __hip_atomic_store(w, (bits & 0xffffffffull) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(w + 1, (bits >> 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#else
    unsigned long long* w = dp_words(x.bufs[p], (int)(x.seq & 1), x.world, x.rank, x.npad, idx);
    __hip_atomic_store(w, (bits & 0xffffffffull) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(w + 1, (bits >> 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
}

// rank r's value of element idx from the OWN buffer: polls until both words carry this exchange's tag.  fail: 0 arrived,
// 1 poison found, 2 the wait overran (t0 = wall_clock64() at the start of the caller's waiting)
__device__ __forceinline__ double dpx_load(const DpX& x, int r, long idx, long long t0, int& fail) {
    char* own = x.bufs[x.rank];
    const DpHeader* hdr = reinterpret_cast<const DpHeader*>(own);
    const unsigned long long* w = dp_words(own, (int)(x.seq & 1), x.world, r, x.npad, idx);
    const unsigned long long tag = dpx_tag(x);
    for (;;) {
        const unsigned long long a = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long b = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // (the poison word travels with the data words, not after them: a rank that arrives late -- its peers gave up a whole
        // timeout ago -- sees it in the same round trip that brings the words)
        if (__hip_atomic_load(&hdr->poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) { fail = 1; break; }
        if ((a >> 32) == tag && (b >> 32) == tag)
            return __longlong_as_double((long long)((a & 0xffffffffull) | (b << 32)));
        if (wall_clock64() - t0 > x.timeout_ticks) { fail = 2; break; }
        __builtin_amdgcn_s_sleep(2);
    }
    return std::numeric_limits<double>::quiet_NaN();
}

// End of a workgroup's exchange: the threads that polled pass what they saw (`fail` of dpx_load); every thread learns the
// workgroup's outcome.  A timeout raises the poison word in every rank's header.  One barrier; `sh_failed`: one int of LDS,
// zeroed by the caller before a barrier that precedes this call.
__device__ __forceinline__ bool dpx_agree(const DpX& x, int fail, int* sh_failed) {
    DpHeader* hdr = reinterpret_cast<DpHeader*>(x.bufs[x.rank]);
    if (fail) __hip_atomic_fetch_max(sh_failed, fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    const int failed = *sh_failed;
    if (failed) {
        const int tid = threadIdx.x;
        if (tid == 0) atomicOr(&hdr->error, 1u);
        if (failed == 2 && tid < x.world)                    // own wait overran: nobody may complete this or any later exchange
            __hip_atomic_store(&reinterpret_cast<DpHeader*>(x.bufs[tid])->poison, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return false;
    }
    return true;
}

// A workgroup exchanges `count` <= kDpBlockValues values, one thread per (value, rank) pair.  local[i] (LDS, visible to the
// workgroup: the caller has passed a barrier since it was written and since *sh_failed was zeroed) = this rank's value of flat
// element idx_of(i); on return xch[i * world + r] (LDS) = rank r's.  Returns false on failure (see the header).
template <class IdxOf>
__device__ __forceinline__ bool dpx_exchange_block(const DpX& x, int count, const double* local, IdxOf idx_of, double* xch,
                                                   int* sh_failed) {
    const int pairs = count * x.world, nthreads = blockDim.x;
    for (int t = threadIdx.x; t < pairs; t += nthreads) {
        const int i = t / x.world, p = t - i * x.world;
        const double v = local[i];
        if (p == x.rank) xch[t] = v;
#ifndef QHEA_DP_LOOPBACK
        else
#endif
            dpx_store(x, p, idx_of(i), v);
    }
    const long long t0 = wall_clock64();
    int fail = 0;
    for (int t = threadIdx.x; t < pairs && !fail; t += nthreads) {
        const int i = t / x.world, r = t - i * x.world;
        if (r != x.rank) xch[t] = dpx_load(x, r, idx_of(i), t0, fail);
    }
    return dpx_agree(x, fail, sh_failed);
}
// sum over the ranks, in rank order, of value i (after dpx_exchange_block returned true)
__device__ __forceinline__ double dpx_sum(const DpX& x, const double* xch, int i) {
    double s = 0.0;
    for (int r = 0; r < x.world; ++r) s += xch[i * x.world + r];
    return s;
}

}  // namespace qhea
