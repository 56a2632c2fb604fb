// hea_dp.hpp -- layout of a data-parallel exchange buffer and the device side of the exchange, shared by the one-workgroup
// exchange kernel (hea_dp.hip) and by the reduce kernel of the data-parallel training step (hea_api.hip), whose blocks
// exchange their own columns.
//
//   [ DpHeader (256 B) | block flags [QHEA_DP_MAX_RANKS][kDpMaxBlocks] u64 | slots [2 parities][world][padded n] f64 ]
//
// A rank publishes a value by storing it into slot [parity of seq][its rank] of EVERY rank's buffer (peer buffers are mapped
// through hipIpc; write-through stores at system scope) and, once all its stores have left the device, a flag = seq in every
// rank's buffer: flag[rank] of the header for the one-workgroup kernel, block flag [rank][block] where every reduce block
// exchanges its own columns.  A collector polls the flags in its OWN buffer and adds the `world` slots in rank order --
// every rank adds the same numbers in the same order: replicas stay bitwise identical, runs are reproducible.
//
// Failure is fatal on EVERY rank: a wait that overruns its wall-clock bound sets the sticky `poison` word in
// every rank's header.  A collector that finds `poison` set -- while it waits or after its flags have arrived -- fails too:
// NaN results, no Adam update, error bit (qhea_dp_status -> QHEA_EEXCHANGE).  The word is never cleared: after one timeout
// every later exchange on these buffers fails on every rank, so a late rank cannot complete the step the others gave up on
// and no replica trains on.
#pragma once
#include <hip/hip_runtime.h>

#include <limits>

#include "quanonet_hea.h"

namespace qhea {

constexpr size_t kDpHeaderBytes = 256;
constexpr int kDpMaxBlocks = 512;                    // reduce blocks that may exchange per launch

struct DpHeader {
    unsigned long long flag[QHEA_DP_MAX_RANKS];      // flag[r] = seq of the last one-workgroup exchange rank r has published here
    unsigned int error;                              // bit 0: a collect on this rank failed since the last qhea_dp_status
    unsigned int poison;                             // sticky: some rank's wait overran -- every later exchange fails
};
static_assert(sizeof(DpHeader) <= kDpHeaderBytes, "header");

__host__ __device__ inline long dp_padded(long n) { return (n + 1) & ~1L; }
__host__ __device__ inline size_t dp_slots_offset() {
    return kDpHeaderBytes + (size_t)QHEA_DP_MAX_RANKS * kDpMaxBlocks * sizeof(unsigned long long);
}
__host__ __device__ inline size_t dp_bytes(long n, int world) {
    return dp_slots_offset() + (size_t)2 * world * dp_padded(n) * sizeof(double);
}
__device__ __forceinline__ double* dp_slot(char* buf, int parity, int world, int r, long npad) {
    return reinterpret_cast<double*>(buf + dp_slots_offset()) + ((long)parity * world + r) * npad;
}
__device__ __forceinline__ unsigned long long* dp_bflag(char* buf, int r, int block) {
    return reinterpret_cast<unsigned long long*>(buf + kDpHeaderBytes) + (long)r * kDpMaxBlocks + block;
}

// one exchange, as seen by one launch
struct DpX {
    char* bufs[QHEA_DP_MAX_RANKS];                   // every rank's exchange buffer as mapped in THIS process
    int rank, world;                                 // world == 0: no exchange
    long npad;                                       // dp_padded(values per rank the buffers were allocated for)
    unsigned long long seq;
    long long timeout_ticks;                         // of wall_clock64() (100 MHz)
};

// this rank's value `v` of element `idx` into every rank's slot (relaxed; dpx_flags_and_wait releases them)
__device__ __forceinline__ void dpx_publish(const DpX& x, long idx, double v) {
    const int parity = (int)(x.seq & 1);
    for (int p = 0; p < x.world; ++p)
        __hip_atomic_store(dp_slot(x.bufs[p], parity, x.world, x.rank, x.npad) + idx, v, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
}

// Called by EVERY thread of the workgroup after its dpx_publish calls: drains the stores, raises flag `flag_of(buffer)` = seq
// in every rank's buffer, waits for the `world` flags in the own buffer.  Returns true when every contribution has
// arrived; false after a timeout (poison raised everywhere) or when poison was found.  `sh_failed`: one int of LDS.
template <class FlagOf>
__device__ __forceinline__ bool dpx_flags_and_wait(const DpX& x, FlagOf flag_of, int* sh_failed) {
    const int tid = threadIdx.x;
    if (tid == 0) *sh_failed = 0;
    // Hand-off without cache maintenance.  Every published value is a system-scope atomic store (`global_store ... sc0 sc1`:
    // written through to the memory it lives in, never left in an L2) and every read of a slot is a system-scope atomic load
    // (dpx_collect), so neither a write-back nor an invalidate is needed -- MI355X_MICROARCH.md, inter-workgroup visibility:
    // "{sc0 sc1 stores and loads both sides}" with (a) every storing wave waiting for its stores (`s_waitcnt vmcnt(0)`: the
    // stores have been acknowledged by the memory they target, the peer's included) and (b) the flag raised only after the
    // wait of EVERY wave it speaks for -- the workgroup barrier below.  A system-scope release fence here (`buffer_wbl2 sc0
    // sc1`, every thread of every reduce block, with the kernel's own plain stores dirty in L2) and the acquire after the
    // wait (`buffer_inv`) cost 30-37 us per step inside the reduce kernel (profiles/r03_dp_loopback.json).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < x.world)
        __hip_atomic_store(flag_of(x.bufs[tid], x.rank), x.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    char* own = x.bufs[x.rank];
    DpHeader* hdr = reinterpret_cast<DpHeader*>(own);
    if (tid < x.world) {
        const unsigned long long* f = flag_of(own, tid);
        const long long t0 = wall_clock64();
        int fail = 0;
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < x.seq) {
            if (__hip_atomic_load(&hdr->poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) { fail = 2; break; }
            if (wall_clock64() - t0 > x.timeout_ticks) { fail = 1; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        if (!fail && __hip_atomic_load(&hdr->poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) fail = 2;
        if (fail) __hip_atomic_fetch_max(sh_failed, fail == 1 ? 2 : 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    const int failed = *sh_failed;
    if (failed) {
        if (tid == 0) atomicOr(&hdr->error, 1u);
        if (failed == 2 && tid < x.world)                    // own wait overran: nobody may complete this or any later exchange
            __hip_atomic_store(&reinterpret_cast<DpHeader*>(x.bufs[tid])->poison, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return false;
    }
    return true;                                             // (the slots are read with system-scope loads: nothing to invalidate)
}

// sum over the ranks, in rank order, of element idx (after dpx_flags_and_wait returned true)
__device__ __forceinline__ double dpx_collect(const DpX& x, long idx) {
    char* own = x.bufs[x.rank];
    const int parity = (int)(x.seq & 1);
    double s = 0.0;
    for (int r = 0; r < x.world; ++r)
        s += __hip_atomic_load(dp_slot(own, parity, x.world, r, x.npad) + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return s;
}

}  // namespace qhea
