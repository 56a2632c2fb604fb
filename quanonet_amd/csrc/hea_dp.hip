// hea_dp.hip -- data-parallel gradient exchange without a collective library call on the step's critical path.
//
// The data-parallel training step sums ONE flat fp64 buffer [gradients | sse | sum y^2] over the ranks
// (19 KB at Q5, 46 KB at Q12) and applies the same Adam update everywhere.  Through RCCL that is a collective launch
// (13-14 us on the device at world size 1 before any link latency, profiles/r02_allreduce_cost_world1_rccl.json) plus
// a separate Adam launch, on a step of ~100 us.  Here it is ONE one-workgroup kernel per rank:
//
//   publish   every rank writes its values into slot [parity][rank] of every OTHER rank's exchange buffer (peer buffers are
//             mapped through hipIpc): two self-validating 8-byte words per value, each carrying the exchange's sequence
//             number as a tag (hea_dp.hpp) -- no flag, nothing to drain;
//   collect   it polls the slots of its OWN buffer until both words of a value carry the tag (all ranks' words of a value
//             requested together; wall-clock bound), adds the `world` values IN RANK ORDER -- every rank adds the same
//             numbers in the same order, so the replicas stay bitwise identical and runs are reproducible -- writes
//             the sums and applies Adam to the parameters.
//
// Exchange buffers are fine-grained device memory the library allocates (hipExtMallocWithFlags: hipMalloc'ed memory is
// cached in the owner's L2, which remote writes do not pass through).  Two slot sets alternate with the parity of seq:
// a rank can be at most one exchange ahead of the slowest rank (its next collect needs everybody's next words, which a
// rank only stores after it has finished reading the previous slots), so seq + 1 never overwrites what a peer still reads.
// A wait that overruns its bound poisons the output with NaN, skips the Adam update and raises an error word that
// qhea_dp_status reports -- it never computes through a missing contribution -- and it raises the sticky poison word in
// EVERY rank's header, so that the late rank and every later exchange fail too (hea_dp.hpp).  The buffer layout and the
// device side of the exchange live in hea_dp.hpp, shared with the reduce kernel of the data-parallel training step
// (hea_api.hip: qhea_model_dp_train_steps), whose blocks exchange their own columns.
#include <cmath>
#include <cstring>
#include <limits>

#include "hea_adam.hpp"
#include "hea_dp.hpp"
#include "quanonet_hea.h"

namespace qhea {
namespace {

constexpr int kDpThreads = 1024;

struct DpArgs {
    DpX x;
    long n, n_adam;                                  // values exchanged; the first n_adam of them are parameters' gradients
    const double* local;
    double* out;
    AdamArgs adam;
};

__global__ __launch_bounds__(kDpThreads) void dp_exchange_kernel(DpArgs a) {
    const int tid = threadIdx.x;
    const DpX& x = a.x;
    __shared__ int failed;
    if (tid == 0) failed = 0;
    __syncthreads();
    // ---- publish: this rank's values into slot [parity][rank] of every other rank's buffer, one (value, peer) pair per thread
    const long pairs = a.n * x.world;
    for (long t = tid; t < pairs; t += kDpThreads) {
        const long i = t / x.world;
        const int p = (int)(t - i * x.world);
#ifndef QHEA_DP_LOOPBACK
        if (p != x.rank)
#endif
            dpx_store(x, p, i, a.local[i]);
    }
    __syncthreads();             // `out` may be `local` (an in-place sum): every value has been read before any is overwritten
    // ---- collect: a thread owns a value; the words of all ranks are requested together (one round trip when everybody has
    // published), stragglers are then waited for one by one; the sum runs in rank order
    const long long t0 = wall_clock64();
    const unsigned long long tag = dpx_tag(x);
    const int parity = (int)(x.seq & 1);
    char* own = x.bufs[x.rank];
    int fail = 0;
    for (long i = tid; i < a.n && !fail; i += kDpThreads) {
        unsigned long long wa[QHEA_DP_MAX_RANKS], wb[QHEA_DP_MAX_RANKS];
#pragma unroll
        for (int r = 0; r < QHEA_DP_MAX_RANKS; ++r)
            if (r < x.world && r != x.rank) {
                const unsigned long long* w = dp_words(own, parity, x.world, r, x.npad, i);
                wa[r] = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                wb[r] = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < QHEA_DP_MAX_RANKS; ++r)
            if (r < x.world) {
                double v;
                if (r == x.rank) v = a.local[i];
                else if ((wa[r] >> 32) == tag && (wb[r] >> 32) == tag)
                    v = __longlong_as_double((long long)((wa[r] & 0xffffffffull) | (wb[r] << 32)));
                else v = dpx_load(x, r, i, t0, fail);
                s += v;
            }
        a.out[i] = s;
    }
    // (a value that arrived in the first request was not checked against the poison word: one look before the agreement)
    if (!fail && __hip_atomic_load(&reinterpret_cast<const DpHeader*>(own)->poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) fail = 1;
    const bool ok = dpx_agree(x, fail, &failed);
    for (long i = tid; i < a.n; i += kDpThreads) {
        if (!ok) { a.out[i] = std::numeric_limits<double>::quiet_NaN(); continue; }
        if (a.adam.p && i < a.n_adam) adam_update(a.adam, i, a.out[i]);      // (a.out[i]: this thread's own store)
    }
}

}  // namespace
}  // namespace qhea

using namespace qhea;

extern "C" {

size_t qhea_dp_buffer_bytes(int64_t n_values, int world) {
    if (n_values < 1 || world < 1 || world > QHEA_DP_MAX_RANKS) return 0;
    return dp_bytes((long)n_values, world);
}

int qhea_dp_alloc(int64_t n_values, int world, void** buffer) {
    if (!buffer) return QHEA_EINVAL;
    *buffer = nullptr;
    const size_t bytes = qhea_dp_buffer_bytes(n_values, world);
    if (bytes == 0) return QHEA_EINVAL;
    void* p = nullptr;
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            return QHEA_ELAUNCH;
        }
    }
    if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipFree(p);
        return QHEA_ELAUNCH;
    }
    *buffer = p;
    return QHEA_OK;
}

int qhea_dp_free(void* buffer) {
    if (!buffer) return QHEA_OK;
    return hipFree(buffer) == hipSuccess ? QHEA_OK : QHEA_ELAUNCH;
}

int qhea_dp_export(void* buffer, void* handle64) {
    static_assert(sizeof(hipIpcMemHandle_t) == QHEA_DP_HANDLE_BYTES, "handle size");
    if (!buffer || !handle64) return QHEA_EINVAL;
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, buffer) != hipSuccess) { (void)hipGetLastError(); return QHEA_ELAUNCH; }
    std::memcpy(handle64, &h, sizeof h);
    return QHEA_OK;
}

int qhea_dp_import(const void* handle64, void** peer_buffer) {
    if (!handle64 || !peer_buffer) return QHEA_EINVAL;
    *peer_buffer = nullptr;
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle64, sizeof h);
    void* p = nullptr;
    if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return QHEA_ELAUNCH; }
    *peer_buffer = p;
    return QHEA_OK;
}

int qhea_dp_close(void* peer_buffer) {
    if (!peer_buffer) return QHEA_OK;
    return hipIpcCloseMemHandle(peer_buffer) == hipSuccess ? QHEA_OK : QHEA_ELAUNCH;
}

int qhea_dp_allreduce_adam(int rank, int world, void* const* buffers, int64_t n_values, int64_t seq,
                           const double* local, double* out, int64_t n_params, double* params, double* exp_avg,
                           double* exp_avg_sq, int64_t step, double lr, double beta1, double beta2, double eps,
                           double weight_decay, double timeout_ms, void* stream) {
    if (world < 1 || world > QHEA_DP_MAX_RANKS || rank < 0 || rank >= world || !buffers || n_values < 1 || seq < 1 ||
        !local || !out || n_params < 0 || n_params > n_values || !(timeout_ms > 0.0))
        return QHEA_EINVAL;
    DpArgs a{};
    for (int r = 0; r < world; ++r) {
        if (!buffers[r]) return QHEA_EINVAL;
        a.x.bufs[r] = static_cast<char*>(buffers[r]);
    }
    a.x.rank = rank; a.x.world = world; a.x.npad = dp_padded((long)n_values);
    a.x.seq = (unsigned long long)seq;
    a.x.timeout_ticks = (long long)(timeout_ms * 1e5);       // wall_clock64: 100 MHz
    a.n = (long)n_values; a.n_adam = (long)n_params; a.local = local; a.out = out;
    if (params) {
        if (!exp_avg || !exp_avg_sq || step < 1) return QHEA_EINVAL;
        const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
        a.adam = AdamArgs{params, exp_avg, exp_avg_sq, lr / bc1, 1.0 / sqrt(bc2), beta1, beta2, eps, weight_decay};
    }
    hipLaunchKernelGGL(dp_exchange_kernel, dim3(1), dim3(kDpThreads), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? QHEA_OK : QHEA_ELAUNCH;
}

int qhea_dp_status(void* buffer, void* stream) {
    if (!buffer) return QHEA_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned int err[2] = {0, 0};                            // error (cleared here), poison (sticky)
    unsigned int* dev = &static_cast<DpHeader*>(buffer)->error;
    if (hipMemcpyAsync(err, dev, sizeof err, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return QHEA_ELAUNCH;
    if (err[0] == 0 && err[1] == 0) return QHEA_OK;
    (void)hipMemsetAsync(dev, 0, sizeof(unsigned int), st);
    (void)hipStreamSynchronize(st);
    return QHEA_EEXCHANGE;
}

}  // extern "C"
