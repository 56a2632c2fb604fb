// Micro-benchmarks calibrating the gate-kernel cost model on gfx950: cycles per wave-instruction
// (s_memtime) for fp64 FMA chains, DPP moves, ds_swizzle, ds_bpermute, at 1 and 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2000;

__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memtime(); }

template <int MODE>
__global__ void k(double* out, unsigned long long* cyc, double a, double b) {
    const int lane = threadIdx.x & 63;
    double x0 = lane * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    int i0 = lane, i1 = lane + 1, i2 = lane + 2, i3 = lane + 3;
    const int addr = (lane ^ 5) << 2;
    unsigned long long t0 = now();
    for (int it = 0; it < ITERS; ++it) {
        if constexpr (MODE == 0) {          // 8 independent fp64 fma chains: issue-bound
            x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
            x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b);
        } else if constexpr (MODE == 1) {   // 1 dependent fp64 fma chain: latency (8 per iter)
            x0 = fma(x0, a, b); x0 = fma(x0, a, b); x0 = fma(x0, a, b); x0 = fma(x0, a, b);
            x0 = fma(x0, a, b); x0 = fma(x0, a, b); x0 = fma(x0, a, b); x0 = fma(x0, a, b);
        } else if constexpr (MODE == 2) {   // 8 independent DPP moves per iter (4 chains x 2)
            i0 = __builtin_amdgcn_mov_dpp(i0, 0xB1, 0xF, 0xF, false); i1 = __builtin_amdgcn_mov_dpp(i1, 0x4E, 0xF, 0xF, false);
            i2 = __builtin_amdgcn_mov_dpp(i2, 0x128, 0xF, 0xF, false); i3 = __builtin_amdgcn_mov_dpp(i3, 0x141, 0xF, 0xF, false);
            i0 = __builtin_amdgcn_mov_dpp(i0, 0x4E, 0xF, 0xF, false); i1 = __builtin_amdgcn_mov_dpp(i1, 0xB1, 0xF, 0xF, false);
            i2 = __builtin_amdgcn_mov_dpp(i2, 0x141, 0xF, 0xF, false); i3 = __builtin_amdgcn_mov_dpp(i3, 0x128, 0xF, 0xF, false);
        } else if constexpr (MODE == 3) {   // dependent DPP chain (8 per iter)
#pragma unroll
            for (int j = 0; j < 4; ++j) { i0 = __builtin_amdgcn_mov_dpp(i0, 0xB1, 0xF, 0xF, false); i0 = __builtin_amdgcn_mov_dpp(i0, 0x128, 0xF, 0xF, false); }
        } else if constexpr (MODE == 4) {   // dependent ds_swizzle chain (8 per iter)
#pragma unroll
            for (int j = 0; j < 8; ++j) i0 = __builtin_amdgcn_ds_swizzle(i0, 0x401F);
        } else if constexpr (MODE == 5) {   // dependent ds_bpermute chain (8 per iter)
#pragma unroll
            for (int j = 0; j < 8; ++j) i0 = __builtin_amdgcn_ds_bpermute(addr, i0);
        } else if constexpr (MODE == 6) {   // 4 independent ds_bpermute then use (like the ring): 2 groups per iter
            i0 = __builtin_amdgcn_ds_bpermute(addr, i0); i1 = __builtin_amdgcn_ds_bpermute(addr, i1);
            i2 = __builtin_amdgcn_ds_bpermute(addr, i2); i3 = __builtin_amdgcn_ds_bpermute(addr, i3);
            i0 += i3; i1 += i2;
            i0 = __builtin_amdgcn_ds_bpermute(addr, i0); i1 = __builtin_amdgcn_ds_bpermute(addr, i1);
            i2 = __builtin_amdgcn_ds_bpermute(addr, i2); i3 = __builtin_amdgcn_ds_bpermute(addr, i3);
            i0 += i3; i1 += i2;
        } else if constexpr (MODE == 7) {   // gate-like: dpp x4 -> 8 fma (2 chains of 4), dependent across gates; 2 gates per iter
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                double qr = __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(x0), 0xB1, 0xF, 0xF, false), __builtin_amdgcn_mov_dpp(__double2loint(x0), 0xB1, 0xF, 0xF, false));
                double qi = __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(x1), 0xB1, 0xF, 0xF, false), __builtin_amdgcn_mov_dpp(__double2loint(x1), 0xB1, 0xF, 0xF, false));
                double nr = a * x0 - b * x1 + b * qr - a * qi;
                double ni = a * x1 + b * x0 + b * qi + a * qr;
                x0 = nr; x1 = ni;
            }
        } else if constexpr (MODE == 8) {   // 2 independent fp64 fma chains (like re/im): 8 per iter
            x0 = fma(x0, a, b); x1 = fma(x1, a, b); x0 = fma(x0, a, b); x1 = fma(x1, a, b);
            x0 = fma(x0, a, b); x1 = fma(x1, a, b); x0 = fma(x0, a, b); x1 = fma(x1, a, b);
        } else if constexpr (MODE == 9) {   // 4 independent fp64 fma chains: 8 per iter
            x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
            x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
        }
    }
    unsigned long long t1 = now();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + i0 + i1 + i2 + i3;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
int run(const char* name, int per_iter) {
    double* out; unsigned long long* cyc;
    CHECK(hipMalloc(&out, sizeof(double) * 256 * 8 * 1024));
    CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * 4 * 8 * 1024));
    for (int wps : {-16, -4, -2, 1, 2, 4, 8}) {   // >0: waves per SIMD on all 256 CUs; <0: only 256/|wps| blocks (1 wave per SIMD on a fraction of the CUs)
        const int blocks = wps > 0 ? 256 * wps : 256 / (-wps);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0000001, 1e-9);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0000001, 1e-9);
        hipEventRecord(e1);
        CHECK(hipDeviceSynchronize());
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 4);
        CHECK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost));
        double avg = 0; for (auto v : h) avg += v; avg /= h.size();
        // s_memtime ticks at 100 MHz constant clock; convert using wall: report both
        printf("%-28s waves/SIMD=%d: %.1f memtime-ticks/instr/wave, wall %.3f ms -> %.2f ns per instr per wave, x%d waves => %.2f ns issue interval\n",
               name, wps, avg / (double)(ITERS * per_iter), ms, ms * 1e6 / (ITERS * per_iter), wps, ms * 1e6 / (ITERS * per_iter) / wps);
    }
    hipFree(out); hipFree(cyc);
    return 0;
}

int main() {
    run<0>("fma64 x8 independent", 8);
    run<9>("fma64 x4 chains", 8);
    run<8>("fma64 x2 chains", 8);
    run<1>("fma64 dependent", 8);
    run<2>("dpp x4 chains", 8);
    run<3>("dpp dependent", 8);
    run<4>("ds_swizzle dependent", 8);
    run<5>("ds_bpermute dependent", 8);
    run<6>("ds_bpermute 4-wide groups", 8);
    run<7>("gate-like (4dpp+8fma) dep", 2);
    return 0;
}
