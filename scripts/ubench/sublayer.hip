// Cost model check for the n = 5 chain step: one wave alone on its SIMD walks `iters` sub-layers of
//   OLD: 5 x fused SU(2) (4 cross-lane moves + 8 fp64 multiply-adds, coefficients as 2 ds_read_b128 per gate) + ring gather
//   NEW: 1 per-lane diagonal (4 fp64 ops, 1 ds_read_b128) + 5 x RY (4 cross-lane moves + 4 fp64 ops, 1 ds_read_b128) + ring gather
// (the ZYZ form RY(c)RZ(b)RY(a) = RZ(alpha)RY(theta)RZ(beta) with all RZ of a layer merged into one diagonal).
// Prints s_memtime ticks (100 MHz) and wall ns per sub-layer for 1 wave per SIMD on 128 CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MASK> __device__ __forceinline__ int xi(int v) {
    if constexpr (MASK == 1) return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, false);
    else if constexpr (MASK == 2) return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, false);
    else if constexpr (MASK == 4) { int t = __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, false); return __builtin_amdgcn_mov_dpp(t, 0x1B, 0xF, 0xF, false); }
    else if constexpr (MASK == 8) return __builtin_amdgcn_mov_dpp(v, 0x128, 0xF, 0xF, false);
    else return __builtin_amdgcn_ds_swizzle(v, 0x401F);
}
template <int MASK> __device__ __forceinline__ double xd(double v) {
    return __hiloint2double(xi<MASK>(__double2hiint(v)), xi<MASK>(__double2loint(v)));
}
__device__ __forceinline__ double gather(double v, int a) {
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(a, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(a, __double2loint(v)));
}

#define NEWGATE(M, G) { const double4 u = row[(G) * 2 + ((lane >> (G)) & 1)]; const double qr = xd<M>(re), qi = xd<M>(im); \
    re = u.x * re - u.y * qr; im = u.x * im - u.y * qi; }

template <int MODE>
__global__ __launch_bounds__(128) void k(double* out, unsigned long long* cyc, int iters) {
    __shared__ double4 tab[2][2 * 64 * 6];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = lane; i < 2 * 64 * 6; i += 64) tab[w][i] = make_double4(0.8 + 1e-3 * i, 0.6 - 1e-3 * i, 0.36, 0.48);
    double re = 1.0 / 8 + lane * 1e-3, im = 0.01 * lane;
    const int ring = ((lane & 32) | ((lane * 5 + 3) & 31)) << 2;
    const double4* t = tab[w];
    double4 cf[6];
#pragma unroll
    for (int g = 0; g < 6; ++g) cf[g] = t[g * 2 + (lane & 1)];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long tl = t0, tg0 = 0, tg1 = 0;
    for (int it = 0; it < iters; ++it) {
        const double4* row = t + (it & 1) * 64 * 6;
        if constexpr (MODE == 0) {
#define OLDGATE(M, G) { const double4 u = row[(G) * 2 * 2 + ((lane >> (G)) & 1) * 2]; const double4 v = row[(G) * 2 * 2 + ((lane >> (G)) & 1) * 2 + 1]; \
            const double qr = xd<M>(re), qi = xd<M>(im); \
            const double nr = u.x * re - u.y * im + v.x * qr - v.y * qi; const double ni = u.x * im + u.y * re + v.x * qi + v.y * qr; re = nr; im = ni; }
            OLDGATE(1, 0) OLDGATE(2, 1) OLDGATE(4, 2) OLDGATE(8, 3) OLDGATE(16, 4)
        } else if constexpr (MODE == 2) {
            // the kernels' current form: wire 4 through v_permlane16_swap (re <-> im rows), no LDS latency on the chain
            { const double4 d = row[5 * 4 + (lane & 31)]; const double nr = d.x * re - d.y * im, ni = d.x * im + d.y * re; re = nr; im = ni; }
            NEWGATE(1, 0) NEWGATE(2, 1) NEWGATE(4, 2) NEWGATE(8, 3)
            { const double4 u = row[4 * 2];
              auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(re), (unsigned)__double2loint(im), false, false);
              auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(re), (unsigned)__double2hiint(im), false, false);
              const double a = __hiloint2double((int)hi[0], (int)lo[0]), b = __hiloint2double((int)hi[1], (int)lo[1]);
              const double o1 = u.x * a - u.y * b, o2 = u.y * a + u.x * b;
              auto l2 = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(o1), (unsigned)__double2loint(o2), false, false);
              auto h2 = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(o1), (unsigned)__double2hiint(o2), false, false);
              re = __hiloint2double((int)h2[0], (int)l2[0]); im = __hiloint2double((int)h2[1], (int)l2[1]); }
        } else if constexpr (MODE == 4 || MODE == 5 || MODE == 6 || MODE == 7) {
            // modes 2 / 3 with the coefficients read one sub-layer ahead, as the kernels do
            const double4* nrow = t + ((it + 1) & 1) * 64 * 6;
            double4 nx[6];
            if constexpr (MODE == 4 || MODE == 6) {
                nx[5] = nrow[5 * 4 + (lane & 31)];
#pragma unroll
                for (int g = 0; g < 4; ++g) nx[g] = nrow[g * 2 + ((lane >> g) & 1)];
                nx[4] = nrow[4 * 2];
                { const double4 d = cf[5]; const double nr = d.x * re - d.y * im, ni = d.x * im + d.y * re; re = nr; im = ni; }
#define PFGATE(M, G) { const double4 u = cf[G]; const double qr = xd<M>(re), qi = xd<M>(im); re = u.x * re - u.y * qr; im = u.x * im - u.y * qi; }
                PFGATE(1, 0) PFGATE(2, 1) PFGATE(4, 2) PFGATE(8, 3)
                { const double4 u = cf[4];
                  auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(re), (unsigned)__double2loint(im), false, false);
                  auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(re), (unsigned)__double2hiint(im), false, false);
                  const double a = __hiloint2double((int)hi[0], (int)lo[0]), b = __hiloint2double((int)hi[1], (int)lo[1]);
                  const double o1 = u.x * a - u.y * b, o2 = u.y * a + u.x * b;
                  auto l2 = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(o1), (unsigned)__double2loint(o2), false, false);
                  auto h2 = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(o1), (unsigned)__double2hiint(o2), false, false);
                  re = __hiloint2double((int)h2[0], (int)l2[0]); im = __hiloint2double((int)h2[1], (int)l2[1]); }
            } else {
                nx[5] = nrow[5 * 4 + lane];
#pragma unroll
                for (int g = 0; g < 4; ++g) nx[g] = nrow[g * 2 + ((lane >> g) & 1)];
                nx[4] = nrow[4 * 2 + ((lane >> 4) & 1)];
#define PFSWAP(BUILTIN, U) { const double4 u = U; \
                  auto lo = BUILTIN((unsigned)__double2loint(re), (unsigned)__double2loint(re), false, false); \
                  auto hi = BUILTIN((unsigned)__double2hiint(re), (unsigned)__double2hiint(re), false, false); \
                  const double a = __hiloint2double((int)hi[0], (int)lo[0]), b = __hiloint2double((int)hi[1], (int)lo[1]); \
                  re = fma(u.y, b, u.x * a); }
#define PFSPLIT(M, G) { const double4 u = cf[G]; const double own = u.x * re; const double q = xd<M>(re); re = fma(-u.y, q, own); }
                PFSWAP(__builtin_amdgcn_permlane32_swap, cf[5])
                PFSPLIT(1, 0) PFSPLIT(2, 1) PFSPLIT(4, 2) PFSPLIT(8, 3)
                PFSWAP(__builtin_amdgcn_permlane16_swap, cf[4])
            }
#pragma unroll
            for (int g = 0; g < 6; ++g) cf[g] = nx[g];
        } else if constexpr (MODE == 3) {
            // split layout: ONE sample per wave, lanes 0..31 hold the real parts, 32..63 the imaginary parts (`re` only)
#define SWAPFORM(BUILTIN, IDX) { const double4 u = row[IDX]; \
              auto lo = BUILTIN((unsigned)__double2loint(re), (unsigned)__double2loint(re), false, false); \
              auto hi = BUILTIN((unsigned)__double2hiint(re), (unsigned)__double2hiint(re), false, false); \
              const double a = __hiloint2double((int)hi[0], (int)lo[0]), b = __hiloint2double((int)hi[1], (int)lo[1]); \
              re = u.x * a + u.y * b; }
#define SPLITGATE(M, G) { const double4 u = row[(G) * 2 + ((lane >> (G)) & 1)]; const double q = xd<M>(re); re = u.x * re - u.y * q; }
            SWAPFORM(__builtin_amdgcn_permlane32_swap, 5 * 4 + lane)
            SPLITGATE(1, 0) SPLITGATE(2, 1) SPLITGATE(4, 2) SPLITGATE(8, 3)
            SWAPFORM(__builtin_amdgcn_permlane16_swap, 4 * 2 + ((lane >> 4) & 1))
        } else {
            { const double4 d = row[5 * 4 + (lane & 31)]; const double nr = d.x * re - d.y * im, ni = d.x * im + d.y * re; re = nr; im = ni; }
            NEWGATE(1, 0) NEWGATE(2, 1) NEWGATE(4, 2) NEWGATE(8, 3) NEWGATE(16, 4)
        }
        unsigned long long ta = 0;
        if constexpr (MODE >= 6) { ta = __builtin_amdgcn_s_memtime(); tg0 += ta - tl; }
        re = gather(re, ring);
        if constexpr (MODE != 3 && MODE != 5 && MODE != 7) im = gather(im, ring);
        if constexpr (MODE >= 6) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tl = __builtin_amdgcn_s_memtime(); tg1 += tl - ta; }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = re + im;
    if (lane == 0) cyc[blockIdx.x * 2 + w] = t1 - t0;
    if constexpr (MODE >= 6) { if (lane == 0 && blockIdx.x == 0 && w == 0) { cyc[2 * gridDim.x] = tg0; cyc[2 * gridDim.x + 1] = tg1; } }
}

template <int MODE> int run(const char* name) {
    double* out; unsigned long long* cyc;
    const int blocks = 256, iters = 4000;
    CHECK(hipMalloc(&out, sizeof(double) * blocks * 128));
    CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * (blocks * 2 + 2)));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(128), 0, 0, out, cyc, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(128), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    CHECK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 2 + 2);
    CHECK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * (blocks * 2 + 2), hipMemcpyDeviceToHost));
    double avg = 0; for (int i = 0; i < blocks * 2; ++i) avg += h[i]; avg /= blocks * 2;
    if (MODE >= 6) printf("    gates part %.1f ticks, gather part %.1f ticks per sub-layer (wave 0)\n", (double)h[blocks * 2] / iters, (double)h[blocks * 2 + 1] / iters);
    printf("%-44s %.2f memtime ticks (= %.0f ns) per sub-layer; wall %.1f ns per sub-layer\n", name, avg / iters, 10.0 * avg / iters, ms * 1e6 / iters);
    return 0;
}
int main() {
    run<0>("OLD  5 x fused SU(2) + ring");
    run<1>("NEW  diagonal + 5 x RY + ring");
    run<2>("NOW  the same, wire 4 by permlane16_swap");
    run<3>("SPLIT re/im in lanes: diagonal + 5 x RY + ring");
    run<4>("NOW, coefficients read one sub-layer ahead");
    run<5>("SPLIT, coefficients read one sub-layer ahead");
    run<6>("NOW (prefetched), timed segments");
    run<7>("SPLIT (prefetched), timed segments");
    return 0;
}
