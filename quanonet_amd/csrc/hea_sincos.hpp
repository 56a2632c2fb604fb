// hea_sincos.hpp -- sin and cos of one fp64 angle in ~45 instructions.
//
// The device library's sincos(double) costs a wave ~2 k clocks (argument reduction with a large-argument path behind a branch,
// two polynomial kernels with their tails); it sits on the critical path twice per training step: every circuit kernel
// starts by filling its (cos, sin) table of the encoding angles, and the reduce kernel that writes the next step's layer
// records needs the half angles of the angles it has just updated.  Angles here are a few pi at most (rotation angles, and
// frequency-layer outputs in*w + b), so:
//
//   k = rint(x * 2/pi);  r = x - k * pi/2 in three fused steps against a three-term split of pi/2 (Cody-Waite; each term a full
//   double, the products exact inside the FMAs: |r| <= pi/4 to within an ulp for |k| < 2^20);
//   sin r = r + r^3 (S1 + r^2 (S2 + ... S6)),  cos r = 1 - r^2/2 + r^4 (C1 + r^2 (C2 + ... C6))   -- the fdlibm kernels'
//   minimax coefficients (|error| < 1 ulp on [-pi/4, pi/4]);  the quadrant k & 3 swaps / negates.
//
// |x| >= 1e5 (or NaN / inf) takes the library's sincos.  Accuracy against libm on the host: tests/test_sincos.py
// (max |error| 1.2e-16 absolute over 4 M angles in [-1e3, 1e3], the same order as libm's own rounding).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

namespace qhea {

__host__ __device__ __forceinline__ void fast_sincos(double x, double* s, double* c) {
#pragma clang fp contract(off)
    if (!(fabs(x) < 1.0e5)) {
        sincos(x, s, c);
        return;
    }
    const double kd = rint(x * 6.36619772367581382433e-01);            // 2/pi
    double r = fma(-kd, 1.57079632679489655800e+00, x);
    r = fma(-kd, 6.12323399573676603587e-17, r);
    r = fma(kd, 1.49738490485916983e-33, r);                           // (the third term of pi/2 is negative)
    const double z = r * r;
    const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                                   2.75573137070700676789e-06), -1.98412698298579493134e-04),
                                 8.33333333332248946124e-03), -1.66666666666666324348e-01);
    const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                                   -2.75573143513906633035e-07), 2.48015872894767294178e-05),
                                 -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double sr = fma(z * r, ps, r);
    const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int q = (int)kd & 3;
    const double s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
    *s = (q & 2) ? -s0 : s0;
    *c = ((q + 1) & 2) ? -c0 : c0;
}

}  // namespace qhea
