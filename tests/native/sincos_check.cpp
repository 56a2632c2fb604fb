// Host-side accuracy check of quanonet_amd/csrc/hea_sincos.hpp against libm (built and run by tests/test_sincos.py with hipcc:
// the function is __host__ __device__, the arithmetic -- fp64 FMAs, no contraction left to the compiler -- is the same on
// both sides).  Prints: max |sin error|, max |cos error|, max |sin^2 + cos^2 - 1|, count.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../quanonet_amd/csrc/hea_sincos.hpp"

int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 4000000;
    const double span = argc > 2 ? atof(argv[2]) : 1000.0;
    std::mt19937_64 gen(12345);
    std::uniform_real_distribution<double> u(-span, span);
    double es = 0, ec = 0, en = 0;
    auto one = [&](double x) {
        double s, c;
        qhea::fast_sincos(x, &s, &c);
        const long double ls = sinl((long double)x), lc = cosl((long double)x);
        es = fmax(es, (double)fabsl(ls - s)); ec = fmax(ec, (double)fabsl(lc - c));
        en = fmax(en, fabs(s * s + c * c - 1.0));
    };
    for (long i = 0; i < n; ++i) one(u(gen));
    const double special[] = {0.0, -0.0, 1e-300, M_PI_4, -M_PI_4, M_PI_2, M_PI, 3 * M_PI_4, 2 * M_PI, 99999.9, -99999.9, 1e5, 3e7, -1e12};
    for (double x : special) one(x);
    double s, c;
    qhea::fast_sincos(NAN, &s, &c);
    const int nan_ok = std::isnan(s) && std::isnan(c);
    printf("%.3e %.3e %.3e %ld %d\n", es, ec, en, n, nan_ok);
    return 0;
}
