// hea_adam.hpp -- the Adam arithmetic shared by adam_kernel, the reduce kernel of qhea_model_train_step (hea_api.hip)
// and the data-parallel exchange kernel (hea_dp.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "hea_device.hpp"

namespace qhea {

// One Adam update (torch.optim.Adam arithmetic) of element i.  Shared by adam_kernel and by the reduce kernel of
// qhea_model_train_step, where the thread that finishes a gradient applies it at once.
struct AdamArgs {
    double* p; double* m; double* v;           // p == nullptr: no update (plain qhea_model_loss_grad)
    double lr_over_bc1, inv_sqrt_bc2, b1, b2, eps, wd;
};
// (pi, m0, v0 = the element's current parameter and moments: callers that know early which element they will update
// load them BEFORE their long reduction, so that the update does not add a dependent memory round trip at the end)
// returns the updated parameter
// Every fused multiply-add is written out and implicit contraction is off: the reduce kernel, adam_kernel and the exchange kernel
// inline this into different surroundings and must round identically (their results are compared bitwise).
__device__ __forceinline__ double adam_update_pre(const AdamArgs& a, long i, double gi, double pi, double m0, double v0) {
#pragma clang fp contract(off)
    if (a.wd != 0.0) gi = fma(a.wd, pi, gi);
    const double mi = fma(a.b1, m0, (1.0 - a.b1) * gi);           // torch: exp_avg.lerp_(grad, 1 - beta1)
    const double vi = fma(a.b2, v0, ((1.0 - a.b2) * gi) * gi);    //        exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    store_through(&a.m[i], mi); store_through(&a.v[i], vi);       // (write-through: the next launch reads them, hea_device.hpp)
    const double denom = fma(sqrt(vi), a.inv_sqrt_bc2, a.eps);
    const double pn = fma(-a.lr_over_bc1, mi / denom, pi);
    store_through(&a.p[i], pn);
    return pn;
}
__device__ __forceinline__ void adam_update(const AdamArgs& a, long i, double gi) {
    adam_update_pre(a, i, gi, a.p[i], a.m[i], a.v[i]);
}

}  // namespace qhea
