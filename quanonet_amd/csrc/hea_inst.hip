// hea_inst.hip -- instantiates the wave-resident forward/backward kernels for ONE qubit count
// (compile with -DQHEA_N=<n>; one object per n keeps the build parallel).
#include "hea_device.hpp"
#include "hea_zyz.hpp"

#ifndef QHEA_N
#error "compile with -DQHEA_N=<qubits>"
#endif
#define QHEA_CAT_(a, b) a##b
#define QHEA_CAT(a, b) QHEA_CAT_(a, b)

namespace qhea {

void QHEA_CAT(launch_fwd_, QHEA_N)(dim3 grid, hipStream_t st, const FwdArgs& a) {
    hipLaunchKernelGGL(fwd_kernel<QHEA_N>, grid, dim3(kWaves * 64), 0, st, a.runs, a.B, a.E, a.cs, a.gates, a.gates_bytes, a.off,
                       a.co, a.diag, a.pauli, a.out, a.state_out, a.bias);
}
void QHEA_CAT(launch_bwd_, QHEA_N)(dim3 grid, hipStream_t st, const BwdArgs& a) {
#if QHEA_N == 8 || QHEA_N == 9
    if (a.dense) {
        hipLaunchKernelGGL((bwd_kernel<QHEA_N, 2>), grid, dim3(kWaves * 64), 0, st, a.runs, a.B, a.E, a.blk, a.cs, a.gates,
                           a.gates_bytes, a.off, a.co, a.diag, a.pauli, a.g, a.state_in, a.y, a.bias, a.inv_bt, a.out,
                           a.grad_x, a.partial);
        return;
    }
#endif
    hipLaunchKernelGGL((bwd_kernel<QHEA_N, 1>), grid, dim3(kWaves * 64), 0, st, a.runs, a.B, a.E, a.blk, a.cs, a.gates, a.gates_bytes,
                       a.off, a.co, a.diag, a.pauli, a.g, a.state_in, a.y, a.bias, a.inv_bt, a.out, a.grad_x, a.partial);
}

void QHEA_CAT(launch_bwd_pair_, QHEA_N)(dim3 grid, hipStream_t st, const BwdArgs& a) {
#if QHEA_N <= 5
    if (a.tri == 1) {
        hipLaunchKernelGGL(bwd_tri_kernel<QHEA_N>, grid, dim3(128 + 64 * kSigmaWaves), 0, st, a.runs, a.B, a.E, a.blk, a.cs, a.gates,
                           a.gates_bytes, a.off, a.co, a.diag, a.pauli, a.g, a.state_in, a.y, a.bias, a.inv_bt, a.out,
                           a.grad_x, a.partial, a.status);
        return;
    }
    hipLaunchKernelGGL(bwd_pair_kernel<QHEA_N>, grid, dim3(128), 0, st, a.runs, a.B, a.E, a.blk, a.cs, a.gates,
                       a.gates_bytes, a.off, a.co, a.diag, a.pauli, a.g, a.state_in, a.y, a.bias, a.inv_bt, a.out, a.grad_x,
                       a.partial, a.status);
#else
    (void)grid; (void)st; (void)a;      // never selected for n > 5 (hea_api.hip: make_layout)
#endif
}

#if QHEA_N <= 5
void QHEA_CAT(launch_fwd_zyz_, QHEA_N)(dim3 grid, size_t dyn_lds, hipStream_t st, const ZFwdArgs& a) {
    hipLaunchKernelGGL(fwd_zyz_kernel<QHEA_N>, grid, dim3((kZFwdWaves + kFwdHelpers) * 64), dyn_lds, st, a);
}
void QHEA_CAT(launch_bwd_ztri_, QHEA_N)(dim3 grid, size_t dyn_lds, hipStream_t st, const ZBwdArgs& a) {
    if (a.pipes == 2) hipLaunchKernelGGL((bwd_ztri_kernel<QHEA_N, 2>), grid, dim3(2 * 64 * kZPipeWaves), dyn_lds, st, a);
    else hipLaunchKernelGGL((bwd_ztri_kernel<QHEA_N, 1>), grid, dim3(64 * kZPipeWaves), dyn_lds, st, a);
}
void QHEA_CAT(launch_fwd_zshared_, QHEA_N)(dim3 grid, size_t dyn_lds, hipStream_t st, const ZFwdArgs& a) {
    hipLaunchKernelGGL(fwd_zshared_kernel<QHEA_N>, grid, dim3(kZPWaves * 64), dyn_lds, st, a);
}
void QHEA_CAT(launch_bwd_zpacked_, QHEA_N)(dim3 grid, size_t dyn_lds, hipStream_t st, const ZBwdArgs& a) {
    hipLaunchKernelGGL(bwd_zpacked_kernel<QHEA_N>, grid, dim3(kZPWaves * 64), dyn_lds, st, a);
}
#if QHEA_N == 5
void launch_fwd_split_5(dim3 grid, size_t dyn_lds, hipStream_t st, const ZFwdArgs& a) {
    hipLaunchKernelGGL(fwd_split_kernel<5>, grid, dim3((kSplitWaves + kSplitHelpers) * 64), dyn_lds, st, a);
}
void launch_bwd_zquad_5(dim3 grid, size_t dyn_lds, hipStream_t st, const ZBwdArgs& a) {
    hipLaunchKernelGGL((bwd_zquad_kernel<kPairRing, kZSigma>), grid, dim3(64 * (4 + kZSigma)), dyn_lds, st, a);
}
#endif
#elif QHEA_N <= 5      // layout-experiment build: the ZYZ kernels need the all-lane layout and are never selected
void QHEA_CAT(launch_fwd_zyz_, QHEA_N)(dim3, size_t, hipStream_t, const ZFwdArgs&) {}
void QHEA_CAT(launch_bwd_ztri_, QHEA_N)(dim3, size_t, hipStream_t, const ZBwdArgs&) {}
void QHEA_CAT(launch_bwd_zpacked_, QHEA_N)(dim3, size_t, hipStream_t, const ZBwdArgs&) {}
void QHEA_CAT(launch_fwd_zshared_, QHEA_N)(dim3, size_t, hipStream_t, const ZFwdArgs&) {}
#if QHEA_N == 5
void launch_fwd_split_5(dim3, size_t, hipStream_t, const ZFwdArgs&) {}
void launch_bwd_zquad_5(dim3, size_t, hipStream_t, const ZBwdArgs&) {}
#endif
#endif

}  // namespace qhea
