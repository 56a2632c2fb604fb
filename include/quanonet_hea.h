/*
 * quanonet_hea.h -- C ABI of the MI355X-native batched HEA circuit simulator.
 *
 * This is the drop-in boundary for the reference's quantum-layer plug-in
 * (Wang-Ruocheng/QuanONet).  The reference has no C ABI of its own: its boundary
 * is the Python factory `_build_quantum_layer(...) -> nn.Module`
 * (core/models_pt.py:71-100) whose returned module computes
 *     forward(x[B,E]) -> out[B,1]          (core/quantum_circuits_tq.py:65-127)
 * and is differentiated by torch autograd (TorchQuantum) or by MindQuantum's
 * adjoint op `get_expectation_with_grad` (core/quantum_circuits_ms.py:229-233).
 * Each entry point below names the reference interface it replaces.
 *
 * All pointers marked DEVICE are HBM addresses valid on the current HIP device;
 * pointers marked HOST are ordinary host memory read before the call returns.
 * All device work is enqueued on `stream` (a hipStream_t passed as void*; NULL =
 * the null stream).  No entry point allocates, frees or synchronises: scratch is
 * the caller-owned `workspace` (size from qhea_workspace_bytes), so every call
 * is hipGraph-capturable.  Nothing is retained past return.
 *
 * Layouts (row-major, fp64):
 *   x      [B, E]        encoding angles, column e = block*n + wire, trunk blocks first
 *   w      [blk, 3, n]   ansatz angles: sub-layer, gate (RY,RZ,RY), wire
 *   out    [B]           <psi|H|psi>   (no model bias)
 *   state  [B, 2^n, 2]   final statevector (re,im), basis index little-endian in the wire number
 *   g      [B]           upstream dL/d out_b
 *   grad_x [B, E]        g_b * d out_b / d x[b,e]
 *   grad_w [blk, 3, n]   sum_b g_b * d out_b / d w          (fully reduced, deterministic)
 *
 * Circuit (core/quantum_circuits_tq.py:79-104): start |0..0>; for each block b:
 *   RX(x[:,col]) on wire j%n for j < enc_per_block[b]; then ld_per_block[b] times
 *   { per wire i: RY(w[s,0,i]) RZ(w[s,1,i]) RY(w[s,2,i]);  for i=0..n-1: CNOT(control=(i+1)%n, target=i) }.
 * Read-out (core/quantum_circuits_tq.py:106-127): H = ham_offset + ham_coeff * sum_i Z_i, or,
 *   when ham_diag != NULL, H = diag(ham_diag[k]) with bit i of k = wire i.
 *
 * Return value: 0 on success, negative QHEA_E* otherwise (qhea_strerror gives text).
 */
#ifndef QUANONET_HEA_H
#define QUANONET_HEA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QHEA_OK            0
#define QHEA_EINVAL       -1   /* bad argument (null pointer, n out of range, ...)   */
#define QHEA_EUNSUPPORTED -2   /* circuit shape not supported by this build           */
#define QHEA_EWORKSPACE   -3   /* workspace too small / missing                       */
#define QHEA_ELAUNCH      -4   /* HIP launch or runtime failure                       */
#define QHEA_ENODEVICE    -5   /* no usable HIP device                                */

#define QHEA_MIN_QUBITS 2      /* n=1 has no entangler in MindQuantum and an undefined one in TQ */
#define QHEA_MAX_QUBITS 12

/* Library version (major*10000 + minor*100 + patch). */
int qhea_version(void);

/* Text for a QHEA_* code. */
const char* qhea_strerror(int code);

/* Number of usable HIP devices (0 when none); never initialises a context on failure. */
int qhea_device_count(void);

/*
 * Bytes of DEVICE scratch the calls below need for this circuit shape and batch.
 * Replaces: nothing in the reference (TorchQuantum allocates per-gate temporaries and
 * autograd saves every intermediate state; core/quantum_circuits_tq.py:74).
 * enc_per_block / ld_per_block: HOST arrays of length n_blocks.
 */
size_t qhea_workspace_bytes(int n_qubits, int n_blocks,
                            const int32_t* enc_per_block, const int32_t* ld_per_block,
                            int64_t batch);

/*
 * Forward: out[b] = <psi_b|H|psi_b>.
 * Replaces `_TQHEACircuit.forward` + `_measure` (core/quantum_circuits_tq.py:65-127)
 * and MindQuantum's forward half of `get_expectation_with_grad`
 * (core/quantum_circuits_ms.py:229-233).
 * state_out may be NULL; when given it receives the final statevectors, which
 * qhea_backward can consume to skip its own forward sweep.
 */
int qhea_forward(int n_qubits, int n_blocks,
                 const int32_t* enc_per_block /*HOST*/, const int32_t* ld_per_block /*HOST*/,
                 int64_t batch,
                 const double* x /*DEVICE [B,E]*/, const double* w /*DEVICE [blk,3,n]*/,
                 double ham_offset, double ham_coeff, const double* ham_diag /*DEVICE [2^n] or NULL*/,
                 double* out /*DEVICE [B]*/, double* state_out /*DEVICE [B,2^n,2] or NULL*/,
                 void* workspace /*DEVICE*/, size_t workspace_bytes, void* stream);

/*
 * Adjoint backward: grad_x, grad_w for upstream g.
 * Replaces torch autograd through the TorchQuantum gate ops
 * (solvers/solver_pt.py:235 `loss.backward()`) and MindQuantum's adjoint gradient
 * (core/quantum_circuits_ms.py:229-233).  O(1) state memory: psi and lambda only.
 * state_in: final statevectors from qhea_forward(state_out) or NULL (recomputed).
 * out may be NULL; when given it receives the forward values as well.
 */
int qhea_backward(int n_qubits, int n_blocks,
                  const int32_t* enc_per_block /*HOST*/, const int32_t* ld_per_block /*HOST*/,
                  int64_t batch,
                  const double* x /*DEVICE [B,E]*/, const double* w /*DEVICE [blk,3,n]*/,
                  double ham_offset, double ham_coeff, const double* ham_diag /*DEVICE or NULL*/,
                  const double* g /*DEVICE [B]*/, const double* state_in /*DEVICE or NULL*/,
                  double* out /*DEVICE [B] or NULL*/,
                  double* grad_x /*DEVICE [B,E]*/, double* grad_w /*DEVICE [blk,3,n]*/,
                  void* workspace /*DEVICE*/, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QUANONET_HEA_H */
