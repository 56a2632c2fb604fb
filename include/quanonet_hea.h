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
 * the null stream).  No entry point allocates, frees or synchronises (except
 * qhea_check_status, whose purpose is to): scratch is the caller-owned `workspace`
 * (size from qhea_workspace_bytes; its first 256 bytes are a header the library
 * maintains -- keep them intact between calls), so every compute call is
 * hipGraph-capturable.  Nothing is retained past return.
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
 * Read-out (core/quantum_circuits_tq.py:106-127): H = ham_offset + ham_coeff * sum_i P_i, or,
 *   when ham_diag != NULL, H = diag(ham_diag[k]) with bit i of k = wire i.
 *   P is the Pauli named by ham_pauli (QHEA_PAULI_Z/X/Y; generate_simple_hamiltonian's `pauli`,
 *   core/quantum_circuits_ms.py:28-39).  The PyTorch back-ends of the reference only read out Z; X and Y are
 *   the MindQuantum path's --ham_pauli option.  ham_diag requires QHEA_PAULI_Z.
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
#define QHEA_EPIPELINE    -6   /* a backward kernel's wave-to-wave hand-off overran (qhea_check_status) */
#define QHEA_EEXCHANGE    -7   /* a data-parallel exchange did not hear from every rank in time (qhea_dp_status) */

#define QHEA_MIN_QUBITS 2      /* n=1 has no entangler in MindQuantum and an undefined one in TQ */
#define QHEA_MAX_QUBITS 12

/* Library version (major*10000 + minor*100 + patch). */
int qhea_version(void);

/* Text for a QHEA_* code. */
const char* qhea_strerror(int code);

/* Number of usable HIP devices (0 when none); never initialises a context on failure. */
int qhea_device_count(void);

/*
 * Backward-kernel variant for n <= 5 (no reference counterpart; the reference has one autograd path).
 * QHEA_BWD_AUTO (default) chooses by batch density: the psi / lambda / sigma wave pipeline while the sample groups
 * leave SIMDs free, the one-wave-per-group kernel otherwise.  The others force a variant (parity tests, batch sweeps).
 * Process-wide.  The choice fixes the layout of the per-wave partial sums, so set it BEFORE qhea_workspace_bytes()
 * and do not change it while calls that use that workspace are being issued.
 */
#define QHEA_BWD_AUTO   0
#define QHEA_BWD_PACKED 1      /* one wave per sample group: forward sweep, then psi and lambda walked back together */
#define QHEA_BWD_PAIR   2      /* psi wave + lambda wave                                                              */
#define QHEA_BWD_TRI    3      /* psi wave + lambda wave + two inner-product (sigma) waves                            */
#define QHEA_BWD_ZTRI   4      /* the same pipeline on the ZYZ form of the gates with in-kernel (cos, sin) tables     */
                               /* (what AUTO runs for eligible shapes; the values 1-3 also select the first-          */
                               /* generation forward kernel, AUTO, ZTRI and ZPACKED the ZYZ-form one)                 */
#define QHEA_BWD_ZPACKED 5     /* one wave per sample group in the ZYZ form (what AUTO runs once the batch fills the  */
                               /* SIMDs, for circuits whose blocks are one full RX chunk + 1 or 2 sub-layers)         */
#define QHEA_BWD_ZTRI2  6      /* ZTRI with two sample groups per workgroup whose gradient sums are added in LDS: half */
                               /* the partial rows.  AUTO does that only where it costs nothing (batches that fill     */
                               /* every CU's two slots); ZTRI2 forces it from one group per CU on, ZTRI never does it   */
#define QHEA_BWD_ZQUAD  7      /* n = 5, block-unrolled shapes: the pipeline with BOTH sweeps of every chain in the split layout */
                               /* (one sample per chain wave, four chain waves per sample group).  AUTO runs it while every    */
                               /* CU holds at most one sample group; ZQUAD forces it at any batch                               */
int qhea_set_backward_variant(int variant);

/*
 * Failure reporting for the pipelined backward kernels (QHEA_BWD_PAIR / QHEA_BWD_TRI).  Their waves hand states to
 * each other through LDS with bounded spins; a wait that overruns its bound aborts the workgroup's pipeline and the
 * kernel ORs a flag into a status word at the start of `workspace`.  The SAME call's reduce kernel then writes NaN
 * into every gradient and into the sse scalar and (qhea_model_train_step) skips the parameter update, so the failure
 * is visible in-band without any host synchronisation.  qhea_check_status() is the explicit check: it copies the
 * status word back on `stream`, WAITS for the stream (the only entry point that synchronises), clears the word and
 * returns QHEA_EPIPELINE if any call since the last check overran, QHEA_OK otherwise (also for a workspace no call has
 * used yet).  Call it where the caller synchronises anyway, e.g. once per epoch.
 */
int qhea_check_status(void* workspace /*DEVICE*/, size_t workspace_bytes, void* stream);

/*
 * Measurement hook (no reference counterpart): the NEXT qhea_backward / qhea_model_loss_grad /
 * qhea_forward / qhea_model_forward call on this thread records `start_event` immediately before and
 * `stop_event` immediately after its circuit kernel (fwd_kernel / bwd_kernel) on the call's stream, then
 * the hook disarms itself.  Both are hipEvent_t handles created with timing enabled; pass NULLs to
 * disarm.  Used by bench.py to time the dominant kernel alone with HIP events.
 */
int qhea_profile_next_circuit_kernel(void* start_event, void* stop_event);

/*
 * Diagnostic (no reference counterpart): n_workgroups one-wave workgroups each time a dependent fp64 FMA chain of `iters`
 * steps with the shader-clock counter and with the constant 100 MHz counter; ticks[2 w] / ticks[2 w + 1] x 100 MHz is the clock
 * workgroup w ran at.  bench.py prints the median, so that a step time can be read against the clock of the device it ran on
 * (devices of one model differ by several percent, and so do the latency-bound kernels here).
 */
int qhea_clock_probe(int n_workgroups, int64_t iters, unsigned long long* ticks /*DEVICE [2 * n_workgroups]*/, void* stream);

/*
 * Bytes of DEVICE scratch the calls below need for this circuit shape and batch.
 * Replaces: nothing in the reference (TorchQuantum allocates per-gate temporaries and
 * autograd saves every intermediate state; core/quantum_circuits_tq.py:74).
 * enc_per_block / ld_per_block: HOST arrays of length n_blocks.
 */
size_t qhea_workspace_bytes(int n_qubits, int n_blocks,
                            const int32_t* enc_per_block, const int32_t* ld_per_block,
                            int64_t batch);

#define QHEA_PAULI_Z 0
#define QHEA_PAULI_X 1
#define QHEA_PAULI_Y 2

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
                 int ham_pauli /*QHEA_PAULI_**/,
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
                  int ham_pauli /*QHEA_PAULI_**/,
                  const double* g /*DEVICE [B]*/, const double* state_in /*DEVICE or NULL*/,
                  double* out /*DEVICE [B] or NULL*/,
                  double* grad_x /*DEVICE [B,E]*/, double* grad_w /*DEVICE [blk,3,n]*/,
                  void* workspace /*DEVICE*/, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Model-level entry points: the same kernels with the reference's classical pre/post-processing and
 * the MSE loss fused in, so that one training step is three launches (prep, circuit, reduce) instead
 * of ~35 framework kernels.  They replace, for the HIP backend,
 *   QuanONetPT.forward / HEAQNNPT.forward            (core/models_pt.py:153-166, 205-213)
 *   _TiledElementWise / _ScaleRepeat                  (core/models_pt.py:14-68)
 *   nn.MSELoss + loss.backward() in the batch loop    (solvers/solver_pt.py:232-236)
 *
 * Parameters travel as ONE flat fp64 DEVICE vector in the order torch's nn.Module.parameters() /
 * state_dict() yields for the reference classes (a module's own parameters before its children's):
 *   QuanONet, trainable_freq=1: [bias (1) | branch_freq.weights (bd*n) | branch_freq.bias (bd*n) |
 *                                trunk_freq.weights (td*n) | trunk_freq.bias (td*n) |
 *                                quantum_layer.ansatz_weights (blk*3*n)]
 *   QuanONet, trainable_freq=0: [bias | quantum_layer.ansatz_weights]
 *   HEAQNN,   trainable_freq=1: [freq.weights (depth*n) | freq.bias (depth*n) | ansatz_weights]
 *   HEAQNN,   trainable_freq=0: [ansatz_weights]
 * and gradients come back in the same layout followed by two scalars [sse, sum_b y_b^2]
 * (grad has qhea_model_param_count()+2 entries) so that a data-parallel caller needs exactly one
 * SUM all-reduce per step.
 * ------------------------------------------------------------------------------------------------ */
#define QHEA_MODEL_QUANONET 0
#define QHEA_MODEL_HEAQNN   1

typedef struct qhea_model_desc {
    int32_t model;            /* QHEA_MODEL_*                                                        */
    int32_t n_qubits;
    int32_t net[4];           /* QuanONet: (branch_depth, branch_ld, trunk_depth, trunk_ld);         */
                              /* HEAQNN:   (depth, linear_depth, 0, 0)                               */
    int32_t branch_in;        /* QuanONet: branch input features; HEAQNN: input features             */
    int32_t trunk_in;         /* QuanONet: trunk input features;  HEAQNN: 0                          */
    int32_t trainable_freq;   /* 1: _TiledElementWise (weights+bias in params); 0: _ScaleRepeat      */
    int32_t ham_pauli;        /* QHEA_PAULI_* read-out basis (0 = Z, the reference's default)          */
    double  scale_coeff;      /* fixed scale when trainable_freq == 0                                */
    double  ham_offset, ham_coeff;   /* H = offset + coeff * sum P_i (ham_diag is passed per call)   */
} qhea_model_desc;

/* Number of trainable scalars for this model (layout above); negative QHEA_E* on a bad descriptor. */
int64_t qhea_model_param_count(const qhea_model_desc* desc);

/* DEVICE scratch bytes for the two calls below. */
size_t qhea_model_workspace_bytes(const qhea_model_desc* desc, int64_t batch);

/*
 * pred[b] = model(branch[b], trunk[b])   (bias included for QuanONet).
 * Replaces QuanONetPT.forward / HEAQNNPT.forward under torch.no_grad()
 * (solvers/solver_pt.py:299-310, infer.py:274-289).  trunk is ignored (may be NULL) for HEAQNN.
 */
int qhea_model_forward(const qhea_model_desc* desc, int64_t batch,
                       const double* branch /*DEVICE [B,branch_in]*/, const double* trunk /*DEVICE [B,trunk_in]*/,
                       const double* params /*DEVICE flat*/, const double* ham_diag /*DEVICE [2^n] or NULL*/,
                       double* pred /*DEVICE [B]*/, void* workspace, size_t workspace_bytes, void* stream);

/*
 * qhea_model_forward over `n_chunks` consecutive row ranges [row_begin[i], row_begin[i+1]) of the same arrays with the
 * SAME parameters -- the chunk loop of PTSolver.evaluate / infer.predict (solvers/solver_pt.py:299-310, infer.py:274-289) in
 * one host call.  The layer records depend on the parameters alone, so one preparation launch serves all chunks of
 * equal size; results are bitwise those of the single calls.  The workspace must fit the largest chunk.
 */
int qhea_model_forward_chunks(const qhea_model_desc* desc, int64_t n_chunks, const int64_t* row_begin /*HOST [n_chunks+1]*/,
                              const double* branch /*DEVICE*/, const double* trunk /*DEVICE or NULL*/,
                              const double* params /*DEVICE flat*/, const double* ham_diag, double* pred /*DEVICE [rows]*/,
                              void* workspace, size_t workspace_bytes, void* stream);

/*
 * One fused loss + gradient evaluation:
 *   pred_b = model(...);  loss contribution = (pred_b - y_b)^2 * inv_batch_total;
 *   grad[0..P) = d/dparams sum_b (pred_b - y_b)^2 * inv_batch_total;  grad[P] = sum_b (pred_b-y_b)^2;
 *   grad[P+1] = sum_b y_b^2.
 * inv_batch_total = 1 / (GLOBAL batch size): a rank's shard passes the global size, and a SUM over
 * ranks reproduces MSELoss(mean) gradients of the whole batch (solvers/solver_pt.py:232-236).
 * pred may be NULL.
 */
int qhea_model_loss_grad(const qhea_model_desc* desc, int64_t batch,
                         const double* branch, const double* trunk, const double* y /*DEVICE [B]*/,
                         const double* params, const double* ham_diag, double inv_batch_total,
                         double* grad /*DEVICE [P+2]*/, double* pred /*DEVICE [B] or NULL*/,
                         void* workspace, size_t workspace_bytes, void* stream);

/*
 * Single-device training step: qhea_model_loss_grad followed by qhea_adam_step on the same flat vectors, with the
 * Adam update applied by the very thread of the reduce kernel that finishes each gradient -- three launches
 * (prep, circuit, reduce+Adam) for `pred = model(...); loss = MSE(pred, y); loss.backward(); optimizer.step()`
 * (solvers/solver_pt.py:232-236).  Bitwise the same parameters as the two separate calls.  Data-parallel runs
 * keep the two calls, with the gradient all-reduce between them.  grad still receives [gradients | sse | sum y^2].
 */
int qhea_model_train_step(const qhea_model_desc* desc, int64_t batch,
                          const double* branch, const double* trunk, const double* y /*DEVICE [B]*/,
                          double* params /*DEVICE flat, updated in place*/, const double* ham_diag,
                          double inv_batch_total, double* grad /*DEVICE [P+2]*/, double* pred /*DEVICE [B] or NULL*/,
                          double* exp_avg /*DEVICE [P]*/, double* exp_avg_sq /*DEVICE [P]*/, int64_t step,
                          double lr, double beta1, double beta2, double eps, double weight_decay,
                          void* workspace, size_t workspace_bytes, void* stream);

/*
 * `n_steps` consecutive qhea_model_train_step calls over CONTIGUOUS rows, issued from one host call: the inner loop of
 * one epoch (solvers/solver_pt.py:226-241, after the epoch's rows have been gathered in batch order).  Step i takes
 * rows [row_begin[i], row_begin[i+1]) of branch / trunk / y, weights its residuals with inv_batch_total[i], leaves its
 * [P+2] buffer (gradients | sse | sum y^2) at grad + i * grad_stride and applies Adam update number first_step + i.
 * Nothing is synchronised; results are bitwise those of the single calls.  The workspace must fit the largest step.
 */
int qhea_model_train_steps(const qhea_model_desc* desc, int64_t n_steps, const int64_t* row_begin /*HOST [n_steps+1]*/,
                           const double* branch /*DEVICE*/, const double* trunk /*DEVICE or NULL*/,
                           const double* y /*DEVICE*/, double* params /*DEVICE flat*/, const double* ham_diag,
                           const double* inv_batch_total /*HOST [n_steps]*/, double* grad /*DEVICE [n_steps][grad_stride]*/,
                           int64_t grad_stride, double* exp_avg, double* exp_avg_sq, int64_t first_step, double lr,
                           double beta1, double beta2, double eps, double weight_decay, void* workspace,
                           size_t workspace_bytes, void* stream);

/*
 * One Adam update of the flat parameter vector, in place, in ONE launch.  Same arithmetic as
 * torch.optim.Adam (amsgrad=False, maximize=False; the reference's optimizer, solvers/solver_pt.py:149-163):
 *   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * with g += weight_decay * p first when weight_decay != 0.  `step` is the 1-based update count t.
 */
int qhea_adam_step(int64_t n, double* params /*DEVICE*/, const double* grads /*DEVICE*/,
                   double* exp_avg /*DEVICE*/, double* exp_avg_sq /*DEVICE*/, int64_t step,
                   double lr, double beta1, double beta2, double eps, double weight_decay, void* stream);

/*
 * Data-parallel gradient exchange (the reference has no multi-device step -- this replaces the
 * `all_reduce` + `optimizer.step()` pair a DistributedDataParallel port of solvers/solver_pt.py:232-237 would run).
 * Every rank owns ONE exchange buffer of fine-grained device memory that every other rank maps through hipIpc; one
 * one-workgroup kernel per rank and step writes the rank's flat buffer [gradients | sse | sum y^2] into every other rank's
 * buffer -- each value as two 8-byte words that carry the exchange's sequence number, so that a value validates itself
 * and nothing has to be drained or flagged --, polls its own buffer for the other ranks' words, sums in rank order (bitwise
 * identical on every rank, reproducible) and applies Adam -- no collective-library launch and no separate optimizer launch
 * on the step's critical path.  Set-up (once): qhea_dp_alloc -> qhea_dp_export -> exchange the 64-byte handles by any means
 * (torch.distributed.all_gather_object) -> qhea_dp_import each peer's.  The library retains nothing: the caller
 * owns the buffer and the mapped pointers and passes them to every call.
 */
#define QHEA_DP_MAX_RANKS    16
#define QHEA_DP_HANDLE_BYTES 64
size_t qhea_dp_buffer_bytes(int64_t n_values, int world);
/* allocate (fine-grained, zeroed) / free this rank's exchange buffer for `n_values` doubles per rank */
int qhea_dp_alloc(int64_t n_values, int world, void** buffer /*out: DEVICE*/);
int qhea_dp_free(void* buffer);
/* inter-process handle of an exchange buffer (QHEA_DP_HANDLE_BYTES bytes, host), and a peer's buffer mapped from one */
int qhea_dp_export(void* buffer /*DEVICE*/, void* handle64 /*HOST out*/);
int qhea_dp_import(const void* handle64 /*HOST*/, void** peer_buffer /*out: DEVICE pointer valid in this process*/);
int qhea_dp_close(void* peer_buffer);
/*
 * out[i] = sum over ranks r = 0..world-1 (in that order) of rank r's local[i], i < n_values; then, if params != NULL,
 * the Adam update of qhea_adam_step on params[0..n_params) with gradient out[i].  `buffers` is a HOST array of `world`
 * device pointers: buffers[rank] this rank's own buffer, the others as returned by qhea_dp_import.  `seq` counts the
 * exchanges on these buffers from 1 and must be the same on every rank for the same step; `local` and `out` may be
 * the same array.  A rank that does not hear from every other rank within `timeout_ms` writes NaN to `out`, skips the
 * update and raises the error that qhea_dp_status reports.
 */
int qhea_dp_allreduce_adam(int rank, int world, void* const* buffers /*HOST array of DEVICE pointers*/,
                           int64_t n_values, int64_t seq, const double* local /*DEVICE*/, double* out /*DEVICE*/,
                           int64_t n_params, double* params /*DEVICE or NULL*/, double* exp_avg, double* exp_avg_sq,
                           int64_t step, double lr, double beta1, double beta2, double eps, double weight_decay,
                           double timeout_ms, void* stream);
/*
 * waits for `stream`; QHEA_EEXCHANGE if an exchange on this buffer failed since the last call.  A timeout is fatal on
 * EVERY rank: the rank whose wait overran raises a sticky word in every rank's buffer, and from then on every exchange on
 * these buffers -- the late rank's included -- fails (NaN results, no update) and this call keeps returning
 * QHEA_EEXCHANGE; the replicas cannot drift apart silently.  Free and re-create the buffers to start over.
 */
int qhea_dp_status(void* buffer /*DEVICE: this rank's own*/, void* stream);

/*
 * The data-parallel training step with the exchange INSIDE the reduce kernel: qhea_model_train_steps for `world` ranks.
 * Step i of this rank trains on its shard rows [row_begin[i], row_begin[i+1]) (never empty) with residual weight
 * inv_batch_total[i] = 1 / GLOBAL batch; every block of the step's reduce kernel publishes the gradients it has just
 * summed to the peers' exchange buffers (exchange number first_seq + i; the qhea_dp_* buffers above, allocated for
 * dp_values >= P + 2 doubles), waits for the peers' blocks, adds the contributions in rank order, leaves the GLOBAL
 * [gradients | sse | sum y^2] in grad + i*grad_stride, applies Adam and -- between steps of equal shard size on the
 * block-unrolled shapes -- writes the next step's layer records: two launches per step (circuit, reduce) instead of
 * prep + circuit + reduce + exchange.  Bitwise the results of qhea_model_loss_grad + qhea_dp_allreduce_adam.  Replaces the
 * `loss.backward(); all_reduce; optimizer.step()` a DistributedDataParallel port of solvers/solver_pt.py:232-237 would run.
 * Returns QHEA_EUNSUPPORTED -- before anything is launched -- when a shard is empty or the reduce grid would not be
 * resident at once (more blocks than CUs: its blocks wait for their peers' blocks): use qhea_model_loss_grad +
 * qhea_dp_allreduce_adam for such a run.  Failure semantics as qhea_dp_allreduce_adam / qhea_dp_status.
 */
int qhea_model_dp_train_steps(const qhea_model_desc* desc, int64_t n_steps, const int64_t* row_begin /*HOST [n_steps+1]*/,
                              const double* branch, const double* trunk, const double* y, double* params,
                              const double* ham_diag, const double* inv_batch_total /*HOST [n_steps]*/,
                              double* grad /*DEVICE [n_steps, grad_stride]*/, int64_t grad_stride,
                              double* exp_avg, double* exp_avg_sq, int64_t first_step, double lr, double beta1,
                              double beta2, double eps, double weight_decay,
                              int rank, int world, void* const* buffers /*HOST array of DEVICE pointers*/,
                              int64_t dp_values, int64_t first_seq, double timeout_ms,
                              void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QUANONET_HEA_H */
