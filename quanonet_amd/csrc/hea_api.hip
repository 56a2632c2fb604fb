// hea_api.hip -- C ABI (include/quanonet_hea.h) of the MI355X HEA simulator: argument checks,
// workspace layout, the batch-invariant prep / reduce kernels and the per-qubit-count dispatch.
#include "hea_device.hpp"

namespace qhea {

// U[s*n+q] = RY(w[s,2,q]) RZ(w[s,1,q]) RY(w[s,0,q]) as (ar,ai,br,bi);  cs[b,e] = (cos, sin)(x[b,e]/2)
__global__ void prep_kernel(int n, int blk, const double* __restrict__ w, double4* __restrict__ U,
                            long BE, const double* __restrict__ x, double2* __restrict__ cs) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long ng = (long)blk * n;
    if (tid < ng) {
        const int s = (int)(tid / n), q = (int)(tid % n);
        const double* ws = w + (long)s * 3 * n;
        double sa, ca, sb, cb, sc, cc;
        sincos(0.5 * ws[q], &sa, &ca);
        sincos(0.5 * ws[n + q], &sb, &cb);
        sincos(0.5 * ws[2 * n + q], &sc, &cc);
        // M = RZ(b) RY(a): M00 = e^{-ib/2} ca, M01 = -e^{-ib/2} sa, M10 = e^{+ib/2} sa, M11 = e^{+ib/2} ca
        const double m00r = cb * ca, m00i = -sb * ca;
        const double m01r = -cb * sa, m01i = sb * sa;
        const double m10r = cb * sa, m10i = sb * sa;
        const double m11r = cb * ca, m11i = sb * ca;
        // U = RY(c) M: U00 = cc M00 - sc M10, U01 = cc M01 - sc M11
        U[tid] = make_double4(cc * m00r - sc * m10r, cc * m00i - sc * m10i,
                              cc * m01r - sc * m11r, cc * m01i - sc * m11i);
    } else if (tid - ng < BE) {
        const long t = tid - ng;
        double s, c;
        sincos(0.5 * x[t], &s, &c);
        cs[t] = make_double2(c, s);
    }
}

// grad_w[s,{0,1,2},q] from the per-wave (X,Y,Z) partial sums.
//   g_c = Y;  g_b = cos(c) Z + sin(c) X;  g_a = cos(b) Y - sin(b) cos(c) X + sin(b) sin(c) Z
// One thread per (s, q); waves are summed in index order (deterministic).
__global__ void reduce_kernel(int n, int blk, int kw, long nwaves, const double* __restrict__ partial,
                              const double* __restrict__ w, double* __restrict__ grad_w) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (long)blk * n) return;
    const int s = (int)(tid / n), q = (int)(tid % n);
    const double* p = partial + (long)s * kw + 3 * q;
    const long stride = (long)blk * kw;
    double X = 0.0, Y = 0.0, Z = 0.0;
    for (long wv = 0; wv < nwaves; ++wv) {
        X += p[0]; Y += p[1]; Z += p[2];
        p += stride;
    }
    const double* ws = w + (long)s * 3 * n;
    double sb, cb, sc, cc;
    sincos(ws[n + q], &sb, &cb);
    sincos(ws[2 * n + q], &sc, &cc);
    double* gs = grad_w + (long)s * 3 * n;
    gs[2 * n + q] = Y;
    gs[n + q] = cc * Z + sc * X;
    gs[q] = cb * Y - sb * cc * X + sb * sc * Z;
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
struct Shape {
    long E = 0, blk = 0;
    Runs runs{};
};

int make_shape(int n, int nb, const int32_t* enc, const int32_t* ld, Shape& sh) {
    if (n < QHEA_MIN_QUBITS || n > QHEA_MAX_QUBITS || nb < 0) return QHEA_EINVAL;
    if (nb > 0 && (!enc || !ld)) return QHEA_EINVAL;
    sh.runs.nruns = 0;
    for (int b = 0; b < nb; ++b) {
        if (enc[b] < 0 || ld[b] < 0) return QHEA_EINVAL;
        sh.E += enc[b];
        sh.blk += ld[b];
        const int k = sh.runs.nruns;
        if (k > 0 && sh.runs.enc[k - 1] == enc[b] && sh.runs.ld[k - 1] == ld[b]) {
            sh.runs.count[k - 1]++;
        } else {
            if (k == kMaxRuns) return QHEA_EUNSUPPORTED;
            sh.runs.count[k] = 1; sh.runs.enc[k] = enc[b]; sh.runs.ld[k] = ld[b];
            sh.runs.nruns = k + 1;
        }
    }
    if (sh.E > INT32_MAX || sh.blk > INT32_MAX) return QHEA_EINVAL;
    return QHEA_OK;
}

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct Layout {
    size_t off_U, off_cs, off_part, total;
    long nwaves;
};

Layout make_layout(int n, const Shape& sh, int64_t B) {
    Layout L{};
    const int spw = n < 6 ? (64 >> n) : 1;
    L.nwaves = (B + spw - 1) / spw;
    const long nwg = (L.nwaves + kWaves - 1) / kWaves;
    L.nwaves = nwg * kWaves;                       // padding waves write zeros
    size_t p = 0;
    L.off_U = p;    p = align_up(p + (size_t)sh.blk * n * sizeof(double4));
    L.off_cs = p;   p = align_up(p + (size_t)B * sh.E * sizeof(double2));
    L.off_part = p; p = align_up(p + (size_t)L.nwaves * sh.blk * padded_3n(n) * sizeof(double));
    L.total = p;
    return L;
}

int launch_prep(int n, const Shape& sh, int64_t B, const double* w, const double* x, char* ws, const Layout& L,
                hipStream_t st) {
    const long total = sh.blk * n + B * sh.E;
    if (total == 0) return QHEA_OK;
    const int threads = 256;
    const long blocks = (total + threads - 1) / threads;
    hipLaunchKernelGGL(prep_kernel, dim3((unsigned)blocks), dim3(threads), 0, st, n, (int)sh.blk, w,
                       reinterpret_cast<double4*>(ws + L.off_U), (long)(B * sh.E), x,
                       reinterpret_cast<double2*>(ws + L.off_cs));
    return hipGetLastError() == hipSuccess ? QHEA_OK : QHEA_ELAUNCH;
}

}  // namespace qhea

using namespace qhea;

extern "C" {

int qhea_version(void) { return 100; }

const char* qhea_strerror(int code) {
    switch (code) {
        case QHEA_OK: return "ok";
        case QHEA_EINVAL: return "invalid argument";
        case QHEA_EUNSUPPORTED: return "unsupported circuit shape";
        case QHEA_EWORKSPACE: return "workspace missing or too small";
        case QHEA_ELAUNCH: return "HIP launch/runtime failure";
        case QHEA_ENODEVICE: return "no usable HIP device";
        default: return "unknown error";
    }
}

int qhea_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

size_t qhea_workspace_bytes(int n_qubits, int n_blocks, const int32_t* enc_per_block,
                            const int32_t* ld_per_block, int64_t batch) {
    Shape sh;
    if (make_shape(n_qubits, n_blocks, enc_per_block, ld_per_block, sh) != QHEA_OK || batch < 0) return 0;
    return make_layout(n_qubits, sh, batch).total;
}

int qhea_forward(int n_qubits, int n_blocks, const int32_t* enc_per_block, const int32_t* ld_per_block,
                 int64_t batch, const double* x, const double* w, double ham_offset, double ham_coeff,
                 const double* ham_diag, double* out, double* state_out, void* workspace,
                 size_t workspace_bytes, void* stream) {
    Shape sh;
    int rc = make_shape(n_qubits, n_blocks, enc_per_block, ld_per_block, sh);
    if (rc != QHEA_OK) return rc;
    if (batch < 0) return QHEA_EINVAL;
    if (batch == 0) return QHEA_OK;
    if (!out || (sh.E > 0 && !x) || (sh.blk > 0 && !w)) return QHEA_EINVAL;
    const Layout L = make_layout(n_qubits, sh, batch);
    if (!workspace || workspace_bytes < L.total) return QHEA_EWORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    rc = launch_prep(n_qubits, sh, batch, w, x, ws, L, st);
    if (rc != QHEA_OK) return rc;
    const dim3 grid((unsigned)(L.nwaves / kWaves));
    const double2* cs = reinterpret_cast<const double2*>(ws + L.off_cs);
    const double4* U = reinterpret_cast<const double4*>(ws + L.off_U);
const FwdArgs fa{sh.runs, (long)batch, (int)sh.E, cs, U, ham_offset, ham_coeff, ham_diag, out, state_out};
    switch (n_qubits) {
#define QHEA_CASE(NN) case NN: launch_fwd_##NN(grid, st, fa); break;
        QHEA_FOR_EACH_N(QHEA_CASE)
#undef QHEA_CASE
        default: return QHEA_EUNSUPPORTED;
    }
    return hipGetLastError() == hipSuccess ? QHEA_OK : QHEA_ELAUNCH;
}

int qhea_backward(int n_qubits, int n_blocks, const int32_t* enc_per_block, const int32_t* ld_per_block,
                  int64_t batch, const double* x, const double* w, double ham_offset, double ham_coeff,
                  const double* ham_diag, const double* g, const double* state_in, double* out,
                  double* grad_x, double* grad_w, void* workspace, size_t workspace_bytes, void* stream) {
    Shape sh;
    int rc = make_shape(n_qubits, n_blocks, enc_per_block, ld_per_block, sh);
    if (rc != QHEA_OK) return rc;
    if (batch < 0) return QHEA_EINVAL;
    if ((sh.blk > 0 && (!w || !grad_w))) return QHEA_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (batch == 0) {
        if (sh.blk > 0 && hipMemsetAsync(grad_w, 0, sizeof(double) * sh.blk * 3 * n_qubits, st) != hipSuccess)
            return QHEA_ELAUNCH;
        return QHEA_OK;
    }
    if (!g || (sh.E > 0 && (!x || !grad_x))) return QHEA_EINVAL;
    const Layout L = make_layout(n_qubits, sh, batch);
    if (!workspace || workspace_bytes < L.total) return QHEA_EWORKSPACE;
    char* ws = static_cast<char*>(workspace);
    rc = launch_prep(n_qubits, sh, batch, w, x, ws, L, st);
    if (rc != QHEA_OK) return rc;
    const dim3 grid((unsigned)(L.nwaves / kWaves));
    const double2* cs = reinterpret_cast<const double2*>(ws + L.off_cs);
    const double4* U = reinterpret_cast<const double4*>(ws + L.off_U);
    double* partial = reinterpret_cast<double*>(ws + L.off_part);
const BwdArgs ba{sh.runs, (long)batch, (int)sh.E, (int)sh.blk, cs, U, ham_offset, ham_coeff, ham_diag, g,
                     state_in, out, grad_x, partial};
    switch (n_qubits) {
#define QHEA_CASE(NN) case NN: launch_bwd_##NN(grid, st, ba); break;
        QHEA_FOR_EACH_N(QHEA_CASE)
#undef QHEA_CASE
        default: return QHEA_EUNSUPPORTED;
    }
    if (hipGetLastError() != hipSuccess) return QHEA_ELAUNCH;
    if (sh.blk > 0) {
        const long nthreads = sh.blk * n_qubits;
        hipLaunchKernelGGL(reduce_kernel, dim3((unsigned)((nthreads + 127) / 128)), dim3(128), 0, st, n_qubits,
                           (int)sh.blk, padded_3n(n_qubits), L.nwaves, partial, w, grad_w);
    }
    return hipGetLastError() == hipSuccess ? QHEA_OK : QHEA_ELAUNCH;
}

}  // extern "C"
