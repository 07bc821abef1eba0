/* nd4hip.h — C ABI of libnd4hip.so: the MI355X (gfx950) backend of nd4js's `nd.la` hot path.
 *
 * The reference (nd4js v1.3.0, pure JavaScript) has no FFI/plugin layer; its boundary for this path
 * is the JS call contract of src/la/{matmul,qr,lu,svd}.js on dense row-major Float64Array /
 * Int32Array buffers. This header is the C ABI a maintainer binds underneath those
 * functions (N-API shim: nd4js_amd/csrc/napi_shim.c; ctypes: nd4js_amd/_lib.py).
 *
 * Conventions
 *  - plain pointers and sizes only; every matrix is dense, row-major, contiguous, fp64; leading
 *    (batch) dimensions are flattened by the host into `batch`;
 *  - inputs are never modified; outputs are caller-allocated;
 *  - return 0 on success, negative on error; nd4hip_last_error() gives the message (thread-local);
 *  - `*_dev` entry points take DEVICE pointers and enqueue on the handle's stream without
 *    synchronising the host (except where a host-visible scalar is returned: noted per function);
 *    the un-suffixed entry points take HOST pointers (what the JS TypedArrays are) and do
 *    H2D -> kernels -> D2H -> stream sync internally;
 *  - one handle = one GPU + one stream + one growable device workspace; calls on one handle are
 *    serialised by the caller (the JS host is single-threaded, like the reference).
 */
#ifndef ND4HIP_H
#define ND4HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct nd4hip_handle nd4hip_handle;

#define ND4HIP_OK            0
#define ND4HIP_ERR_ARG      -1   /* bad argument / shape */
#define ND4HIP_ERR_HIP      -2   /* HIP runtime error (message has the hipError string) */
#define ND4HIP_ERR_NOCONV   -3   /* Jacobi SVD hit the sweep limit */
#define ND4HIP_ERR_NODEV    -4   /* no usable GPU */
#define ND4HIP_ERR_SINGULAR -5   /* Cholesky met a NaN pivot: 'Matrix contains NaNs or is (near) singular.' */
#define ND4HIP_ERR_XCHG     -6   /* an exchange between co-resident workgroups INSIDE a kernel timed out (the row-split QR panels, the
                                    multi-workgroup LU panels, the one-launch Hessenberg reduction / bidiagonalisation of one matrix of
                                    128 .. 2048 rows and columns): the results of the call are invalid — Q / R hold NaN, the permutation
                                    vector of an LU holds -1, U / H / B / V are unspecified — and the handle stays usable. Reported by the next entry point that
                                    synchronises: every host-pointer form, nd4hip_synchronize, nd4hip_timer_stop. The _dev forms
                                    enqueue only and return 0; their callers see the NaN / -1 markers or the code at the next
                                    nd4hip_synchronize. (The reference never returns a half-valid factorisation: lu.js:24-81.) */

/* ---- lifecycle --------------------------------------------------------------------------- */
int  nd4hip_device_count(void);
/* device < 0: use the current HIP device. Creates a private non-blocking stream. */
int  nd4hip_create(nd4hip_handle** out, int device);
void nd4hip_destroy(nd4hip_handle* h);
/* One handle over SEVERAL devices of the node (SURVEY.md §8b `nd4hip_init(handle**, device_ids, n_dev)`, §8e): the HOST-pointer
 * batched entry points cut the leading batch axis — the reference's loops over independent matrices, src/la/qr.js:43-49,
 * lu.js:34-40, svd_dc.js:918-925, matmul.js:44-70 — into contiguous blocks, one per device (nd4hip_partition), each block
 * uploaded, computed and downloaded by its own host thread on its own device and streams; results land directly in the
 * caller's arrays, so there is no data-path collective. The *_dev entry points, nd4hip_malloc / nd4hip_set_stream and single
 * matrices (batch 1: "replicas only") use device_ids[0]. nd4hip_destroy releases all devices. */
int  nd4hip_create_multi(nd4hip_handle** out, const int* device_ids, int n_dev);
/* number of devices behind the handle; their ids are written to device_ids[0..capacity) when it is not NULL */
int  nd4hip_device_list(nd4hip_handle* h, int* device_ids, int capacity);
/* the block [lo, hi) of a batch that device number `index` (0-based position in the handle's list) of n_dev devices processes:
 * contiguous, sizes differ by at most one, remainder to the low devices; devices beyond the batch get an empty block */
int  nd4hip_partition(int64_t batch, int n_dev, int index, int64_t* lo, int64_t* hi);
/* run on a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream). NULL is a
 * valid value: HIP's default (null) stream. nd4hip_reset_stream goes back to the private stream. */
int  nd4hip_set_stream(nd4hip_handle* h, void* hip_stream);
int  nd4hip_reset_stream(nd4hip_handle* h);
int  nd4hip_synchronize(nd4hip_handle* h);
const char* nd4hip_last_error(void);
const char* nd4hip_version(void);

/* device memory for hosts without their own allocator (the N-API shim, C/C++ drivers) */
int  nd4hip_malloc(nd4hip_handle* h, size_t bytes, void** dev_ptr);
int  nd4hip_free(nd4hip_handle* h, void* dev_ptr);
int  nd4hip_memcpy_h2d(nd4hip_handle* h, void* dst_dev, const void* src_host, size_t bytes);
int  nd4hip_memcpy_d2h(nd4hip_handle* h, void* dst_host, const void* src_dev, size_t bytes);

/* hipEvent timing on the handle's stream (bench.py roofline leg): start, enqueue work, stop -> ms */
int  nd4hip_timer_start(nd4hip_handle* h);
int  nd4hip_timer_stop(nd4hip_handle* h, float* ms_out);   /* synchronises the stream */

/* synthetic inputs: out[i] = u(seed, offset+i) in [-1,1), bit-identical to nd4js_amd/rng.py */
int  nd4hip_fill_uniform_dev(nd4hip_handle* h, uint32_t seed, uint32_t offset, int64_t n, double* out_dev);

/* ---- matmul2: replaces the matmul2_RR hot loop, src/la/matmul.js:31-74 (:49-53) -------------
 * C[b] (I x J) = A[b] (I x K) * B[b] (K x J), b = 0..batch-1; element strides strideA/strideB
 * between consecutive batch members, 0 = the operand is broadcast (the reference's odometer,
 * matmul.js:44-70, is flattened into (batch, stride) groups by the host wrapper). C is dense
 * [batch, I, J]. */
int nd4hip_dgemm_batched_dev(nd4hip_handle* h, int64_t batch, int64_t I, int64_t K, int64_t J,
                             const double* A, int64_t strideA, const double* B, int64_t strideB, double* C);
int nd4hip_dgemm_batched    (nd4hip_handle* h, int64_t batch, int64_t I, int64_t K, int64_t J,
                             const double* A, int64_t strideA, const double* B, int64_t strideB, double* C);

/* general strided form used by the decompositions and exposed for tests:
 * C = alpha * op(A) * op(B) + beta * C, row-major with leading dimensions; transA/transB != 0
 * means the stored matrix is the transpose of the operand (A stored K x M, B stored N x K). */
int nd4hip_dgemm_ex_dev(nd4hip_handle* h, int transA, int transB, int64_t M, int64_t N, int64_t K,
                        double alpha, const double* A, int64_t lda, const double* B, int64_t ldb,
                        double beta, double* C, int64_t ldc);

/* ---- lu_decomp: replaces src/la/lu.js:24-81 ---------------------------------------------------
 * A [batch,N,N] -> LU [batch,N,N] (unit-L below the diagonal, U on/above) and the PERMUTATION
 * VECTOR P [batch,N] int32 with A[P[i],:] = (L*U)[i,:] (not LAPACK ipiv); pivot = first maximum of
 * |x| down the column (lu.js:48-52). N > 2048 uses panels whose rows are spread over co-resident workgroups with one in-kernel
 * exchange per column: should such an exchange time out, every entry of P is -1 (never a half-valid permutation) and the next
 * synchronising entry point returns ND4HIP_ERR_XCHG; the host-pointer form returns it itself. */
int nd4hip_dgetrf_batched_dev(nd4hip_handle* h, int64_t batch, int64_t N, const double* A, double* LU, int32_t* P);
int nd4hip_dgetrf_batched    (nd4hip_handle* h, int64_t batch, int64_t N, const double* A, double* LU, int32_t* P);

/* ---- lu_solve: replaces src/la/lu.js:84-177 -------------------------------------
 * X [batch,N,J] = U^-1 L^-1 Y[P,:] for LU/P as returned by dgetrf. strides in ELEMENTS between consecutive
 * batch members, 0 = the operand is broadcast over the batch (lu.js:148-163 broadcasting, flattened by
 * the host wrapper). */
int nd4hip_dgetrs_batched_dev(nd4hip_handle* h, int64_t batch, int64_t N, int64_t J, const double* LU, int64_t strideLU,
                              const int32_t* P, int64_t strideP, const double* Y, int64_t strideY, double* X);
int nd4hip_dgetrs_batched    (nd4hip_handle* h, int64_t batch, int64_t N, int64_t J, const double* LU, int64_t strideLU,
                              const int32_t* P, int64_t strideP, const double* Y, int64_t strideY, double* X);

/* ---- tril_solve / triu_solve: replace src/la/tri.js:155-290 (kernels :45-95) ----------------------
 * X [batch,M,J] = T^-1 Y with T [batch,M,M] lower (upper = 0) or upper (upper != 0) triangular; only that
 * triangle of T is read; unit_diag != 0 treats the diagonal as ones. strideT / strideY = 0 broadcast. */
int nd4hip_dtrsm_batched_dev(nd4hip_handle* h, int upper, int unit_diag, int64_t batch, int64_t M, int64_t J,
                             const double* T, int64_t strideT, const double* Y, int64_t strideY, double* X);
int nd4hip_dtrsm_batched    (nd4hip_handle* h, int upper, int unit_diag, int64_t batch, int64_t M, int64_t J,
                             const double* T, int64_t strideT, const double* Y, int64_t strideY, double* X);

/* ---- cholesky_decomp: replaces src/la/cholesky.js:51-71 (kernel :27-48)    -------------
 * S [batch,N,N] symmetric positive definite, only the lower triangle is read (:65-67) -> L [batch,N,N] lower
 * triangular with exact zeros above the diagonal, S = L L^T. A NaN pivot in any matrix returns ND4HIP_ERR_SINGULAR
 * with the reference's message (:43-44); L then holds NaNs from that pivot on. Both forms synchronise once
 * (a batch-sized flag read-back) to decide that. */
int nd4hip_dpotrf_batched_dev(nd4hip_handle* h, int64_t batch, int64_t N, const double* S, double* L);
int nd4hip_dpotrf_batched    (nd4hip_handle* h, int64_t batch, int64_t N, const double* S, double* L);

/* ---- cholesky_solve: replaces src/la/cholesky.js:74-150 ------------------------------------------------
 * X [batch,N,J] = L^-T L^-1 Y (forward substitution tri.js:45-71, then backward with the transpose :100-125).
 * strides in elements, 0 = broadcast. */
int nd4hip_dpotrs_batched_dev(nd4hip_handle* h, int64_t batch, int64_t N, int64_t J, const double* L, int64_t strideL,
                              const double* Y, int64_t strideY, double* X);
int nd4hip_dpotrs_batched    (nd4hip_handle* h, int64_t batch, int64_t N, int64_t J, const double* L, int64_t strideL,
                              const double* Y, int64_t strideY, double* X);

/* ---- ldl_decomp / ldl_solve: replace src/la/ldl.js:67-90 (kernel :47-64) and :133-201 (kernel :93-130) ----
 * S [batch,N,N] symmetric (lower triangle read), no pivoting -> packed LD [batch,N,N]: unit-lower L below the
 * diagonal, D on it, exact zeros above; S = L D L^T. A zero pivot propagates Inf/NaN exactly like the reference
 * (no check there either). dldltrs: X [batch,N,J] = L^-T D^-1 L^-1 Y; strides in elements, 0 = broadcast. */
int nd4hip_dldltrf_batched_dev(nd4hip_handle* h, int64_t batch, int64_t N, const double* S, double* LD);
int nd4hip_dldltrf_batched    (nd4hip_handle* h, int64_t batch, int64_t N, const double* S, double* LD);
int nd4hip_dldltrs_batched_dev(nd4hip_handle* h, int64_t batch, int64_t N, int64_t J, const double* LD, int64_t strideLD,
                               const double* Y, int64_t strideY, double* X);
int nd4hip_dldltrs_batched    (nd4hip_handle* h, int64_t batch, int64_t N, int64_t J, const double* LD, int64_t strideLD,
                               const double* Y, int64_t strideY, double* X);

/* ---- bidiag_decomp: replaces src/la/bidiag.js:245-319 (kernels :32-242) -----------------------------------
 * A [batch,M,N] -> U [batch,M,K], B [batch,K,J] upper bidiagonal (exact zeros elsewhere), V [batch,J,N] with A = U B V,
 * K = min(M,N), J = K for M >= N and K+1 for M < N; U has orthonormal columns, V orthonormal rows. The signs follow
 * the reference's three branches (see csrc/bidiag.hip), so U, B, V agree with it to rounding. One matrix with 128 <= M, N <= 2048
 * is reduced by ONE launch whose 256 workgroups exchange inside the kernel (csrc/bidiag.hip: bdp): ND4HIP_ERR_XCHG applies. */
int nd4hip_dgebrd_batched_dev(nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, const double* A, double* U, double* B, double* V);
int nd4hip_dgebrd_batched    (nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, const double* A, double* U, double* B, double* V);

/* ---- hessenberg_decomp: replaces src/la/hessenberg.js:89-115 (kernel :27-86) -------------------------------
 * A [batch,N,N] -> U, H [batch,N,N] with A = U H U^T, U orthogonal (last row and column = unit vector), H upper
 * Hessenberg with exact zeros below the sub-diagonal. Same reflectors as the reference (rows finished from the bottom
 * up, sign chosen against cancellation), so U and H agree with it to rounding. H may alias A in the _dev form. One matrix with
 * 128 <= N <= 2048 is reduced by ONE launch whose 256 workgroups exchange inside the kernel (csrc/hess.hip: hessp): ND4HIP_ERR_XCHG
 * applies. */
int nd4hip_dgehrd_batched_dev(nd4hip_handle* h, int64_t batch, int64_t N, const double* A, double* U, double* H);
int nd4hip_dgehrd_batched    (nd4hip_handle* h, int64_t batch, int64_t N, const double* A, double* U, double* H);

/* ---- qr_lstsq: replaces src/la/qr.js:186-273 -------------------------------------
 * X [batch,I,J] = R[0:L,0:L]^-1 (Q^T Y)[0:L,:], L = min(M,I), rows L..I-1 zero; Q [batch,N,M], R [batch,M,I]
 * (as returned by dgeqrf_q for an N x I system: M = min(N,I)), Y [batch,N,J]. I > N is refused like qr.js:209.
 * strides in elements, 0 = broadcast. One TN GEMM + one blocked triangular solve on the device. */
int nd4hip_dqrls_batched_dev(nd4hip_handle* h, int64_t batch, int64_t N, int64_t M, int64_t I, int64_t J,
                             const double* Q, int64_t strideQ, const double* R, int64_t strideR,
                             const double* Y, int64_t strideY, double* X);
int nd4hip_dqrls_batched    (nd4hip_handle* h, int64_t batch, int64_t N, int64_t M, int64_t I, int64_t J,
                             const double* Q, int64_t strideQ, const double* R, int64_t strideR,
                             const double* Y, int64_t strideY, double* X);

/* ---- svd_lstsq / svd_solve: replace src/la/svd.js:66-228 ---------------------------------------------
 * X [batch,I,J] = V[0:r,:]^T diag(1/sv[0:r]) U[:,0:r]^T Y with r = first index with |sv_r| <= sqrt(eps)|sv_0|
 * (svd_rank, svd.js:31-63), U [batch,N,M], sv [batch,M], V [batch,M,I], Y [batch,N,J]. The host form
 * refuses non-finite singular values (svd.js:171-172, ND4HIP_ERR_ARG); the _dev form does not look. */
int nd4hip_dsvdls_batched_dev(nd4hip_handle* h, int64_t batch, int64_t N, int64_t M, int64_t I, int64_t J,
                              const double* U, int64_t strideU, const double* sv, int64_t strideSv,
                              const double* V, int64_t strideV, const double* Y, int64_t strideY, double* X);
int nd4hip_dsvdls_batched    (nd4hip_handle* h, int64_t batch, int64_t N, int64_t M, int64_t I, int64_t J,
                              const double* U, int64_t strideU, const double* sv, int64_t strideSv,
                              const double* V, int64_t strideV, const double* Y, int64_t strideY, double* X);

/* ---- qr_decomp: replaces src/la/qr.js:80-145 / qr_decomp_full :27-77 ----------------------------
 * A [batch,M,N] -> Q [batch,M,L], R [batch,L,N], L = min(M,N); blocked Householder with the
 * reference's Givens sign convention restored (R_jj >= 0 wherever a column had something to
 * eliminate, det(Q)=+1 for M <= N: SURVEY.md §8 A4). */
int nd4hip_dgeqrf_q_batched_dev(nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, const double* A, double* Q, double* R);
int nd4hip_dgeqrf_q_batched    (nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, const double* A, double* Q, double* R);

/* ---- qr_decomp_full: replaces src/la/qr.js:27-77 for every shape -----------------
 * A [batch,M,N] -> Q [batch,M,M], R [batch,M,N]. For M <= N identical to dgeqrf_q. For M > N the Givens-full
 * convention applies (R_jj >= 0 wherever something was eliminated, rows N.. of R zero) and the trailing M-N
 * columns of Q are AN orthonormal completion: the reference's own depends on its rotation order, so only
 * Q[:, 0:N], R and the properties Q Q^T = I, Q R = A are comparable. */
int nd4hip_dgeqrf_full_batched_dev(nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, const double* A, double* Q, double* R);
int nd4hip_dgeqrf_full_batched    (nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, const double* A, double* Q, double* R);

/* ---- _qr_decomp_inplace: replaces src/la/qr.js:146-183 (used by opt/_trust_region_solver_tls.js:1126) ----
 * In place: A [batch,M,N] <- R (upper trapezoid, zeros below), Y [batch,M,L] <- Q^T Y with the full M x M Q of
 * dgeqrf_full. Rows N.. of the result Y (only when M > N) are expressed in this library's completion basis. */
int nd4hip_dgeqrf_qty_batched_dev(nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, int64_t L, double* A, double* Y);
int nd4hip_dgeqrf_qty_batched    (nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, int64_t L, double* A, double* Y);

/* ---- svd_decomp: replaces the output contract of src/la/svd.js:25 (= svd_dc.js:883-932) ----------
 * A [batch,M,N] -> U [batch,M,L], sv [batch,L] (>= 0, descending), V [batch,L,N] (rows = right
 * singular vectors), L = min(M,N); one-sided Jacobi with the reference's Jacobi post-processing
 * contract (_svd_jac_utils.js:123-188). sweeps_out / offnorm_out are HOST pointers (may be NULL):
 * max sweeps over the batch and the largest remaining |a_p.a_q| / (|a_p||a_q|). Both forms
 * synchronise the stream (the sweep loop is host-driven). */
int nd4hip_dgesvdj_batched_dev(nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, const double* A,
                               double* U, double* sv, double* V, int* sweeps_out, double* offnorm_out);
int nd4hip_dgesvdj_batched    (nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, const double* A,
                               double* U, double* sv, double* V, int* sweeps_out, double* offnorm_out);

/* ---- one Householder panel on its own: geqr2 + larft of a 16-column panel (the inner step of qr_decomp, src/la/qr.js:54-68
 * eliminates the same entries column by column with Givens rotations) ---------------------------------------------
 * A [batch,M,16] (1 <= M <= 2048) is overwritten with R in its top 16 x 16 (entries below the diagonal are left as the kernel
 * leaves them); V [batch,M,16] receives the panel's reflector block, T [batch,16,16] its factor: Q_panel = I - V T V^T is
 * orthogonal and Q_panel^T A = [R; 0]. Panels of fewer than 64 rows take the thread-per-row Householder kernel: V explicit unit
 * lower trapezoidal, T upper triangular (compact WY). Everything else is CholeskyQR2 + a compact orthogonal completion:
 * V = Q - [S; 0] with a full top block, T a full 16 x 16 matrix — up to 8 panels as the row-split launch (the rows of a panel over
 * co-resident workgroups), more than 8 as ONE workgroup per panel on the matrix cores (round 4, qr_batched_panel.h), which reads
 * every element of A once and writes V once and nothing else: the rows of A below the top 16 are left untouched. A panel the Gram
 * route must not take (nearly dependent columns, a column that is zero below the top block, non-finite data) is factorised by
 * the thread-per-row kernel in the same launch (compact WY form, rows below R zeroed).
 * Device pointers. Exposed for the panel roofline of bench.py (16 M b bytes per panel). */
int nd4hip_dgeqr2_panel_batched_dev(nd4hip_handle* h, int64_t batch, int64_t M, int64_t b, double* A, double* V, double* T);

/* Executed-work audit of the LAST nd4hip_dgesvdj_batched[_dev] call on this handle ("must print sweeps and
 * rotations actually applied"; the reference's loop is svd_jac_2sided.js:95-134): sweeps = max over the batch, rotations =
 * plane rotations applied over all matrices and sweeps, offnorm = largest |a_p.a_q| / (|a_p||a_q|) over the row pairs as they
 * were found during the last sweep (<= N*eps at convergence). Any pointer may be NULL. */
int nd4hip_dgesvdj_last_info(nd4hip_handle* h, int* sweeps, unsigned long long* rotations, double* offnorm);

/* ---- per-call profile (SURVEY.md 8b `nd4hip_profile_last`; the reference times calls with performance.now(),
 * benchmarks/bench_la_decomps.html:218-224) ---------------------------------------------------------------------
 * nd4hip_profile_enable(h, 1): from now on every device-side entry point (the *_dev forms; the host-pointer forms run them per
 * chunk and per device) brackets its kernels with two HIP events on the handle's stream and records the ALGORITHMIC work of the
 * call (SURVEY.md 8d conventions: matmul 2 I K J, LU 2/3 N^3, QR with explicit Q 4 (M N^2 - N^3 / 3) for M >= N, SVD nominal
 * 4 M^2 N + 8 M N^2 + 9 N^3, ...; bytes = every operand read once and every result written once). Off by default (two event
 * records per call). nd4hip_profile_last fills one record per device of the handle (capacity entries at most; a multi-device
 * handle's devices each report the last block they ran), waits for those kernels to finish, and returns the number of devices.
 * For a host-pointer call that was cut into chunks a record describes the LAST chunk of that device. */
typedef struct nd4hip_prof {
  double kernel_ms;   /* HIP-event time between the first and the last kernel of the call on this device */
  double flops;       /* algorithmic flops of the call (0 for pure data movement) */
  double bytes;       /* algorithmic HBM bytes of the call */
  int    device;      /* HIP device id */
  int    valid;       /* 0: nothing recorded on this device yet */
  char   op[32];      /* entry point without the nd4hip_ prefix, e.g. "dgetrf_batched" */
} nd4hip_prof;
int nd4hip_profile_enable(nd4hip_handle* h, int on);
int nd4hip_profile_last(nd4hip_handle* h, nd4hip_prof* out, int capacity);

#ifdef __cplusplus
}
#endif
#endif /* ND4HIP_H */
