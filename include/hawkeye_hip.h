/* hawkeye_hip.h - C ABI of libhawkeye_hip.so (MI355X / gfx950 kernels for the
 * Hawkeye high-order pooling + attention-pooling hot path).
 *
 * The reference (Hawkeye-FineGrained/Hawkeye) is pure Python/PyTorch: it has no
 * FFI of its own.  Each entry point below therefore replaces a *sequence of ATen
 * calls* issued by one reference function; the function is cited as
 * "replaces <file>:<lines>" (paths relative to the reference checkout).  The
 * Python binding a maintainer adds on the reference side is a ctypes stub inside
 * a torch.autograd.Function - see INTEGRATION.md.
 *
 * Conventions
 *  - all pointers are DEVICE pointers to contiguous fp32 (or int32 where said)
 *    buffers owned by the caller; nothing is allocated or freed by the library;
 *  - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream);
 *    every call only enqueues kernels on it: no host synchronisation.  Process-wide
 *    state is limited to (a) the test-only hk_tuning_* knobs and (b) one helper HIP
 *    queue + event pair per device, owned by the library, created once
 *    (std::call_once) on the first batched Newton-Schulz call and used inside
 *    hk_ns_sqrtm_fwd / _bwd only: the two halves of the batch are forked onto it
 *    and joined back onto `stream` before the entry point returns - also on its
 *    error paths.  A per-device mutex is held from fork to join, so several host
 *    threads may drive one device (they take turns on the helper queue);
 *  - scratch memory is passed in (`ws`, `ws_bytes`); the matching
 *    hk_*_ws_bytes() tells how much is needed; contents need not be preserved
 *    between calls unless stated ("saved for backward" buffers are explicit
 *    arguments);
 *  - return value: 0 on success, a negative HK_ERR_* for bad arguments, or a
 *    positive hipError_t from the launch;
 *  - reductions use fixed orders (no float atomics): reruns are bit-identical.
 */
#ifndef HAWKEYE_HIP_H
#define HAWKEYE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* hk_stream_t; /* hipStream_t */

#define HK_OK 0
#define HK_ERR_BAD_ARG (-1)
#define HK_ERR_WORKSPACE (-2)
#define HK_ERR_UNSUPPORTED (-3)

/* library / build identification: returns e.g. "hawkeye_hip 0.1 gfx950" */
const char* hk_version(void);

/* A/B levers for tests, benchmarks and profiling - never needed by a caller of the ops.  Each knob selects between
 * implementations that produce the same results (bit-identical or to rounding); the defaults are the measured winners.
 *   "bcnn_generic"  1 = generic GEMM path for the Gram / covariance / CBP kernels and the three-kernel CIN forward
 *   "bwd_v"         Gram backward (BCNN / signed-sqrt / covariance / CBP): 0 automatic (gram_bwd3_kernel / cbp_bwd3_kernel
 *                   where their row blocks fill the chip, else the 64-row panel kernel); 1 panel kernel
 *   "cbp_bin"       CBP forward: -1 automatic (the fused Gram + binning kernel where the plan allows it, else by batch
 *                   size), 0 row-sketch, 1 CSR gather, 2 row-scatter, 3 fused, 4 fused without paired tile steps
 *   "roi_bwd"       0 = LDS-staged ROI refinement kernels, 1 = the round-1 table kernels
 *   "linear_slabs"  split-K slabs of hk_linear_fwd (0 automatic, -1 generic tiles)
 *   "ns_tn"         tile width of the Newton-Schulz products (0 automatic, 64, 128)
 *   "ns_streams"    n: the batch runs the Newton-Schulz chain in n + 1 parts on n + 1 HIP queues (default 1, at most 3)
 *   "ns_sym"        1 (default): hk_ns_sqrtm_fwd_sym skips the tiles below the diagonal blocks; 0: it computes all
 *   "lin_walk"      classifier backward: 1 workgroup s walks the 64-feature chunks s, s + S, ..; 0 a contiguous slab per
 *                   workgroup; -1 (default) the measured winner per kernel
 *   "sched_b"       > 0: batch-size dependent work splits behave as if the batch were this (tests)
 * Values are seeded once from the environment (HK_<NAME>) when the library is first used; the launch paths never read
 * the environment.  Returns HK_ERR_BAD_ARG for an unknown name.  Process-wide: set them only while no other thread is
 * launching. */
int hk_tuning_set(const char* name, int value);
int hk_tuning_get(const char* name, int* value);

/* ---------------------------------------------------------------- BCNN ----
 * Bilinear pooling: G = X X^T / HW ; z = sqrt(G + 1e-5) ; y = z / max(|z|_2, 1e-12).
 * replaces model/methods/BCNN.py:13-27 (BilinearPooling.forward: view, transpose,
 * bmm, div, add, sqrt, F.normalize) and its autograd backward.
 *   x        [B, C, HW]   input feature map (NCHW contiguous, HW = H*W)
 *   y        [B, C*C]     output
 *   inv_norm [B]          1/max(|z|_2,1e-12)            (saved for backward)
 *   colsum   [B, HW]      sum_c x[b,c,hw]               (saved for backward)
 */
size_t hk_bcnn_pool_ws_bytes(int B, int C, int HW);
int hk_bcnn_pool_fwd(const float* x, float* y, float* inv_norm, float* colsum, int B, int C, int HW,
                     void* ws, size_t ws_bytes, hk_stream_t stream);
/* dx [B, C, HW] = d loss / d x given dy [B, C*C] */
int hk_bcnn_pool_bwd(const float* x, const float* y, const float* dy, const float* inv_norm,
                     const float* colsum, float* dx, int B, int C, int HW, void* ws, size_t ws_bytes,
                     hk_stream_t stream);
/* The same backward when dy is the input gradient of a linear layer ON y (BCNN.py:54: logits = classifier(y)), i.e.
 * dy = g W: the inner product t = <y, dy> the normalisation's backward needs (autograd of F.normalize, BCNN.py:26) is then
 *     t[b] = sum_k ta[b,k] * (tb[b,k] - tc[k])          ta = g [B,K], tb = logits [B,K], tc = bias [K] (nullable)
 * - B*K multiply-adds instead of a pass over the 2 x 4 C^2 bytes per image of y and dy: one launch, the rank-1 term applied
 * while dX is written.  Any (ta, tb, tc) with that property may be passed; with another it computes dX for THAT t. */
int hk_bcnn_pool_bwd_tdot(const float* x, const float* y, const float* dy, const float* inv_norm,
                          const float* colsum, const float* ta, const float* tb, const float* tc, int K,
                          float* dx, int B, int C, int HW, void* ws, size_t ws_bytes, hk_stream_t stream);

/* Signed-sqrt variant - the second normalisation the reference keeps (commented out) next to the one it runs:
 *     G = X X^T / HW ; u = sign(G) sqrt(|G| + 1e-10) ; y = u / max(|u|_2, 1e-12)
 * replaces model/methods/BCNN.py:18,23-26 (the `sign * sqrt(abs + 1e-10)` alternative; = CBCNN.py:132 on the 512x512
 * Gram) and its autograd backward.  Raw Gram on the panel kernel, then two elementwise passes (the norm of u has no
 * closed form from column sums); backward: t = <y, dy> partials, then the same GEMM-shaped launch as the default
 * variant with P = (dy + dy^T - 2 t y) / |y| * inv_norm^2 / (2 HW), zero where y = 0 (torch: sign'(0) = abs'(0) = 0).
 *   inv_norm [B]  saved for backward ;  ws: hk_bcnn_ssqrt_ws_bytes */
size_t hk_bcnn_ssqrt_ws_bytes(int B, int C, int HW);
int hk_bcnn_ssqrt_pool_fwd(const float* x, float* y, float* inv_norm, int B, int C, int HW, void* ws, size_t ws_bytes,
                           hk_stream_t stream);
int hk_bcnn_ssqrt_pool_bwd(const float* x, const float* y, const float* dy, const float* inv_norm, float* dx, int B, int C,
                           int HW, void* ws, size_t ws_bytes, hk_stream_t stream);
/* The same pooling with the l2 scale left to the consumer ("fold the per-sample 1 / |z| into the classifier's
 * epilogue"): _fwd_unscaled writes u = sign(G) sqrt(|G| + 1e-10) and inv_norm = 1 / max(|u|_2, 1e-12) - the scale pass over
 * the 4 C^2 bytes per image is not launched; hk_linear_fwd_scaled(u, ..., row_scale = inv_norm) applies it to the
 * [B, classes] logits.  _bwd_unscaled takes that u and dy = dL/d(inv_norm u) (what hk_linear_bwd_scaled returns).
 * (The default sqrt(G + 1e-5) variant needs none of this: its norm is known before the Gram is formed.) */
int hk_bcnn_ssqrt_pool_fwd_unscaled(const float* x, float* u, float* inv_norm, int B, int C, int HW, void* ws, size_t ws_bytes,
                                    hk_stream_t stream);
int hk_bcnn_ssqrt_pool_bwd_unscaled(const float* x, const float* u, const float* dy, const float* inv_norm, float* dx, int B,
                                    int C, int HW, void* ws, size_t ws_bytes, hk_stream_t stream);
/* ... and with the norm left entirely to the consumer: ONE launch that writes u and *nparts partial sums of u^2 per image
 * (ss_part [B, 64]); hk_linear_fwd_ssq (below) adds them up in its reduce launch, applies 1 / |u| to the logits and writes
 * inv_norm.  HK_ERR_UNSUPPORTED (nothing launched) outside the panel kernel's shapes. */
int hk_bcnn_ssqrt_pool_fwd_parts(const float* x, float* u, float* ss_part, int* nparts, int B, int C, int HW, hk_stream_t stream);
/* Either backward when dy = g W comes from a linear layer on the normalised pooled vector: the inner product <y, dy> of the
 * l2-normalisation's backward is then t[b] = sum_k ta[b,k] (tb[b,k] - tc[k]) (ta = g, tb = logits, tc = bias, nullable) and the
 * pass over y and dy that adds it up (2 x 4 C^2 bytes per image) is not launched.  unscaled = 0: y as hk_bcnn_ssqrt_pool_fwd
 * returned it; != 0: the u / dy of the _unscaled pair. */
int hk_bcnn_ssqrt_pool_bwd_tdot(const float* x, const float* y, const float* dy, const float* inv_norm, const float* ta,
                                const float* tb, const float* tc, int K, int unscaled, float* dx, int B, int C, int HW,
                                void* ws, size_t ws_bytes, hk_stream_t stream);

/* The two stages of each direction, individually callable (hk_bcnn_pool_fwd = colsum_norm + gram_norm,
 * hk_bcnn_pool_bwd = bwd_gemm + bwd_rank1); bench.py times them separately.
 *   tpart [B, ceil(C/64)] partial sums of <y,dy> written by bwd_gemm, consumed by bwd_rank1 */
/* ws (optional, hk_bcnn_pool_ws_bytes): enables the two-stage column sum (B*C/64 workgroups instead of B) */
int hk_bcnn_colsum_norm(const float* x, float* colsum, float* inv_norm, int B, int C, int HW, void* ws,
                        size_t ws_bytes, hk_stream_t stream);
int hk_bcnn_gram_norm(const float* x, const float* inv_norm, float* y, int B, int C, int HW, hk_stream_t stream);
int hk_bcnn_bwd_gemm(const float* x, const float* y, const float* dy, const float* inv_norm, float* dx,
                     float* tpart, int B, int C, int HW, hk_stream_t stream);
int hk_bcnn_bwd_rank1(float* dx, const float* tpart, const float* inv_norm, const float* colsum, int B, int C,
                      int HW, hk_stream_t stream);

/* ------------------------------------------------------------ Fast MPN-COV ----
 * Covariance pooling  cov = (1/M) (X - mu 1^T) X^T.
 * replaces model/methods/MPNCOV.py:105-119 (Covpool.forward) / :121-134 (backward).
 *   x [B, C, M] ; cov [B, C, C] ; mu [B, C] channel means (saved for backward)
 */
int hk_cov_pool_fwd(const float* x, float* cov, float* mu, int B, int C, int M, hk_stream_t stream);
int hk_cov_pool_bwd(const float* x, const float* mu, const float* dcov, float* dx, int B, int C, int M,
                    hk_stream_t stream);

/* Newton-Schulz matrix square root, iter_n coupled iterations with trace
 * pre-normalisation and sqrt(trace) post-compensation.
 * replaces model/methods/MPNCOV.py:137-164 (Sqrtm.forward) / :166-202 (backward,
 * including its per-sample python loop :198-201).
 *   a      [B, d, d]  input (covariance)
 *   out    [B, d, d]  sqrtm
 *   norm_a [B]        trace(a)                                  (saved)
 *   ysave, zsave [B, max(iter_n-1,1), d, d]  Y_i / Z_i iterates (saved; unused when iter_n < 2)
 */
size_t hk_ns_sqrtm_ws_bytes(int B, int d, int iter_n, int backward);
int hk_ns_sqrtm_fwd(const float* a, float* out, float* norm_a, float* ysave, float* zsave, int B, int d,
                    int iter_n, void* ws, size_t ws_bytes, hk_stream_t stream);
/* hk_ns_sqrtm_fwd_sym: the same, for a SYMMETRIC input `a` (what Covpool produces and all MPNCOV.forward ever feeds
 * Sqrtm, MPNCOV.py:73-76).  Every product of the chain is then a symmetric matrix (a polynomial in `a`): the launches
 * compute only the tiles that touch the 128 x 128 blocks on or above the diagonal (3 of 4 at d = 256) and write the
 * blocks right of the diagonal twice, transposed.  Results differ from hk_ns_sqrtm_fwd by the rounding asymmetry of a
 * product of commuting symmetric matrices (~1e-7 relative); with an input that is not symmetric the result is WRONG -
 * use hk_ns_sqrtm_fwd.  d % 128 != 0 falls back to the full schedule.  Saved iterates are interchangeable with
 * hk_ns_sqrtm_fwd's for either backward. */
int hk_ns_sqrtm_fwd_sym(const float* a, float* out, float* norm_a, float* ysave, float* zsave, int B, int d,
                        int iter_n, void* ws, size_t ws_bytes, hk_stream_t stream);
/* Sqrtm followed by Triuvec, as MPNCOV.forward applies them (MPNCOV.py:88-92): the chain's last product writes `out`
 * [B, d, d] (saved for the backward) AND its row-major packed upper triangle tv [B, d (d + 1) / 2] - what
 * hk_triu_vec_fwd(out) would return, bit for bit, without the extra pass.  symmetric != 0: the schedule of
 * hk_ns_sqrtm_fwd_sym (input must be symmetric), else that of hk_ns_sqrtm_fwd.  Backward: hk_triu_vec_bwd then
 * hk_ns_sqrtm_bwd. */
int hk_ns_sqrtm_triu_fwd(const float* a, float* out, float* tv, float* norm_a, float* ysave, float* zsave, int B, int d,
                         int iter_n, int symmetric, void* ws, size_t ws_bytes, hk_stream_t stream);
/* hk_ns_sqrtm_bwd executes 34 products instead of the reference's 38: Z_i Y_i is taken from the accumulator of Y_i Z_i.
 * That is exact for ANY input `a`, symmetric or not: every iterate is a polynomial in the one matrix A = a / tr(a)
 * (Y_0 = A (3I - A)/2, Z_0 = (3I - A)/2, and each step multiplies polynomials in A), and polynomials in one matrix
 * commute.  Measured on non-symmetric inputs: 6.7e-7 from the reference, the same as the 38-product form
 * (tests/test_gpu_kernels.py::test_ns_general_input_backward).  The upstream gradient `dout` may be anything.
 * hk_ns_sqrtm_bwd_general: same arguments, Z_i Y_i as its own product - all 38 of MPNCOV.py:174-194, literally; kept
 * for A/B checks (it costs four more products). */
int hk_ns_sqrtm_bwd(const float* a, const float* out, const float* norm_a, const float* ysave,
                    const float* zsave, const float* dout, float* da, int B, int d, int iter_n, void* ws,
                    size_t ws_bytes, hk_stream_t stream);
int hk_ns_sqrtm_bwd_general(const float* a, const float* out, const float* norm_a, const float* ysave,
                            const float* zsave, const float* dout, float* da, int B, int d, int iter_n, void* ws,
                            size_t ws_bytes, hk_stream_t stream);

/* Row-major upper-triangle (incl. diagonal) vectorisation, index in closed form.
 * replaces model/methods/MPNCOV.py:205-218 (Triuvec.forward; index built on the
 * CPU every call :213-214) / :220-230 (backward).
 *   x [B, d, d] -> y [B, d(d+1)/2]
 */
int hk_triu_vec_fwd(const float* x, float* y, int B, int d, hk_stream_t stream);
int hk_triu_vec_bwd(const float* dy, float* dx, int B, int d, hk_stream_t stream);

/* ------------------------------------------------------- Compact bilinear ----
 * Tensor-sketch compact bilinear pooling, evaluated through the exact identity
 *   c[b,k] = sum_{i,j : (h1[i]+h2[j]) mod D = k} s1[i] s2[j] (X X^T)[b,i,j]
 * (= ifft(fft(S1^T x) * fft(S2^T x)) summed over positions), then
 *   u = sign(c) sqrt(|c| + 1e-10) ; y = u / max(|u|_2, 1e-12).
 * replaces model/methods/CBCNN.py:96-135 (CompactBilinearPooling.forward) and its
 * autograd backward.
 * hk_cbp_plan_build: host+device one-time setup from the count-sketch hashes
 * (CBCNN.py:76-91): writes the hash tables, a CSR "bin -> (i*C+j, sign)" table and
 * (C % 64 == 0) the sorted per-tile gather lists of the fused forward into `plan`
 * (device memory, hk_cbp_plan_bytes(C, D) bytes).  h1,h2 int32 [C] in [0,D);
 * s1,s2 fp32 [C] of +-1; all four are HOST pointers.  The library remembers, per
 * plan ADDRESS, whether the fused lists could be built for these hashes (a host-side
 * directory under a mutex - the blob itself is device memory); hk_cbp_fwd on a blob
 * the caller has copied elsewhere takes the unfused kernels: same results to rounding.
 * hk_cbp_plan_destroy(plan) forgets the address: call it BEFORE freeing or reusing the
 * plan's memory, so that another blob placed there later cannot inherit the entry.
 * hk_cbp_fwd: Gram and binning in one kernel (the Gram matrix never reaches HBM),
 * then two small finishing launches; ws holds the per-workgroup partial bin vectors.
 */
size_t hk_cbp_plan_bytes(int C, int D);
int hk_cbp_plan_destroy(const void* plan);
int hk_cbp_plan_build(const int32_t* h1, const float* s1, const int32_t* h2, const float* s2, int C, int D,
                      void* plan, hk_stream_t stream);
size_t hk_cbp_ws_bytes(int B, int C, int HW, int D);
/* x [B,C,HW] -> y [B,D]; c_raw [B,D] pre-normalisation sums and inv_norm [B] saved for backward */
int hk_cbp_fwd(const float* x, const void* plan, float* y, float* c_raw, float* inv_norm, int B, int C, int HW,
               int D, void* ws, size_t ws_bytes, hk_stream_t stream);
int hk_cbp_bwd(const float* x, const void* plan, const float* y, const float* c_raw, const float* inv_norm,
               const float* dy, float* dx, int B, int C, int HW, int D, void* ws, size_t ws_bytes,
               hk_stream_t stream);

/* The two forms of CompactBilinearPooling.forward Hawkeye's own CBCNN never takes (it calls it with one input and
 * sum_pool = True, CBCNN.py:23,33): two DIFFERENT inputs (CBCNN.py:96-102) and sum_pool = False (:127-130).  Same identity:
 *   hk_cbp_bin_matrix    c_raw[b,k]  = sum_{(i,j) -> k} s1_i s2_j G[b,i,j]      for a given G [B,C,C] (the cross Gram X1 X2^T:
 *                                                                              hk_bgemm_f32) - the plan's CSR gather
 *   hk_cbp_unbin_matrix  dG[b,i,j]   = s1_i s2_j dc[b, (h1_i + h2_j) mod D]     its transpose (then dX1 = dG X2, dX2 = dG^T X1)
 *   hk_cbp_loc_fwd       c[b,p,k]    = sum_{(i,j) -> k} s1_i s2_j x1[b,i,p] x2[b,j,p]     c [B,HW,D], no sum over the map
 *   hk_cbp_loc_bwd       dx1, dx2 [B,C,HW] from dc [B,HW,D] (either output nullable)
 * The signed square root and F.normalize (CBCNN.py:132-133) are the caller's; C (= both input widths) <= 1024 for _loc_. */
int hk_cbp_bin_matrix(const float* G, const void* plan, float* c_raw, int B, int C, int D, hk_stream_t stream);
int hk_cbp_unbin_matrix(const float* dc, const void* plan, float* dG, int B, int C, int D, hk_stream_t stream);
int hk_cbp_loc_fwd(const float* x1, const float* x2, const void* plan, float* c, int B, int C, int HW, int D,
                   hk_stream_t stream);
int hk_cbp_loc_bwd(const float* x1, const float* x2, const float* dc, const void* plan, float* dx1, float* dx2, int B,
                   int C, int HW, int D, hk_stream_t stream);

/* input_dim1 != input_dim2: CompactBilinearPooling(C1, C2, D) (model/methods/CBCNN.py:68-94 builds one sketch matrix per
 * input width; :104-105 asserts bottom1 has C1 and bottom2 C2 channels).  The same identity over the C1 x C2 cross Gram
 * G = X1 X2^T (hk_bgemm_f32):   c[b,k] = sum_{(i,j) : (h1[i] + h2[j]) mod D = k} s1[i] s2[j] G[b,i,j].
 *   hk_cbp_rect_plan_build   hashes h1 [C1], h2 [C2] in [0,D), signs s1 [C1], s2 [C2] (HOST pointers) -> `plan` (DEVICE memory,
 *                            hk_cbp_rect_plan_bytes(C1, C2, D) bytes: the hashes, the signs and the CSR table bin -> (i*C2+j, sign))
 *   hk_cbp_rect_bin_matrix   G [B,C1,C2] -> c_raw [B,D]            hk_cbp_rect_unbin_matrix   dc [B,D] -> dG [B,C1,C2]
 *                            (then dX1 = dG X2, dX2 = dG^T X1)
 *   hk_cbp_rect_loc_fwd/bwd  sum_pool = False: x1 [B,C1,HW], x2 [B,C2,HW] <-> c [B,HW,D]   (C1, C2 <= 1024; either gradient nullable)
 * A rect plan holds no host-side state (nothing to destroy).  The signed square root and F.normalize stay the caller's. */
size_t hk_cbp_rect_plan_bytes(int C1, int C2, int D);
int hk_cbp_rect_plan_build(const int32_t* h1, const float* s1, int C1, const int32_t* h2, const float* s2, int C2, int D,
                           void* plan, hk_stream_t stream);
int hk_cbp_rect_bin_matrix(const float* G, const void* plan, float* c_raw, int B, int C1, int C2, int D, hk_stream_t stream);
int hk_cbp_rect_unbin_matrix(const float* dc, const void* plan, float* dG, int B, int C1, int C2, int D, hk_stream_t stream);
int hk_cbp_rect_loc_fwd(const float* x1, const float* x2, const void* plan, float* c, int B, int C1, int C2, int HW, int D,
                        hk_stream_t stream);
int hk_cbp_rect_loc_bwd(const float* x1, const float* x2, const float* dc, const void* plan, float* dx1, float* dx2, int B,
                        int C1, int C2, int HW, int D, hk_stream_t stream);

/* ------------------------------------------------------- VGG trunk epilogues ----
 * What PyTorch-ROCm runs around every convolution of the VGG-16 trunk (model/backbone/vgg.py:24-57: Conv2d + bias, ReLU,
 * five MaxPool2d(2, 2); BCNN.py:38-39 / CBCNN.py:22 take all of `features`) as one full-tensor pass per elementwise op -
 * bias add, ReLU, pool; pool backward, ReLU backward, bias-gradient reduction - fused to ONE pass per convolution and
 * direction.  channels_last tensors: x [rows = N H W][C], C % 4 == 0.  Replaces, with the same arithmetic,
 *   hk_bias_relu_fwd        x = max(x + bias, 0) IN PLACE                 (nn.Conv2d's bias add + nn.ReLU(inplace=True)); `mask`
 *                           (nullable, [rows][C/4] bytes): bit t of a byte = channel t of the quad came out positive
 *   hk_bias_relu_bwd        dx = dy where y > 0 else 0 ; dbias = sum dx   (threshold_backward + the convolution's bias gradient;
 *                           dx may alias dy); the sign from `mask` when given (1/16 of the bytes of y), else from y
 *   hk_bias_relu_pool_fwd   p = maxpool2x2(max(x + bias, 0)), argmax [N][H/2][W/2][C/4] bytes (2 bits per channel: window position
 *                           2 dh + dw of the FIRST maximum, ATen's tie rule); the full-resolution activation is not written
 *   hk_bias_relu_pool_bwd   dx [N][H][W][C] = dp routed to the argmax where p > 0, zeros elsewhere ; dbias = sum dx
 * The backward kernels need C / 4 to divide 256 (C = 64 .. 512 in VGG) and a workspace of hk_trunk_ws_bytes(C) bytes (partial
 * column sums, added in a fixed order: deterministic).  HK_ERR_UNSUPPORTED (nothing launched) for other shapes / unaligned
 * pointers: the caller keeps the framework's own ops for those. */
size_t hk_trunk_ws_bytes(int C);
int hk_bias_relu_fwd(float* x, const float* bias, uint8_t* mask, long long rows, int C, hk_stream_t stream);
int hk_bias_relu_bwd(const float* dy, const float* y, const uint8_t* mask, float* dx, float* dbias, long long rows, int C, void* ws,
                     size_t ws_bytes, hk_stream_t stream);
int hk_bias_relu_pool_fwd(const float* x, const float* bias, float* p, uint8_t* argmax, int N, int H, int W, int C,
                          hk_stream_t stream);
int hk_bias_relu_pool_bwd(const float* dp, const float* p, const uint8_t* argmax, float* dx, float* dbias, int N, int H, int W,
                          int C, void* ws, size_t ws_bytes, hk_stream_t stream);
/* The end of a ResNet bottleneck, `out += identity; out = relu(out)` (model/backbone/resnet.py:89-136; the trunk of MPN, AP-CNN,
 * OSMENet, CIN): hk_add_relu_fwd  a = max(a + b, 0) IN PLACE on a, n elements of any dense layout the two share, n % 4 == 0;
 * hk_relu_mask_bwd  g = dy where y > 0 else 0 - the gradient of both operands. */
/* The trunk's FIRST convolution with its epilogue, Conv2d(Cin <= 3, 64, 3, padding=1) + bias + ReLU (model/backbone/vgg.py:24-57, layer 0
 * of `features`), as one kernel per direction: the layer is the write of its 3.29 GB output / the read of that output's gradient.
 *   hk_conv1_bias_relu_fwd   x [N][H][W][Cin] (channels_last), wt [9 Cin][64] = the layer's weight permuted to tap-major
 *                            (tap = (kh * 3 + kw) * Cin + c), bias [64] -> y [N][H][W][64] = max(conv + bias, 0), mask (nullable,
 *                            [N][H][W][16] bytes: the sign bits hk_bias_relu_fwd writes)
 *   hk_conv1_bias_relu_bwd   dy, mask, x -> dwt [9 Cin][64] (same layout as wt), dbias [64]; no input gradient (images need none);
 *                            workspace hk_conv1_ws_bytes(Cin) bytes; the masked gradient map is never written
 * Replaces the library convolution + hk_bias_relu_fwd and hk_bias_relu_bwd + the library's weight gradient for that layer. */
size_t hk_conv1_ws_bytes(int Cin);
int hk_conv1_bias_relu_fwd(const float* x, const float* wt, const float* bias, float* y, uint8_t* mask, int N, int H, int W, int Cin,
                           int Cout, hk_stream_t stream);
int hk_conv1_bias_relu_bwd(const float* dy, const uint8_t* mask, const float* x, float* dwt, float* dbias, int N, int H, int W, int Cin,
                           int Cout, void* ws, size_t ws_bytes, hk_stream_t stream);
int hk_add_relu_fwd(float* a, const float* b, long long n, hk_stream_t stream);
int hk_relu_mask_bwd(const float* dy, const float* y, float* g, long long n, hk_stream_t stream);

/* ------------------------------------------------------------------ AP-CNN ----
 * Attention pooling.  The reference materialises A = a_s*F + a_c*F and only ever
 * consumes its global average (cls3/4/5 start with AdaptiveAvgPool2d(1)), so
 *   gap [b,c] = mean_hw F[b,c,hw]              (also ChannelGate / Concate input)
 *   sgap[b,c] = mean_hw a_s[b,hw] F[b,c,hw]
 * and pooled = sgap + a_c * gap.  One pass over F.
 * replaces model/methods/APCNN.py:256,261,266 + :377-405 first layer + :533-538.
 *   f [B,C,HW] ; a_s [B,HW] (nullable: gap only) ; gap, sgap [B,C]
 */
int hk_att_pool_fwd(const float* f, const float* a_s, float* gap, float* sgap, int B, int C, int HW,
                    hk_stream_t stream);
/* df [B,C,HW] = (dsgap*a_s + dgap)/HW ; da_s [B,HW] = sum_c dsgap*F / HW (nullable with a_s).
 * a_s == NULL (plain GAP backward, df = dgap / HW): dsgap must be NULL too and f is not read (may be NULL). */
int hk_att_pool_bwd(const float* f, const float* a_s, const float* dgap, const float* dsgap, float* df,
                    float* da_s, int B, int C, int HW, hk_stream_t stream);

/* The three pyramid levels of PyramidAttentions.forward (APCNN.py:256-266) in one launch per direction: the levels'
 * poolings are independent (only the channel gates chain), and at 28 x 28 / 14 x 14 a launch per level is launch-bound.
 *   f_l [B,C,HW_l] ; a_l [B,HW_l] ; gap, sgap [3][B][C] (level-major) ; dgap, dsgap likewise ; df_l, da_l like f_l, a_l
 * Same arithmetic per row / column as hk_att_pool_fwd / _bwd: bit-identical results. */
int hk_att_pool3_fwd(const float* f0, const float* f1, const float* f2, const float* a0, const float* a1, const float* a2,
                     float* gap, float* sgap, int B, int C, int HW0, int HW1, int HW2, hk_stream_t stream);
int hk_att_pool3_bwd(const float* f0, const float* f1, const float* f2, const float* a0, const float* a1, const float* a2,
                     const float* dgap, const float* dsgap, float* df0, float* df1, float* df2, float* da0, float* da1,
                     float* da2, int B, int C, int HW0, int HW1, int HW2, hk_stream_t stream);

/* Attention -> ROI: border mask, one square anchor per cell, keep score > mean,
 * greedy NMS (IoU < thr keeps, area without +1, highest score first; ties:
 * highest cell index first), top-k, clamp to the image.  One workgroup per image,
 * no host synchronisation.
 * replaces model/methods/APCNN.py:444-476 (get_att_roi) + model/methods/nms.py:4-93.
 *   att   [B, h*w]      spatial attention (sigmoid output)
 *   rois  [B, topk, 5]  x1,y1,x2,y2,score (rows >= count are zero)
 *   count [B] int32     number of valid rows
 *   keep_r0..keep_c1: rows [r0,r1) x cols [c0,c1) that survive the border mask
 *       (= int(0.2*h), int(0.8*h), ... for 200 classes, 0.1/0.9 otherwise; the
 *       caller evaluates the reference's python expression, APCNN.py:451-455)
 */
int hk_att_roi_select(const float* att, float* rois, int32_t* count, int B, int h, int w,
                      int feature_stride, float anchor_size, int img_h, int img_w, int keep_r0, int keep_r1,
                      int keep_c0, int keep_c1, float iou_thr, int topk, hk_stream_t stream);

/* The three pyramid levels of one forward in ONE launch (grid B x 3): the reference calls get_att_roi once per level
 * (APCNN.py:256-266), each call a chain of top-k dependent rounds on one workgroup per image - side by side they take
 * as long as the longest.  Arrays of three (HOST arrays; the pointers in them are device pointers): att[i], rois[i],
 * count[i], h[i], w[i], feature_stride[i], anchor_size[i], topk[i]; keep[4 i .. 4 i + 3] = keep_r0, keep_r1, keep_c0,
 * keep_c1 of level i.  Results are those of three hk_att_roi_select calls, bit for bit. */
int hk_att_roi_select3(const float* const* att, float* const* rois, int32_t* const* count, int B, const int* h,
                       const int* w, const int* feature_stride, const float* anchor_size, int img_h, int img_w,
                       const int* keep, float iou_thr, const int* topk, hk_stream_t stream);

/* ROI-guided zoom-in (+ drop block): crop the union box of an image's ROIs,
 * zero one dropped ROI (training), rescale by c*h*w/sum(mask), bilinear resize
 * (align_corners=False) back to H x W.
 * replaces model/methods/APCNN.py:478-531 (get_roi_crop_feat).
 *   x    [B, C, H, W]
 *   box  [B, 4] fp32   x1,y1,x2,y2 of the union box in feature coordinates (already /scale)
 *   drop [B, 4] fp32   box to zero, or x2<=x1 for "no drop" (ignored when training == 0)
 *   y    [B, C, H, W]
 * Integer truncation of the box corners and the scale-rate rule follow the reference.
 */
int hk_roi_crop_resize_fwd(const float* x, const float* box, const float* drop, float* y, int B, int C, int H,
                           int W, int training, hk_stream_t stream);
int hk_roi_crop_resize_bwd(const float* dy, const float* box, const float* drop, float* dx, int B, int C, int H,
                           int W, int training, hk_stream_t stream);
/* Union / drop boxes from the fixed-layout ROI tables of hk_att_roi_select, on
 * device (no host read of the counts).
 *   rois_l [B,k_l,5], cnt_l [B] for the three pyramid levels; scale = 8 (coordinates are divided by it);
 *   u01 [B,2] uniforms in [0,1): u[.,0] picks the branch (<0.3 level-3 drop,
 *   <0.6 level-4 drop, else none), u[.,1] the ROI index floor(u*n).
 */
int hk_roi_boxes(const float* rois3, const int32_t* cnt3, int k3, const float* rois4, const int32_t* cnt4, int k4,
                 const float* rois5, const int32_t* cnt5, int k5, const float* u01, float scale, float* box,
                 float* drop, int B, hk_stream_t stream);

/* -------------------------------------------------------------------- OSME ----
 * One-squeeze multi-excitation: z = GAP(x) (one pass), and per attention p the
 * channel re-scaling s_p = m_p (.) x written for the following FC.
 * replaces model/methods/OSME.py:19-24 (avg_pool / broadcast mul).
 *   x [N,C,HW] ; z [N,C] ; m [P,N,C] sigmoid gates ; s [P,N,C,HW]
 */
int hk_osme_gap(const float* x, float* z, int N, int C, int HW, hk_stream_t stream);
int hk_osme_scale_fwd(const float* x, const float* m, float* s, int P, int N, int C, int HW, hk_stream_t stream);
/* dx [N,C,HW] = sum_p m_p * ds_p (+ dz/HW if dz != NULL) ; dm [P,N,C] = sum_hw ds_p * x */
int hk_osme_scale_bwd(const float* x, const float* m, const float* ds, const float* dz, float* dx, float* dm,
                      int P, int N, int C, int HW, hk_stream_t stream);

/* ------------------------------------------------------- classifier (8f-1) ----
 * out = y W^T + bias for the wide pooled vector and its backward.  Forward: the
 * feature axis is cut into slabs, partial results are added in slab order
 * (deterministic); wide classifiers (J % 32 == 0, enough work) stream y and W once
 * through LDS-DMA, other shapes take the generic f32-MFMA tiles.  Backward: streaming
 * kernels for up to 64 samples x 208 classes from 65536 features, plain tiles otherwise.
 * replaces nn.Linear at model/methods/BCNN.py:42,54 ; CBCNN.py:26,34 ; MPNCOV.py:31 ;
 * OSME.py:34,43 (same operand layouts: W is [K][J] as in nn.Linear.weight).
 *   y [B,J] ; w [K,J] ; bias [K] or NULL ; out [B,K] ; ws: hk_linear_ws_bytes(B,J,K)
 *   g [B,K] = dL/dout ; dy [B,J], dw [K,J], db [K]: each may be NULL (skipped)
 */
size_t hk_linear_ws_bytes(int B, int J, int K);
int hk_linear_fwd(const float* y, const float* w, const float* bias, float* out, int B, int J, int K, void* ws,
                  size_t ws_bytes, hk_stream_t stream);
int hk_linear_bwd(const float* y, const float* w, const float* g, float* dy, float* dw, float* db, int B, int J, int K,
                  hk_stream_t stream);
/* With a per-sample scale folded in (row_scale [B], e.g. the 1 / |z| of a pooled vector handed over unnormalised):
 *   out = row_scale[b] * (y W^T) + bias ;  dy = g W  (= dL/d(row_scale y)) ;  dW = (row_scale g)^T y ;  db = sum_b g.
 * hk_linear_bwd_scaled is served by the one-launch kernel only (up to 64 samples, up to 208 outputs, J % 64 == 0,
 * J >= 16384); HK_ERR_UNSUPPORTED - nothing launched - otherwise. */
int hk_linear_fwd_scaled(const float* y, const float* w, const float* bias, const float* row_scale, float* out, int B, int J,
                         int K, void* ws, size_t ws_bytes, hk_stream_t stream);
int hk_linear_bwd_scaled(const float* y, const float* w, const float* g, const float* row_scale, float* dy, float* dw, float* db,
                         int B, int J, int K, hk_stream_t stream);
/* hk_linear_fwd_scaled with the scale formed in the reduce launch: 1 / max(sqrt(sum of ss_part[b, 0 .. nparts)), 1e-12), written
 * to inv_norm [B] (the partial sums of |u|^2 of hk_bcnn_ssqrt_pool_fwd_parts: no launch in between). */
int hk_linear_fwd_ssq(const float* u, const float* w, const float* bias, const float* ss_part, int nparts, float* inv_norm,
                      float* out, int B, int J, int K, void* ws, size_t ws_bytes, hk_stream_t stream);

/* ------------------------------------------------------ MAMC n-pairs loss (8f-4) ----
 * loss = NPairsLoss(parts, targets) and dx = d loss / d parts in one call.
 * replaces model/loss/MAMC_loss.py:34-90 (python loop over the b*p anchors).
 *   x [b*p, D] part features (row i = sample i / p, attention i % p) ; labels int32 [b]
 *   loss [1] ; dx [b*p, D] ; ws: hk_npairs_ws_bytes(b*p, D)
 */
size_t hk_npairs_ws_bytes(int n, int D);
int hk_npairs_loss(const float* x, const int32_t* labels, float* loss, float* dx, int b, int p, int D, void* ws,
                   size_t ws_bytes, hk_stream_t stream);

/* ------------------------------------------------ CIN channel interaction (8f-2) ----
 * SCI: W = softmax_rows(-X X^T / HW), Y = W X ; CCI: Yc[b] = |W[b] - w_b W[(b + B/2) % B]| X[b].
 * replaces the bmm / softmax / abs / bmm parts of ChannelInteractionModule.forward,
 * model/methods/CIN.py:24-60 (conv 3x3, residual and the 1-output fc stay on PyTorch).
 *   x, y, dy, dx [B,C,HW] ; w, dw [B,C,C] ; wt, dwt [B] ; B even for the CCI pair.
 *   hk_cin_sci_bwd: dwbuf [B,C,C] scratch; holds the gradient reaching W from the CCI
 *   branch on entry when has_extra != 0 (it is overwritten).
 *   hk_cin_sci_fwd, C % 64 == 0: ONE kernel at 7x7 / 8x8 / 6x6 maps (scores recomputed on the matrix pipe, W written
 *   once); at 14x14 / 12x12 / 10x10 maps and C % 128 == 0 three - Gram panel kernel, row statistics, softmax applied on the way into the
 *   second product (W written once, in place of the scores); any other shape: Gram, row softmax, product on the generic tile.
 *   hk_cin_sci_bwd at those larger maps: W^T dY and (dG + dG^T) X / HW each stream their C x C operand once through the
 *   forward's pipeline (the second as ONE product).
 */
int hk_cin_sci_fwd(const float* x, float* w, float* y, int B, int C, int HW, hk_stream_t stream);
int hk_cin_sci_bwd(const float* x, const float* w, const float* dy, float* dwbuf, int has_extra, float* dx, int B, int C,
                   int HW, hk_stream_t stream);
int hk_cin_cci_fwd(const float* x, const float* w, const float* wt, float* y, int B, int C, int HW, hk_stream_t stream);
size_t hk_cin_cci_ws_bytes(int B, int C);
int hk_cin_cci_bwd(const float* x, const float* w, const float* wt, const float* dy, float* dx, float* dw, float* dwt,
                   int B, int C, int HW, void* ws, size_t ws_bytes, hk_stream_t stream);

/* ------------------------------------------------------ input finalisation (8f-3) ----
 * uint8 HWC crops -> normalised fp32 images on the device, with the random-erasing
 * rectangle applied: PILToTensor + ConvertImageDtype + Normalize + RandomErasing(value 0)
 * of dataset/transforms.py:38-46 in one pass; bit-identical to the CPU order of operations.
 *   u8 [B,H,W,3] (device) ; mean3 / std3: HOST pointers to three floats ;
 *   erase int32 [B,4] = top, left, h, w (device; h or w <= 0: nothing erased; NULL: none)
 *   out fp32 [B,3,H,W], or the channels_last storage of the same tensor ([B,H,W,3]) if channels_last != 0
 */
int hk_image_finalize(const uint8_t* u8, const float* mean3, const float* std3, const int32_t* erase, float* out, int B,
                      int H, int W, int channels_last, hk_stream_t stream);

/* ------------------------------------------------------- generic primitive ----
 * Batched fp32 GEMM on the f32 MFMA path (exact fp32 fma chain):
 *   C[b] = alpha * op(A[b]) op(B[b]) + beta * C[b] + diag * I
 * row-major operands; trans_a: A stored K x M; trans_b: B stored N x K.
 * Exposed for the tests and for composing the Newton-Schulz chain.
 */
int hk_bgemm_f32(const float* a, int lda, long long stride_a, int trans_a, const float* b, int ldb,
                 long long stride_b, int trans_b, float* c, int ldc, long long stride_c, int M, int N, int K,
                 int batch, float alpha, float beta, float diag, hk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HAWKEYE_HIP_H */
