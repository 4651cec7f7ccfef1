/*
 * bde_hip.h -- C ABI of libbde_hip.so: the MI355X (gfx950) posterior-update
 * kernels behind BayesianOptimizer.step()/complete_epoch()/sample_parameters().
 *
 * The reference (Feuermagier/Beyond_Deep_Ensembles) is pure Python over PyTorch
 * and has no FFI for this path; its "operator API" is the Python optimizer
 * surface in src/algos.  Each entry point below replaces the ATen op sequence
 * at the cited reference lines.  The Python shells in
 * beyond_deep_ensembles_amd/ bind these symbols with ctypes (INTEGRATION.md
 * shows the stub a reference maintainer would add).
 *
 * Conventions (all entry points):
 *   - plain device pointers + sizes, fp32 data, no torch types;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *     work is only ENQUEUED on it: no allocation, no free, no synchronisation,
 *     no host<->device copies -- every call is hipGraph-capturable;
 *   - return 0 on success, a negative hipError_t on a launch failure, or
 *     BDE_ERR_INVALID (-1) for a rejected argument (nothing is enqueued);
 *   - every vector pointer must be 16-byte aligned (bde_gauss_draw_fwd/bwd also
 *     accept unaligned operands and then run a scalar path); leading dimensions
 *     `ld` are in floats and must be multiples of 4 with ld >= D;
 *   - scratch memory is supplied by the caller (sizes from bde_*_ws_bytes);
 *   - results are deterministic run to run (two-stage reductions, no float
 *     atomics).
 */
#ifndef BDE_HIP_H
#define BDE_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BDE_ERR_INVALID (-1)
#define BDE_MAX_PARTICLES 64   /* SVGD: M <= 64; M <= 16 runs the single-tile fast path,
                                  17..64 a blocked generic path (several passes)        */
#define BDE_FAST_PARTICLES 16
#define BDE_GMAT_DOUBLES 257    /* bde_svgd_gram_finish: 16 x 16 Gram matrix (row-major, padded) + its padded M */
#define BDE_MAX_RANK 256       /* SWAG: deviation_samples K <= 256           */
#define BDE_MAX_BATCH 32       /* SWAG batched sampling: S <= 32 per call    */

/* Library/ABI version (major*10000 + minor*100 + patch). */
int bde_version(void);
/* gfx target the device code was compiled for, e.g. "gfx950". */
const char* bde_arch(void);
/* Load every code object of the library on the CURRENT device now.  HIP defers that to the first launch of a kernel;
 * call this once per device, from one thread, before communication threads (torch.distributed) start or other
 * processes share the device, so that no first launch coincides with them.  Idempotent.  (No reference counterpart:
 * PyTorch loads its kernels the same lazy way.) */
int bde_init(void);
/* bde_init() fails only for the code objects whose kernels have been verified on a device.  The ones that have not (no default
 * call launches them: the small-model SVGD kernel, the fused convolution kernels) are uploaded too, but a failure there is
 * only recorded here -- bit 0 svgd_small, bit 1 conv_lrt, bit 2 conv_lrt_bwd of the LAST bde_init(); 0 = all resident. */
int bde_init_optional_failures(void);

/* ------------------------------------------------------------------ SVGD --
 * One SVGD posterior update over M flattened particles P [M, ld] with
 * gradients G [M, ld] (D valid floats per row).
 *
 * Replaces src/algos/svgd.py:14-32 (rbf) and :83-89 (stack/cat, prior term,
 * phi) -- torch.cdist**2, torch.quantile, exp, kernel.sum, two matmuls and
 * ~6 elementwise passes -- by three launches:
 *   gram    : mean-centred Gram partials of P on the f32 MFMA (reads P once)
 *   kstats  : fixed-order fp64 reduction of the partials, d2, median
 *             bandwidth h (torch.quantile semantics, diagonal included),
 *             K = exp(-d2/(2h^2)), and the two M x M coefficient matrices
 *   combine : out = CG @ G + CP @ P in one streaming pass (reads P, G; writes
 *             out; out may alias G).
 */

/* Bytes of scratch `ws` needed by the SVGD entry points for M particles.  The first 1.5 KB are a
 * header; it must be ZERO-FILLED once after allocation (it holds the hand-off counters of the
 * single-launch path, which only ever count up).  One `ws` serves one stream at a time. */
size_t bde_svgd_ws_bytes(int M);

/* Number of floats of the `kstat` result block for M particles.  Layout:
 *   [0, M*M)        K        row-major kernel matrix      (svgd.py:21)
 *   [M*M, 2M*M)     d2       squared distances            (svgd.py:15)
 *   [2M*M, 2M*M+M)  rowsum   sum_j K_ij                   (svgd.py:23)
 *   next 4 floats   h, median(d2), s = kernel_grad_scale/(dataset_size*h^2), M
 *   next M*M        CG^T     coefficient of G, stored [j][i]
 *   next M*M        CP^T     coefficient of P, stored [j][i]
 */
size_t bde_svgd_kstat_floats(int M);

/* Stage 1: per-workgroup partial Gram matrices of the mean-centred particles
 * into ws.  P [M, ld]. */
int bde_svgd_gram(const float* P, int M, int64_t D, int64_t ld, void* ws, void* stream);
/* Tuning hook: bytes of the Gram walk's tail loaded cacheably (so that the combine pass finds them in the 256 MB
 * Infinity Cache); the head of the walk is loaded non-temporally.  Default 240 MB; 0 = all non-temporal; < 0 restores
 * the default.  Process-wide.  tools/gram_split_ab.py sweeps it at M = 5, 8 and 16 (profiles/r04_gram_split_*.txt). */
int bde_svgd_set_gram_keep_bytes(int64_t bytes);

/* Stage 2: reduce the partials, bandwidth, kernel, coefficients -> kstat.
 * phi = K @ (-(G + l2_reg/2 * P)) + kernel_grad_scale * gradK / dataset_size
 * (svgd.py:86-89); the coefficients produce `sign * phi` (sign = -1 gives the
 * gradient the reference hands to the base optimizer, svgd.py:95).
 * If h_override > 0 it replaces the median bandwidth (rbf's h_override).
 * mode 0: step coefficients (CG = -sign*K ...); mode 1: rbf coefficients, i.e.
 * CG = 0 and CP = (diag(rowsum) - K) / h^2 so that combine() yields grad_kernel
 * (svgd.py:23,31). */
int bde_svgd_kstats(const void* ws, int M, float l2_reg, float kernel_grad_scale,
                    float dataset_size, float sign, float h_override, int mode,
                    float* kstat, void* stream);

/* Dimension-sharded multi-GPU update (every rank owns a column slice of all M particles): after
 * bde_svgd_gram over the slice, gram_finish reduces the partials to gmat_out [BDE_GMAT_DOUBLES] (fp64, fixed
 * order); the ranks exchange those blocks (one tiny all-gather); kstats_gmat sums `n_mats` blocks (stride
 * `mat_stride` doubles) in order and evaluates the same statistics as bde_svgd_kstats.  M <= 16. */
int bde_svgd_gram_finish(const void* ws, int M, double* gmat_out, void* stream);
int bde_svgd_kstats_gmat(const double* gmats, int n_mats, int64_t mat_stride, int M, float l2_reg,
                         float kernel_grad_scale, float dataset_size, float sign, float h_override, int mode,
                         float* kstat, void* stream);

/* Stage 3: out[i, :] = sum_j CG[i][j] * G[j, :] + CP[i][j] * P[j, :].
 * P and out have leading dimension ld, G has ldg (0 = ld): with P / out offset to a column chunk and G a
 * staging buffer the multi-GPU exchange gathers into, the update runs chunk by chunk behind the collective.
 * G may be NULL (CG ignored).  out may alias G (not P) for M <= 16 when ldg == ld; for M > 16 the
 * rows are produced in chunks of 16 that re-read all of G, so out must not alias G (and ldg must equal ld). */
int bde_svgd_combine(const float* P, const float* G, float* out, int M, int64_t D,
                     int64_t ld, int64_t ldg, const float* kstat, void* stream);

/* One SVGD posterior update (svgd.py:86-89) on `stream`: the three stages back to back, at every size.  (Until ABI 405
 * this entry point chose the small-model kernel by itself when bde_svgd_small_supported(M, D); that choice now belongs
 * to the caller -- bde_svgd_step_small* below -- so that a default call never reaches a kernel that has not been
 * parity-tested on a device: beyond_deep_ensembles_amd/device_verified.py.) */
int bde_svgd_step(const float* P, const float* G, float* out, int M, int64_t D, int64_t ld,
                  float l2_reg, float kernel_grad_scale, float dataset_size, float sign,
                  void* ws, float* kstat, void* stream);

/* Small models (M <= 8 and D <= 524,288, e.g. the CIFAR ResNet-20 of the reference: D = 273,610): the update as TWO
 * launches of one kernel, one workgroup per CU -- (1) per-workgroup Gram partials of the columns a workgroup owns,
 * (2) every workgroup reduces all partials in the same fixed order, evaluates the kernel statistics redundantly and
 * combines its own columns (which the first launch left in L2).  12*M*D bytes of HBM traffic.  Same results as the
 * three-stage path up to the order of the partial sums.  mode / h_override as in bde_svgd_kstats (mode 1: G may
 * be NULL, out = grad_kernel).  bde_svgd_small_supported() answers for the CURRENT device (CUs x resident workgroups
 * per CU, queried once per device), so a partition of the chip (CPX mode) gets a smaller limit on D or the
 * three-stage path.  (Rounds 2-3 also offered both halves as one persistent launch with an in-kernel hand-off; it was
 * no faster once its wait was bounded and is gone.) */
int bde_svgd_small_supported(int M, int64_t D);
int bde_svgd_step_small(const float* P, const float* G, float* out, int M, int64_t D, int64_t ld,
                        float l2_reg, float kernel_grad_scale, float dataset_size, float sign,
                        float h_override, int mode, void* ws, float* kstat, void* stream);

/* The second launch continued through the M shared-state base-optimizer applications (svgd.py:92-103, semantics of
 * bde_svgd_fused_sgd / bde_svgd_fused_adam): the updated particles are written back over P -- the whole
 * SVGDOptimizer.step minus forward/backward in two launches for small models, (12*M + 8)*D bytes (SGD). */
int bde_svgd_step_small_sgd(float* P, const float* G, float* momentum_buf, int M, int64_t D, int64_t ld,
                            float l2_reg, float kernel_grad_scale, float dataset_size, double lr, double momentum,
                            double dampening, double weight_decay, int nesterov, int first, void* ws, float* kstat,
                            void* stream);
int bde_svgd_step_small_adam(float* P, const float* G, float* exp_avg, float* exp_avg_sq, int M, int64_t D, int64_t ld,
                             float l2_reg, float kernel_grad_scale, float dataset_size, double lr, double beta1,
                             double beta2, double eps, double weight_decay, int64_t step0, void* ws, float* kstat,
                             void* stream);

/* ---- gradients handed over WITHOUT a copy (the "_store_grads" clones of svgd.py:129-133 removed) ----
 * The flat row of a particle is the concatenation of its parameter tensors, each starting on a float4 boundary.
 * Segment s = tensor s = columns [col0_s, col0_s + numel_s) of the row.  seg_ptrs (device memory, n_seg * M
 * pointers) holds, for segment s and particle j, the address seg_ptrs[s * M + j] of that gradient tensor as autograd
 * produced it (fp32, contiguous, 16-byte aligned); the caller substitutes the address of the segment inside a flat
 * gradient row for gradients that are missing / unaligned / strided, after zeroing or copying there.  `chunks`
 * (device memory, static per layout) lists pieces of at most 256 float4 columns of ONE segment:
 *   c4    first float4 column of the piece in the flat row         loc4  the same, counted from the segment's start
 *   seg   segment index                                            nflt  valid floats in the piece (1..1024)
 * Columns of the row that no chunk covers (alignment padding) are left untouched; they hold zeros.
 * D (columns in use, padding included) must be a multiple of 4.  Same arithmetic per element as the flat-G entry
 * points, so results are bit-identical to copying the gradients into G [M, ld] first. */
typedef struct bde_seg_chunk {
  int64_t c4;
  int64_t loc4;
  int32_t seg;
  int32_t nflt;
  int64_t reserved;
} bde_seg_chunk;
/* bde_svgd_combine with segmented gradients; out [M, ld] (may be the flat gradient rows the substituted pointers
 * point into). */
int bde_svgd_combine_seg(const float* P, const void* const* seg_ptrs, const bde_seg_chunk* chunks, int64_t n_chunks,
                         float* out, int M, int64_t D, int64_t ld, const float* kstat, void* stream);
/* bde_svgd_fused_sgd / bde_svgd_fused_adam with segmented gradients (M <= 16; ws_next as there). */
int bde_svgd_fused_sgd_seg(float* P, const void* const* seg_ptrs, const bde_seg_chunk* chunks, int64_t n_chunks,
                           float* momentum_buf, int M, int64_t D, int64_t ld, const float* kstat, double lr,
                           double momentum, double dampening, double weight_decay, int nesterov, int first,
                           void* ws_next, void* stream);
int bde_svgd_fused_adam_seg(float* P, const void* const* seg_ptrs, const bde_seg_chunk* chunks, int64_t n_chunks,
                            float* exp_avg, float* exp_avg_sq, int M, int64_t D, int64_t ld, const float* kstat,
                            double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step0,
                            void* ws_next, void* stream);
/* Pack the segmented gradients of particles [row0, row0 + n_rows) into the flat rows G [M, ld] in ONE launch (what the
 * collective exchanges and the single-launch kernel read); pieces that already live there are skipped. */
int bde_svgd_gather_seg(const void* const* seg_ptrs, const bde_seg_chunk* chunks, int64_t n_chunks, float* G, int M,
                        int row0, int n_rows, int64_t ld, void* stream);
/* out[0] = ((*v[0] + *v[1]) + *v[2]) + ... in fp32, in this order: the loss a step returns (svgd.py:66,72 `total_loss +=
 * loss` per particle, :105) summed by ONE launch instead of one torch add per particle.  ``scalars`` is a HOST array of n
 * (1 <= n <= 64) device pointers to fp32 scalars; it is read during the call (the pointers travel in the kernel's
 * argument block), so it may be a temporary.  `out` may be one of the inputs. */
int bde_sum_scalars(const float* const* scalars, int n, float* out, void* stream);
/* The same sum "divided" by `divisor` (> 0) in the same launch: the value SVGDOptimizer.step returns, svgd.py:105
 * `total_loss / particle_count`, without a second launch (ABI 405).  Rounded as the reference rounds it ON A GPU: torch's
 * kernel for `tensor / python_number` multiplies by fl(1 / divisor), so this computes sum * fl(1.f / divisor) (ABI 406;
 * 405 divided in IEEE fp32, one ulp off for e.g. 5 particles).  With divisor = 1 it is bde_sum_scalars. */
int bde_mean_scalars(const float* const* scalars, int n, float divisor, float* out, void* stream);

/* Shared-state base-optimizer apply for the M particles, in particle order
 * (svgd.py:92-103 with ONE torch.optim.SGD / Adam whose state is keyed on the
 * model's parameters and therefore shared by all particles).
 * grad [M, ld] is the gradient handed to the optimizer (= -phi), P [M, ld] the
 * particles (updated in place), state buffers are [D]-sized (ld_state = ld).
 * sgd: torch.optim.SGD semantics (momentum, dampening, nesterov, weight_decay,
 * maximize=False); `first` != 0 means the momentum buffer is uninitialised
 * (torch initialises it with the first gradient it sees).
 * adam: torch.optim.Adam semantics (amsgrad=False, maximize=False), `step0` is
 * the shared step counter before this call (it advances by M).
 * Hyper-parameters are doubles: torch forms step_size = lr / (1 - beta1^t) etc.
 * in Python double arithmetic before rounding to fp32. */
int bde_svgd_apply_sgd(float* P, const float* grad, float* momentum_buf, int M, int64_t D, int64_t ld,
                       double lr, double momentum, double dampening, double weight_decay, int nesterov,
                       int first, void* stream);
int bde_svgd_apply_adam(float* P, const float* grad, float* exp_avg, float* exp_avg_sq, int M, int64_t D,
                        int64_t ld, double lr, double beta1, double beta2, double eps, double weight_decay,
                        int64_t step0, void* stream);

/* Posterior update + shared-state base-optimizer apply in ONE pass (svgd.py:86-103):
 * equivalent to bde_svgd_combine(P, G, tmp) followed by bde_svgd_apply_sgd/adam(P, tmp)
 * with the coefficients of `kstat` (from bde_svgd_kstats, sign = -1), but -phi is never
 * written: (12*M + 8)*D bytes instead of (24*M + 8)*D.  If ws_next != NULL (M <= 8 only,
 * see bde_svgd_fused_gram_supported) the kernel also leaves the Gram partials of the
 * UPDATED particles in ws_next, in the format bde_svgd_kstats reads, so the next step can
 * skip bde_svgd_gram as long as nothing else modifies P in between.
 * Single-tile path only: M <= BDE_FAST_PARTICLES.  ldg = leading dimension of G (0 = ld), as in bde_svgd_combine. */
int bde_svgd_fused_gram_supported(int M);
int bde_svgd_fused_sgd(float* P, const float* G, float* momentum_buf, int M, int64_t D, int64_t ld, int64_t ldg,
                       const float* kstat, double lr, double momentum, double dampening, double weight_decay,
                       int nesterov, int first, void* ws_next, void* stream);
int bde_svgd_fused_adam(float* P, const float* G, float* exp_avg, float* exp_avg_sq, int M, int64_t D, int64_t ld,
                        int64_t ldg, const float* kstat, double lr, double beta1, double beta2, double eps,
                        double weight_decay, int64_t step0, void* ws_next, void* stream);

/* ------------------------------------------------------------------ SWAG --
 * Statistics live on the device: mean [D], sq [D], and the deviation matrix as
 * a ring dev [K, ld] (row = one iterate) instead of the reference's CPU
 * [D, K] matrix that is physically rolled every update (swag.py:103).
 * Logical column c of the reference (0 = oldest ... K-1 = newest) is physical
 * row (head + c) mod K, where `head` is the row the NEXT update overwrites.
 *
 * Rows are contiguous (round 3 kept them interleaved in 16 KB pieces; no bench.py record showed a gain and the
 * layout is gone -- DESIGN.md section 8). */

/* One moment update, n = the already incremented `__updates` counter
 * (swag.py:97-104): mean = (n*mean + theta)/(n+1); sq = (n*sq + theta^2)/(n+1);
 * dev_row[:] = theta - mean_new.  The caller passes dev_row = the ring row `head` and
 * advances head.  theta is contiguous.  Bit-exact with the reference's fp32 CPU arithmetic. */
int bde_swag_update(const float* theta, float* mean, float* sq, float* dev_row, int64_t n, int64_t D, void* stream);

/* One posterior sample (swag.py:57,112-114 + LowRankMultivariateNormal.rsample):
 *   out = mean + sum_c dev[col c] * eps_w[c] / sqrt(2 (K-1))
 *              + sqrt(0.5 * (relu(sq - mean^2) + 1e-6)) * eps_d
 * eps_w [K] is indexed by LOGICAL column.  eps_w / eps_d may be NULL: the
 * noise then comes from the in-kernel Philox4x32-10 stream (seed, stream_id),
 * element e of the stream being a pure function of (seed, stream_id, e).
 * out and eps_d are contiguous [D] (out is the vector the model's parameters view). */
int bde_swag_sample(const float* mean, const float* sq, const float* dev, int K, int64_t ld, int head,
                    const float* eps_w, const float* eps_d, uint64_t seed, uint64_t stream_id,
                    float* out, int64_t D, void* stream);

/* S <= BDE_MAX_BATCH samples in ONE pass over the statistics (MFMA low-rank product; Philox
 * streams stream_id0 + s, identical to S calls of bde_swag_sample).  eps_w [S, K] / eps_d [S, ld_eps] (contiguous
 * rows) may be NULL.  out: S contiguous rows ld_out floats apart.  K <= 20: the next tile's ring rows travel through
 * the LDS-DMA while the current tile's noise epilogue runs. */
int bde_swag_sample_batched(const float* mean, const float* sq, const float* dev, int K, int64_t ld, int head,
                            const float* eps_w, const float* eps_d, int64_t ld_eps, uint64_t seed, uint64_t stream_id0,
                            float* out, int64_t ld_out, int S, int64_t D, void* stream);

/* In-kernel noise: Philox4x32 (Salmon et al., SC'11) keyed by `seed`, counter = (float4 index, stream id, domain),
 * normals by Box-Muller on the top 24 bits of each word.  `rounds`: 10 = the published default, used by every draw of
 * the BBB / iVON / layer kernels; bde_swag_philox_rounds() (= 7, the fewest rounds the paper reports as
 * Crush-resistant) is what bde_swag_sample / bde_swag_sample_batched use for their S x D normals per pass.
 *
 * The normals a kernel would use, written out (for tests and for callers that want the noise): eps_w [K] (the
 * low-rank weights' domain) and/or eps_d [D] (either may be NULL). */
int bde_swag_philox_rounds(void);
int bde_philox_normal(uint64_t seed, uint64_t stream_id, float* eps_w, int K, float* eps_d, int64_t D, int rounds,
                      void* stream);

/* The raw Philox4x32 words behind every in-kernel noise draw: for g in [0, n_groups)
 * out[4g .. 4g+3] = Philox4x32-<rounds>(counter = (lo32(idx0+g), hi32(idx0+g), lo32(stream_id),
 * hi32(stream_id) ^ domain), key = (lo32(seed), hi32(seed))).  Lets the tests pin the
 * generator against the published Random123 known-answer vectors (10 rounds; domain 0 = the diagonal /
 * element noise, 0x80000000 = the low-rank weights of the SWAG sampler). */
int bde_philox_bits(uint64_t seed, uint64_t stream_id, uint32_t domain, uint64_t idx0, uint32_t* out,
                    int64_t n_groups, int rounds, void* stream);

/* ------------------------------------------------- mean-field Gaussian (BBB) --
 * src/algos/util.py:151-183 (GaussianParameter) and bbb.py:18-21 (KL). */

/* w = mean + softplus(rho) * eps   (util.py:170-171,181-183).
 * eps == NULL: eps comes from Philox(seed, stream_id) and, if eps_out != NULL,
 * is also written there. */
int bde_gauss_draw_fwd(const float* mean, const float* rho, const float* eps, uint64_t seed,
                       uint64_t stream_id, float* w, float* eps_out, int64_t n, void* stream);

/* Backward of the draw: gmean (+)= g ; grho (+)= g * eps * sigmoid(rho).
 * accumulate != 0 adds into gmean/grho, else overwrites.  eps == NULL
 * regenerates the Philox noise of the forward call. */
int bde_gauss_draw_bwd(const float* g, const float* rho, const float* eps, uint64_t seed,
                       uint64_t stream_id, float* gmean, float* grho, int accumulate, int64_t n,
                       void* stream);

/* Bytes of scratch for the reductions of bde_gauss_kl / bde_l2. */
size_t bde_reduce_ws_bytes(void);

/* Closed-form KL(q || N(prior_mu, prior_sigma^2)) summed over n elements
 * (bbb.py:18-21) -> kl_out[0] (if kl_out != NULL), fused with its analytic
 * gradients (if gmean/grho != NULL):
 *   gmean (+)= c * (mean - prior_mu) / prior_sigma^2
 *   grho  (+)= c * (-1/sigma + sigma/prior_sigma^2) * sigmoid(rho)
 * with c = grad_scale * (grad_scale_dev ? *grad_scale_dev : 1) -- the host
 * part is pi = kl_rescaling/dataset_size (bbb.py:78), the device part an AMP
 * GradScaler's scale tensor. */
int bde_gauss_kl(const float* mean, const float* rho, float prior_mu, float prior_sigma,
                 float grad_scale, const float* grad_scale_dev, float* gmean, float* grho,
                 int accumulate, float* kl_out, void* ws, int64_t n, void* stream);

/* Plain-parameter L2 term (bbb.py:75-76): val_out[0] = l2_scale/2 * sum p^2,
 * g (+)= c * l2_scale * p. */
int bde_l2(const float* p, float l2_scale, float grad_scale, const float* grad_scale_dev, float* g,
           int accumulate, float* val_out, void* ws, int64_t n, void* stream);

/* MixturePrior.kl_divergence (bbb.py:23-37): -sum over the means of log(pi N(x; 0, sigma1) + (1 - pi) N(x; 0, sigma2))
 * with each component's log-density clamped to [-23, 0], and its gradient wrt the means
 * (gmean = or += grad_scale * [grad_scale_dev[0]] * d/dmean; rho receives no gradient from this prior).
 * val_out / gmean may be NULL. */
int bde_mixture_nll(const float* mean, float pi, float sigma1, float sigma2, float grad_scale,
                    const float* grad_scale_dev, float* gmean, int accumulate, float* val_out, void* ws,
                    int64_t n, void* stream);

/* Epilogue of the local-reparameterisation layers (bbb_layers.py:70-80: activation_mean +
 * sqrt(activation_var) * eps after the mean and variance GEMMs / convs), fused:
 *   fwd: out = mean + sqrt(var) * eps          bwd: gvar = g * eps / (2 sqrt(var))   (gmean = g)
 * eps == NULL: Philox(seed, stream_id) noise, regenerated in the backward call. */
int bde_local_reparam_fwd(const float* mean, const float* var, const float* eps, uint64_t seed,
                          uint64_t stream_id, float* out, int64_t n, void* stream);
int bde_local_reparam_bwd(const float* g, const float* var, const float* eps, uint64_t seed,
                          uint64_t stream_id, float* gvar, int64_t n, void* stream);

/* Operands of the variance product of the local-reparameterisation layers (bbb_layers.py:66-67,71,150-153), one pass
 * each instead of 2-3 ATen launches (and their autograd nodes):
 *   mode 0: out = clamp(v^2, 1e-4)               (v = the layer input)     bwd: gv = g * 2 v * [v^2 >= 1e-4]
 *   mode 1: out = clamp(softplus(v)^2, 1e-4)     (v = rho of the weights)  bwd: gv = g * [sigma^2 >= 1e-4] * 2 sigma sigmoid(v)
 *   mode 2: out = softplus(v)^2                  (BBBConv2d's bias)        bwd: gv = g * 2 sigma sigmoid(v)
 * Contiguous, 16-byte aligned buffers of n floats; outputs are overwritten. */
int bde_var_operand_fwd(const float* v, int mode, float* out, int64_t n, void* stream);
int bde_var_operand_bwd(const float* g, const float* v, int mode, float* gv, int64_t n, void* stream);

/* The local-reparameterisation CONVOLUTION layer (BBBConv2d, bbb_layers.py:146-154), forward and backward.
 *
 * bde_conv_lrt_prep -- once per weight VERSION (the weights change at base_optimizer.step(); a BBB step runs mc_samples
 * forward / backward passes per version, bbb.py:63-67): sigma^2 = clamp(softplus(W_rho)^2, 1e-4), its rho-derivative,
 * the bias variance, and both weight matrices re-arranged the way the kernels stage them (k-major, output channels padded
 * to 32, for the forward; transposed and flipped for the input gradient).  wbuf: bde_conv_lrt_prep_floats() floats, 16-byte aligned,
 * ZERO-INITIALISED by the caller once (the padding is never written).
 *
 * bde_conv_lrt_fwd -- both convolutions of lines 146-147, conv2d(x, W_mu, b_mu) and conv2d(clamp(x^2, 1e-4), sigma^2,
 * b_var), as ONE implicit GEMM with two accumulators per output tile over the same staged input windows (zero padding
 * applied after the clamp, as F.conv2d pads the clamped tensor), then out = mean + sqrt(var) * eps (lines 148-154) in
 * the epilogue.  x [N, C, H, W], b_mu [O] or NULL; has_bias_var != 0: the bias variance softplus(b_rho)^2 (NOT clamped:
 * line 147) that bde_conv_lrt_prep evaluated from its b_rho argument (NULL there = no bias variance) is added,
 * out / var_out [N, O, Ho, Wo] (var_out = the total activation variance, which the backward pass needs); all fp32,
 * contiguous NCHW, 16-byte aligned.  eps [N, O, Ho, Wo] or NULL = Philox(seed, stream_id) with the element numbering
 * of bde_local_reparam_fwd over the flat output (so bde_local_reparam_bwd regenerates the same noise).
 *
 * Backward, given g = dL/d out and g_var = g eps / (2 sqrt(var)) (bde_local_reparam_bwd on g, var_out and the noise):
 *   bde_conv_lrt_bwd_data:   g_x = convT(g, W_mu) + 2 x [x^2 >= 1e-4] convT(g_var, sigma^2) -- the same dual-accumulator
 *                            implicit GEMM over g / g_var dilated by the stride, the clamp's derivative in the epilogue;
 *   bde_conv_lrt_bwd_weight: g_wmu = corr(x, g), g_wrho = corr(clamp(x^2), g_var) * [sigma^2 >= 1e-4] 2 sigma sigmoid(rho):
 *                            dual-accumulator implicit GEMM reducing over the output pixels, per-share partial blocks in
 *                            `ws` (bde_conv_lrt_bwd_weight_ws_bytes) summed in a fixed order by a finish pass.
 * The bias gradients are plain channel sums of g and g_var (callers: torch.sum / bde_var_operand_bwd mode 2).
 *
 * bde_conv_lrt_supported: 1 when tilings exist for the layer and its input gradient (kernel <= 7 x 7, stride / padding
 * per axis with padding <= kernel - 1, no dilation / groups). */
int bde_conv_lrt_supported(int N, int C, int H, int W, int O, int KH, int KW, int stride_h, int stride_w, int pad_h, int pad_w);
/* The tilings the kernels run with, for tests and tools (tests/conv_emulator.py replays the kernels' index arithmetic
 * on the CPU with exactly these numbers).  bde_conv_lrt_plan: which = 0 forward, 1 input gradient; out[16] = MF, PT, NI, TH,
 * bands, CC, PH, PWP, WP, WK, tiles_per_img, kcpad_max, grid.x, grid.y, LDS bytes, 0.  bde_conv_lrt_bwd_weight_plan:
 * out[16] = MF, CT_MAX, NI, TH, bands, PS, CT, colgroups, PH, PWP, cmax, GP, npix, grid.y, LDS bytes, otiles. */
int bde_conv_lrt_plan(int which, int N, int C, int H, int W, int O, int KH, int KW, int stride_h, int stride_w, int pad_h,
                      int pad_w, int* out);
int bde_conv_lrt_bwd_weight_plan(int N, int C, int H, int W, int O, int KH, int KW, int stride_h, int stride_w, int pad_h,
                                 int pad_w, int* out);
/* Tuning hooks of the fused convolution (tools/conv_autotune.py; without them the planners' own scores decide).  The scores'
 * weights are hand-set; on the device every candidate tiling of a layer can be timed and the winner pinned.
 * A LAUNCH geometry is what one launch of the convolution kernel convolves: geo[15] = N, C, H, W, O, KH, KW, sh, sw, ph, pw,
 * dh, dw, Ho, Wo (dh, dw: dilation of the input image, > 1 only in the dilated input-gradient pass; Ho, Wo: the output extent
 * the launch writes).  A LAYER geometry is layer[11] = N, C, H, W, O, KH, KW, stride_h, stride_w, pad_h, pad_w.
 *   bde_conv_lrt_pass_geos    the launch geometries of a pass of a layer -- which = 0 forward (one), 1 input gradient over the
 *                             zero-dilated gradient (one), 2 input gradient per phase (up to sh * sw) -- into out[max][15];
 *                             returns their number
 *   bde_conv_lrt_candidates   every tiling the planner considers for a launch geometry: out[i][6] = WK, TH, NI, CC, PT, LDS bytes;
 *                             *chosen = the index it runs (the pinned one, else the best score); returns the number of candidates
 *   bde_conv_lrt_set_tiling   pins (WK, TH, NI, CC) for a launch geometry, process-wide; wk = 0 removes the pin; a tiling that is
 *                             not a candidate is refused (BDE_ERR_INVALID)
 *   bde_conv_lrt_wgrad_candidates / _set_tiling   the same for the weight-gradient pass, keyed by the layer geometry:
 *                             out[i][5] = CT, TH, NI, PS, LDS bytes.  Size the partials buffer AFTER pinning
 *                             (bde_conv_lrt_bwd_weight_ws_bytes follows the pin). */
int bde_conv_lrt_pass_geos(int which, const int* layer, int* out, int max);
int bde_conv_lrt_candidates(const int* geo, int* out, int max, int* chosen);
int bde_conv_lrt_set_tiling(const int* geo, int wk, int th, int ni, int cc);
int bde_conv_lrt_wgrad_candidates(const int* layer, int* out, int max, int* chosen);
int bde_conv_lrt_wgrad_set_tiling(const int* layer, int ct, int th, int ni, int ps);
size_t bde_conv_lrt_prep_floats(int O, int C, int KH, int KW);
int bde_conv_lrt_prep(const float* w_mu, const float* w_rho, const float* b_rho, int O, int C, int KH, int KW, float* wbuf,
                      void* stream);
/* bde_conv_lrt_prep for a layer whose stride is known (the caller's layer object): additionally writes the input-gradient
 * matrices of a strided layer split by PHASE of the output pixel grid -- pixel (a + sh i, b + sw j) of g_x only sees the taps
 * r = (a + ph) mod sh + sh r', q likewise, and for those the pass is a stride-1 convolution of the undilated g with that
 * sub-kernel.  bde_conv_lrt_bwd_data_phases (wbuf prepared by THIS function with the same stride / padding) runs one launch per
 * phase: the matrix work of the layer's useful flops instead of sh * sw times that (bde_conv_lrt_bwd_data convolves the
 * zero-dilated g; it stays the fallback for geometries without a per-phase tiling and for wbufs prepared without a stride). */
int bde_conv_lrt_prep_strided(const float* w_mu, const float* w_rho, const float* b_rho, int O, int C, int KH, int KW, int stride_h,
                              int stride_w, int pad_h, int pad_w, float* wbuf, void* stream);
int bde_conv_lrt_bwd_data_phases(const float* g_out, const float* g_var, const float* wbuf, const float* x, float* g_x, int N,
                                 int C, int H, int W, int O, int KH, int KW, int stride_h, int stride_w, int pad_h, int pad_w,
                                 void* stream);
/* (var_out may be NULL: a forward pass nobody will differentiate -- torch.no_grad() -- does not write the variance.) */
int bde_conv_lrt_fwd(const float* x, const float* wbuf, const float* b_mu, int has_bias_var, const float* eps, uint64_t seed,
                     uint64_t stream_id, float* out, float* var_out, int N, int C, int H, int W, int O, int KH, int KW,
                     int stride_h, int stride_w, int pad_h, int pad_w, void* stream);
int bde_conv_lrt_bwd_data(const float* g_out, const float* g_var, const float* wbuf, const float* x, float* g_x, int N, int C,
                          int H, int W, int O, int KH, int KW, int stride_h, int stride_w, int pad_h, int pad_w, void* stream);
/* The first pass of BBBConv2d's backward: g_var = g eps / (2 sqrt(var)) over the layer output [N, O, HW] (eps supplied, or the
 * forward's Philox stream: float4 group e >> 2 of the flat output, as bde_local_reparam_bwd) and, in the same pass, the bias
 * gradients (b_rho != NULL): g_bmu[o] = sum g, g_brho[o] = (sum g_var) 2 softplus(b_rho) sigmoid(b_rho) -- the reference's
 * autograd does two reductions over the gradient and the chain rule of bbb_layers.py:147 separately.  ws:
 * bde_conv_lrt_gvar_ws_bytes(N, O) bytes, 8-byte aligned (fixed-order fp64 partials). */
size_t bde_conv_lrt_gvar_ws_bytes(int N, int O);
int bde_conv_lrt_gvar_bias(const float* g, const float* var, const float* eps, uint64_t seed, uint64_t stream_id, float* gvar,
                           const float* b_rho, float* g_bmu, float* g_brho, void* ws, int N, int O, int64_t HW, void* stream);
size_t bde_conv_lrt_bwd_weight_ws_bytes(int N, int C, int H, int W, int O, int KH, int KW, int stride_h, int stride_w, int pad_h,
                                        int pad_w);
/* `ws_bytes` (ABI 406): the size of `ws` as the caller allocated it.  The launch re-plans, so a tiling pinned between
 * the sizing call and the launch (bde_conv_lrt_wgrad_set_tiling from another thread) may need more partial slots than
 * the buffer holds: that is refused with BDE_ERR_INVALID instead of written out of bounds. */
int bde_conv_lrt_bwd_weight(const float* x, const float* g_out, const float* g_var, const float* w_rho, void* ws,
                            size_t ws_bytes, float* g_wmu, float* g_wrho, int N, int C, int H, int W, int O, int KH, int KW,
                            int stride_h, int stride_w, int pad_h, int pad_w, void* stream);

/* The whole local-reparameterisation forward of a mean-field LINEAR layer (bbb_layers.py:61-80, sampling =
 * "activations") for small batches (B <= 128): W_mu / W_rho [O, I] row-major are streamed ONCE, sigma^2 =
 * clamp(softplus(rho)^2, 1e-4) and clamp(x^2, 1e-4) are formed on the fly and both products run on the f32 MFMA
 * (split-K partials in `ws`, fixed-order finish):
 *   out[b, o] = (x W_mu^T + b_mu)[b, o] + sqrt((clamp(x^2) clamp(sigma_W^2)^T + var_b)[b, o]) * eps[b, o]
 * with var_b = softplus(b_rho)^2, clamped at 1e-4 iff clamp_bias_var (BBBLinear clamps it, BBBConv2d does not).
 * x [B, I] with row stride ldx; b_mu / b_rho both NULL for a bias-free layer; eps [B, O] or NULL (Philox: element
 * e = b * O + o of stream stream_id, as bde_philox_normal writes it); var_out [B, O] (may be NULL) receives the
 * activation variance the backward pass needs. */
int bde_lrt_linear_supported(int B, int I, int O);
size_t bde_lrt_linear_ws_bytes(int B, int I, int O);
int bde_lrt_linear_fwd(const float* x, int64_t ldx, const float* w_mu, const float* w_rho, const float* w_s2,
                       const float* b_mu, const float* b_rho, int clamp_bias_var, const float* eps, uint64_t seed,
                       uint64_t stream_id, float* out, float* var_out, int B, int I, int O, void* ws, void* stream);
/* sigma^2 cache for WIDE layers (bde_lrt_sigma_cache_wanted(I, O): O * I >= 2^20, I % 4 == 0; there the kernels are
 * co-bound by the softplus / sigmoid evaluations, the fp32 MFMA and HBM).  The weights only change at
 * base_optimizer.step(), but BBB runs mc_samples forward + backward passes per step (bbb.py:63-67): one pass
 *   s2 = clamp(softplus(rho)^2, 1e-4),   ds2 = [sigma^2 >= 1e-4] * 2 sigma sigmoid(rho)      (ds2 may be NULL)
 * per weight VERSION, then every forward (w_s2) and backward (w_s2, w_ds2) of that version reads these instead of
 * evaluating the transcendentals per weight per pass -- same bytes read, identical bits.  Pass NULL to compute on
 * the fly; narrow layers ignore the cache. */
int bde_lrt_sigma_cache_wanted(int I, int O);
int bde_lrt_sigma_cache(const float* w_rho, float* s2, float* ds2, int64_t n, void* stream);

/* Backward of bde_lrt_linear_fwd: the autograd graph of bbb_layers.py:61-80 in two launches for layers of up to 2^20
 * weights (one pass over the weights for all three matrix gradients), three or four for larger ones.  With g = d loss / d out [B, O] and
 * gvar = g * eps / (2 sqrt(var)) (var = the var_out of the forward call; eps as supplied there, or NULL to regenerate
 * the forward's Philox noise from (seed, stream_id)):
 *   g_x    [B, I] = g W_mu + (gvar clamp(sigma_W^2)) * 2 x * [x^2 >= 1e-4]           (NULL: not computed)
 *   g_wmu  [O, I] = g^T x
 *   g_wrho [O, I] = (gvar^T clamp(x^2)) * [sigma_W^2 >= 1e-4] * 2 sigma_W sigmoid(rho_W)
 *   g_bmu  [O]    = sum_b g,     g_brho [O] = (sum_b gvar) * [clamp_bias_var ? sigma_b^2 >= 1e-4 : 1] * 2 sigma_b sigmoid(rho_b)
 * (b_rho, g_bmu, g_brho all NULL for a bias-free layer).  Every output is OVERWRITTEN (autograd accumulates).
 * Reductions are MFMA tiles and fixed-order sums: bit-reproducible.  ws: bde_lrt_linear_bwd_ws_bytes(B, I, O). */
size_t bde_lrt_linear_bwd_ws_bytes(int B, int I, int O);
int bde_lrt_linear_bwd(const float* x, int64_t ldx, const float* w_mu, const float* w_rho, const float* w_s2,
                       const float* w_ds2, const float* b_rho, int clamp_bias_var, const float* g, const float* var,
                       const float* eps, uint64_t seed, uint64_t stream_id, float* g_x, float* g_wmu, float* g_wrho,
                       float* g_bmu, float* g_brho, int B, int I, int O, void* ws, void* stream);

/* ------------------------------------------------------------------ iVON --
 * src/algos/ivorn.py:102-115 (weight-noise draw) and :66-89 (update). */

/* delta = eps / sqrt(n_eff * max(prec, 1e-4)); param = mean + delta;
 * delta_sum = (first ? delta : delta_sum + delta).  eps == NULL: Philox.
 * deterministic != 0: delta = 0 (ivorn.py:109-110). */
int bde_ivon_sample(const float* mean, const float* prec, const float* eps, uint64_t seed, uint64_t stream_id,
                    float n_eff, int deterministic, int first, float* param, float* delta_sum, int64_t n,
                    void* stream);

/* The fused natural-gradient update (ivorn.py:76-89), one pass, 32 B/param.
 * The scalars are the Python-double expressions of the reference, each rounded
 * to fp32 by the caller exactly where PyTorch rounds them (a Python scalar
 * meets an fp32 tensor):
 *   lam = tempering*prior_prec/n_eff (line 74), n_eff = N*augmentation (72),
 *   mc = mc_samples, omb1 = 1-beta1, omb2 = 1-beta2, c2 = 0.5*(1-beta2)^2,
 *   bc1 = 1-beta1^t, bc2 = 1-beta2^t (84-85).
 * mean/momentum/prec are updated in place. */
int bde_ivon_update(float* mean, float* momentum, float* prec, const float* delta_sum, const float* acc_grad,
                    float lam, float n_eff, float mc, float beta1, float omb1, float omb2, float c2,
                    float bc1, float bc2, float lr, float damping, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BDE_HIP_H */
