/* libmcmc_hip.so - C ABI of the MI355X (gfx950) Monte Carlo predictive path.
 *
 * The reference (fintkz/mcmc) has no FFI layer: the path sits behind the Python class
 * GPUBayesianEnsemble (models/bayesian_ensemble.py:95-256) and utils.peak_weighted_loss
 * (utils.py:7-28).  These entry points are what a binding for that path calls (ctypes in
 * mcmc_amd/_lib.py; INTEGRATION.md shows the stub).  Each one cites the reference lines it
 * replaces.
 *
 * Conventions
 *   - every function returns 0 on success, a negative MCMC_E* code on error; the message is
 *     available from mcmc_last_error() (thread-local);
 *   - all buffers are DEVICE pointers owned by the caller (torch tensors' data_ptr()) unless a
 *     parameter says "host"; nothing is allocated or freed behind the caller's back;
 *   - `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream); calls only enqueue;
 *   - a context is bound to one device, is not thread-safe, and must be used with that device
 *     current - one context per device per thread, like the reference's single-threaded use
 *     (main.py:248-251).
 *
 * Vocabulary: member = one BayesianNetwork of the ensemble; sample = one Monte Carlo weight
 * draw; pair = (member, sample); unit = pair x timestep-row.  Pairs are processed in the order
 * q = 0 .. M*S-1 with member = q % M, sample = q / M (every chunk of consecutive q holds all
 * members evenly); the GLOBAL pair id g = member*S + sample is the reference's row index in
 * all_predictions (bayesian_ensemble.py:242), indexes injected eps and keys Philox.
 */
#ifndef MCMC_HIP_H
#define MCMC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCMC_OK 0
#define MCMC_EINVAL (-1)  /* bad argument (shape, dtype, null pointer, alignment) */
#define MCMC_EHIP (-2)    /* a HIP runtime call or kernel launch failed           */
#define MCMC_ESTATE (-3)  /* context used on the wrong device / destroyed         */

#define MCMC_DT_F32 0  /* fp32 operands, fp32 MFMA accumulate (exact fp32 FMA chain)      */
#define MCMC_DT_BF16 1 /* bf16 operands, fp32 MFMA accumulate, exact ELU (v_exp_f32)       */
/* bf16 operands, fp32 MFMA accumulate, ELU with its exponential built in the bf16 exponent field
 * (no transcendental: |elu error| <= 0.03, predictive mean / std within 6e-3 of fp32 at the reference's
 * own shape against 2e-3 for MCMC_DT_BF16; ~1.17x the throughput).  Eval-mode forward only: mcmc_forward
 * with masks and mcmc_forward_dropout reject it.  xpack / wpack images are per dtype: pack_x,
 * sample_weights and forward of one predict must name the same dtype. */
#define MCMC_DT_BF16_FAST 2

#define MCMC_NOISE_NONE 0     /* W = mu            (sample=False, bayesian_ensemble.py:45-47) */
#define MCMC_NOISE_INJECTED 1 /* W = mu + eps*exp(rho), eps supplied by the caller          */
#define MCMC_NOISE_PHILOX 2   /* eps from Philox4x32-10 keyed (seed; g, param index)        */

typedef struct mcmc_ctx mcmc_ctx;
typedef void* mcmc_stream_t;

/* Thread-local text of the last error (never NULL). */
const char* mcmc_last_error(void);
/* ABI version of this header (bumped on any signature change). */
int mcmc_abi_version(void);

/* One context per device.  Replaces nothing in the reference (it keeps its state in torch);
 * holds the device id and cached device properties. */
int mcmc_ctx_create(int device, mcmc_ctx** out);
int mcmc_ctx_destroy(mcmc_ctx* ctx);

/* ---- layout queries (host only, no device work) ---------------------------------------- */
/* Parameters of one member in the flat layout = noise draws per pair:
 * 44,289 at input_dim 11 (bayesian_ensemble.py:17-22,61-70). */
int64_t mcmc_param_count(int input_dim);
/* Bytes of one packed input matrix / of one pair's packed effective weights. */
int64_t mcmc_xpack_bytes(int64_t n_rows, int input_dim, int dtype);
int64_t mcmc_wpack_bytes(int input_dim, int dtype);

/* ---- piece 1: member x sample forward ---------------------------------------------------- */
/* X [n_rows, input_dim] fp32 row-major -> kernel input layout (padded, bias column, dtype).
 * Replaces X.to(device) / x.rename(None) staging (bayesian_ensemble.py:219, :76). */
int mcmc_pack_x(mcmc_ctx* ctx, const float* X, int64_t n_rows, int input_dim, int dtype,
                void* xpack, mcmc_stream_t stream);

/* Effective weights of pairs q = pair_begin .. pair_begin+pair_count-1:
 *   W = mu + eps * exp(rho)   (bayesian_ensemble.py:43-44; exp, not softplus)
 * written as MFMA operand fragments, wpack[pair_count][mcmc_wpack_bytes].
 *   mu, rho : [n_members, P] fp32 flat layout (W then b per layer, layers in order)
 *   eps     : MCMC_NOISE_INJECTED: [n_members*n_samples, P] fp32 indexed by global pair id g
 *             (same flat order = the reference's draw order); otherwise ignored (may be NULL)
 *   seed    : MCMC_NOISE_PHILOX key. */
int mcmc_sample_weights(mcmc_ctx* ctx, const float* mu, const float* rho, int n_members,
                        int n_samples, int input_dim, int64_t pair_begin, int pair_count,
                        int noise_mode, const float* eps, uint64_t seed, int dtype, void* wpack,
                        mcmc_stream_t stream);

/* Same as mcmc_sample_weights with MCMC_NOISE_PHILOX, but the 64-bit key is read from device memory
 * when the kernel runs: a captured hipGraph can be replayed with a new seed (no counterpart in the
 * reference, whose randn_like stream advances by itself, bayesian_ensemble.py:43-44). */
int mcmc_sample_weights_seedptr(mcmc_ctx* ctx, const float* mu, const float* rho, int n_members,
                                int n_samples, int input_dim, int64_t pair_begin, int pair_count,
                                const uint64_t* seed_dev, int dtype, void* wpack, mcmc_stream_t stream);

/* Raw Philox noise eps[pair_count, P] fp32 for global pair ids g = g_begin.. (test/inspection:
 * what MCMC_NOISE_PHILOX feeds into mcmc_sample_weights). */
int mcmc_philox_eps(mcmc_ctx* ctx, uint64_t seed, int64_t g_begin, int pair_count, int input_dim,
                    float* eps, mcmc_stream_t stream);

/* The fused MLP forward of BayesianNetwork (bayesian_ensemble.py:73-92) for pair_count pairs:
 *   h = X; 3 x { h = elu(h W^T + b); h = h * mask * keep_scale }; y = h w3^T + b3
 * Y[c*ldY + row] = y of pair c, row (fp32).  ldY >= n_rows, multiple of 4.
 *   mask0/1/2 : optional uint8 keep-masks [pair_count, n_rows, 256/128/64] (train-mode dropout,
 *               bayesian_ensemble.py:86); NULL = identity (eval mode, what predict() runs,
 *               bayesian_ensemble.py:215-216).  keep_scale = fp32(1/(1-p)).  4-byte aligned.
 *               MCMC_DT_F32 and MCMC_DT_BF16 apply them inside the MFMA kernel's activation stage. */
int mcmc_forward(mcmc_ctx* ctx, int dtype, const void* xpack, int64_t n_rows, int input_dim,
                 const void* wpack, int pair_count, const uint8_t* mask0, const uint8_t* mask1,
                 const uint8_t* mask2, float keep_scale, float* Y, int64_t ldY,
                 mcmc_stream_t stream);

/* The same forward with counter-based dropout masks (train-mode forward, bayesian_ensemble.py:86,
 * :158; an opt-in MC-dropout predict - the reference's predict() itself runs in eval mode): the
 * keep flag of hidden activation elem (0..447 over the three layers) of global pair id
 * g = (q % n_members) * n_samples + q / n_members, q = pair_begin + c, and batch row `row` is
 * a 16-bit half of Philox4x32-10(counter (((elem>>4)<<1) | ((elem>>2)&1), g, 'MASK', row_begin + row),
 * key mask_seed) - half slot&1 of output word slot>>1, slot = 4*((elem>>3)&1) + (elem&3) - compared
 * < floor((1 - p_drop) * 65536): one Philox block per eight elements (ABI 21; 32 bits per element
 * before).  It is the stream of mcmc_train_forward's mask_mode 2: predict and train see the same masks.
 * row_begin: global index of the launch's first row (0 unless the launch is one row block of a
 * larger input: the masks then do not depend on how the rows are cut into launches). */
int mcmc_forward_dropout(mcmc_ctx* ctx, int dtype, const void* xpack, int64_t n_rows,
                         int input_dim, const void* wpack, int n_members, int n_samples,
                         int64_t pair_begin, int pair_count, uint64_t mask_seed, float p_drop,
                         int64_t row_begin, float* Y, int64_t ldY, mcmc_stream_t stream);

/* ---- piece 2: streaming per-timestep statistics ------------------------------------------- */
/* Replaces torch.stack + reshape + mean(0) + std(0) (bayesian_ensemble.py:234-246) without
 * keeping all M*S samples: the caller feeds chunks Y[n_pairs, ldY] and the per-row state
 *   sum, sumsq : [n_rows] fp64      hist : [n_rows, nbins] uint16 (optional, NULL = no quantiles)
 *   lo, inv_w  : [n_rows] fp32 histogram placement (mcmc_reduce_set_edges)
 * is updated in place.  State must be zeroed by the caller before the first chunk.
 * nbins: 64, 128, 256 or 512; at most 65,535 samples per row in total (uint16 counts). */
int mcmc_reduce_update(mcmc_ctx* ctx, const float* Y, int64_t ldY, int n_pairs, int64_t n_rows,
                       double* sum, double* sumsq, const float* lo, const float* inv_w,
                       uint16_t* hist, int nbins, mcmc_stream_t stream);

/* Sum of n_parts partial states of the SAME rows: the local half of the multi-GPU combine (the reference
 * combines nothing - it stacks every sample on one device, bayesian_ensemble.py:234-246).  After the
 * row exchange a rank holds sum_parts / sumsq_parts [n_parts, n_rows] fp64 and hist_parts [n_parts, n_rows, nbins]
 * uint16 (NULL = no histogram); the sums run in the fixed order p = 0..n_parts-1.  A count above 65,535 saturates and
 * sets *overflow (device int32, zeroed by the caller; may be NULL) to 1.  All buffers 16-byte aligned. */
int mcmc_reduce_merge(mcmc_ctx* ctx, int n_parts, int64_t n_rows, const double* sum_parts,
                      const double* sumsq_parts, const uint16_t* hist_parts, int nbins, double* sum,
                      double* sumsq, uint16_t* hist, int32_t* overflow, mcmc_stream_t stream);

/* Histogram placement from pilot moments (n0 samples per row already in sum/sumsq): interior
 * bins 1..nbins-2 tile [mean - k*std, mean + k*std), bins 0 and nbins-1 catch the tails. */
int mcmc_reduce_set_edges(mcmc_ctx* ctx, int64_t n_rows, int64_t n0, const double* sum,
                          const double* sumsq, int nbins, float range_sigmas, float* lo,
                          float* inv_w, mcmc_stream_t stream);

/* mean = sum/n, std = unbiased (n-1) like torch.std (bayesian_ensemble.py:245-246; NaN at n=1),
 * quant[k, row] = histogram quantile at q_levels[k] (host fp64 array, n_q <= 8): torch.quantile's
 * 'linear' rule - interpolate order statistics floor(q*(n-1)) and +1 - with each order statistic
 * placed inside its own bin, so the result is within one bin width of the exact quantile.
 * hist may be NULL iff n_q == 0. */
int mcmc_reduce_finalize(mcmc_ctx* ctx, int64_t n_rows, int64_t n, const double* sum,
                         const double* sumsq, const uint16_t* hist, const float* lo,
                         const float* inv_w, int nbins, const double* q_levels, int n_q,
                         float* mean, float* std, float* quant, mcmc_stream_t stream);

/* ---- piece 3: fused peak-weighted loss (utils.py:7-28) ------------------------------------ */
/* Three streaming passes over this rank's slice p,t [n]; the partial sums live in device
 * memory so a data-parallel caller can all-reduce them between passes (n_total = global N).
 *   pass 1: stats[0..2] += { sum t, sum t^2, sum |p-t| }                     (utils.py:17-18,22-23)
 *   pass 2: part[0..1]  += { sum d^2 w, G = sum k d^2 g' }                   (utils.py:19-28)
 *   pass 3: grad[j] = grad_scale/N * [2 d w + (alpha-1) sign(d) (k d^2 g' - G/N)]
 * loss = part[0]/n_total.  stats/part are fp64[4] each, zeroed by the caller. */
int mcmc_loss_stats(mcmc_ctx* ctx, const float* p, const float* t, int64_t n, double* stats,
                    mcmc_stream_t stream);
int mcmc_loss_forward(mcmc_ctx* ctx, const float* p, const float* t, int64_t n, int64_t n_total,
                      const double* stats, float alpha, float beta, double* part,
                      mcmc_stream_t stream);
int mcmc_loss_backward(mcmc_ctx* ctx, const float* p, const float* t, int64_t n, int64_t n_total,
                       const double* stats, const double* part, float alpha, float beta,
                       float grad_scale, float* grad, mcmc_stream_t stream);

/* The same loss and gradient for ONE rank (n = N, the reference's own use: utils.py:7-28 is called on whole tensors)
 * in one call: the three passes back to back, their partial sums as per-workgroup slots of `workspace`
 * (mcmc_loss_workspace_bytes() bytes of device memory, 8-byte aligned, contents irrelevant before and after) that
 * every later pass re-reduces in a fixed order - no atomics, nothing to zero, bitwise reproducible.
 *   loss [1] device fp32 = mean(d^2 w);  grad [n] = grad_scale * dL/dp, or NULL for the forward value alone. */
int64_t mcmc_loss_workspace_bytes(void);
int mcmc_loss_fused(mcmc_ctx* ctx, const float* p, const float* t, int64_t n, float alpha, float beta,
                    float grad_scale, float* loss, float* grad, void* workspace, mcmc_stream_t stream);

/* mcmc_loss_fused without a workspace argument and without any allocation.  It has NO counterpart in the reference: it
 * is for callers that cannot allocate - code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.
 * The per-workgroup partial sums live in a 40 KB file-scope device array that every GPU the library runs on holds until the
 * process ends (its per-device address is taken once, in mcmc_ctx_create).  Same kernels, same bits as mcmc_loss_fused.
 * n <= MCMC_LOSS_STATIC_MAX_N (larger: MCMC_EINVAL).  All calls on one GPU share that array: calls that overlap on one GPU
 * (different streams or threads) race; calls in stream order are safe, and nothing needs clearing between them (every slot
 * a call reads it has written first). */
#define MCMC_LOSS_STATIC_MAX_N (((int64_t)1 << 31) - 1)
int mcmc_loss_fused_static(mcmc_ctx* ctx, const float* p, const float* t, int64_t n, float alpha,
                           float beta, float grad_scale, float* loss, float* grad,
                           mcmc_stream_t stream);

/* ---- next row: one training step of all members --------------------------- */
/* Replaces the inner loop of GPUBayesianEnsemble.train (bayesian_ensemble.py:160-181) for all
 * n_members at once (the reference trains them one after the other, :135).  Pairs are indexed
 * pair = member*n_samples + sample.  All buffers are caller-owned device memory:
 *   mu, rho, sigma : [M,P]    parameters and sigma = exp(rho) (mcmc_train_sigma refreshes it)
 *   eps, W, WT, dW : [M*S,P]  noise, effective weights ([out][in] and [in][out] per layer), and
 *                             the gradient w.r.t. the effective weights of every pair
 *   xb [M,rows,F], yb [M,rows] : each member's mini-batch (DataLoader order differs per member)
 *   Y [M*S,rows], delta [M,rows], loss [M]
 * mask_mode: 0 none, 1 host uint8 keep-masks [M*S,rows,256/128/64] (4-byte aligned), 2 Philox keyed
 * (mask_seed; pair, row, element) with keep probability 1-p_drop (bayesian_ensemble.py:71,86). */
int mcmc_train_sigma(mcmc_ctx* ctx, const float* rho, float* sigma, int64_t n, mcmc_stream_t stream);
/* eps (injected copy or Philox), W = mu + eps*sigma in both layouts (bayesian_ensemble.py:43-44). */
int mcmc_train_sample(mcmc_ctx* ctx, const float* mu, const float* sigma, int n_members,
                      int n_samples, int input_dim, int noise_mode, const float* eps_in,
                      uint64_t seed, float* eps, float* W, float* WT, mcmc_stream_t stream);
/* Train-mode forward of every pair over its member's batch (bayesian_ensemble.py:169-171). */
int mcmc_train_forward(mcmc_ctx* ctx, const float* xb, int64_t rows, int input_dim, int n_members,
                       int n_samples, const float* WT, int mask_mode, const uint8_t* mask0,
                       const uint8_t* mask1, const uint8_t* mask2, uint64_t mask_seed,
                       float p_drop, float* Y, mcmc_stream_t stream);
/* pred_mean over samples, peak_weighted_loss per member, delta = dLoss/dY of one sample
 * (bayesian_ensemble.py:173-176, utils.py:7-28). */
int mcmc_train_loss(mcmc_ctx* ctx, const float* Y, const float* yb, int n_members, int n_samples,
                    int64_t rows, float alpha, float beta, float* delta, float* loss,
                    mcmc_stream_t stream);
/* mcmc_train_loss that also adds every member's loss to loss_acc (fp64 [M]): the epoch sums of
 * bayesian_ensemble.py:181 / temporal_fusion.py:257 kept on the device. */
int mcmc_train_loss_acc(mcmc_ctx* ctx, const float* Y, const float* yb, int n_members, int n_samples,
                        int64_t rows, float alpha, float beta, float* delta, float* loss,
                        double* loss_acc, mcmc_stream_t stream);
/* Backward of every pair (recomputes its forward with the same masks): dW (loss.backward, :178). */
int mcmc_train_backward(mcmc_ctx* ctx, const float* xb, int64_t rows, int input_dim,
                        int n_members, int n_samples, const float* W, const float* WT,
                        int mask_mode, const uint8_t* mask0, const uint8_t* mask1,
                        const uint8_t* mask2, uint64_t mask_seed, float p_drop,
                        const float* delta, float* dW, mcmc_stream_t stream);
/* Large batches (BASELINE config 5: 10^5..10^6 rows per rank and step; the reference's batch is 32 rows,
 * bayesian_ensemble.py:124, so it has no counterpart): each pair's rows are split into row_blocks runs of
 * whole tiles (64 rows when row_blocks > 1, eight waves per workgroup), one workgroup each, and
 * dW [M*S*row_blocks, P] receives one partial per block (mcmc_train_grad sums them).
 * mcmc_train_row_blocks(rows) is the split the library recommends (1 up to 2,047 rows).  mcmc_train_forward
 * splits its rows the same way internally. */
int mcmc_train_row_blocks(int64_t rows);
int mcmc_train_backward_blocks(mcmc_ctx* ctx, const float* xb, int64_t rows, int input_dim,
                               int n_members, int n_samples, const float* W, const float* WT,
                               int mask_mode, const uint8_t* mask0, const uint8_t* mask1,
                               const uint8_t* mask2, uint64_t mask_seed, float p_drop,
                               const float* delta, int row_blocks, float* dW, mcmc_stream_t stream);
/* The same forward / backward pair with autograd's saved tensors instead of the recompute (loss.backward() at :178
 * works on what the forward kept): mcmc_train_forward_save also leaves every tile's activations in
 * act [M*S*mcmc_train_act_floats(rows)] (1.8 KB per pair and row: 9.5 GB for 80 pairs x 65,536 rows) and
 * mcmc_train_backward_saved (row_blocks = mcmc_train_row_blocks(rows)) reloads them.  Same numbers, bit for bit, as
 * mcmc_train_forward / mcmc_train_backward_blocks; 1.2x faster per step when the buffer fits. */
int mcmc_train_forward_save(mcmc_ctx* ctx, const float* xb, int64_t rows, int input_dim, int n_members,
                            int n_samples, const float* WT, int mask_mode, const uint8_t* mask0,
                            const uint8_t* mask1, const uint8_t* mask2, uint64_t mask_seed,
                            float p_drop, float* Y, float* act, mcmc_stream_t stream);
int mcmc_train_backward_saved(mcmc_ctx* ctx, const float* xb, int64_t rows, int input_dim,
                              int n_members, int n_samples, const float* W, const float* WT,
                              int mask_mode, const uint8_t* mask0, const uint8_t* mask1,
                              const uint8_t* mask2, uint64_t mask_seed, float p_drop,
                              const float* delta, int row_blocks, float* dW, const float* act,
                              mcmc_stream_t stream);
/* Opt-in bf16-operand variant of the large-batch forward / backward pair (no counterpart in the reference, whose step is fp32 on
 * 32 rows, bayesian_ensemble.py:124,165-178): the same train-mode forward and back-propagation on bf16 MFMA operands with fp32
 * accumulation; master weights, mcmc_train_sample, mcmc_train_loss, mcmc_train_grad and mcmc_train_apply stay fp32.  fp32 MFMA
 * runs at 1/16 of the bf16 rate, which bounds the fp32 kernels at 10^5 .. 10^6 rows per step.  Reads only W ([out][in] per layer,
 * 16-byte aligned) and writes dW [M*S*row_blocks, P] partials like mcmc_train_backward_blocks.
 * row_blocks: any count in [1, 65535], the same in both calls (these kernels cut a pair's rows into row_blocks runs of whole 64-row
 * tiles themselves; blocks beyond the last tile write zero partials) - mcmc_train_row_blocks(rows) is not required here.
 * act: optional device buffer of n_members*n_samples*mcmc_train_act_bytes_bf16(rows) bytes (16-byte aligned; 0.9 KB per pair
 * and row): the forward kernel leaves every tile's activations there and the backward kernel reloads them - autograd's saved
 * tensors, :178; dropped elements are stored as -0.0, so the masks are not drawn again.  NULL in both calls: the backward kernel
 * recomputes each tile's forward pass with the same masks (same numbers).  Y / loss within 2e-2, gradients within 2e-2 of each
 * tensor's largest entry of the fp32 path (tests/test_gpu_train.py). */
int64_t mcmc_train_act_bytes_bf16(int64_t rows);
int mcmc_train_forward_bf16(mcmc_ctx* ctx, const float* xb, int64_t rows, int input_dim, int n_members,
                            int n_samples, const float* W, int mask_mode, const uint8_t* mask0,
                            const uint8_t* mask1, const uint8_t* mask2, uint64_t mask_seed, float p_drop,
                            int row_blocks, float* Y, void* act, mcmc_stream_t stream);
int mcmc_train_backward_bf16(mcmc_ctx* ctx, const float* xb, int64_t rows, int input_dim, int n_members,
                             int n_samples, const float* W, int mask_mode, const uint8_t* mask0,
                             const uint8_t* mask1, const uint8_t* mask2, uint64_t mask_seed, float p_drop,
                             const float* delta, int row_blocks, float* dW, const void* act,
                             mcmc_stream_t stream);
/* gmu = sum_s dW, grho = sum_s dW*eps*sigma; clip_grad_norm_(max_norm) per member (:179); AdamW
 * with torch's update rule (:180) on members whose active flag is set (NULL = all); refreshes
 * sigma.  step = 1-based step count for the bias corrections.  adam_m/adam_v: [M,2P] (mu then
 * rho), norm_scratch: fp64 [M*mcmc_train_norm_blocks()], gnorm: optional [M] pre-clip norms. */
int mcmc_train_norm_blocks(void);
int mcmc_train_update(mcmc_ctx* ctx, float* mu, float* rho, float* sigma, int n_members,
                      int n_samples, int input_dim, const float* eps, const float* dW, float* gmu,
                      float* grho, float* adam_m, float* adam_v, double* norm_scratch,
                      const uint8_t* active, float lr, float beta1, float beta2, float adam_eps,
                      float weight_decay, float max_norm, int64_t step, float* gnorm,
                      mcmc_stream_t stream);

/* The two halves of mcmc_train_update for the data-parallel step (BASELINE config 5, SURVEY.md 8e:
 * rows of each member's batch sharded over ranks): mcmc_train_grad forms this rank's (gmu, grho)
 * from its dW [M*S*row_blocks, P] (row_blocks = 1 with mcmc_train_backward); the caller sum-all-reduces them (RCCL; the reference has no counterpart - it is
 * single-process, loss.backward() at bayesian_ensemble.py:178 sees the whole batch);
 * mcmc_train_apply recomputes the norm of the combined gradient, then clips and applies AdamW
 * exactly as mcmc_train_update does (:179-180). */
int mcmc_train_grad(mcmc_ctx* ctx, int n_members, int n_samples, int row_blocks, int input_dim,
                    const float* eps, const float* sigma, const float* dW, float* gmu, float* grho,
                    double* norm_scratch, mcmc_stream_t stream);
int mcmc_train_apply(mcmc_ctx* ctx, float* mu, float* rho, float* sigma, int n_members,
                     int input_dim, const float* gmu, const float* grho, float* adam_m,
                     float* adam_v, double* norm_scratch, const uint8_t* active, float lr,
                     float beta1, float beta2, float adam_eps, float weight_decay, float max_norm,
                     int64_t step, float* gnorm, mcmc_stream_t stream);

/* The whole step of bayesian_ensemble.py:165-181 for every member in ONE call: mcmc_train_sample ->
 * mcmc_train_forward -> mcmc_train_loss (run inside the backward kernel: same arithmetic, `delta` is not
 * written) -> mcmc_train_backward -> mcmc_train_update on `stream`, for
 * batches of one row block (mcmc_train_row_blocks(rows) == 1; larger batches and the data-parallel step
 * use the separate entry points).  xb / yb need not be packed: member m's batch starts at
 * xb + m*x_member_stride and yb + m*y_member_stride (floats), so a slice [:, b0:b1] of the epoch's
 * shuffled [M,N,F] / [M,N] tensors (the DataLoader batches of :122-128) is passed in place.
 * loss_acc: optional fp64 [M]; loss_acc[m] += (double)loss[m] (the epoch_loss of :181), else NULL.
 * act: optional scratch of act_floats >= M*S*mcmc_train_act_floats(rows) floats; when given, the forward
 * kernel leaves each tile's activations there (57.7 KB per pair and 32 rows, L2-resident at the reference's
 * batch size) and the backward kernel reloads them instead of recomputing the forward pass - the same
 * numbers, bit for bit (autograd's saved tensors, :178).  NULL or too small: recompute. */
int64_t mcmc_train_act_floats(int64_t rows);
typedef struct mcmc_train_buffers {
    float *mu, *rho, *sigma;    /* [M,P] */
    float *eps, *W, *WT, *dW;   /* [M*S,P] */
    float *gmu, *grho;          /* [M,P] */
    float *adam_m, *adam_v;     /* [M,2P] */
    double* norm_scratch;       /* [M*mcmc_train_norm_blocks()] */
    float *Y, *delta;           /* [M*S,rows], [M,rows] */
    float *loss, *gnorm;        /* [M]; gnorm may be NULL */
    float* act;                 /* optional, see above */
    int64_t act_floats;
} mcmc_train_buffers;
typedef struct mcmc_train_hyper {
    float p_drop, alpha, beta;                           /* dropout (:71), peak_weighted_loss (utils.py:7) */
    float beta2, adam_eps, weight_decay, max_norm;       /* AdamW (:137-141), clip_grad_norm_ (:179) */
} mcmc_train_hyper;
int mcmc_train_step(mcmc_ctx* ctx, const mcmc_train_buffers* buffers, const mcmc_train_hyper* hyper,
                    const float* xb, int64_t x_member_stride, const float* yb, int64_t y_member_stride,
                    int64_t rows, int input_dim, int n_members, int n_samples, int noise_mode,
                    const float* eps_in, uint64_t seed, int mask_mode, const uint8_t* mask0,
                    const uint8_t* mask1, const uint8_t* mask2, uint64_t mask_seed,
                    const uint8_t* active, float lr, float beta1, int64_t step, double* loss_acc,
                    mcmc_stream_t stream);

/* Early stopping of bayesian_ensemble.py:183-193 for all members after an epoch, on the device (so the
 * training loop need not wait for the epoch's losses): a member whose active flag is set either improves
 * best[m] (bad[m] = 0) or counts a bad epoch, and its flag is cleared once bad[m] >= patience.
 * epoch_loss, best: fp64 [M]; bad: int64 [M]; active: uint8 [M] (the flags mcmc_train_step reads). */
int mcmc_train_early_stop(mcmc_ctx* ctx, const double* epoch_loss, double* best, int64_t* bad,
                          uint8_t* active, int n_members, int64_t patience, mcmc_stream_t stream);

/* ---- next row: the Temporal Fusion Transformer leg ------------------------ */
/* models/temporal_fusion.py.  Parameters are ONE flat fp32 vector in the reference module's
 * named_parameters() order (input_projection, feature_grn, temporal_grn, attention, final_grn,
 * output_projection; 67,585 values at 11 features); PT is the same vector with every weight
 * matrix transposed (mcmc_tft_transpose).  Windows are rows start[b] .. start[b]+seq_len-1 of
 * X [n_rows, num_features] (TFTModel.create_sequences :167-203 without materialising them);
 * seq_len <= 32; rows past n_rows (or an invalid start) read as zero.  Only the last timestep's output is produced - what train (:249) and predict
 * (:293) read.  mask_mode: 0 eval, 1 host uint8 keep-masks in the reference's draw order
 * (feature GRN [B,T,64], temporal GRN [B,T,64], attention [B*4,T,T], final GRN [B,T,64]),
 * 2 Philox keyed (seed; window start, row, element). */
int64_t mcmc_tft_param_count(int num_features);
int mcmc_tft_transpose(mcmc_ctx* ctx, const float* P, float* PT, int num_features, mcmc_stream_t stream);
/* TemporalFusionTransformer.forward (:120-153) + [:, -1, 0] -> pred [n_windows]. */
int mcmc_tft_forward(mcmc_ctx* ctx, const float* X, int64_t n_rows, const int32_t* start, int n_windows, int seq_len,
                     int num_features, const float* P, const float* PT, int mask_mode,
                     const uint8_t* mask_feature, const uint8_t* mask_temporal,
                     const uint8_t* mask_attention, const uint8_t* mask_final, uint64_t seed,
                     float p_drop, float* pred, mcmc_stream_t stream);
/* loss.backward() (:252): recomputes the window's forward with the same masks; delta [n_windows]
 * = dLoss/dpred (mcmc_train_loss with n_members = n_samples = 1); dP [n_windows, n_params] gets
 * every window's parameter gradient (all entries written). */
int mcmc_tft_backward(mcmc_ctx* ctx, const float* X, int64_t n_rows, const int32_t* start, int n_windows, int seq_len,
                      int num_features, const float* P, const float* PT, int mask_mode,
                      const uint8_t* mask_feature, const uint8_t* mask_temporal,
                      const uint8_t* mask_attention, const uint8_t* mask_final, uint64_t seed,
                      float p_drop, const float* delta, float* dP, mcmc_stream_t stream);
/* The same pair with autograd's saved tensors instead of the recompute (:249-252: forward, then
 * loss.backward() on what the forward kept): mcmc_tft_forward_save also leaves every window's
 * activations in act [n_windows * mcmc_tft_act_floats()] (124 KB per window), and
 * mcmc_tft_backward_saved reloads them.  Same numbers, bit for bit, as mcmc_tft_forward /
 * mcmc_tft_backward with the same masks. */
int64_t mcmc_tft_act_floats(void);
int mcmc_tft_forward_save(mcmc_ctx* ctx, const float* X, int64_t n_rows, const int32_t* start, int n_windows,
                          int seq_len, int num_features, const float* P, const float* PT, int mask_mode,
                          const uint8_t* mask_feature, const uint8_t* mask_temporal,
                          const uint8_t* mask_attention, const uint8_t* mask_final, uint64_t seed,
                          float p_drop, float* pred, float* act, mcmc_stream_t stream);
int mcmc_tft_backward_saved(mcmc_ctx* ctx, const float* X, int64_t n_rows, const int32_t* start, int n_windows,
                            int seq_len, int num_features, const float* P, const float* PT, int mask_mode,
                            const uint8_t* mask_feature, const uint8_t* mask_temporal,
                            const uint8_t* mask_attention, const uint8_t* mask_final, uint64_t seed,
                            float p_drop, const float* delta, float* dP, const float* act,
                            mcmc_stream_t stream);
/* loss = peak_weighted_loss(pred, yb) (:250-251 -> utils.py:7-28) and loss.backward() (:252) in ONE launch: every
 * window's workgroup repeats the batch's two loss reductions over pred [n_windows] / yb [n_windows] in its prologue and
 * derives its own dLoss/dpred (the arithmetic of mcmc_train_loss with n_members = n_samples = 1, no delta buffer);
 * workgroup 0 stores loss[0] and, if loss_acc is not NULL, adds it to the fp64 epoch sum loss_acc[0] (:257).
 * act: the activations mcmc_tft_forward_save left, or NULL to recompute the forward pass as mcmc_tft_backward does. */
int mcmc_tft_backward_loss(mcmc_ctx* ctx, const float* X, int64_t n_rows, const int32_t* start, int n_windows,
                           int seq_len, int num_features, const float* P, const float* PT, int mask_mode,
                           const uint8_t* mask_feature, const uint8_t* mask_temporal,
                           const uint8_t* mask_attention, const uint8_t* mask_final, uint64_t seed,
                           float p_drop, const float* pred, const float* yb, float alpha, float beta,
                           float* loss, double* loss_acc, float* dP, const float* act, mcmc_stream_t stream);
/* grad = sum over the n_partials rows of dP; clip_grad_norm_(max_norm) (:254); torch's AdamW step
 * (:255) on a flat parameter vector.  norm_scratch: fp64 [mcmc_flat_norm_blocks()]. */
int mcmc_flat_norm_blocks(void);
int mcmc_flat_adamw(mcmc_ctx* ctx, float* param, int64_t n, const float* dP, int n_partials, float* grad,
                    float* adam_m, float* adam_v, double* norm_scratch, float lr, float beta1,
                    float beta2, float adam_eps, float weight_decay, float max_norm, int64_t step,
                    float* gnorm, mcmc_stream_t stream);
/* mcmc_flat_adamw on a TFT parameter vector P that also refreshes the transposed image PT in the same kernel
 * (saves the mcmc_tft_transpose launch of every training step). */
int mcmc_tft_adamw(mcmc_ctx* ctx, float* P, float* PT, int num_features, const float* dP, int n_partials,
                   float* grad, float* adam_m, float* adam_v, double* norm_scratch, float lr, float beta1,
                   float beta2, float adam_eps, float weight_decay, float max_norm, int64_t step,
                   float* gnorm, mcmc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MCMC_HIP_H */
