/* crank_hip.h - C ABI of libcrank_hip.so, the MI355X (gfx950) implementation of the
 * VQ-VAE voice-conversion training step's hot path.
 *
 * The reference (k2kobayashi/crank) has no FFI layer: its step calls torch modules
 * from Python.  Each entry point below replaces one torch-op cluster of that step and
 * cites the reference code it stands in for; INTEGRATION.md shows the ctypes binding
 * a maintainer of the reference would add (crank_amd/_lib.py is that binding).
 *
 * Conventions
 *  - plain C types only: device pointers, sizes, strides; no torch types.
 *  - every function returns 0 on success (CRK_OK) or a CRK_ERR_* code; none throws.
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it; the compute entry
 *    points neither synchronise the device nor allocate: what a net handle needs for a batch
 *    shape is made by crk_net_reserve, outside the step.
 *  - activations are "frames x channels" row-major fp32, frame n = b*T + t, with an
 *    explicit row stride `ld*` in elements (the reference's (B,T,C) tensors as they
 *    are; its internal (B,C,T) transposes disappear).
 */
#ifndef CRANK_HIP_H
#define CRANK_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define CRK_OK 0
#define CRK_ERR_ARG 1
#define CRK_ERR_HIP 2
#define CRK_ERR_UNSUPPORTED 3

/* flags for crk_net_forward / crk_net_backward */
#define CRK_FLAG_PRECISE 1      /* bf16x3 split operands (~fp32 accuracy) instead of plain bf16 */
#define CRK_FLAG_NO_PARAM_GRAD 2 /* skip weight gradients (they would be discarded) */
#define CRK_FLAG_NO_SAVE 4       /* forward only: no backward will follow, do not store per-layer activations */
#define CRK_FLAG_DEFER_WNORM 8   /* backward only: leave the weight-norm backward (partial sums -> dg, dv, dbias) to a
                                  * crk_nets_wnorm_bwd call over all nets of the model; `params` / `grads` must stay
                                  * valid until then */

#define CRK_FLAG_SEED_ON_DEVICE 16 /* `seed` is the address of a device-resident uint64 (written by crk_seed_next on the same
                                  * stream) instead of the value itself: nothing per call lives in kernel arguments, so
                                  * a call with dropout can sit in a captured HIP graph and draw fresh masks every replay */
#define CRK_FLAG_FWD_PRECISE 32  /* backward only, without CRK_FLAG_PRECISE: the forward of this call ran with
                                  * CRK_FLAG_PRECISE | CRK_FLAG_BWD_PLAIN; the backward arithmetic is plain bf16 */
#define CRK_FLAG_BWD_PLAIN 64    /* forward only, with CRK_FLAG_PRECISE: the backward of this call will run WITHOUT
                                  * CRK_FLAG_PRECISE (and with CRK_FLAG_FWD_PRECISE) - the "bf16x3f" pairing: split-operand
                                  * forward (losses within 1e-3 of the fp32 reference), plain-bf16 backward.  Lets the forward
                                  * save only what that backward reads, in the layout it reads it */

/* ---- convolutional stacks -----------------------------------------------------
 * Replaces the parallel_wavegan networks the reference instantiates (third-party,
 * un-vendored): ParallelWaveGANGenerator (crank/net/module/vqvae2.py:237-273),
 * ResidualParallelWaveGANDiscriminator (crank/bin/train.py:108-118),
 * ParallelWaveGANDiscriminator (crank/net/module/spkradv.py:49-60,
 * crank/bin/train.py:78-89), all with torch.nn.utils.weight_norm on every conv. */
typedef struct crk_net_desc {
  int kind;          /* 0 gated-residual generator, 1 gated-residual discriminator, 2 plain conv stack */
  int in_ch, out_ch, kernel_size, layers, stacks;
  int res_ch, gate_ch, skip_ch, aux_ch;   /* kinds 0/1: must be 64/128/64; aux_ch <= 0: none */
  int conv_ch;       /* kind 2 hidden width */
  int causal;        /* use_causal_conv */
  int use_bias;
  float slope;       /* LeakyReLU negative slope */
  float dropout;     /* residual-block input dropout (kind 1) */
} crk_net_desc;

void* crk_net_create(const crk_net_desc* desc);
void crk_net_destroy(void* net);
long long crk_net_param_count(void* net);
int crk_net_conv_count(void* net);
/* out[9] = cout, cin, k, off_bias(-1: none), off_g, off_v, dilation, role, layer;
 * offsets are element offsets into the net's flat fp32 parameter block; weight_v is
 * stored (cout, cin, k) like torch's Conv1d.weight_v, weight_g (cout). */
int crk_net_conv_info(void* net, int i, long long* out9);
long long crk_net_saved_bytes(void* net, int B, int T);
/* The handle's own device memory for batch shape (B, T) - the backward's gradient planes, the weight-gradient partial sums
 * (crk_net_scratch_bytes of them) and the per-shape descriptor tables - allocated HERE, once per shape and outside the step
 * (it calls hipMalloc / hipMemcpy: a device-wide synchronisation; not inside a stream capture).  crk_net_forward /
 * crk_net_backward* never allocate: at a shape that was not reserved (and that the buffers of a larger reserved shape with
 * the same slot counts do not happen to cover) they launch nothing and return CRK_ERR_ARG.  Buffers outgrown by a later
 * reserve stay alive until crk_net_destroy (a captured HIP graph may still hold them). */
int crk_net_reserve(void* net, int B, int T);
long long crk_net_scratch_bytes(void* net, int B, int T);
/* Optional: run the weight gradients of crk_net_backward on `stream` (null: on the call's stream).
 * They only read buffers the data-gradient chain has finished with, so they overlap the next
 * stack's chain.  The caller must make its consumers of `grads` (optimizer, all-reduce) and the
 * release of `saved` wait for that stream; the library orders its own buffers itself. */
int crk_net_set_wgrad_stream(void* net, void* stream);
/* y[N,out_ch] = net(x[N,in_ch], c[N,aux_ch]); `saved` (crk_net_saved_bytes) keeps what
 * the backward needs; `version` changes whenever `params` was modified. */
int crk_net_forward(void* net, const float* params, unsigned long long version, const float* x, int ldx,
                    const float* c, int ldc, float* y, int ldy, float* saved, int B, int T, int flags,
                    unsigned long long seed, void* stream);
/* accumulates parameter gradients into `grads` (same layout as params) and writes the
 * input gradients dx (scaled by dx_scale: gradient reversal of spkradv.py:63-72 is
 * dx_scale = -lambda) and dc (aux) when non-null. */
int crk_net_backward(void* net, const float* params, unsigned long long version, float* grads, const float* x,
                     int ldx, const float* c, int ldc, const float* dy, int lddy, float* dx, int lddx, float dx_scale,
                     float* dc, int lddc, const float* saved, int B, int T, int flags, unsigned long long seed,
                     void* stream);
/* crk_net_backward of dy * (dy_num[0] / dy_den[1]), the factor read on the device: the backward of a mean cross entropy
 * (crank/net/trainer/utils.py:26, trainer_vqvae.py:177-198) on the net's output - dy = softmax - onehot as crk_ce_fwd wrote
 * it, dy_num = upstream gradient, dy_den = crk_ce_fwd's out2 {loss, count} - without a scaling launch in between.  Chains
 * of plain convs (kind 2) on the fused path only: CRK_ERR_UNSUPPORTED otherwise (crk_ce_bwd, then crk_net_backward). */
int crk_net_backward_scaled(void* net, const float* params, unsigned long long version, float* grads, const float* x,
                            int ldx, const float* c, int ldc, const float* dy, int lddy, float* dx, int lddx, float dx_scale,
                            float* dc, int lddc, const float* saved, int B, int T, int flags, unsigned long long seed,
                            const float* dy_num, const float* dy_den, void* stream);

/* Dropout seeds on the device (D's ResidualBlock dropout, crank/bin/train.py:114; torch draws its Philox offsets on the
 * host, which a captured graph would freeze): *out = mix(*state), *state advances.  One thread; `state` is a uint64 the
 * caller seeds once (per net), `out` the uint64 a forward call and its backward both receive (CRK_FLAG_SEED_ON_DEVICE). */
int crk_seed_next(unsigned long long* state, unsigned long long* out, void* stream);

/* The sub-nets of one model (the generator has four stacks) share an optimizer step; these do for all of them in
 * ONE launch what the per-net calls do in one launch each.  crk_nets_wnorm_bwd: the weight-norm backward of every net
 * with one pending (CRK_FLAG_DEFER_WNORM); must precede any reader of the gradients.  crk_nets_prepare: the weight
 * preparation of every net whose (params[i], version) changed - otherwise crk_net_forward / _backward prepare their
 * net on their first call after the change.  bump_step (may be NULL): an Adam step count to advance by one in the same
 * launch (crk_adam_step, clear_grads bit 1). */
int crk_nets_wnorm_bwd(int n_nets, void* const* nets, void* stream);
int crk_nets_prepare(int n_nets, void* const* nets, const float* const* params, unsigned long long version,
                     float* bump_step, void* stream);

/* ---- VQ codebook (crank/net/module/vqvae2.py:286-347) --------------------------- */
/* Quantizer.vq + lookup + straight-through value: idx[n] = argmin_k ||x_n - w_k||^2
 * (reference fp32 expression, first index on ties), e = W[idx], qx = x + (e - x). */
int crk_vq_forward(const float* x, int ldx, const float* codebook, int N, int D, int K, long long* idx, float* e,
                   int lde, float* qx, int ldq, void* stream);
/* crk_vq_forward with the quantizer's surroundings in the same launch (D = 64, K <= 512; CRK_ERR_UNSUPPORTED otherwise -
 * compose the separate entry points then): add (NULL: none) - the quantizer's input is x + add ("enc[n] + dec",
 * crank/net/module/vqvae2.py:177), written to xsum (NULL: not kept); commit_out2 (NULL: none) = {mean over the frames
 * mask selects (NULL: all) of (input - e)^2, element count}: the commitment loss of trainer_vqvae.py:227-237, exactly what
 * crk_masked_loss_fwd(input, e, mask, mode 1) returns up to summation order; scratch: crk_loss_scratch_floats() floats.
 * xsum may alias x or add (the sum formed in place): every output is that of the inputs as they were passed. */
int crk_vq_forward_fused(const float* x, int ldx, const float* add, int ldadd, float* xsum, int ldsum, const float* codebook,
                         int N, int D, int K, long long* idx, float* e, int lde, float* qx, int ldq,
                         const unsigned char* mask, float* commit_out2, float* scratch, const void* image, void* stream);
/* What the search derives from the codebook alone (Quantizer.vq's `w ** 2` term and the operands of its `x @ w.t()`,
 * crank/net/module/vqvae2.py:338-347, in the search kernel's form: split-f16 fragment planes, squared norms, per-code scales),
 * prepared ONCE per codebook update instead of by every workgroup of every call.  crk_vq_image_bytes: size of the caller-owned
 * buffer (0: shape without an image - pass image = NULL).  crk_vq_image_build_multi: up to 4 codebooks in one launch (the
 * quantizers of a forward).  The caller keeps the image valid: rebuild after ANY write to the codebook (EMA blend,
 * load_state_dict, optimizer step of a trained codebook).  crk_vq_forward_fused(image = NULL) derives everything per call as
 * before; indices, e and qx are identical either way. */
long long crk_vq_image_bytes(int K, int D);
int crk_vq_image_build_multi(int nq, const float* const* codebooks, const int* K, int D, void* const* images, void* stream);
/* vqvae2.py:316-321: counts[K] (int32) and sums[D][K] (int64, 2^-28 fixed point: integer
 * sums are exact and order independent).  `scratch` holds per-chunk partial tables
 * (crk_vq_ema_scratch_bytes; -1: unsupported K).  Under data parallelism all-reduce counts
 * and sums between stats and apply. */
long long crk_vq_ema_scratch_bytes(int N, int D, int K);
int crk_vq_ema_stats(const float* x, int ldx, const long long* idx, int N, int D, int K, int* counts, long long* sums,
                     void* scratch, void* stream);
/* vqvae2.py:316-330: EMA blend, Laplace smoothing (stored back), codebook refresh.
 * ema_w is (D,K), codebook (K,D), like the reference buffers. */
int crk_vq_ema_apply(const int* counts, const long long* sums, float* ema_size, float* ema_w, float* codebook, int D,
                     int K, double decay, double eps, void* stream);
/* The same update for every quantizer of a generator forward (vqvae2.py:171-190 calls the quantizers one after the
 * other; nothing reads a codebook between them) with fewer launches than four per quantizer:
 * crk_vq_ema_partial per quantizer (per-chunk tables into scratch[q]), ONE crk_vq_ema_reduce_multi (tables ->
 * counts[q], sums[q]; all-reduce them here under data parallelism), ONE crk_vq_ema_apply_multi (a size launch and a
 * blend launch for all quantizers).  nq <= 4. */
int crk_vq_ema_partial(const float* x, int ldx, const long long* idx, int N, int D, int K, void* scratch, void* stream);
int crk_vq_ema_reduce_multi(int nq, const void* const* scratch, const int* N, const int* D, const int* K,
                            int* const* counts, long long* const* sums, void* stream);
int crk_vq_ema_apply_multi(int nq, const int* const* counts, const long long* const* sums, float* const* ema_size,
                           float* const* ema_w, float* const* codebook, const int* D, const int* K, double decay,
                           double eps, void* stream);
/* Fewer launches for the same update in a single process (nothing is all-reduced between the steps): the per-chunk tables
 * of all quantizer calls of a forward in one launch (crk_vq_ema_partial_multi, <= 4 calls), tables -> counts / sums AND the
 * cluster-size update (the first half of crk_vq_ema_apply_multi, same arithmetic) in one launch
 * (crk_vq_ema_reduce_size_multi), then the blend (crk_vq_ema_blend_multi): 3 launches per generator forward instead of 5. */
int crk_vq_ema_partial_multi(int nq, const float* const* x, const int* ldx, const long long* const* idx, const int* N,
                             const int* D, const int* K, void* const* scratch, void* stream);
int crk_vq_ema_reduce_size_multi(int nq, const void* const* scratch, const int* N, const int* D, const int* K,
                                 int* const* counts, long long* const* sums, float* const* ema_size, double decay, double eps,
                                 void* stream);
int crk_vq_ema_blend_multi(int nq, const long long* const* sums, float* const* ema_size, float* const* ema_w,
                           float* const* codebook, const int* D, const int* K, double decay, void* stream);
/* crk_vq_ema_blend_multi that also leaves each codebook's search image (crk_vq_image_bytes(K, 64) bytes, see
 * crk_vq_image_build_multi) current - the bytes crk_vq_image_build_multi would derive from the blended codebook - in the same
 * launch; nq <= 4, D = 64 and K <= 512 for every quantizer (CRK_ERR_UNSUPPORTED otherwise: blend, then build). */
int crk_vq_ema_blend_image_multi(int nq, const long long* const* sums, float* const* ema_size, float* const* ema_w,
                                 float* const* codebook, const int* D, const int* K, double decay, void* const* images,
                                 void* stream);

/* ---- losses ----------------------------------------------------------------------- */
/* `scratch` of the loss entry points: crk_loss_scratch_floats() floats; calls on one stream may share it. */
int crk_loss_scratch_floats(void);
/* masked mean of |x-y| (mode 0) or (x-y)^2 (mode 1): CustomFeatureLoss l1/mse
 * (crank/net/module/loss.py:30-47) and the masked_select + MSELoss pairs of
 * trainer_vqvae.py:227-237, trainer_lsgan.py:154-170 (y == NULL: constant target).
 * mask: one byte per frame or NULL.  out2 = {mean, element count}. */
int crk_masked_loss_fwd(const float* x, int ldx, const float* y, int ldy, float yconst, const unsigned char* mask,
                        long long N, int D, int mode, float* out2, float* scratch, void* stream);
int crk_masked_loss_bwd(const float* x, int ldx, const float* y, int ldy, float yconst, const unsigned char* mask,
                        long long N, int D, int mode, const float* stat2, const float* gout, float* dx, int lddx,
                        float* dy, int lddy, void* stream);
/* L1 and MSE means of the same (x, y, mask) in one pass: out4 = {L1 mean, count, MSE mean, count}; each half is a
 * stat2 for crk_masked_loss_bwd with the matching mode (the trainers take both of the decoded features,
 * trainer_vqvae.py:215-225) */
int crk_masked_loss_both_fwd(const float* x, int ldx, const float* y, int ldy, const unsigned char* mask, long long N,
                             int D, float* out4, float* scratch, void* stream);
/* the same with dx = add * add_scale[0] + (loss gradient): a second gradient of x (add, row stride ldadd; NULL: none;
 * add_scale: device scalar, NULL = 1) joins in the launch instead of a separate addition - the commitment loss next to
 * the quantizer's straight-through gradient (trainer_vqvae.py:227-237 + vqvae2.py:343-347), the L1 loss of the decoded
 * features next to the unit gradient of the STFT loss (trainer_vqvae.py:215-225) */
int crk_masked_loss_bwd_acc(const float* x, int ldx, const float* y, int ldy, float yconst, const unsigned char* mask,
                            long long N, int D, int mode, const float* stat2, const float* gout, float* dx, int lddx,
                            float* dy, int lddy, const float* add, int ldadd, const float* add_scale, void* stream);
/* Backward of the commitment loss (trainer_vqvae.py:227-237: mse of the quantizer's input x against e.detach(), masked
 * mean; stat2 / gout as for crk_masked_loss_bwd, mode mse) where the other gradients of the same tensors meet it
 * (vqvae2.py:171-190): t = (a1 + a2) + loss gradient -> dsum (the gradient of the tensor that was added to x inside the
 * quantizer op; NULL: not wanted), t + a3 -> dx.  a1: the straight-through gradient of qx, a2: a second consumer of qx
 * (the last decoder's concatenation), a3: a second consumer of x (the speaker-adversarial net, spkradv.py:74-76); each
 * optional, each with its own row stride.  The sums are the ones autograd's accumulation makes, bit for bit.
 * CRK_ERR_UNSUPPORTED unless D, the strides and the pointers are multiples of 4 floats. */
int crk_vq_commit_bwd(const float* x, int ldx, const float* e, int lde, const unsigned char* mask, long long N, int D,
                      const float* stat2, const float* gout, float* dx, int lddx, float* dsum, int ldsum,
                      const float* a1, int ld1, const float* a2, int ld2, const float* a3, int ld3, void* stream);
/* nn.CrossEntropyLoss(ignore_index) over frames (crank/net/trainer/utils.py:26). */
int crk_ce_fwd(const float* logits, int ldl, const long long* target, long long N, int C, int ignore_index,
               float* out2, float* dlogits_unscaled, float* scratch, void* stream);
int crk_ce_bwd(float* dlogits_unscaled, long long N, int C, const float* stat2, const float* gout, float* dlogits,
               void* stream);
/* one resolution of the STFT-magnitude loss along the frame axis
 * (crank/net/module/loss.py:50-114); n_fft/hop_length/win_length as torch.stft
 * receives them (i.e. after the reference's argument shuffle). */
int crk_stft_loss_fwd(const float* x, int ldx, const float* y, int ldy, int B, int T, int D, int n_fft,
                      int hop_length, int win_length, const float* window, float logratio, float weight,
                      int accumulate, float* out1, float* scratch, void* stream);
int crk_stft_loss_bwd(const float* x, int ldx, const float* y, int ldy, int B, int T, int D, int n_fft,
                      int hop_length, int win_length, const float* window, float logratio, float weight,
                      const float* gout, float* dx, int lddx, void* stream);

/* every resolution of MultiSizeSTFTLoss (loss.py:88-114) in one launch per direction: out1[0] = mean over resolutions.
 * CRK_ERR_UNSUPPORTED (win_length > 64 or more than 4 resolutions): loop over the single-resolution entry points. */
int crk_stft_loss_multi_fwd(const float* x, int ldx, const float* y, int ldy, int B, int T, int D, int nres,
                            const int* n_fft, const int* hop_length, const int* win_length, const float* const* windows,
                            float logratio, float* out1, float* scratch, void* stream);
/* loss and gradient in one pass: dx_unit (zero-initialised) += d out1 / d x; the backward of the loss is then
 * upstream gradient * dx_unit (no second evaluation of the DFTs) */
int crk_stft_loss_multi_fwd_grad(const float* x, int ldx, const float* y, int ldy, int B, int T, int D, int nres,
                                 const int* n_fft, const int* hop_length, const int* win_length,
                                 const float* const* windows, float logratio, float* out1, float* dx_unit, int lddx,
                                 float* scratch, void* stream);
int crk_stft_loss_multi_bwd(const float* x, int ldx, const float* y, int ldy, int B, int T, int D, int nres,
                            const int* n_fft, const int* hop_length, const int* win_length, const float* const* windows,
                            float logratio, const float* gout, float* dx, int lddx, void* stream);

/* L1 mean, MSE mean and the multi-resolution STFT loss of the decoded features against their target - the three terms
 * trainer_vqvae.py:215-225 (calculate_vqvae_loss) forms on the same pair with CustomFeatureLoss / MultiSizeSTFTLoss
 * (crank/net/module/loss.py:18-114) - as ONE launch (+ a finishing one) per direction, without atomics:
 *   out5 = {L1 mean, count, MSE mean, count, STFT loss};
 *   grad (crk_recon_grad_floats floats; NULL: no gradient wanted) = d STFT loss / d x for an upstream gradient of 1, in a
 *   compact per-frame layout that only crk_recon_loss_bwd reads;
 *   tables[r] = crk_stft_twiddles(n_fft[r], win_length[r], hann window of resolution r), crk_stft_twiddle_floats floats,
 *   built once per criterion.
 * Geometry: crk_recon_supported (hop_length >= win_length + 3: frames that do not overlap - what quirk Q1 of
 * MultiSizeSTFTLoss makes of the default stft_params; win_length <= 64; <= 4 resolutions); otherwise
 * CRK_ERR_UNSUPPORTED and the caller uses crk_masked_loss_both_fwd + crk_stft_loss_multi_*.
 * crk_recon_loss_bwd: dx = g1[0] dL1 + g2[0] dMSE + g3[0] dSTFT (device scalars; NULL = term not differentiated). */
int crk_recon_supported(int T, int nres, const int* n_fft, const int* hop_length, const int* win_length);
long long crk_stft_twiddle_floats(int n_fft, int win_length);
int crk_stft_twiddles(int n_fft, int win_length, const float* window, float* table, void* stream);
long long crk_recon_grad_floats(int B, int T, int D, int nres, const int* hop_length, const int* win_length);
int crk_recon_loss_fwd(const float* x, int ldx, const float* y, int ldy, const unsigned char* mask, int B, int T, int D,
                       int nres, const int* n_fft, const int* hop_length, const int* win_length,
                       const float* const* tables, float logratio, float* out5, float* grad, float* scratch,
                       void* stream);
int crk_recon_loss_bwd(const float* x, int ldx, const float* y, int ldy, const unsigned char* mask, int B, int T, int D,
                       int nres, const int* n_fft, const int* hop_length, const int* win_length, const float* out5,
                       const float* grad, const float* g1, const float* g2, const float* g3, float* dx, int lddx,
                       void* stream);

/* The trainers' loss totals ("loss[G] += alpha * term", crank/net/trainer/basetrainer.py:200-206 and the
 * _parse_*_loss functions of trainer_vqvae.py / trainer_lsgan.py / trainer_cyclegan.py): out[0] = sum_i weights[i] *
 * terms[i][0] + constant over <= 16 device scalars in one launch; backward grads[i] = weights[i] * gout[0]. */
int crk_weighted_sum(int n, const float* const* terms, const float* weights, float constant, float* out, void* stream);
int crk_weighted_sum_bwd(int n, const float* weights, const float* gout, float* grads, void* stream);

/* ---- optimiser / glue -------------------------------------------------------------- */
/* torch.optim.Adam defaults on one flat block (crank/net/trainer/utils.py:40-58);
 * lr_dev[0] and step_dev[0] live in device memory (no host sync, graph friendly).  clear_grads bit 0: the gradient
 * block is zeroed as it is consumed (optimizer.zero_grad() of the next step without a memset launch); bit 1: the step
 * count is NOT advanced by this call - the caller hands step_dev to the crk_nets_prepare call that follows the update
 * (one launch less per optimizer step). */
int crk_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, long long n, const float* lr_dev,
                  float* step_dev, float beta1, float beta2, float eps, int clear_grads, void* stream);
/* torch_optimizer.RAdam(lr) (crank/net/trainer/utils.py:44-45; the package is absent from the reference tree: its
 * published update, Liu et al. Alg. 2 - betas (0.9, 0.999), eps 1e-8, no weight decay, rectified adaptive update where the
 * approximated SMA length N_sma >= 5, momentum-only update below).  Arguments and clear_grads bits as crk_adam_step;
 * the betas and eps are the host's doubles (the packages form 1 - beta in double before the fp32 arithmetic). */
int crk_radam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, long long n, const float* lr_dev,
                   float* step_dev, double beta1, double beta2, double eps, int clear_grads, void* stream);
/* pytorch_lamb.Lamb(lr) (crank/net/trainer/utils.py:46-47; absent third party: its published update, You et al. Alg. 2 as
 * that package states it - betas (0.9, 0.999), eps 1e-6, no weight decay, no bias correction, per parameter tensor
 * trust ratio clamp(||w||, 0, 10) / ||m / (sqrt(v) + eps)||, 1 where a norm is 0).  The caller describes the block's
 * parameter tensors once: tiles[4 * t] = {offset, length <= crk_lamb_tile(), tensor, 0} (no tile crosses a tensor, the
 * tiles of a tensor are consecutive; 16-byte aligned), tensors[2 * s] = {first tile, tiles}; upd: one float per element of
 * the block, part: 2 * n_tiles floats (both
 * caller-owned scratch), ratio_out (may be NULL): the trust ratio of every tensor.  clear_grads bits as crk_adam_step;
 * the gradient is cleared over the tiles. */
int crk_lamb_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, float* upd, const int* tiles, int n_tiles,
                  const int* tensors, int n_tensors, float* part, float* ratio_out, const float* lr_dev, float* step_dev,
                  double beta1, double beta2, double eps, int clear_grads, void* stream);
int crk_lamb_tile(void);
/* out[n,:] = [a[n,:ca] | b[n,:cb] | table[idx[n],:E]] (vqvae2.py:154-158,
 * trainer_lsgan.py:194-206) and the embedding-table gradient. */
int crk_concat_embed(const float* a, int lda, int ca, const float* b, int ldb, int cb, const float* table, int E,
                     const long long* idx, long long N, float* out, int ldo, void* stream);
/* dtable[r,:] += sum of dcat[n, c0:c0+E] over frames with idx[n] == r; fixed summation order
 * (bit-reproducible); scratch: crk_embed_bwd_scratch_floats floats. */
long long crk_embed_bwd_scratch_floats(long long N, int E, int n_rows);
int crk_embed_bwd(const float* dcat, int ld, int c0, int E, const long long* idx, long long N, int n_rows,
                  float* dtable, float* scratch, void* stream);
/* The same two with frames [j * run, (j + 1) * run) all carrying the label idx[j * run]: the reference overwrites every
 * frame's label with its utterance's first (basetrainer.py:303-308 "h[:, :] = h[:, 0:1]", trainer_lsgan.py:202-204)
 * before the lookup; with run = frames per utterance the lookup reads the label tensor as the batch holds it (-100 pads
 * behind the first frame included) and no filled copy is made.  run = 1: one label per frame. */
int crk_concat_embed_run(const float* a, int lda, int ca, const float* b, int ldb, int cb, const float* table, int E,
                         const long long* idx, long long run, long long N, float* out, int ldo, void* stream);
int crk_embed_bwd_run(const float* dcat, int ld, int c0, int E, const long long* idx, long long run, long long N,
                      int n_rows, float* dtable, float* scratch, void* stream);

/* ---- on-the-fly log-mel front end (crank/net/module/mlfb.py:134-171, use_raw) ------ */
/* raw[B, n_samples] -> logmel[B*T, n_mels]; STFT n_fft with a win_length window, |.|, mel
 * matvec, clamp(eps), log10, optional (x-mean)/std.  center = 0: frame t starts at sample
 * t*hop (the training step, vqvae2.py:60); center = 1: frame t is centred on t*hop over the
 * reflect-padded signal, T = 1 + n_samples / hop (the offline extraction,
 * crank/feature/feature.py:126-145 -> parallel_wavegan logmelfilterbank). */
int crk_logmel_fwd(const float* raw, int ld_raw, int B, int n_samples, int T, int n_fft, int hop, int win_length,
                   const float* window, const float* mel_basis /* [n_bins][n_mels] */, int n_mels, float eps,
                   const float* mean, const float* std, float* out, int ldo, int center, void* stream);

/* ---- batch assembly from an HBM-resident corpus -----------------
 * Replaces the numpy work of BaseDataset.__getitem__ in the reference's DataLoader workers,
 * torch's default collate and the per-batch host->device copy
 * (crank/net/trainer/dataset.py:58-139 sample dict, :158-198 + :239-258 pad / crop rule,
 * :229-236 speaker codes, :288-293 convert_f0).  The corpus lives in HBM once: every
 * utterance's frames packed one after the other, frame-major fp32 rows. */

/* sklearn StandardScaler.transform (inverse = 0: fl32(fl64(fl32(fl64(x) - mean)) / scale)) and
 * inverse_transform (inverse = 1: fl32(fl64(fl32(fl64(x) * scale)) + mean)) exactly as sklearn
 * evaluates them on a float32 array with float64 statistics (dataset.py:146-150,
 * basetrainer.py:341-345).  x[N, D] row stride ldx -> y[N, D] row stride ldy; y may be x.
 * mean / scale: D doubles in device memory. */
int crk_scaler_apply(const float* x, int ldx, float* y, int ldy, long long N, int D, const double* mean,
                     const double* scale, int inverse, void* stream);

#define CRK_COLLATE_MAX_STREAMS 8
/* one continuous feature of the batch dict: columns [col0, col0 + ncols) of the packed
 * rows src[F_total, ld] -> dst (B, T, ncols), tail-padded with 0.0 */
typedef struct crk_collate_stream {
  const float* src;
  int ld, col0, ncols;
  float* dst;
} crk_collate_stream;
typedef struct crk_collate_desc {
  int n_streams;
  crk_collate_stream streams[CRK_COLLATE_MAX_STREAMS];
  const long long* utt_start; /* [n_utt + 1] first packed frame of each utterance */
  const int* utt_spk;         /* [n_utt] speaker index (position in scp["train"]["spkrs"]) */
  int n_utt, n_spk;
  const float* lcf0_raw;        /* [F_total] unscaled continuous log-F0, for cv_lcf0 (null: not produced) */
  const double* spk_lcf0_mean;  /* [n_spk] scaler[spkr]["lcf0"].mean_ */
  const double* spk_lcf0_std;   /* [n_spk] sqrt(scaler[spkr]["lcf0"].var_) */
  const float* raw;             /* use_raw: packed waveforms [sum samples] (null: none) */
  const long long* raw_start;   /* [n_utt + 1] first sample of each utterance */
  int fftl, hop;                /* conf["feature"]["fftl"], ["hop_size"] */
} crk_collate_desc;
/* picks (device, int32 [3][B]): utterance index, first kept frame p (used when the utterance is
 * longer than T; the reference draws it with random.choice, dataset.py:161) and conversion-target
 * speaker (dataset.py:84-86) of every batch row.  Outputs, each optional (null: skip):
 * cv_lcf0 (B,T) fp32 = convert_f0 in float64 rounded once, 0 on padding; org_h / cv_h (B,T) int64
 * with -100 on padding; one-hot codes (B,T,n_spk) fp32; mask (B,T) bytes 1/0; flen (B) int64 =
 * the utterance's own length (also when cropped); raw_out (B, fftl + hop*T - 1) fp32 = the waveform
 * padded / cropped like padding_raw (dataset.py:261-285), incl. its "no left padding when the crop
 * start is 0" quirk. */
int crk_collate_batch(const crk_collate_desc* desc, const int* picks, int B, int T, float* cv_lcf0,
                      long long* org_h, long long* cv_h, float* org_onehot, float* cv_onehot,
                      unsigned char* mask, long long* flen, float* raw_out, void* stream);

/* ---- decode-side F0 post-processing ------------------------------
 * BaseTrainer._store_features / _get_cvf0 (crank/net/trainer/basetrainer.py:311-320, :372-385):
 * lcf0 (B,T) fp32 normalised by the global lcf0 scaler -> inverse scaler (float32, as sklearn) ->
 * convert_f0 org_spk[b] -> cv_spk[b] in float64 -> cv_lcf0, f0 = exp(cv_lcf0) * uv and
 * normed_lcf0 = (cv_lcf0 - mean) / scale, all (B,T) float64, each optional.
 * has_lcf0_scaler = 0: "lcf0" is listed in ignore_scaler.  The spectral features use
 * crk_scaler_apply(inverse = 1). */
int crk_decode_f0(const float* lcf0, const float* uv, int B, int T, const int* org_spk, const int* cv_spk,
                  double lcf0_mean, double lcf0_scale, int has_lcf0_scaler, const double* spk_lcf0_mean,
                  const double* spk_lcf0_std, double* cv_lcf0, double* f0, double* normed_lcf0, void* stream);

/* ---- MCD evaluation with FastDTW alignment --------------------------
 * crank/bin/evaluate_mcd.py:61-77 for P utterance pairs at once: cv / gt are the voiced-frame
 * mel-cepstra (float64, [sum n][D] packed, pair p owns rows off[p]..off[p+1]); the warping path is
 * the third-party `fastdtw(cv, gt, dist=euclidean)` of the reference (radius 1 by default;
 * restated from its published algorithm, oracle/mcd.py) and mcd[p] = mean over the path of
 * 10 / ln 10 * sqrt(2 * sum_d (cv - gt)^2).  path_out (optional): path_stride ints per pair,
 * (i, j) pairs from start to end; path_len[p] = number of pairs.  status[p] = 1: pair too long
 * for the scratch / LDS sizing (max_nx, max_ny must bound every pair).  One wavefront per pair. */
long long crk_mcd_scratch_bytes(int P, int max_nx, int max_ny, int D, int radius);
int crk_mcd_fastdtw(const double* cv, const long long* cv_off, const double* gt, const long long* gt_off, int P, int D,
                    int radius, int max_nx, int max_ny, double* mcd, int* path_len, int* path_out,
                    long long path_stride, void* scratch, int* status, void* stream);

/* ---- measurement -------------------------------------------------------------------
 * HIP-event timing of the conv kernels on their launch stream (bench.py's roofline leg),
 * one class per kernel: 0 conv_tile_kernel (generic per-layer conv), 1 stack_fwd_kernel,
 * 2 stack_bwd_kernel, 3 wgrad_kernel (table), 4 pstack_kernel, 5 stack_wgrad_kernel,
 * 6 pstack_wgrad_kernel, 7 the VQ codebook search (crk_vq_forward / _fused; bytes = 520 B per frame), 8 logmel_kernel
 * (crk_logmel_fwd).  crk_prof_enable(1) resets and starts recording, crk_prof_report
 * synchronises on the recorded events. */
int crk_prof_enable(int on);
int crk_prof_report(int cls, long long* count, double* total_ms, double* total_flops);
/* summed ALGORITHMIC HBM bytes of the class's launches (inputs read once + outputs and saved planes written
 * once), for the bandwidth side of the roofline */
int crk_prof_report_bytes(int cls, double* total_bytes);

/* The nearest-code search of crk_vq_forward / crk_vq_forward_fused (D = 64, K <= 512) runs on the f16 matrix pipe with
 * split operands and re-scores, with the exact fp32 chain, the frames whose two best candidates it cannot separate
 * (vq_kernels.hip: vq_forward_f16_kernel; indices identical to the exact search by construction).
 * crk_debug_vq_set_f16(0) selects the exact fp32-MFMA search instead (A/B runs, the equality test);
 * crk_debug_vq_flags: frames decided by [1] the two-candidate re-scoring, [2] the full exact scan since the last reset. */
int crk_debug_vq_set_f16(int on);
int crk_debug_vq_flags(unsigned long long* host_out3, int reset);

/* Measurement aid: bytes > 0 puts a read-modify-write pass over a private buffer of that size (choose > 256 MiB, the
 * Infinity Cache) in front of every conv-stack kernel launch from here on, 0 removes it again.  A/B runs only: with the
 * pass in place no kernel finds its producer's output in a cache (tools/mall_ab.sh). */
int crk_debug_flush_before(long long bytes);

/* number of device allocations net handles have made since the library was loaded (tests pin "none inside the step") */
long long crk_debug_alloc_count(void);

/* which kernel generation crk_net_forward / crk_net_backward pick for batch shape (B, T): bit 0 a generator stack (kind 0) runs
 * the channel-split kernels in plain bf16, bit 1 its CRK_FLAG_PRECISE | CRK_FLAG_BWD_PLAIN forward runs the channel-split
 * split-operand kernel, bit 2 a discriminator (kind 1) runs channel-split, bit 3 a chain of plain convs (kind 2) runs fused.
 * The fallbacks compute the same values more slowly; tests pin the bits at the benchmark shape. */
int crk_debug_net_paths(void* net, int B, int T);

const char* crk_version(void);

#ifdef __cplusplus
}
#endif
#endif
