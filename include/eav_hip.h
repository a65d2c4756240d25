/* libeav_hip.so - C ABI of the MI355X (gfx950) kernels behind the EAV trainer API.
 *
 * The reference (nubcico/EAV) has no native/FFI layer: its hot path is Python calling torch ops.
 * Each entry point below therefore names the *reference op call site* it replaces
 * (file:line under /root/reference).  The Python mirror of the reference's trainer classes
 * (eav_amd/eegnet.py, audio.py, vision.py) binds these with ctypes - see INTEGRATION.md.
 *
 * Conventions: every function returns 0 on success or a negative EAV_E* code (message from
 * eav_last_error(), thread-local).  All pointers are caller-owned DEVICE pointers unless said
 * otherwise; nothing is allocated inside; `stream` is a hipStream_t (NULL = default stream);
 * launches are asynchronous.  Tensors are dense, row-major, fp32 unless noted.
 * The only process-wide state are the eav_*_set_* tuning hooks of include/eav_hip_tuning.h (tile shape, resident-block caps;
 * test / benchmark use only, declared apart from this product ABI): plain globals read at
 * launch time, meant to be set once before work is issued (benchmarks / experiments), not synchronised across threads.
 */
#ifndef EAV_HIP_H
#define EAV_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EAV_ABI_VERSION 3

const char* eav_last_error(void);
int eav_abi_version(void);

/* ---- generic ------------------------------------------------------------------------------ */
/* out[i] = scale * sum_p part[p*stride + i], fp64 accumulation in fixed order (deterministic). */
int eav_reduce_partials(const float* part, int nparts, int64_t stride, int n, float scale, float* out, void* stream);

/* nn.BatchNorm2d forward statistics (EEGNet_tor.py:25,29,38).  part[p][0..nch) = sum x,
 * part[p][nch..2nch) = sum x^2.  Writes bn[0..nch)=mean, invstd, scale=gamma*invstd,
 * shift=beta-mean*scale as four separate arrays; training!=0 also updates the running stats
 * (momentum, unbiased variance); training==0 uses the running stats. */
int eav_bn_finalize(const float* part, int nparts, int nch, double count, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, int training, float momentum, float eps, float* mean,
                    float* invstd, float* scale, float* shift, void* stream);
/* BatchNorm backward sums -> dgamma, dbeta and the two means of the input-gradient formula. */
int eav_bn_bwd_finalize(const float* part, int nparts, int nch, double count, int training, float* dgamma,
                        float* dbeta, float* m1, float* m2, void* stream);
/* eav_reduce_partials(part, nparts, stride, n, 1, out) and one or two eav_bn_bwd_finalize jobs (part_b = NULL: one) in ONE
 * launch - the same arithmetic on disjoint block ranges, the same bits (autograd of EEGNet_tor.py:52-54: depthwiseConv.weight
 * and firstBN - and, in the eval-mode step, depthwiseBN - behind eav_eegnet_dw_bwd_fused). */
int eav_reduce_and_bn_bwd_finalize(const float* part, int nparts, int64_t stride, int n, float* out,
                                   const float* part_a, int nparts_a, int nch_a, double count_a, int training_a,
                                   float* dgamma_a, float* dbeta_a, float* m1_a, float* m2_a,
                                   const float* part_b, int nparts_b, int nch_b, double count_b, int training_b,
                                   float* dgamma_b, float* dbeta_b, float* m1_b, float* m2_b, void* stream);
/* weight.data.renorm_(p=2, dim=0, maxnorm) - the max-norm hooks, EEGNet_tor.py:33-34,47-48. */
int eav_renorm_rows(float* w, int rows, int cols, float maxnorm, void* stream);
/* both hooks of EEGNet_tor.forward (depthwiseConv.weight, dense.weight) in one launch */
int eav_renorm_rows2(float* w_a, int rows_a, int cols_a, float* w_b, int rows_b, int cols_b, float maxnorm,
                     void* stream);

/* ---- EEGNet block 1 ----------------------------------------------------------------------- */
/* firstConv forward: nn.Conv2d(1,8,(1,K<=300),padding='same',bias=False), EEGNet_tor.py:24,51.
 * x [B,C,S] -> y1 [B,8,C,S]; stat_part [eav_eegnet_fir_fwd_nparts()][16] = per-filter sum / sum sq. */
int eav_eegnet_fir_fwd_nparts(int B, int C, int S);
int eav_eegnet_fir_fwd(const float* x, const float* w1, float* y1, float* stat_part, int B, int C, int S, int klen,
                       void* stream);
/* No-grad evaluation of block 1 (Trainer_uni.validate, EEGNet_tor.py:118-135; model.eval(): BatchNorm on running
 * statistics, no dropout): x [B,C,S] (or rows xidx of a resident [*,C,S] array; xidx may be NULL) -> p2 [B,64,S/4] =
 * AvgPool(1,4)(ELU(depthwiseBN(depthwiseConv(ELU(firstBN(firstConv(x))))))) in one pass - the FIR output, its ELU and the
 * depthwise output never exist in memory.  bn1 / bn2: mean, invstd, scale, shift (8 / 64 floats each) as eav_bn_finalize
 * writes them in eval mode; w2 = depthwiseConv.weight [64,C].  C <= 32, S % 4 == 0, klen <= 300. */
int eav_eegnet_block1_infer(const float* x, const int64_t* xidx, const float* w1, const float* bn1, const float* w2,
                            const float* bn2, float* p2, int B, int C, int S, int klen, void* stream);
/* firstConv weight gradient fused with firstBN backward (autograd of EEGNet_tor.py:51-52):
 * bn_params = mean, invstd, scale, shift, m1, m2 (8 floats each); part [nparts][8][klen].
 * y1 == NULL: BatchNorm in eval mode (m1 = m2 = 0, the gradient through firstBN is scale * g1; y1 is not read). */
int eav_eegnet_fir_wgrad_nparts(int B, int C, int S);
int eav_eegnet_fir_wgrad(const float* x, const float* y1, const float* g1, const float* bn_params, float* part,
                         int B, int C, int S, int klen, void* stream);
/* The same two kernels with the batch addressed in place: sample b of the batch = row xidx[b] of x [*,C,S] (the
 * trainer's HBM-resident data set) - the per-step batch copy (EEGNet_tor.py:100-101: a host->device copy in the
 * reference, a device gather otherwise) disappears.  xidx: device int64 [B]; NULL = x is the batch. */
/* firstConv and its weight gradient by overlap-save FFT (csrc/eegnet_fir_fft.hip): the same exact-fp32 linear maps as
 * eav_eegnet_fir_fwd / eav_eegnet_fir_wgrad (nn.Conv2d(1, 8, (1, kernLength), padding='same', bias=False) and autograd's
 * weight gradient of it, CNN_torch/EEGNet_tor.py:24,51,109) in ~13 x fewer flops - HBM-bound instead of MFMA-bound; up to
 * eav_eegnet_fir_fft_max_taps() = 321 taps, any C / S.  xidx (optional): the batch is x[xidx[0..B)].
 * fwd: stat_part [eav_eegnet_fir_fwd_fft_nparts(B,C,S)][16] (8 sums, 8 sums of squares per row) for eav_bn_finalize; NULL:
 * no statistics are collected (firstBN in eval mode uses its running statistics).
 * wgrad: dW [8, klen] is WRITTEN (no partials to reduce); ws: eav_eegnet_fir_wgrad_fft_ws_floats(B,C,S) floats;
 * y1 = NULL selects BatchNorm in eval mode (dy = scale g).  Bit-reproducible run to run. */
int eav_eegnet_fir_fft_max_taps(void);
int eav_eegnet_fir_fwd_fft_nparts(int B, int C, int S);
int eav_eegnet_fir_fwd_fft(const float* x, const int64_t* xidx, const float* w1, float* y1, float* stat_part, int B, int C,
                           int S, int klen, void* stream);
int64_t eav_eegnet_fir_wgrad_fft_ws_floats(int B, int C, int S);
int eav_eegnet_fir_wgrad_fft(const float* x, const int64_t* xidx, const float* y1, const float* g1, const float* bn_params,
                             float* ws, float* dW, int B, int C, int S, int klen, void* stream);
int eav_eegnet_fir_fwd_indexed(const float* x, const int64_t* xidx, const float* w1, float* y1, float* stat_part, int B,
                               int C, int S, int klen, void* stream);
int eav_eegnet_fir_wgrad_indexed(const float* x, const int64_t* xidx, const float* y1, const float* g1,
                                 const float* bn_params, float* part, int B, int C, int S, int klen, void* stream);
/* firstBN -> ELU -> depthwiseConv (EEGNet_tor.py:52-54): y1 -> z [B,64,S];
 * stat_part [B*ceil(S/1024)][128]. */
int eav_eegnet_dw_fwd(const float* y1, const float* bn1, const float* w2, float* z, float* stat_part, int B, int C,
                      int S, void* stream);
/* eav_eegnet_dw_fwd that also leaves p2 [B,64,S/4] = AvgPool(1,4)(ELU(depthwiseBN(z))) (EEGNet_tor.py:55-57) when
 * depthwiseBN is in EVAL mode: bn2 = its parameter block as eav_bn_finalize(training = 0) leaves it (scale / shift from the
 * running statistics, known before the launch); no dropout (identity in eval mode).  Saves eav_bn_elu_pool_fwd's pass over
 * z in the eval-mode training steps the reference runs from its second epoch on.  S % 4 == 0. */
int eav_eegnet_dw_fwd_pool_eval(const float* y1, const float* bn1, const float* w2, float* z, float* stat_part,
                                const float* bn2, float* p2, int B, int C, int S, void* stream);
/* backward of the above: g1 [B,8,C,S] = dL/d(firstBN out); stat_part [B*ceil(S/1024)][16];
 * w_part [B*ceil(S/1024)][64*C]. */
int eav_eegnet_dw_bwd(const float* y1, const float* dz, const float* bn1, const float* w2, float* g1,
                      float* stat_part, float* w_part, int B, int C, int S, void* stream);

/* the same with the backward of depthwiseBN -> ELU -> AvgPool(1,4) -> Dropout (EEGNet_tor.py:55-58) folded in: dz is
 * formed from z [B,64,S] and dp2 [B,64,S/4] on the fly (no eav_bn_elu_pool_bwd_apply launch, no dz tensor).
 * bn2 = mean, invstd, scale, shift, m1, m2 of depthwiseBN (64 floats each). */
int eav_eegnet_dw_bwd_fused(const float* y1, const float* z, const float* dp2, const float* bn2, const float* bn1,
                            const float* w2, float* g1, float* stat_part, float* w_part, int B, int C, int S,
                            float drop_p, uint64_t seed, const uint8_t* mask, const uint64_t* seed_dev, void* stream);

/* the same for an eval-mode step (depthwiseBN on its running statistics, what the reference's loop runs from its second
 * epoch on: model.eval() in validate(), EEGNet_tor.py:118, is never undone): dz = scale2 g needs no batch sums first, so the
 * sums themselves (the depthwiseBN weight / bias gradients) leave from this pass - bn2_part [B*ceil(S/1024)][2*64], finished
 * by eav_bn_bwd_finalize(training = 0) AFTER this launch; no eav_bn_elu_pool_bwd_reduce pass; bn2's m1 / m2 are not read. */
int eav_eegnet_dw_bwd_fused_eval(const float* y1, const float* z, const float* dp2, const float* bn2, const float* bn1,
                                 const float* w2, float* g1, float* stat_part, float* w_part, float* bn2_part, int B, int C,
                                 int S, float drop_p, uint64_t seed, const uint8_t* mask, const uint64_t* seed_dev,
                                 void* stream);

/* ---- BN -> ELU -> AvgPool(1,P) -> Dropout (EEGNet_tor.py:55-58, 60-63), P in {4,8} -------- */
/* bn = mean, invstd, scale, shift (CH each).  mask: optional uint8 keep-mask [B,CH,T/P]
 * (NULL = counter-based generator keyed by seed); drop_p = 0 disables dropout; drop_p < 0 = nn.Dropout2d with
 * probability -drop_p: one keep decision per (sample, channel) row (the reference's dropoutType != 'Dropout',
 * EEGNet_tor.py:21) - eav_bn_elu_pool_* and eav_eegnet_dw_bwd_fused. */
/* seed_dev (optional, device uint64): effective seed = seed + 2 * (*seed_dev) - a device-resident step counter,
 * so that a captured hipGraph draws a fresh mask on every replay. */
int eav_bn_elu_pool_fwd(const float* in, const float* bn, float* out, int B, int CH, int T, int P, float drop_p,
                        uint64_t seed, const uint8_t* mask, const uint64_t* seed_dev, void* stream);
int eav_bn_elu_pool_bwd_reduce(const float* dp, const float* u, const float* bn, float* part /*[B][2*CH]*/, int B,
                               int CH, int T, int P, float drop_p, uint64_t seed, const uint8_t* mask,
                               const uint64_t* seed_dev, void* stream);
int eav_bn_elu_pool_bwd_apply(const float* dp, const float* u, const float* bn, const float* m12 /*m1[CH],m2[CH]*/,
                              float* du, int B, int CH, int T, int P, float drop_p, uint64_t seed,
                              const uint8_t* mask, const uint64_t* seed_dev, void* stream);
/* BatchNorm on its running statistics (eval-mode step): du = scale g and the sums of eav_bn_elu_pool_bwd_reduce (part
 * [B][2*CH], to be finished by eav_bn_bwd_finalize(training = 0)) in ONE pass over u and dp. */
int eav_bn_elu_pool_bwd_eval(const float* dp, const float* u, const float* bn, float* du, float* part /*[B][2*CH]*/, int B,
                             int CH, int T, int P, float drop_p, uint64_t seed, const uint8_t* mask,
                             const uint64_t* seed_dev, void* stream);

/* ---- separableConv: dense 64->64, 16 taps, 'same' (EEGNet_tor.py:37,59) ------------------- */
int eav_conv64_prep_weights(const float* w /*[64,64,16]*/, float* wT_fwd /*[1024,64]*/, float* wT_bwd, void* stream);
/* eav_conv64_prep_weights and eav_counter_inc4 (the dropout / num_batches_tracked step counters; NULL = skip) in one
 * launch: the per-step prologue of EEGNet_tor.forward when the direct separableConv kernels run */
int eav_eegnet_step_prologue(const float* w, float* wT_fwd, float* wT_bwd, int64_t* c0, int64_t* c1, int64_t* c2,
                             int64_t* c3, void* stream);
int eav_conv64_ntiles(int T);
/* out[b,o,t] = sum wT[(i*16+k)][o]*in[b,i,t+k-padl]; stat_part (may be NULL) [eav_conv64_fwd_nparts()][128]. */
int eav_conv64_fwd_nparts(int B, int T);
int eav_conv64_fwd(const float* in, const float* wT, float* out, float* stat_part, int B, int T, int padl,
                   void* stream);
/* separableConv and its data gradient in the frequency domain (csrc/eegnet_conv64_fft.hip): the same exact-fp32 linear maps
 * as eav_conv64_fwd with wT_fwd / wT_bwd (nn.Conv2d(64, 64, (1,16), padding='same', bias=False), CNN_torch/EEGNet_tor.py:37,59
 * and autograd's input gradient of it) by overlap-save blocks of 64 samples: per frequency bin the channel mixing is one
 * [128 x 128] real matrix - 6 x fewer multiply-adds than the 1024-deep direct contraction.  w = separableConv.weight
 * [64,64,16] (no prepared copy needed); bwd = 0 forward (stat_part [eav_conv64_fft_nparts(B,T)][128] or NULL), bwd = 1 data
 * gradient, bwd = 2 data gradient re-using the filter spectra the forward call of the same step prepared in ws; ws: eav_conv64_fft_ws_floats(B,T) floats, 16-byte aligned (it keeps the input spectra of both calls for the
 * weight gradient). */
int64_t eav_conv64_fft_ws_floats(int B, int T);
int eav_conv64_fft_nparts(int B, int T);
int eav_conv64_fft_fwd(const float* in, const float* w, float* out, float* stat_part, float* ws, int B, int T, int bwd,
                       void* stream);
/* dW [64,64,16] = d loss / d separableConv.weight, WRITTEN (replaces eav_conv64_wgrad + eav_reduce_partials); needs the
 * forward input's spectra of the same step in ws (eav_conv64_fft_fwd with bwd = 0 was called on it). */
int eav_conv64_fft_wgrad(const float* du, float* dW, float* ws, int B, int T, void* stream);
int eav_conv64_wgrad_nparts(int B, int T);
/* part [nparts][64*64*16]; sum over parts = dL/dW[o,i,k]. */
int eav_conv64_wgrad(const float* du, const float* in, float* part, int B, int T, int padl, void* stream);

/* ---- canonical EEGNet (CNN_torch/CNN_EEG.py:7-67): run-time F1<=16, D<=8, F2<=64, K1<=512, K2<=32 ----------- */
/* block1[0] nn.Conv2d(1,F1,(1,K),padding='same',bias=False) (CNN_EEG.py:22): x [B,C,S] -> y1 [B,F1,C,S];
 * stat_part [eav_tconv_fwd_nparts()][2*F1] = per-filter sum / sum of squares (input of eav_bn_finalize).
 * F1 == 8 with K <= 300 runs the fp32-MFMA Toeplitz kernels of eav_eegnet_fir_*; other shapes a direct kernel. */
int eav_tconv_fwd_nparts(int B, int C, int S, int F1, int K);
int eav_tconv_fwd(const float* x, const float* w, float* y1, float* stat_part, int B, int C, int S, int F1, int K,
                  void* stream);
/* its weight gradient with the block1[1] BatchNorm backward folded in; bn_params = mean, invstd, scale, shift,
 * m1, m2 (F1 each); part [eav_tconv_wgrad_nparts()][F1*K]. */
int eav_tconv_wgrad_nparts(int B, int C, int S, int F1, int K);
int eav_tconv_wgrad(const float* x, const float* y1, const float* g1, const float* bn_params, float* part, int B,
                    int C, int S, int F1, int K, void* stream);
/* block1[1..2]: BatchNorm affine -> depthwise nn.Conv2d(F1,D*F1,(Chans,1),groups=F1) (CNN_EEG.py:23-25):
 * y1 -> z [B,D*F1,S]; stat_part [eav_spatial_nparts()][2*D*F1].  elu != 0 puts an ELU between the BatchNorm and the
 * conv: firstBN -> ELU -> depthwiseConv of EEGNet_tor (EEGNet_tor.py:52-54) for widths the specialised
 * eav_eegnet_dw_* kernels do not cover. */
int eav_spatial_nparts(int B, int S);
int eav_spatial_fwd(const float* y1, const float* bn1, const float* wd, float* z, float* stat_part, int B, int C,
                    int S, int F1, int D, int elu, void* stream);
/* backward: g1 [B,F1,C,S] = dL/d(BN output); stat_part [nparts][2*F1]; w_part [nparts][D*F1*C]. */
int eav_spatial_bwd(const float* y1, const float* dz, const float* bn1, const float* wd, float* g1, float* stat_part,
                    float* w_part, int B, int C, int S, int F1, int D, int elu, void* stream);
/* Dense temporal conv nn.Conv2d(Cin,Cout,(1,K),padding='same',bias=False), K <= 16, channels <= 64: the
 * "separableConv" of EEGNet_tor (EEGNet_tor.py:37,59) at widths other than 64 -> 64 (those run eav_conv64_*).
 * in [B,Cin,T], w [Cout,Cin,K] -> out [B,Cout,T]; stat_part (or NULL) [eav_dconv_fwd_nparts()][2*Cout] = per-channel
 * sum / sum of squares.  transposed != 0: the data gradient - in = dL/dout [B,Cin,T] (Cin = the conv's OUTPUT
 * channels), out = dL/din [B,Cout,T], w = the forward weight [Cin,Cout,K]. */
int eav_dconv_fwd_nparts(int B, int T);
int eav_dconv_fwd(const float* in, const float* w, float* out, float* stat_part, int B, int Cin, int Cout, int T,
                  int K, int transposed, void* stream);
/* its weight gradient: part [B][Cout*Cin*K] (finish with eav_reduce_partials over the B rows). */
int eav_dconv_wgrad(const float* dy, const float* x, float* part, int B, int Cin, int Cout, int T, int K, void* stream);
/* block2[0..1]: depthwise (1,K2) 'same' conv then pointwise 1x1 conv (CNN_EEG.py:35-37): a [B,C2,T] ->
 * d3 [B,C2,T] (kept for the backward) and z [B,F2,T]; stat_part [eav_sepconv_fwd_nparts()][2*F2]. */
int eav_sepconv_fwd_nparts(int B, int T);
int eav_sepconv_fwd(const float* a, const float* wdw, const float* wp, float* d3, float* z, float* stat_part, int B,
                    int C2, int F2, int T, int K2, void* stream);
/* pointwise backward: dd3 [B,C2,T] and w_part [eav_pointwise_bwd_nparts()][F2*C2]. */
int eav_pointwise_bwd_nparts(int B, int T);
int eav_pointwise_bwd(const float* du, const float* d3, const float* wp, float* dd3, float* w_part, int B, int C2,
                      int F2, int T, void* stream);
/* depthwise temporal conv backward: da [B,C2,T] and w_part [B][C2*K2]. */
int eav_dwt_bwd(const float* dd3, const float* a, const float* wdw, float* da, float* w_part, int B, int C2, int T,
                int K2, void* stream);

/* ---- head, loss, optimiser ---------------------------------------------------------------- */
/* nn.Linear + nn.Softmax(dim=1) (EEGNet_tor.py:65-66); logits or probs may be NULL. */
int eav_dense_softmax_fwd(const float* in, const float* w, const float* bias, float* logits, float* probs, int B,
                          int NF, int NC, void* stream);
/* probs != NULL: dout is dL/dprobs (softmax backward applied first); NULL: dout is dL/dlogits. */
int eav_dense_softmax_bwd(const float* dout, const float* probs, const float* in, const float* w, float* dw,
                          float* dbias, float* din, int B, int NF, int NC, void* stream);
/* nn.CrossEntropyLoss (mean) on [B,NC] rows + gradient (din may be NULL); *ncorrect += #argmax hits (may be NULL).
 * A class index outside [0,NC) never indexes anything: it is reported through *bad_label (device int, may be NULL;
 * label+1 for label >= 0, the label itself if negative) - torch raises at this point, the Python wrapper does too. */
int eav_ce_fwd_bwd(const float* in, const int64_t* y, float* loss, float* din, int* ncorrect, int* bad_label, int B,
                   int NC, void* stream);
/* v[i] *= *scalar (device scalar): the upstream gradient applied to the stored d loss / d scores. */
int eav_scale_by_scalar(float* v, const float* scalar, int64_t n, void* stream);
/* torch.optim.Adam (decoupled=0) / AdamW (decoupled=1) update of one flat tensor; step >= 1.
 * step_dev (optional, device int64): take the step count from device memory instead (graph-capturable). */
int eav_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int64_t step, int decoupled, const int64_t* step_dev, void* stream);
/* Batch assembly in HBM: out[i,:] = src[idx[i],:] (rows of row_elems floats) / out[i] = src[idx[i]] (labels).
 * Replaces the per-batch host->device copies of the reference loops (EEGNet_tor.py:100-101). */
int eav_gather_rows(const float* src, const int64_t* idx, float* out, int nrows, int64_t row_elems, void* stream);
int eav_gather_i64(const int64_t* src, const int64_t* idx, int64_t* out, int n, void* stream);
/* *counter += 1 on the stream (device-resident step counters for hipGraph replay). */
int eav_counter_inc(int64_t* counter, void* stream);
int eav_counter_inc4(int64_t* c0, int64_t* c1, int64_t* c2, int64_t* c3, void* stream);   /* distinct counters; NULL = skip */
/* The start of a captured training step in one launch (EEGNet_tor.py:100-104): up to five DISTINCT step counters + 1 (NULL =
 * skip: dropout stream, the three BatchNorm num_batches_tracked, the optimiser's step count) and out[i] = labels[idx[i]] for
 * i < n (n = 0: counters only). */
int eav_step_begin(int64_t* c0, int64_t* c1, int64_t* c2, int64_t* c3, int64_t* c4, const int64_t* labels,
                   const int64_t* idx, int64_t* out, int n, void* stream);

/* ---- AST / ViT encoders (HF ASTForAudioClassification / ViTForImageClassification as called at
 *      Transformer_Audio.py:22,72 and Transformer_Vision.py:29,92) ------------------------------ */
/* Batched fp32 GEMM on the fp32 matrix cores: C[z] = epilogue(alpha * opA(A[z]) . opB(B[z])).
 * transA=0: A is [M,K]; 1: A is [K,M].  transB=0: B is [N,K] (nn.Linear weight layout); 1: B is [K,N].
 * z = zb*heads + zh, operand z at base + zb*s?b + zh*s?h (element strides, multiples of 4).
 * Epilogue: +bias[n]; pre (optional) receives the value before GELU; gelu=1 applies erf-GELU; +resid;
 * accumulate=1 adds into C.  Replaces nn.Linear fwd/bwd, the patch-embedding conv (with eav_im2col),
 * Q.K^T, P.V and the attention backward products. */
int eav_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                 int transA, int transB, int batch, int heads, int64_t sAb, int64_t sAh, int64_t sBb, int64_t sBh,
                 int64_t sCb, int64_t sCh, float alpha, const float* bias, int gelu, float* pre, const float* resid,
                 int ldr, int accumulate, void* stream);
/* Split-K form for weight gradients: C[M,N] = opA(A).opB(B) with the contraction cut into
 * eav_gemm_f32_splitk_plan(M,N,K) slices; ws holds [nsplit][M][N] partials, summed in fixed order. */
int eav_gemm_f32_splitk_plan(int M, int N, int K);
int eav_gemm_f32_splitk(const float* A, const float* B, float* C, float* ws, int M, int N, int K, int lda, int ldb,
                        int transA, int transB, void* stream);
/* fp32-grade GEMM on the fp16 matrix cores with split operands (csrc/gemm_sp.hip) - the same call sites as eav_gemm_f32
 * (nn.Linear forward / data gradient / weight gradient inside HF's ASTLayer / ViTLayer, Transformer_Audio.py:72,
 * Transformer_Vision.py:92).  Operands are "sp16 planes": X[R,K] with the contraction index along K stored as
 * uint16 [R][Kp/8][2][8] (8 hi halves, then 8 lo halves; Kp = eav_sp_kpad(K), zero beyond K) of sigma*X, hi =
 * fp16(sigma x), lo = fp16(sigma x - hi).  A device slot of EAV_SP_SLOT floats per tensor holds 64 shards of the bits of
 * max|x| (one per 128-byte line), sigma and 1/sigma: zero it, let producers atomicMax the shards
 * (eav_sp_absmax or a GEMM's amax_slot), then eav_sp_convert writes the planes of X (dst: contraction over columns)
 * and / or of X^T (dstT: contraction over rows) and fills sigma. */
#define EAV_SP_SLOT 4128   /* shard i at word 32*i (one 128-byte line each), sigma at word 2048, 1/sigma at 2049; words
                            * 2080.. : max|x| bits per 32-row block, 3104.. : boost exponent per 32-row block (1024 each; blocks alias beyond 32768 rows) */
int eav_sp_kpad(int K);
int eav_sp_absmax(const float* src, int R, int C, int64_t ld, float* slot, void* stream);
int eav_sp_convert(const float* src, int R, int C, int64_t ld, float* slot, void* dst, void* dstT, void* stream);
/* planes of GELU(src): src = pre-activations written by eav_gemm_sp with gelu = 3, slot = max |GELU(src)| */
int eav_sp_convert_gelu(const float* src, int R, int C, int64_t ld, float* slot, void* dst, void* dstT, void* stream);
/* the same pass also emitting bias-gradient partials: colsum_part [eav_sp_convert_colsum_nparts(R)][C] (column sums of
 * 64-row tiles; finish with eav_reduce_partials) */
int eav_sp_convert_colsum_nparts(int R);
int eav_sp_convert_colsum(const float* src, int R, int C, int64_t ld, float* slot, void* dst, void* dstT,
                          float* colsum_part, void* stream);
/* eav_sp_absmax + eav_sp_convert for a whole table of dense matrices in two launches (the weight refresh of an encoder
 * after an optimiser step: 49 matrices).  jobs: n entries in DEVICE memory; slots zeroed by the caller; every C % 4 == 0,
 * sources 16-byte aligned; maxR / maxC = the largest R and C in the table. */
typedef struct EavPlaneJob {
  const float* src;   /* [R, C] dense fp32 */
  void* dst;          /* planes [R][Cp/8][2][8] or NULL */
  void* dstT;         /* planes of the transpose [C][Rp/8][2][8] or NULL */
  float* slot;        /* EAV_SP_SLOT floats */
  int R, C;
} EavPlaneJob;
int eav_sp_refresh_planes(const void* jobs, int n, int maxR, int maxC, void* stream);
/* producers that also accumulate max|output| into an operand-scale slot (zeroed by the caller).  Both LayerNorm forward forms
 * that take a slot (amax_slot here, scale_slot of eav_layernorm_fwd_planes) ALSO WRITE max(rstd) into word 1 of the slot's 64
 * shard lines (atomicMax of the float bits): the slot is an output of the forward in that respect even where it is declared
 * const, and eav_layernorm_bwd_planes / _bound read it through slot_rstd (a slot whose word 1 is still zero is not trusted:
 * the backward then walks rstd[M]). */
int eav_layernorm_fwd_amax(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                           int M, int D, float eps, float* amax_slot, void* stream);
int eav_layernorm_bwd_amax(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                           float* dx, int accumulate, float* part, int M, int D, float* amax_slot, void* stream);
/* eav_layernorm_bwd_amax whose stored value (dx after the accumulation) also leaves as the row planes [M][Dp/8][2][8] of the
 * gradient products that consume it - no conversion pass over the residual-stream gradient (modeling_vit.py /
 * modeling_audio_spectrogram_transformer.py layernorm_before / layernorm_after backward through HF autograd) - scaled by
 * slot's sigma = the rigorous bound of eav_layernorm_bwd_bound, which the kernel forms itself when slot_dy is given (every
 * workgroup from the same slot words, published in slot by the first; a separate one-block launch queued 15-40 us behind the
 * persistent GEMMs) or which a call of eav_layernorm_bwd_bound put there before (slot_dy NULL):
 * max|dx_old| + (2 + sqrt(D)) max|gamma| max(rstd) max|dy| (slot_old: shards of the measured max|dx_old| or NULL; slot_dy:
 * shards of max|dy|; max(rstd) from slot_rstd - the scale slot the forward LayerNorm launch had, whose shard lines carry it in
 * word 1 - or, slot_rstd NULL, from a walk over rstd[M]).  part is [nparts][3 D]: dgamma | dbeta | column sums of the stored value (a bias gradient).  slot's
 * shards receive the measured maximum of the stored value. */
int eav_layernorm_bwd_planes(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                             float* dx, int accumulate, float* part, int M, int D, float* slot, void* planes,
                             const float* slot_old, const float* slot_dy, const float* slot_rstd, void* stream);
int eav_layernorm_bwd_bound(float* slot_out, const float* slot_old, const float* slot_dy, const float* gamma,
                            const float* rstd, int M, int D, const float* slot_rstd, void* stream);
int eav_gelu_bwd_amax(float* dact, const float* pre, int64_t n, float* amax_slot, void* stream);
/* C[z][m,n] = epilogue(alpha * sum_k A[z][m,k] B[n,k]) from planes A [M,Kp], B [N,Kp]; epilogue as eav_gemm_f32
 * (bias, erf-GELU with pre-activation store, residual, accumulate); amax_slot (optional) receives max |C| bits.
 * gelu = 2: backward through GELU - the product is multiplied by gelu'(pre[m,n]), `pre` (ldc) is READ (the forward's
 * stored pre-activation): fc2's data gradient and eav_gelu_bwd in one pass.
 * gelu = 3: C receives the pre-activation itself and amax_slot max |GELU(C)| - the activation is formed by
 * eav_sp_convert_gelu while it splits (no fp32 activation tensor); pre / resid / accumulate must be unset. */
int eav_gemm_sp(const void* A, const void* B, float* C, const float* slotA, const float* slotB, int M, int N, int K,
                int ldc, int batch, int64_t sA_bytes, int64_t sC, float alpha, const float* bias, int gelu, float* pre,
                const float* resid, int ldr, int accumulate, float* amax_slot, void* stream);
/* eav_gemm_sp that also - or only (C = NULL) - writes the stored value (after the activation) as the row planes
 * [M][Np/8][2][8] of the NEXT product, scaled by planes_slot[EAV_SLOT sigma]: no fp32 activation tensor, no conversion
 * pass.  The scale must be known before the launch: eav_tf_forward_scales derives it from a rigorous bound of |output|.
 * N % 8 == 0, batch 1.  (fc1 of HF's ASTMLP / ViTMLP: bias + erf-GELU, pre-activation kept in `pre` for the backward.) */
int eav_gemm_sp_planes(const void* A, const void* B, float* C, const float* slotA, const float* slotB, int M, int N, int K,
                       int ldc, int batch, int64_t sA_bytes, int64_t sC, float alpha, const float* bias, int gelu,
                       float* pre, const float* resid, int ldr, int accumulate, float* amax_slot, void* planes_out,
                       const float* planes_slot, void* stream);
/* nn.LayerNorm whose output leaves as the row planes of the product that consumes it (y optional, may be NULL);
 * scale_slot as above; D % 8 == 0. */
int eav_layernorm_fwd_planes(const float* x, const float* gamma, const float* beta, float* y, void* planes,
                             const float* scale_slot, float* mean, float* rstd, int M, int D, float eps, void* stream);
/* max over the rows of ||w_r||_2 into *out (float bits combined with atomicMax: zero it first) */
int eav_rownorm_max(const float* w, int R, int C, int64_t ld, float* out, void* stream);
/* the same over the COLUMNS of w [R, C] (C, ld multiples of 4) */
int eav_colnorm_max(const float* w, int R, int C, int64_t ld, float* out, void* stream);
/* A table of eav_rownorm_max / eav_colnorm_max jobs in one launch: jobs = device array of njobs rows of 5 int64
 * {w, ld, out, R | C << 32, cols} (cols 0: max row norm, 1: max column norm; outputs zeroed by the caller), max_blocks >=
 * the largest job's block count (rows: min(ceil(R / 4), 128); columns: ceil(C / 64)).  The weight refresh after an optimiser
 * step forms all the norms behind the a-priori operand scales with it. */
int eav_norm_max_multi(const void* jobs, int njobs, int max_blocks, void* stream);
/* sigma, 1/sigma of slot_out from the bound  factor * max|x| * *norm  (max|x| = the maximum held by amax_slot's shards, norm a
 * device scalar): the operand scale of a tensor BEFORE it is produced, so that its producer can write planes directly.
 * Used for the MLP hidden-state gradient dact = (dh W2) o gelu'(pre): |dact| <= 1.13 sqrt(D) max|dh| max_j ||W2[:,j]||_2. */
int eav_sp_bound_scale(float* slot_out, const float* amax_slot, const float* norm, float factor, void* stream);
/* sigma, 1/sigma of the operand slots of LayerNorm-before output (k_y1), LayerNorm-after output (k_y2) and the MLP's GELU
 * output (k_act) of `layers` encoder layers from rigorous bounds: |LN out| <= sqrt(D) max|gamma| + max|beta|,
 * |GELU(y2 W1^T + b1)| <= (sqrt(D) max|gamma2| + ||beta2||_2) max_n ||W1_n||_2 + max|b1|.  params: first float of layer 0
 * in the flat parameter buffer, layer_stride floats per layer, off_*: offsets of layernorm_before.{weight,bias},
 * layernorm_after.{weight,bias}, mlp.fc1.bias within a layer; wnorm_fc1 [layers]: eav_rownorm_max of mlp.fc1.weight;
 * slots: layer 0's first forward slot, slot_stride floats per layer. */
int eav_tf_forward_scales(const float* params, int64_t layer_stride, int layers, int off_g1, int off_b1, int off_g2,
                          int off_b2, int off_bfc1, int D, int FF, const float* wnorm_fc1, float* slots,
                          int64_t slot_stride, int k_y1, int k_y2, int k_act, void* stream);
int eav_tf_forward_scales_qkv(const float* params, int64_t layer_stride, int layers, int off_g1, int off_b1, int off_g2,
                              int off_b2, int off_bfc1, int off_bqkv, int D, int FF, const float* wnorm_fc1,
                              const float* wnorm_qkv, float* slots, int64_t slot_stride, int k_y1, int k_y2, int k_act,
                              int k_qkv, void* stream);
/* long-contraction form (weight gradients): C[M,N] = sum_t A[t,m] B[t,n] over ROW planes A [Tp][Mp/8][2][8], B
 * [Tp][Np/8][2][8] - the contraction runs over the rows (tokens), read with transposing LDS loads; Tp = T rounded up to
 * 32 and the rows >= T must be ZERO.  eav_gemm_sp_splitk_plan(M,N,T) token slices, ws [nsplit][M][N] partials summed in
 * fixed order (deterministic); C[M,N] dense (ldc = N). */
int eav_gemm_sp_splitk_plan(int M, int N, int K);
int eav_gemm_sp_splitk(const void* A, const void* B, float* C, float* ws, const float* slotA, const float* slotB, int M,
                       int N, int T, int accumulate, void* stream);
/* One-term forms: the hi.hi product alone - the operands rounded to fp16 under the planes' scales (11-bit mantissas, fp32
 * accumulation), a third of the matrix work on the SAME planes, same epilogues.  The opt-in precision of the backward
 * products (Encoder.grad_terms = 1): gradients then carry ~2^-11 relative rounding per operand, the forward (logits)
 * stays on the three-term product unless Encoder.fwd_terms = 1 asks for the plain fp16 forward as well (comparison leg of
 * bench.py: 16-bit matrix operands everywhere, as BASELINE.json's configs name them). */
int eav_gemm_sp_x1(const void* A, const void* B, float* C, const float* slotA, const float* slotB, int M, int N, int K,
                   int ldc, int batch, int64_t sA_bytes, int64_t sC, float alpha, const float* bias, int gelu, float* pre,
                   const float* resid, int ldr, int accumulate, float* amax_slot, void* stream);
/* eav_gemm_sp_planes with option flags */
#define EAV_GEMM_ONE_TERM 1    /* the hi.hi term alone */
#define EAV_GEMM_PLANES_NOLIFT 4 /* planes_out with lo = fp16(sigma x - hi), no 2^11 lift: the row planes the fused attention
                                  * reads (the fused q/k/v projection writes them directly, scale from eav_tf_forward_scales_qkv) */
#define EAV_GEMM_NO_BLOCKMAX 8  /* amax_slot receives the tensor-wide maximum only, no 32-row block entries: for outputs whose
                                 * consumer takes one scale per tensor (the attention operand preparation of dO) */
#define EAV_GEMM_SHARED_GPU 2  /* a second persistent GEMM runs beside this one (the backward's data gradients next to the
                                * side stream's weight gradients): prefer the 256 x 128 one-workgroup-per-CU form */
/* colsum_part (optional): [ceil(M / 64)][N], row p = column sums of the stored value over rows [64 p, 64 p + 64) - bias-gradient
 * partials for eav_reduce_partials; batch 1. */
int eav_gemm_sp_ex(const void* A, const void* B, float* C, const float* slotA, const float* slotB, int M, int N, int K,
                   int ldc, int batch, int64_t sA_bytes, int64_t sC, float alpha, const float* bias, int gelu, float* pre,
                   const float* resid, int ldr, int accumulate, float* amax_slot, void* planes_out,
                   const float* planes_slot, float* colsum_part, int flags, void* stream);
int eav_gemm_sp_splitk_x1(const void* A, const void* B, float* C, float* ws, const float* slotA, const float* slotB, int M,
                          int N, int T, int accumulate, void* stream);
/* eav_gemm_sp_splitk on TWO terms, hi_A.hi_B + lo_A.hi_B: operand B rounded to fp16 (its lo pieces are not read), operand A
 * at full split precision.  For weight gradients with A = the gradient tensor and B = the activation (autograd of
 * Transformer_Audio.py:74-79 / Transformer_Vision.py:94-99 through the HF nn.Linear layers): two thirds of the matrix work. */
int eav_gemm_sp_splitk_x2(const void* A, const void* B, float* C, float* ws, const float* slotA, const float* slotB, int M,
                          int N, int T, int accumulate, void* stream);
/* (the process-global tile / slice-count overrides the kernel benchmarks use are NOT part of this header: eav_hip_tuning.h) */
/* The same fused attention on the fp16 matrix cores with split operands (csrc/attention_sp.hip; fp32-grade, 3 MFMAs per
 * product).  eav_attn_sp_prep converts an fp32 activation src [B*N, ncols] (qkv or dO; slot holds its max|x| shards, see
 * EAV_SP_SLOT) into row planes [B*N][ncols/8][2][8] f16 and, for the column sections (of secw columns) selected by
 * tmask, per-head transposed planes [B][ncols/64][64][Npad/8][2][8] (Npad = eav_attn_sp_npad(N), zero beyond N); here
 * lo = fp16(sigma x - hi) without the 2^11 lift of the GEMM planes.  amax_slot (optional) receives max|output| shards. */
int eav_attn_sp_npad(int N);
int eav_attn_sp_prep(const float* src, float* slot, void* rowp, void* tp, int B, int N, int ncols, int secw,
                     unsigned tmask, void* stream);
/* Since round 3 the three attention kernels take their token-contracting operands (V in the forward, K in the dQ kernel,
 * Q and dO in the dK,dV kernel) from the ROW tiles with transposing LDS reads (ds_read_b64_tr_b16): the transposed planes
 * `tp` / `dotp` are not read any more and may be NULL (the parameters stay for binary compatibility; eav_attn_sp_prep
 * still writes them on request). */
int eav_attn_fwd_sp(const void* rowp, const void* tp, const float* slot, float* ao, float* lse, float* amax_slot, int B,
                    int H, int N, int head_dim, float scale, void* stream);
/* the same, the output also - or only (ao = NULL: forward-only passes) - as the GEMM operand planes [B*N][D/8][2][8] of the
 * o-proj products (lo lifted by 2^11 like eav_sp_convert's), scaled with qkv's own sigma (O is a convex combination of V
 * rows: |O| <= max|V| <= max|qkv|), which the kernel copies into ao_slot: no conversion pass for the attention output. */
int eav_attn_fwd_sp_planes(const void* rowp, const void* tp, const float* slot, float* ao, float* lse, float* amax_slot,
                           void* ao_planes, float* ao_slot, int B, int H, int N, int head_dim, float scale, void* stream);
/* dorow / dotp: planes of dO [B*N, H*64]; slot_ds: zeroed scratch slot (max|dS| travels from the dQ to the dK,dV kernel);
 * ao, dout: fp32 O and dO for delta = rowsum(dO o O); delta: scratch [B*H, N]; dqkv [B*N, 3*H*64] fp32. */
int eav_attn_bwd_sp(const void* rowp, const void* tp, const void* dorow, const void* dotp, const float* slot,
                    const float* slot_do, float* slot_ds, const float* ao, const float* dout, const float* lse,
                    float* delta, float* dqkv, float* amax_slot, int B, int H, int N, int head_dim, float scale,
                    void* stream);
/* eav_attn_bwd_sp that also (or only: dqkv = NULL) writes dqkv as the operand planes [B*N][3D/8][2][8] of the q/k/v
 * projection's gradient products - no fp32 dqkv, no conversion pass - scaled by a sigma the kernels form themselves and
 * publish in planes_slot (eav_attn_dqkv_bound computes the same number stand-alone) from a rigorous bound |dqkv| <= N max|dO| max(1, 128 scale max|qkv|^2) (slot_do: the
 * shards of max|dO|; slot_qkv: the forward's qkv slot); colsum_part (optional) [B * ceil(N/32)][3D] receives the column sums
 * of every 32-row tile (bias-gradient partials: finish with eav_reduce_partials).  ao_planes + ao_slot (optional): the attention
 * output as the planes eav_attn_fwd_sp_planes wrote - delta = dO . O then comes from planes and ao / dout may be NULL (the
 * training step keeps no fp32 attention output).  Replaces, in the split step, the
 * eav_sp_convert_colsum pass over dqkv (Transformer_Audio.py:72 / Transformer_Vision.py:92 through HF's backward). */
int eav_attn_dqkv_bound(float* slot_out, const float* slot_do, const float* slot_qkv, int N, float scale, void* stream);
int eav_attn_bwd_sp_planes(const void* rowp, const void* tp, const void* dorow, const void* dotp, const float* slot,
                           const float* slot_do, float* slot_ds, const float* ao, const float* dout, const float* lse,
                           float* delta, float* dqkv, float* amax_slot, void* planes, float* planes_slot,
                           float* colsum_part, const void* ao_planes, const float* ao_slot, int B, int H, int N,
                           int head_dim, float scale, void* stream);
/* Fused multi-head self-attention (head_dim 64), exact fp32 MFMA, flash-style: softmax(Q K^T scale) V per
 * (image, head) of qkv [B*N, 3*H*64] (HF eager_attention_forward).  ao [B*N, H*64]; lse [B*H, N] saved for
 * the backward. */
int eav_attn_fwd(const float* qkv, float* ao, float* lse, int B, int H, int N, int head_dim, float scale,
                 void* stream);
/* Backward of eav_attn_fwd: dqkv [B*N, 3*H*64] <- (dQ | dK | dV) from dout [B*N, H*64]; ao / lse as produced by
 * the forward; delta: scratch [B*H, N]. */
int eav_attn_bwd(const float* qkv, const float* ao, const float* dout, const float* lse, float* delta, float* dqkv,
                 int B, int H, int N, int head_dim, float scale, void* stream);
/* nn.LayerNorm(D, eps) forward over M rows; mean/rstd [M] saved for the backward (may be NULL). */
int eav_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                      int M, int D, float eps, void* stream);
int eav_layernorm_bwd_nparts(int M);
/* dx (accumulate=1: dx += ...) and per-block partials part[nparts][2*D] = (dgamma, dbeta); part may be NULL. */
int eav_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                      float* dx, int accumulate, float* part, int M, int D, void* stream);
/* softmax over rows of length N (leading dimension ld), in place; backward dS = P o (dP - sum dP*P) in place on dP. */
int eav_softmax_fwd(float* s, int64_t rows, int N, int ld, void* stream);
int eav_softmax_bwd(const float* P, float* dP, int64_t rows, int N, int ld, void* stream);
/* dact *= GELU'(pre) (exact erf form, HF GELUActivation), in place. */
int eav_gelu_bwd(float* dact, const float* pre, int64_t n, void* stream);
/* bias gradient: part[eav_colsum_nparts(M)][N] partial column sums of dy [M,N]. */
int eav_colsum_nparts(int M);
int eav_colsum(const float* dy, float* part, int M, int N, int ld, void* stream);
/* patch extraction for the Conv2d patch embedding (HF AST :53-61 with transposed=1, HF ViT :60-69). */
int eav_im2col(const float* x, float* col, int B, int C, int H, int W, int P, int sy, int sx, int transposed,
               void* stream);
/* h[b,t] = (t<nextra ? token_t : h[b,t]) + pos[t]  (HF AST :93-97, ViT :146-157) and its backward. */
int eav_embed_finish(float* h, const float* cls, const float* dist, const float* pos, int B, int ntok, int D,
                     int nextra, void* stream);
int eav_embed_bwd(const float* dh, float* dpos, float* demb, int B, int ntok, int D, int nextra, void* stream);
/* gather (scatter=0) / scatter (1) the first nextra token rows of every image: rows[b*nextra+e] <-> h[b,e]. */
int eav_token_rows(float* h, float* rows, int B, int ntok, int D, int nextra, int scatter, void* stream);
/* AST pooled = (cls + dist)/2 (HF AST :304); backward=1 writes dseq from dpooled. */
int eav_pair_mean(float* seq, float* pooled, int B, int D, int backward, void* stream);

/* ---- ShallowConvNet + single-head transformer (Transformer_torch/Transformer_EEG.py:14-148) ------------------- */
/* conv(1->NF,(1,KC),valid) fused with PatchEmbedding's per-filter Linear(Chans->1) (:117,:28-35):
 * x [B,C,S] -> u [B,NF,S] (channel-projected input, kept for the backward) and v [B,S-KC+1,NF] tokens. */
int eav_shallow_embed_nparts(int B, int S);
int eav_shallow_embed_fwd(const float* x, const float* wc, const float* wv /*[NF] rows, stride ldv*/, int ldv, float* u,
                          float* v, int B, int C, int S, int NF, int KC, void* stream);
/* weight gradients: part_c [nparts][NF*KC] (conv taps), part_v [nparts][NF*C] (channel projections). */
int eav_shallow_embed_bwd(const float* dv, const float* x, const float* u, const float* wc, float* part_c,
                          float* part_v, int B, int C, int S, int NF, int KC, void* stream);
/* nn.ReLU -> nn.Dropout in place (:84-85) and its backward (act = the forward's output). */
int eav_relu_dropout(float* h, int64_t n, float drop_p, uint64_t seed, const uint8_t* mask, const uint64_t* seed_dev,
                     void* stream);
int eav_relu_dropout_bwd(float* dact, const float* act, int64_t n, float drop_p, void* stream);
/* out = resid + Dropout(y) (:103-104); resid NULL: out = Dropout(y) (also the backward of the dropout branch). */
int eav_dropout_add(const float* y, const float* resid, float* out, int64_t n, float drop_p, uint64_t seed,
                    const uint8_t* mask, const uint64_t* seed_dev, void* stream);
/* out[m,0:n) = a[m,0:n) (+ b[m,0:n)) with leading dimensions - the "+ V" residual of MultiHeadAttention (:74-76). */
int eav_add_strided(const float* a, int lda, const float* b, int ldb, float* out, int ldo, int64_t M, int n,
                    void* stream);
/* per-column sum / sum of squares of x [M,N<=256]: part [eav_colstats_nparts(M)][2N] (BatchNorm over tokens, :136). */
int eav_colstats_nparts(int64_t M);
int eav_colstats(const float* x, float* part, int64_t M, int N, int ld, void* stream);
/* head (:136-144): BatchNorm affine -> square -> AvgPool(1,win)/stride -> log(clamp(lo,hi)) -> Dropout.
 * v [B,T,NF] tokens -> pooled [B,NF,NP] (pre-log means, kept for the backward) and out [B,NF*NP]. */
int eav_sqpool_log_fwd(const float* v, const float* bn, float* pooled, float* out, int B, int T, int NF, int NP,
                       int win, int stride, float lo, float hi, float drop_p, uint64_t seed, const uint8_t* mask,
                       const uint64_t* seed_dev, void* stream);
/* g [B,T,NF] = dL/d(BatchNorm output); part [B][2*NF] = sums for eav_bn_bwd_finalize. */
int eav_sqpool_log_bwd(const float* dy, const float* pooled, const float* v, const float* bn, float* g, float* part,
                       int B, int T, int NF, int NP, int win, int stride, float lo, float hi, float drop_p,
                       uint64_t seed, const uint8_t* mask, const uint64_t* seed_dev, void* stream);
/* BatchNorm input gradient on token-major rows; bn = mean, invstd, scale, shift, m1, m2 (NF each). */
int eav_bn_rows_bwd(const float* g, const float* v, const float* bn, float* dx, int64_t M, int NF, void* stream);

/* ---- pre-processing (SURVEY section 8f "next" rows) ------------------------------------------------ */
/* HF image processor on a batch of uint8 HWC frames (Transformer_Vision.py:52-59): Pillow-exact 8-bit
 * bilinear resize (coefficient tables kx/ky [out][ksize] int32 and bounds [out][2] = (first, count), built
 * on the host like Pillow's precompute_coeffs), * rescale (float64), (x - mean) / std -> out [n,C,OH,OW] fp32.
 * mean3 / std3 are HOST pointers. */
int eav_resize_normalize_u8(const uint8_t* frames, const int* kx, const int* boundsx, const int* ky,
                            const int* boundsy, float* out, int n, int H, int W, int C, int OH, int OW, int ksize_x,
                            int ksize_y, double rescale, const float* mean3, const float* std3, void* stream);

/* AST log-mel front-end, HF ASTFeatureExtractor numpy path (Transformer_Audio.py:38-42): wav [n,L] fp32 (16 kHz)
 * -> out [n,max_len,nmel] fp32 = (log-mel - mean) / std2 with frames beyond the clip zero-padded before the
 * normalisation.  window400 (Hann, symmetric), twiddle256 ([k] = cos, -sin of 2 pi k/512) and melT [nmel][257]
 * are float64 DEVICE tables built by the host (eav_amd/preprocess.py). */
int eav_ast_fbank(const float* wav, const double* window400, const double* twiddle256, const double* melT, float* out,
                  int n, int L, int max_len, int nmel, double preemph, double mel_floor, float mean, float std2,
                  void* stream);

/* EEG pre-processing, float64 like scipy (Dataload_eeg.py:85-121).
 * scipy.signal.resample_poly(x, 1, down): y[c][m] = sum_j h[j] x[c][m*down + center - j], x [nch][n_in]. */
int eav_decimate_fir_f64(const double* x, const double* h, double* y, int nch, int64_t n_in, int64_t n_out, int down,
                         int ntaps, int center, void* stream);
/* scipy.signal.sosfilt(sos, x) along the last axis of x [nch][n] (zero initial state), exact chunk-parallel form:
 * sos [nsec][6]; H [Lc][2 nsec] = cascade output at step k from unit initial state j; AL [2 nsec][2 nsec] = state
 * after Lc zero-input steps; zend / zstart: scratch [nch][ceil(n/Lc)][2 nsec]. */
int eav_sosfilt_f64(const double* x, double* y, const double* sos, const double* H, const double* AL, double* zend,
                    double* zstart, int nch, int64_t n, int nsec, int Lc, void* stream);

/* ---- measured peaks (bench.py): register-only fp32 MFMA loop (FLOP = blocks*4 waves*iters*4*4096) and a float4
 *      streaming copy, to quote roofline fractions against what this chip sustains. */
int eav_peak_mfma_f32(float* sink, int blocks, int iters, void* stream);
int eav_peak_mfma_f16(float* sink, int blocks, int iters, void* stream);
/* L2 -> CU read ceiling probe (bench.py measured_peaks / tuning): mode 0 = 16-byte loads into registers, 1 = LDS-DMA;
 * every wave reads 1-KB chunks of the first footprint_kb KB of src, 8 in flight; bytes = blocks * 4 * iters * 8192. */
int eav_peak_l2_read(const void* src, int footprint_kb, int mode, int iters, int blocks, float* sink, void* stream);
int eav_peak_copy_variant(const float* src, float* dst, int64_t n, int variant, int blocks, void* stream);
int eav_peak_copy(const float* src, float* dst, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EAV_HIP_H */
