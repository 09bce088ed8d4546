/*
 * ds2hip.h -- C ABI of libds2hip.so: the MI355X (gfx950) DeepSpeech2 hot path.
 *
 * The reference (igormq/aes-lac-2018) has no FFI of its own: its hot path bottoms out in
 * torch.nn modules (cuDNN/ATen), librosa and the warp-ctc binding.  Each entry point below
 * replaces one of those call sites; the citation gives the reference line it stands in for.
 * Bindings: aes-lac-2018_amd/ds2hip/lib.py (ctypes); INTEGRATION.md shows the stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - all tensors are dense fp32 row-major unless stated, int32 for lengths/labels;
 *   - the caller owns every buffer (inputs, outputs, saved-for-backward tensors, workspaces);
 *     the library never allocates or frees device memory and keeps no mutable global state;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); calls only enqueue work;
 *   - return value 0 = ok, negative = error (DS2_ERR_*); ds2_last_error() returns a
 *     thread-local message.  No exceptions cross the boundary.
 *
 * Shapes: B batch, T_in spectrogram frames, F=161 bins, T1 = (T_in+9)/2+1 frames after conv1,
 * T = T1-10 frames after conv2, H GRU hidden size, A alphabet size (blank = 0).
 */
#ifndef DS2HIP_H
#define DS2HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DS2_OK 0
#define DS2_ERR_ARG (-1)
#define DS2_ERR_LAUNCH (-2)
#define DS2_ERR_UNSUPPORTED (-3)

const char* ds2_last_error(void);
/* The ABI revision this header describes.  ds2_version() of the loaded library must return exactly this number: signatures
 * have changed between revisions without a change of symbol name (round 3: an amplitude-scale argument in
 * ds2_pcm16_to_float / ds2_gain_requantize; round 4: ds2_conv2_dgrad takes the size of its workspace), so a binding built
 * against another revision mis-passes arguments.  ds2hip/lib.py refuses to load a library whose number differs. */
#define DS2_ABI_VERSION 402
int ds2_version(void);
/* A digest of the sources the loaded binary was built from (csrc/build.py: source_id(); "unstamped" for a build made without
 * build.py).  The Python binding recomputes it from the tree beside it and refuses a binary built from other sources, so a
 * stale .so cannot be tested or benchmarked by accident (DS2_SKIP_BUILD_CHECK=1 overrides, for binary-only installs). */
const char* ds2_build_id(void);

/* ------------------------------------------------------------------ frontend
 * Replaces ToSpectrogram.__call__ (librosa branch), codes/transforms.py:94-119, run per
 * utterance in codes/data.py:61-62, and the zero-padding collate of codes/data.py:132-152.
 *   wav          concatenated clips, clip b = wav[wav_offsets[b] .. wav_offsets[b+1])
 *   wav_offsets  (B+1) int64 device
 *   out          (B, t_max, 161), zero-filled past each clip's 1 + L_b/160 frames
 *   stats_ws     workspace, >= ds2_spectrogram_ws_bytes(B, t_max) bytes
 * STFT: reflect-pad 160, frame 320 / hop 160, symmetric Hann, |rFFT|, log1p, then per-clip
 * (S - mean) / (std_unbiased + eps).
 */
size_t ds2_spectrogram_ws_bytes(int B, int t_max);
int ds2_spectrogram_fwd(const float* wav, const int64_t* wav_offsets, int B, int t_max, int normalize,
                        float eps, float* out, void* stats_ws, void* stream);

/* ------------------------------------------------------------------ waveform decode + augmentation
 * Replace the per-clip host work of ToTensor (codes/transforms.py:130-224): torchaudio.load of a 16-bit PCM
 * file, and -- for training clips with augment=True -- `sox ... -b 16 -e si <out> tempo T gain G` through a
 * temporary file (:185-218).  The loader ships int16 samples; these run after collate on the whole minibatch.
 *   ds2_pcm16_to_float   out[i] = pcm[i] * scale                                  (n samples, any concatenation).  The
 *                        amplitude scale is torchaudio.load's contract (codes/transforms.py:156-161), which changed between
 *                        torchaudio versions: 1/32768 gives [-1, 1); 65536 gives the un-normalised int32-range floats of
 *                        the mid-2018 master (a 16-bit sample read by sox as a 32-bit one).  log1p is not scale invariant.
 *   ds2_wsola_tempo      time-scale change without pitch change (WSOLA, sox's default 82 / 14.68 / 12 ms segment /
 *                        search / overlap -> seg, half = search/2, ovl in samples).  Clip b = x[in_offsets[b] ..
 *                        in_offsets[b+1]) -> out[out_offsets[b] .. out_offsets[b+1]).  The segment schedule is data
 *                        independent: bases[base_offsets[b] + k] is the rounded ideal input position of segment k
 *                        (running sum of tempo * (seg - ovl)); zero segments = copy.  The output length follows from
 *                        the schedule: (nseg + 1) * (seg - ovl) + ovl.  Correlations in float64, first maximum wins.
 *                        The algorithm is specified in oracle/audio.py (sox's own implementation is not in the
 *                        reference tree: parity with sox is not claimed).
 *   ds2_gain_requantize  y = x * gain[b]; out = clip(rint(y * 32768), -32768, 32767) * out_scale   (x in [-1, 1); the gain
 *                        in linear units, 10^(dB/20); rounding half to even; out_scale as ds2_pcm16_to_float's scale),
 *                        in place allowed
 */
int ds2_pcm16_to_float(const int16_t* pcm, size_t n, float scale, float* out, void* stream);
int ds2_wsola_tempo(const float* x, const int64_t* in_offsets, const int64_t* out_offsets, const int32_t* bases,
                    const int32_t* base_offsets, int B, int seg, int ovl, int half, float* out, void* stream);
int ds2_gain_requantize(const float* x, const int64_t* offsets, const float* gain, int B, float out_scale, float* out,
                        void* stream);

/* ------------------------------------------------------------------ generic fp32 GEMM (MFMA)
 * C[M,N] = op(A) * op(B) + beta * C, row-major with leading dimensions.  op(A)=A (M x K, lda) or
 * A^T (A stored K x M); op(B)=B (K x N, ldb) or B^T (B stored N x K).  beta is 0 or 1.
 * Stands in for the cuBLAS calls behind nn.Linear / nn.GRU's input projection
 * (codes/model.py:51-52,178-180).  split_k > 1 splits K over workgroups and accumulates the
 * partial products with float atomics (C is zero-filled first when beta = 0; the last bits then
 * depend on arrival order); split_k = 0 picks a split that fills the chip when M*N alone gives
 * fewer than ~384 tiles (the dW = dG^T X products: small M*N, K = T*B).
 *
 * Arithmetic.  Operands, accumulator and result are fp32.  Two kernel families compute the products:
 *   mode 0      the f32-input matrix instruction (v_mfma_f32_32x32x2_f32): an fp32 FMA chain;
 *   mode 6 / 9  every fp32 operand element is split without error into three bf16 terms a = a1 + a2 + a3
 *               (round-to-nearest at each step: the three 8-bit significands carry all 24 bits), and a * b is summed from
 *               partial products ai * bj on the bf16 matrix instruction, each exact in its fp32 accumulator.  Mode 9 adds
 *               all nine; mode 6 (the default) leaves out a2 b3 + a3 b2 + a3 b3: at most 2^-24.2 |a b| and 6e-9 rms over 10^6 random
 *               products (tests/test_split_cpu.py), where a correctly rounded fp32 multiply is off by up to 2^-24 and 2.5e-8 rms -- less than one fp32
 *               rounding.  Measured against fp64 the three agree to the last digit shown (tools/gemm_split_check.py:
 *               max error / sum|a||b| 2.9e-7 at K = 672 in modes 6 and 9, 3.2e-7 in mode 0).  Not handled like mode 0:
 *               an operand element beyond bf16's largest finite value (3.39e38) or an infinity gives NaN, and terms below
 *               2^-126 are flushed (operands below ~2^-108 lose low bits).
 * ds2_gemm_split_mode(mode) selects the family for later calls (0, 6 or 9; anything else only queries) and returns the one
 * in effect; the process default is DS2_GEMM_SPLIT or 6.  Mode 6 / 9 needs 16-byte aligned operands with leading dimensions
 * that are multiples of 4 and, for an operand stored with K contiguous, K % 16 == 0; other calls use mode 0's kernels.
 */
int ds2_gemm_split_mode(int mode);
int ds2_gemm_f32(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda, const float* B,
                 int ldb, float* C, int ldc, float beta, int split_k, void* stream);
/* Up to four TN problems C_p[M_p, N] = A_p^T B_p that share N and K in ONE launch (A_p stored K x M_p with leading
 * dimension lda_p, B_p stored K x N): the four weight-gradient GEMMs dW_hh of a BiGRU layer -- two directions x
 * {r|z rows, n rows} of dGH^T h_prev, codes/model.py:51-52's backward -- which as separate launches are too small to fill
 * the chip.  The pointer / size arrays are HOST arrays of `count` entries holding device pointers; K is split over
 * workgroups and accumulated with float atomics into C -- zero-filled by the call, or, with accumulate != 0, added to what
 * C holds (the caller zeroed the whole flat gradient with one fill: C += A^T B). */
int ds2_gemm_f32_tn_group(int count, const float* const* A_host, const int* lda_host, const int* M_host,
                          const float* const* B_host, const int* ldb_host, float* const* C_host, const int* ldc_host,
                          int N, int K, int accumulate, void* stream);

/* ------------------------------------------------------------------ conv stack
 * Replaces nn.Conv2d at codes/model.py:143-144 (cuDNN).  Layouts are NCHW with time innermost:
 *   conv1: x_t (B,161,T_in) [the transposed input, codes/model.py:184] -> y (B,32,61,T1)
 *          kernel (32,1,41,11), stride (2,2), padding (0,10)
 *   conv2: a1 (B,32,61,T1) -> y (B,32,21,T); kernel (32,32,21,11), stride (2,1), no padding
 * which = 1 or 2.  Weights are passed in the torch layout (Cout,Cin,KF,KT) plus bias (32).
 * wt_ws: workspace of ds2_conv_wt_ws_floats(which) floats for the re-laid-out filter.  ds2_conv2_dgrad takes its workspace
 * WITH its size: ds2_conv2_dgrad_ws_floats(B, T1) floats let it choose its gather form, which keeps a zero-bordered copy of
 * d_out there (several million floats); with fewer -- but at least ds2_conv_wt_ws_floats(2) -- it runs the direct kernel;
 * with fewer than that it returns DS2_ERR_ARG.  It never writes past ws_floats.
 * Arithmetic of conv2's forward pass and data gradient: fp32 operands, accumulator and result; by default the products run
 * on the bf16 matrix pipe after the error-free three-way operand split described at ds2_gemm_f32 (six exact partial
 * products; DS2_CONV_SPLIT=9 all nine, =0 the direct kernels on the f32-input matrix instruction; the data gradient takes
 * that form from B * T1 >= 3500 input columns, DS2_CONV_SPLIT_DGRAD=0 / 1 forces a choice).
 */
int ds2_transpose_btf_to_bft(const float* x, int B, int T, int F, float* x_t, void* stream);
size_t ds2_conv_wt_ws_floats(int which);
int ds2_conv_fwd(int which, const float* in, const float* weight, const float* bias, int B, int t_in_frames,
                 float* out, float* wt_ws, void* stream);
/* dgrad (conv2 only): d_in (B,32,61,T1) from d_out (B,32,21,T) */
size_t ds2_conv2_dgrad_ws_floats(int B, int T1);
int ds2_conv2_dgrad(const float* d_out, const float* weight, int B, int T1, float* d_in, float* wt_ws,
                    size_t ws_floats, void* stream);
/* wgrad: d_weight (Cout,Cin,KF,KT) and d_bias (32) OVERWRITTEN (d_weight zeroed inside) */
int ds2_conv_wgrad(int which, const float* in, const float* d_out, int B, int t_in_frames, float* d_weight,
                   float* d_bias, void* stream);

/* ------------------------------------------------------------------ BatchNorm
 * BatchNorm2d(32)+Hardtanh(0,20) of codes/model.py:143-145 on (B,C,D,T) and the sequence-wise
 * BatchNorm1d of codes/model.py:27-34,50,59-60,178 on (rows=T*B, F) -- both including padded
 * frames, eps 1e-5, momentum 0.1, running_var fed the unbiased estimate.
 *
 * Training: *_stats computes batch mean / biased var into stat[0:C], stat[C:2C] (=mean, invstd
 * after finalisation) and updates running_mean/var; *_apply normalises.  Eval: pass
 * use_running=1 to *_stats to derive mean/invstd from the running buffers instead.
 * ws: >= ds2_bn_ws_bytes(C) bytes.
 */
size_t ds2_bn_ws_bytes(int C);
/* conv flavour, x (B,C,inner) with inner = D*T */
int ds2_bn2d_stats(const float* x, int B, int C, int inner, float eps, float momentum, int use_running,
                   float* running_mean, float* running_var, float* mean_invstd, void* ws, void* stream);
/* y = hardtanh(bn(x), 0, 20); layout_tbf=0: y (B,C,inner); layout_tbf=1 (conv2 -> GRU input, fuses
 * codes/model.py:189-192): x (B,C,D,T) -> y (T,B,C*D) */
int ds2_bn2d_apply_htanh(const float* x, const float* mean_invstd, const float* gamma, const float* beta,
                         int B, int C, int D, int T, int layout_tbf, float* y, void* stream);
/* backward of the above: dy (B,C,D,T) [a (T,B,C*D) gradient is first brought back with
 * ds2_transpose2d(dy, T, B*C*D)] -> dx (B,C,D,T); dgamma,dbeta (C) overwritten */
int ds2_bn2d_htanh_bwd(const float* x, const float* dy, const float* mean_invstd, const float* gamma,
                       const float* beta, int B, int C, int D, int T, float* dx, float* dgamma,
                       float* dbeta, void* ws, void* stream);
/* sequence flavour.  x = xa (+ xb if xb != NULL): the sum of the two GRU directions
 * (codes/model.py:64-67) is folded into the BN read.  rows = T*B. */
int ds2_bn1d_stats(const float* xa, const float* xb, int rows, int F, float eps, float momentum,
                   int use_running, float* running_mean, float* running_var, float* mean_invstd, void* ws,
                   void* stream);
int ds2_bn1d_apply(const float* xa, const float* xb, const float* mean_invstd, const float* gamma,
                   const float* beta, int rows, int F, float* y, void* stream);
/* dx (rows,F) = BN backward of dy; x recomputed from xa(+xb).  dgamma, dbeta overwritten. */
int ds2_bn1d_bwd(const float* xa, const float* xb, const float* dy, const float* mean_invstd,
                 const float* gamma, int rows, int F, float* dx, float* dgamma, float* dbeta, void* ws,
                 void* stream);

/* ------------------------------------------------------------------ bidirectional GRU recurrence
 * Replaces nn.GRU(bias=False, bidirectional=True) (codes/model.py:51-52,62) -- the cuDNN RNN.
 * The input projection gi = X W_ih^T is a ds2_gemm_f32 call made by the caller into G.
 *   G     (T,B,2,3H)  in: gi (gate order r,z,n; dir 0 forward, dir 1 reverse)
 *                     out: the gate activations r,z,n (saved for backward)
 *   ghn   (T,B,2,H)   out: W_hn h_{t-1} (saved for backward)
 *   hout  (2,T,B,H)   out: hidden state of each direction at every step; h_0 = 0; the reverse
 *                     direction runs t = T-1 .. 0 over the PADDED length (no packing, :62)
 *   w_hh  (2,3H,H)    both directions' recurrent weights, dir-major
 * The caller sums hout[0]+hout[1] (codes/model.py:64-67) inside the next BN read.
 */
int ds2_gru_bidir_fwd(float* G, float* ghn, float* hout, const float* w_hh, int T, int B, int H,
                      void* stream);
/* BPTT.  d_out (T,B,H) is the gradient w.r.t. the direction SUM (so it feeds both directions).
 *   G     in: r,z,n  out: d(gi) = [dr_pre, dz_pre, dn_pre]      (T,B,2,3H)
 *   ghn   in: W_hn h  out: d(gh_n) = dn_pre * r                  (T,B,2,H)
 *   w_hh_t (2,H,3H)   transposed recurrent weights (ds2_transpose2d per direction)
 *   dh_ws  workspace of 2*2*B*H floats
 */
int ds2_gru_bidir_bwd(float* G, float* ghn, const float* hout, const float* d_out, const float* w_hh_t,
                      float* dh_ws, int T, int B, int H, void* stream);
int ds2_transpose2d(const float* in, int rows, int cols, float* out, void* stream);
/* The same for `count` (<= 8) separately placed inputs of `batch` row-major (rows, cols) matrices each, in ONE launch:
 * in_host = HOST array of `count` device pointers; out = (count, batch, cols, rows).  The training forward pass transposes the
 * recurrent weights of all five layers (two directions each, adjacent in the flat parameter buffer) for the backward recurrence
 * with one call instead of ten (ABI revision 402). */
int ds2_transpose2d_group(int count, const float* const* in_host, int batch, int rows, int cols, float* out, void* stream);
/* Persistent form of the two calls above: ONE launch per layer pass; every workgroup keeps its slice
 * of the recurrent weights in registers for all T steps and hands h_t (forward) / d(gh)_t (backward)
 * to the other workgroups of its direction (and batch part) inside the launch: write-through stores
 * into an exchange ring inside sync_ws, one arrival add per workgroup per step on sharded counters,
 * one polling wave per consumer, write-through-coherent loads straight into MFMA operands.  The form
 * (4x4x1 or 16x16x4 MFMA, 1-3 batch parts) is chosen from B and H; results do not depend on it beyond
 * fp32 summation order.  Same arguments and results as the per-step calls; sync_ws is a caller-owned
 * device buffer of ds2_gru_sync_ws_bytes(B, H) bytes, 16-byte aligned, that the CALLER ZEROES ONCE
 * when it allocates it: a launch that completes leaves its counters zero for the next one (the last
 * workgroup to leave resets them), so no memset runs between launches.  Every spin is bounded (5 s):
 * on a timeout the 32-bit word at byte offset ds2_gru_sync_error_offset() of sync_ws is set to 1 and
 * the launch ends.  EVERY output of a launch that sets the word -- G, ghn, hout and the exchange ring -- is UNDEFINED: the
 * step in which the time-out happened still publishes a payload computed from the missing fragment and stores its
 * activations before the workgroups return.  The word is STICKY -- later launches never clear it --
 * so one check after synchronising at the end of a step covers every launch of the step; after a
 * reported timeout the caller zeroes the whole workspace before using it again.
 * All workgroups of a launch must be resident at once: ds2_gru_persistent_supported(B, H) answers for
 * the CURRENT device from its compute-unit count (grid <= 15/16 of the CUs, so a concurrent RCCL
 * kernel keeps some) and the shape limits (H % 16 == 0, B <= 64); the launch itself re-checks the
 * chosen kernel with hipOccupancyMaxActiveBlocksPerMultiprocessor.  DS2_ERR_UNSUPPORTED = use the
 * per-step calls above instead (partitioned or smaller devices). */
size_t ds2_gru_sync_ws_bytes(int B, int H);
size_t ds2_gru_sync_error_offset(void);
int ds2_gru_persistent_supported(int B, int H);
int ds2_gru_bidir_fwd_persistent(float* G, float* ghn, float* hout, const float* w_hh, void* sync_ws, int T,
                                 int B, int H, void* stream);
int ds2_gru_bidir_bwd_persistent(float* G, float* ghn, const float* hout, const float* d_out,
                                 const float* w_hh_t, void* sync_ws, int T, int B, int H, void* stream);
/* The same with a hint (ABI revision 401): spare_cus = compute units the caller wants left FREE beside this launch for work
 * it has queued on other streams (the weight-gradient GEMMs of the layer above, a collective); < 0 = the library's default.
 * For B = 9 .. 12 the kernel exists with 20, 24 or 28 hidden units per workgroup (240 / 204 / 174 workgroups at H = 800;
 * 2.65 / 2.82 / 3.0 us per time step stand-alone at B = 10): the launch takes the widest grid that leaves spare_cus free.
 * Same results, same workspace; elsewhere the hint is ignored. */
int ds2_gru_bidir_bwd_persistent_ex(float* G, float* ghn, const float* hout, const float* d_out,
                                    const float* w_hh_t, void* sync_ws, int T, int B, int H, int spare_cus, void* stream);
/* The d(h) hand-off form of the backward recurrence (ABI revision 402; the reference: cuDNN's GRU backward under
 * codes/model.py:51-52,62).  d(gh)_t[b, g, j] = dh_t[b, j] * c_g[t, b, j], where the three coefficient planes
 *     c_r = (1 - z)(1 - n^2) gh_n r (1 - r),   c_z = (h_prev - n) z (1 - z),   c_n = (1 - z)(1 - n^2) r
 * depend on the forward pass's saved activations only.  With `coef` (T, B, 2, 3H) = (c_r | c_z | c_n) per (t, b, direction)
 * row given, the workgroups hand off dh_t (H values per batch row and step) instead of d(gh)_t (3H) and rebuild d(gh) at
 * the consumer from coefficient fragments loaded a step ahead.  Same inputs / outputs / workspace / error contract as
 * ds2_gru_bidir_bwd_persistent_ex; results equal up to fp32 rounding of the re-associated products.
 *   ds2_gru_bwd_coef:          coef from G (= r, z, n), ghn, hout as ds2_gru_bidir_fwd* left them (an elementwise pass)
 *   ds2_gru_bwd_dh_supported:  1 where the form is built (H = 800, 5 <= B <= 12; H = 64 for tests), else 0 -- the caller
 *                              then uses ds2_gru_bidir_bwd_persistent_ex; the launch itself returns DS2_ERR_UNSUPPORTED */
int ds2_gru_bwd_dh_supported(int B, int H);
/* ds2_gru_bidir_fwd_persistent with the coefficient planes as a fourth output: coef (T, B, 2, 3H) or NULL.  Where the forward
 * form that runs can (B = 9 .. 12 at H = 800) its own gate threads write them; otherwise ds2_gru_bwd_coef runs behind the
 * launch on the same stream.  A training forward pass that will call ds2_gru_bidir_bwd_persistent_dh asks for them here. */
int ds2_gru_bidir_fwd_persistent_ex(float* G, float* ghn, float* hout, const float* w_hh, float* coef, void* sync_ws, int T,
                                    int B, int H, void* stream);
int ds2_gru_bwd_coef(const float* G, const float* ghn, const float* hout, float* coef, int T, int B, int H, void* stream);
int ds2_gru_bidir_bwd_persistent_dh(float* G, float* ghn, const float* hout, const float* d_out, const float* w_hh_t,
                                    const float* coef, void* sync_ws, int T, int B, int H, int spare_cus, void* stream);

/* ------------------------------------------------------------------ output head helpers
 * softmax over the last dim of (rows, A) (eval branch, codes/model.py:204-205) and the argmax
 * of GreedyDecoder.decode (codes/decoder.py:154; ties -> lowest index). */
int ds2_softmax_rows(const float* x, int rows, int A, float* y, void* stream);
int ds2_argmax_rows(const float* x, int rows, int A, int32_t* idx, void* stream);
/* greedy collapse on device: best (B,T) int32 (batch-major), sizes (B) -> out_ids (B,T) and
 * out_offsets (B,T) compacted per row, out_lens (B)   (codes/decoder.py:123-140) */
int ds2_greedy_collapse(const int32_t* best, const int32_t* sizes, int B, int T, int blank,
                        int32_t* out_ids, int32_t* out_offsets, int32_t* out_lens, void* stream);

/* ------------------------------------------------------------------ host-side scoring / decoding helpers
 * HOST pointers, no stream.
 * ds2_edit_distance: Levenshtein distance of two int32 sequences (>= 0; negative = error).  Replaces the
 *   python-Levenshtein calls under Decoder.wer / Decoder.cer (codes/decoder.py:20,49-78).
 * ds2_ctc_beam_search: CTC prefix beam search over ONE utterance's (T, A) probabilities (log_input = 0) or
 *   log-probabilities (log_input = 1); writes the best labelling (<= out_cap labels), the frame each label first
 *   appeared at, its length and log-probability.  Not in the reference (test.py:21 offers greedy / none only);
 *   SURVEY.md 8f rank 4. */
int ds2_edit_distance(const int32_t* a, int na, const int32_t* b, int nb);
int ds2_ctc_beam_search(const float* probs, int T, int A, int blank, int beam_width, int log_input,
                        int32_t* out_labels, int32_t* out_offsets, int out_cap, int* out_len, float* out_logp);

/* ------------------------------------------------------------------ CTC
 * Replaces warpctc_pytorch.CTCLoss (train.py:179, codes/engine.py:22, codes/metrics.py:51):
 * softmax over A inside, blank 0, costs[b] = -log p(labels_b | acts[:act_lens[b], b]),
 * grad = grad_scale * d(sum_b costs[b]) / d acts (zero for t >= act_lens[b]; zero for an utterance
 * whose alignment is infeasible, whose cost is +inf).  grad_scale carries the 1/B of
 * codes/engine.py:23,80 so no separate scaling pass is needed.
 * zero_batch_if_inf = 1 adds the training step's rule (_sanitize_loss, codes/engine.py:24-30): when
 * the batch loss is +-inf the reference replaces it by 0 * loss, so the WHOLE batch's gradient is
 * zero (the optimizer step still runs, on momentum alone); 0 = warp-ctc's per-utterance contract.
 *   acts (T,B,A); labels flat int32 (sum label_lens); label_offsets (B) int32 start of each
 *   utterance's labels; ws >= ds2_ctc_ws_bytes(T,B,A,max_label_len) bytes
 */
size_t ds2_ctc_ws_bytes(int T, int B, int A, int max_label_len);
int ds2_ctc_loss_grad(const float* acts, const int32_t* labels, const int32_t* label_offsets,
                      const int32_t* label_lens, const int32_t* act_lens, int T, int B, int A,
                      int max_label_len, float grad_scale, int zero_batch_if_inf, float* costs,
                      float* grad, void* ws, void* stream);

/* ------------------------------------------------------------------ optimiser
 * clip_grad_norm_(params, max_norm) + SGD(momentum, nesterov) (codes/engine.py:87-90) over ONE
 * flat fp32 buffer of n elements (the model keeps params / grads / momentum flat).
 *   ds2_sumsq: partial sums of squares -> out[0] (double) ; ws >= ds2_sumsq_ws_bytes(n)
 *   ds2_clip_sgd_nesterov: coef = min(1, max_norm / (sqrt(sumsq[0] * norm_scale^2) + 1e-6)) is
 *   computed ON DEVICE from sumsq (no host sync); g' = g * grad_scale * coef;
 *   buf = first_step ? g' : momentum*buf + g'; p -= lr * (g' + momentum*buf)
 * grad_scale lets the data-parallel average (1/world) fold into the update.
 */
size_t ds2_sumsq_ws_bytes(size_t n);
int ds2_sumsq(const float* x, size_t n, double* out, void* ws, void* stream);
int ds2_clip_sgd_nesterov(float* p, const float* g, float* buf, size_t n, const double* sumsq,
                          float grad_scale, float max_norm, float lr, float momentum, int first_step,
                          void* stream);

/* Everything the host reads back after a training step (codes/engine.py:92-94: synchronize, loss.item()),
 * gathered by ONE launch into out[0..3] (double): sum_b costs[b]; sumsq[0] (the squared gradient norm, or 0
 * when sumsq is NULL); the OR of the n_err 32-bit words err_words[i] point at (device pointers held in a
 * device array; the persistent recurrence's sticky timeout flags); the number of +-inf costs. */
int ds2_step_stats(const float* costs, int B, const double* sumsq, const uint32_t* const* err_words, int n_err,
                   double* out, void* stream);

/* misc elementwise used by the layer glue */
int ds2_add2(const float* a, const float* b, size_t n, float* out, void* stream);

/* A HIP stream of the given priority (hipDeviceGetStreamPriorityRange: 1 = low, 0 = normal, -1 = high on
 * gfx950).  torch.cuda.Stream clamps priorities to [-1, 0], so the LOW-priority stream the weight-gradient GEMMs run
 * on (under the next layer's recurrence kernel, without taking CUs from the critical chain) is created here and
 * wrapped with torch.cuda.ExternalStream.  The caller owns the stream. */
int ds2_stream_create(int priority, void** stream_out);
int ds2_stream_destroy(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DS2HIP_H */
