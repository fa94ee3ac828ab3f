/* libslamhip.so -- C ABI of the MI355X (gfx950) hot path behind SLAM-LLM's model plugin surface.
 *
 * The reference (X-LANCE/SLAM-LLM) has NO native code and NO FFI: every operator below replaces a
 * PyTorch / HF-transformers / peft / openai-whisper library call made from the reference's Python glue.
 * Each entry point cites the reference call site (file:line relative to the reference root) it serves.
 * The ctypes binding a maintainer adds on the reference side is shown in INTEGRATION.md
 * (slam_llm_amd/lib.py is that binding).
 *
 * Conventions
 *  - all pointers are DEVICE pointers owned by the caller (torch tensors kept alive by the caller);
 *    the library never allocates, never frees, never synchronises the stream or the device;
 *  - `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream), calls are thread-agnostic;
 *  - bf16 tensors are raw uint16 bit patterns; leading dimensions (ld*) are in ELEMENTS;
 *  - return 0 on success, < 0 on error; slam_last_error() returns a thread-local message;
 *  - the binding must raise (RuntimeError) on non-zero -- there is no CPU fallback anywhere;
 *  - collectives are NOT part of this ABI, by design: the one exchange of the path (mean all-reduce of the flat fp32 gradient
 *    buffer per optimizer step; reference: DDP at src/slam_llm/pipeline/finetune.py:181-184) runs through torch.distributed's
 *    backend "nccl" = RCCL over xGMI on the very device buffers these entry points write (slam_llm_amd/train.py: GradSync, or
 *    torch's DistributedDataParallel around the module).  SURVEY 8(b) sketched `slam_allreduce_flat(buf, n, dtype, rcclComm_t,
 *    stream)`; it would wrap the same ncclAllReduce and make the caller own a second communicator next to torch's, so it is not
 *    exported.
 */
#ifndef SLAM_HIP_H
#define SLAM_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLAM_BF16 0
#define SLAM_F32 1
#define SLAM_ACT_NONE 0
#define SLAM_ACT_GELU 1 /* exact erf GELU (F.gelu, src/slam_llm/models/encoder.py:18-19) */
#define SLAM_ACT_RELU 2 /* nn.ReLU, src/slam_llm/models/projector.py:25 */

const char* slam_last_error(void);
int slam_abi_version(void);
const char* slam_target_arch(void); /* "gfx950" */
/* every process-global tuning knob (slam_gemm_set_config / _set_group_m, slam_attn_set_fwd_qf / _set_bwd_variant) back to the default
 * it is defined with: the defaults live in the library, not in its callers (tests reset between cases; no reference counterpart:
 * the reference has no kernel-selection state) */
int slam_reset_tuning(void);
/* dropout masks under hipGraph replay: every dropout-aware kernel (slam_dropout_bf16, slam_lora_a_fwd, slam_lora_hop_dropout,
 * slam_skinny_gram, the DROP attention kernels) XORs *device_word into its seed when one is registered; a captured step bumps the word
 * at its end so that each replay draws fresh masks (torch's dropout under CUDA graphs does the same through its philox offset).
 * null (default) detaches: seeds are used as passed.  Reference: peft lora_dropout / Blip2QFormer dropout draw from torch's RNG. */
int slam_set_dropout_salt(const unsigned long long* device_word);

/* ---- a1: log-mel front end ------------------------------------------------------------------
 * whisper.pad_or_trim + whisper.log_mel_spectrogram, src/slam_llm/datasets/speech_dataset.py:101-103,
 * src/slam_llm/datasets/speech_dataset_large.py:102-104.
 * audio [B, ld_audio] f32; n_valid[b] (nullable) = samples of clip b that are real (rest treated as the
 * zero padding of pad_or_trim); n_samples = padded length (480000); window400 = periodic Hann;
 * twiddle_folded = the folded real DFT's twiddles, fp32 [13 bin tiles][cos | sin][13 k-quads][64 lanes][4]: lane 16 g + li, k-step
 * ks = 4 kq + j -> bin 16 tile + li at sample index n = 4 ks + g: cos(2 pi bin n / 400) for n <= 200 / sin for n < 200, zero beyond and
 * for bins > 200 (slam_llm_amd/ops.py logmel_twiddle_table builds it); mel_filters_T [201, n_mels];
 * out_mel [B, n_samples/160, n_mels] f32; workspace: slam_logmel_workspace_bytes(B) bytes.
 * per_clip != 0: pad_or_trim OFF (speech_dataset_large.py:102-104 with pad_or_trim=false): each clip's STFT runs over its own
 * n_valid[b] samples, it owns n_valid[b]/160 frames, the remaining rows are the collator's mel-space zeros. */
int64_t slam_logmel_workspace_bytes(int64_t B);
int slam_logmel_fwd(const float* audio, int64_t ld_audio, const int32_t* n_valid, int64_t n_samples,
                    const float* window400, const float* twiddle_folded, const float* mel_filters_T,
                    int64_t n_mels, float* out_mel, int32_t* workspace, int64_t B, int per_clip, void* stream);

/* ---- GEMM: every Linear / Conv1d-as-GEMM / lm_head on the path ---------------------------------
 * C[M,N] = epilogue(alpha * A[M,K] . B[N,K]^T): +bias[N] (f32), act, +residual[(m % res_row_mod), n] (bf16),
 * optional accumulate into C, C bf16 or f32.  K % 64 == 0, N % 4 == 0, 16-byte aligned operands.
 * lda may be SMALLER than K: the rows of A then overlap -- row t of a strided Conv1d's im2col matrix over a row-major [T, C]
 * signal is the k * C contiguous elements from row stride * t on, i.e. A = signal, lda = stride * C, K = k * C, no copy.
 * act: 0 none, 1 GELU(erf), 2 ReLU, 3 SwiGLU backward: the product is dL/dh of h = silu(gate) * up (HF LlamaMLP), `residual`
 * holds the forward's [gate | up] ([M, 2N]); dL/dgate goes to C[:, :N] and dL/dup to C[:, N:2N] (no residual add).
 * Sites: Whisper linears/convs (src/slam_llm/models/encoder.py:18-29), projector (projector.py:24-26),
 * Llama linears + lm_head (slam_model.py:400), and all their backward products (via stored W^T). */
int slam_gemm_bf16_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                      int64_t M, int64_t N, int64_t K, const float* bias, const void* residual,
                      int64_t ldr, int64_t res_row_mod, int act, float alpha, int out_dtype,
                      int accumulate, void* stream);
/* tile configuration override: 0 auto, 1 128x128/4 waves, 2 256x128/8 waves, 3 128x64, 4 256x256/8 waves (2-stage loop),
 * 5/6 256x256 register-double-buffered pipeline (6 = shipped schedule) */
/* Split-K tail of the 4-wave 256x256 kernel ("stream-K" for the partial last round of tiles): when T tiles leave R = T mod 256
 * (<= 128) for the last round, those R tiles are computed by R x S workgroups over 1/S of K each, inside the SAME launch; partial
 * sums go to fp32 slabs, the last arriver of a tile adds them in slice order (bit-reproducible) and runs the epilogue.  The slabs
 * live in a caller-owned device buffer (the library never allocates): `workspace` = 4096 bytes of ZEROED arrival counters followed
 * by up to 256 slabs of 256 KiB (64 MiB + 4 KiB serves every plan; smaller buffers shrink S).  No workspace = no split.  All
 * launches that may split must be ordered on one stream.  null / 0 detaches.  The plan is OFF by default
 * (slam_gemm_set_config(300) = auto plan, 302..316 = forced slices): measured on MI355X the hand-off costs what the idle CUs of the
 * last round would have saved (profiles/r03_gemm_splitk.md). */
int slam_gemm_set_workspace(void* workspace, int64_t bytes);
int slam_gemm_set_group_m(int group_m);   /* tuning knob: M-tiles per raster group of the 256x256 kernels; 0 (default) = the per-shape rule, 1..64 forced */
int slam_gemm_set_group_m_rule(int short_k_many_cols, int long_k_narrow, int wide_k4096, int other);   /* tools: the per-shape rule's four values (defaults 12, 4, 4, 8) */
int slam_gemm_set_config(int cfg);   /* also: 100+v / 200+v = the 256x256 kernel of the auto rule for K > 2048 / <= 2048; 300 / 301 / 302..316 = split-K tail auto / off / forced slices; 400 / 401 = cycle stamps off / on (tools) */
/* tools only (after slam_gemm_set_config(401): production launches never write the stamps): {shader cycles, 100 MHz ticks} at
 * entry and exit of workgroup 0 of the last pipelined-kernel launch (synchronise first); effective shader clock of the launch = d(cycles) / d(ticks) x 100 MHz -- how the DVFS cost of a variant is read */
int slam_gemm_debug_clock(unsigned long long* out6);   /* [4] = cycles at the k-loop's start, [5] = after the epilogue */

/* ---- conv front end (src/slam_llm/models/encoder.py:18-19): k=3, pad=1 im2col, stride 1|2 ---------
 * in [B, Tin, C] (f32 or bf16) -> out [B*Tout, Kp] bf16, column j*C + c = in[b, t*stride + j - 1, c]. */
int slam_conv1d_k3_im2col(const void* in, int in_dtype, void* out, int64_t B, int64_t Tin, int64_t C,
                          int64_t stride, int64_t Kp, const int32_t* n_valid, void* stream);
/* n_valid (nullable, [B]): frames at or beyond n_valid[b] read as zero (ragged encoder: each clip sees the zero padding it
 * would see alone; the reference zero-pads in mel space, speech_dataset_large.py:194-197, and has no length argument). */
/* adjoint of the above for unfrozen-encoder training (train_config.freeze_encoder=false, src/slam_llm/models/slam_model.py:110-113:
 * autograd then runs F.conv1d's backward, models/encoder.py:18-19): dcols [B*Tout, ldc >= 3C] bf16 = dz . W -> dx [B, Tin, C] bf16,
 * dx[b, t, c] = sum of dcols[b, o, j*C + c] over stride*o + j - 1 == t. */
int slam_conv1d_k3_col2im(const void* dcols, int64_t ldc, void* dx, int64_t B, int64_t Tin, int64_t C, int64_t stride,
                          void* stream);

/* row gather, bf16: dst[r, 0:width) = src[idx[r]*src_stride + 0:width), idx[r] < 0 -> zeros.  Packs the valid rows of a padded
 * batch, un-packs them (inverse index), and builds the projector's k-frame windows (models/projector.py:15-23) over a packed
 * encoder output (width = k*d spans k consecutive rows). */
int slam_gather_rows_bf16(const void* src, int64_t src_stride, const int32_t* idx, void* dst, int64_t ld_dst, int64_t n,
                          int64_t width, void* stream);

/* general conv1d im2col (HuBERT/WavLM feature encoder and grouped positional conv, models/slam_model.py:335-341):
 * rows (b*Tin+t) with stride ld_in, channel slice [c0, c0+C) -> out [B*Tout, Kp] bf16, col = j*C + c. */
int slam_conv1d_im2col(const void* in, int in_dtype, int64_t ld_in, int64_t c0, int64_t C, void* out, int64_t B,
                       int64_t Tin, int64_t k, int64_t stride, int64_t pad, int64_t Kp, int64_t Tout_limit,
                       void* stream);

/* ---- norms -------------------------------------------------------------------------------------
 * LayerNorm (Whisper blocks + ln_post, encoder.py:26-29; fp32 statistics), RMSNorm fwd/bwd (Llama). */
/* act: 0 none, 1 exact GELU fused after the affine (HuBERT feature-encoder LayerNorm+GELU) */
int slam_layernorm_fwd(const void* x, int64_t ldx, const float* weight, const float* bias, void* y,
                       int64_t ldy, int64_t M, int64_t d, float eps, int act, float* mean_out, float* rstd_out,
                       void* stream);
/* LayerNorm backward (trainable Q-Former projector, projector.py:51-80): dx (nullable) and dgamma/dbeta (nullable, f32) */
int slam_layernorm_bwd(const void* x, int64_t ldx, const float* mean, const float* rstd, const float* weight,
                       const void* dy, int64_t lddy, void* dx, int64_t lddx, float* dgamma, float* dbeta, int64_t M,
                       int64_t d, int accumulate, void* stream);
/* exact GELU forward/backward on the pre-activation z (Q-Former feed-forward) */
int slam_gelu_fwd(const void* z, int64_t ldz, void* y, int64_t ldy, int64_t M, int64_t N, void* stream);
int slam_gelu_bwd(const void* z, int64_t ldz, const void* dy, int64_t lddy, void* dz, int64_t lddz, int64_t M,
                  int64_t N, void* stream);
int slam_rmsnorm_fwd(const void* x, int64_t ldx, const float* weight, void* y, int64_t ldy, float* rstd,
                     int64_t M, int64_t d, float eps, void* stream);
/* dx = rmsnorm'(dy) * (*grad_scale or 1) + dres (nullable) */
int slam_rmsnorm_bwd(const void* x, int64_t ldx, const float* rstd, const float* weight, const void* dy,
                     int64_t lddy, const void* dres, int64_t lddres, void* dx, int64_t lddx,
                     const float* grad_scale, int64_t M, int64_t d, void* stream);

/* ---- RoPE + head transposes (HF apply_rotary_pos_emb; positions = arange(T)) -------------
 * src rows (b*T+t), columns col0 + h*D + d, rotated IN PLACE when cos/sin tables [T, D/2] are given
 * (inverse != 0 applies the transposed rotation = RoPE backward); dstT (nullable) [B,H,D,Tp] gets the
 * (rotated) values transposed, zero padded to Tp (multiple of 64).
 * positions (nullable, int32 [B*T]): explicit RoPE positions (HF generate derives them from the attention mask). */
int slam_head_rope_transpose(void* src, int64_t ld, int64_t col0, const float* cos_table,
                             const float* sin_table, int inverse, void* dstT, int64_t B, int64_t T,
                             int64_t Tp, int64_t H, int64_t D, const int32_t* positions, void* stream);
int slam_transpose_bf16(const void* in, int64_t ldi, void* out, int64_t ldo, int64_t R, int64_t C,
                        int64_t Rp, void* stream);

/* ---- attention (Whisper blocks: bidirectional, no mask; Llama: causal ^ key padding, GQA) -----------
 * Q/K/V/O/dO row-major [B*T, ld] with head h at column h*D; Vt/Kt/Qt/dOt = [B,H,D,Tp] transposed copies (optional since round 4, see below);
 * LSE/Delta [B,Hq,Tqp] f32; key_mask [B,Tkp] uint8 (1 = attend, zero padded) or NULL; D in {64,128}.
 * Query rows are (b*Tq + t), key/value rows (b*Tk + t): Tq != Tk is cross-attention (Q-Former, projector.py:69-80);
 * Tqp/Tkp are the 64-padded lengths used by the transposed copies, LSE/Delta and the mask.
 * slam_attn_bwd rope_cos/rope_sin (nullable, [T, D/2] f32): when set, dQ and dK are the gradients w.r.t. the
 * PRE-RoPE q/k (the backward of HF apply_rotary_pos_emb is applied in the epilogue; position = row index, or
 * rope_pos[row] when given).
 * Packed ("varlen") batches: B = 1 with several sequences concatenated along T (no pad tokens, SURVEY 8d ragged set);
 * seg_lo[q] = first row of q's sequence, seg_hi[k] = one past the last row of k's sequence (int32, non-decreasing,
 * nullable): query q attends keys seg_lo[q] <= k <= q only.  Equals the right-padded batch of the reference's
 * MultiTaskDataset collator (speech_dataset_large.py:180-233) on every valid token. */
int slam_attn_set_fwd_qf(int qf);   /* tools: 0 = auto, 1 / 2 = query fragments per wave of the forward kernel; 10 / 11 = register-staged / LDS-DMA tiles; 20 / 21 = hardware round-robin / XCD-aware (shipped) workgroup numbering of ALL attention kernels; 30 / 31 = general / mask-free Whisper instantiation; 40 / 41 = round-3 kernels on the transposed copies / transposed-read kernels (shipped) */
int slam_attn_debug_clock(unsigned long long* out256);   /* tools: cycle stamps written by the probe form of the dQ kernel (variant 14): [wave 0|3][tile < 16][8] */
int slam_attn_set_bwd_variant(int variant);   /* tools: 0 = DMA-ring backward kernels (shipped), 1 = the round-1 register-staged kernels, 2 = ring dK/dV with the register-staged dQ, 14 = dQ kernel with cycle stamps, 11/12/15 = timing ablations (wrong results) */
/* Round 4: the kernels read their transposed MFMA operands with ds_read_b64_tr_b16 from the ROW-MAJOR tiles, so the [B,H,D,Tp] copies are
 * optional: slam_attn_fwd takes V row-major (V, ldv; like K) and / or Vt -- with V given it never reads Vt; slam_attn_bwd reads Qt / Kt / dOt
 * only in the configurations slam_attn_needs_transposed reports (bit 1: <= 64 queries, attention dropout, the relative position bias,
 * tensors beyond 2 GiB, the tools' variants), NULL otherwise.  flags: bit 0 = dropout, bit 1 = relative position bias. */
int slam_attn_needs_transposed(int64_t B, int64_t Tq, int64_t Tk, int64_t Hq, int64_t Hkv, int64_t D, int64_t ldq, int64_t ldk,
                               int64_t ldv, int64_t lddo, int flags);
int slam_attn_fwd(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* Vt, const void* V, int64_t ldv, void* O,
                  int64_t ldo, float* LSE, const uint8_t* key_mask, int64_t B, int64_t Tq, int64_t Tk,
                  int64_t Tqp, int64_t Tkp, int64_t Hq, int64_t Hkv, int64_t D, int causal, float scale,
                  const int32_t* seg_lo, const int32_t* seg_hi, const float* rp_gate, const float* rp_tab, int64_t rp_T,
                  int64_t rp_ld, float drop_p, uint64_t drop_seed, void* stream);
/* scale < 0 (forward only, round 5): Q arrives multiplied by |scale| * log2(e) -- the caller folded the softmax scale and the base change
 * of the exponent into a FROZEN query projection (W_q, b_q scaled in fp32 when the checkpoint is loaded: Q is still rounded to bf16 once,
 * Whisper blocks of models/encoder.py:26-27); the kernels then take the scores of the first product as they are, and LSE-less launches of
 * the mask-free bidirectional D = 64 form start their score accumulators at -max (no multiply-add per score in the softmax).  LSE, when
 * requested, keeps its meaning (natural-log units of the scaled scores).  Not with rp_gate or drop_p. */
/* drop_p > 0 (forward and backward, D = 64 bidirectional): dropout on the attention PROBABILITIES (HF Blip2QFormer
 * attention_probs_dropout_prob inside EncoderProjectorQFormer, models/projector.py:51-67, train mode): P keeps its full-row
 * normalisation, element (b, h, q, k) is kept with slam_dropout_bf16's counter-based mask at index ((b*Hq + h)*Tqp + q)*Tkp + k
 * (seed = drop_seed, offset 0) and scaled by 1/(1-p); slam_attn_bwd recomputes the same mask. */
/* rp_gate / rp_tab (nullable, forward only, D = 64, bidirectional): WavLM's gated relative position bias
 * (src/slam_llm/models/wavlm/modules.py:504-533, called from models/slam_model.py:333-334 through models/encoder.py:109-127):
 * score(q, k) = scale q.k + rp_gate[b][h][q] * rp_tab[h * rp_ld + (k - q + rp_T - 1)]; rp_gate [B, Hq, Tqp] f32 from
 * slam_wavlm_gate, rp_tab = the bucketed `relative_attention_bias` of layer 0 laid out by relative distance (host table),
 * with 64 readable floats before and after each row's 2 rp_T - 1 entries. */
int slam_wavlm_gate(const void* x, int64_t ldx, const float* w, const float* bias, const float* grep_a, float* gate,
                    int64_t B, int64_t T, int64_t H, int64_t Tp, void* stream);
/* backward of slam_wavlm_gate (unfrozen WavLM): from dgate [B, H, Tp] f32 -> dv [B*T*H, 8] bf16 (dL/d grep_linear outputs: the weight
 * gradient is the tall-skinny gram dv^T x_h, the bias gradient its column sums), da_term [B*T, Hp] bf16 (column sums = d grep_a; the
 * caller zero-fills the Hp - H padding columns), dx [B*T, lddx] bf16 (the gate's share of dL/d(attention input), every column of the
 * H*64 written). */
/* chain rule of nn.utils.weight_norm(dim = 2) (WavLM's pos_conv.0.weight_g / weight_v, models/wavlm/WavLM.py:378-386): dw, v, dv are
 * [rows, K] f32 views (rows = d * channels per group, K = taps), g / dg [K]; accumulate adds into dg / dv. */
int slam_weight_norm_bwd(const float* dw, const float* v, const float* g, float* dg, float* dv, int64_t rows, int64_t K, int accumulate,
                         void* stream);
/* gradient of layers.0.self_attn.relative_attention_bias.weight [num_buckets, H] from the gradient of the bias table over the relative
 * distances (d_table[h][r], r < n = 2T - 1; buckets[r] = bucket of distance r - (T - 1), modules.py:417-455). */
int slam_relpos_bucket_grad(const float* d_table, int64_t ld, const int32_t* buckets, int64_t n, int64_t H, int64_t num_buckets, float* out,
                            int accumulate, void* stream);
int slam_wavlm_gate_bwd(const void* x, int64_t ldx, const float* w, const float* bias, const float* grep_a, const float* dgate,
                        void* dv, void* da_term, void* dx, int64_t lddx, int64_t B, int64_t T, int64_t H, int64_t Tp, int64_t Hp,
                        void* stream);
/* GroupNorm with one group per channel over time + exact GELU: first conv layer of the "default" feature extractor (WavLM Base,
 * reference src/slam_llm/models/wavlm/WavLM.py:428-441, Fp32GroupNorm(dim, dim)).  x fp32 [B*T, ldx] (time rows), y bf16 [B*T, ldy];
 * every (clip, channel) is normalised over its T rows (biased variance); workspace of slam_groupnorm_time_workspace_bytes bytes. */
int64_t slam_groupnorm_time_workspace_bytes(int64_t B, int64_t T, int64_t C);
int slam_groupnorm_time_gelu(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t B, int64_t T, int64_t C,
                             const float* weight, const float* bias, float eps, float* workspace, void* stream);
/* backward of slam_groupnorm_time_gelu (unfrozen base-geometry encoders): x is the fp32 conv output the forward normalised, stats the
 * [B, 2, C] (mean, rstd) block the forward left at workspace + B * ceil(T / 256) * 2 * C floats, dy / dx bf16 [B*T, C];
 * dgamma / dbeta [C] f32 (accumulate adds).  workspace: slam_groupnorm_time_workspace_bytes(B, T, C). */
int slam_groupnorm_time_gelu_bwd(const float* x, int64_t ldx, const void* dy, int64_t lddy, const float* stats, const float* weight,
                                 const float* bias, void* dx, int64_t lddx, float* dgamma, float* dbeta, int64_t B, int64_t T, int64_t C,
                                 int accumulate, float* workspace, void* stream);
/* gate[b][h][t] = a * (g * grep_a[h] - 1) + 2 with (a, g) = sigmoid of the two 4-sums of grep_linear(x[b, t, h*64:(h+1)*64])
 * (x = the layer's attention INPUT [B*T, H*64] bf16, w [8, 64] / bias [8] f32 = grep_linear; modules.py:522-531). */
int slam_attn_bwd(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv,
                  const void* Qt, const void* Kt, const void* O, int64_t ldo, const void* dO,
                  int64_t lddo, const void* dOt, const float* LSE, float* Delta, const uint8_t* key_mask,
                  void* dQ, int64_t lddq, void* dK, int64_t lddk, void* dV, int64_t lddv, int64_t B,
                  int64_t Tq, int64_t Tk, int64_t Tqp, int64_t Tkp, int64_t Hq, int64_t Hkv, int64_t D,
                  int causal, float scale, const float* rope_cos, const float* rope_sin, const int32_t* rope_pos,
                  const int32_t* seg_lo, const int32_t* seg_hi, float drop_p, uint64_t drop_seed, const float* rp_gate,
                  const float* rp_tab, int64_t rp_T, int64_t rp_ld, float* rp_ds, float* d_gate, float* d_tab, void* stream);
/* rp_gate (nullable) ... d_tab: backward of WavLM's gated relative position bias (unfrozen WavLM; models/wavlm/modules.py:504-533;
 * head_dim 64, bidirectional MHA): rp_gate / rp_tab / rp_T / rp_ld as in slam_attn_fwd; rp_ds = scratch [B, Hq, Tq, Tkp] f32 that
 * receives dL/d(score); d_gate [B, Hq, Tqp] f32 receives dL/d(gate); d_tab [Hq, rp_ld] f32 is ACCUMULATED into (the table is
 * shared by all layers: zero it once per backward). */

/* ---- a11: grouped positional convolution of HuBERT / WavLM as one implicit-GEMM launch ------------------------------------
 * fairseq `pos_conv` (Conv1d(d, d, k = taps, padding = taps / 2, groups) + SamePad + GELU) and the residual add around it
 * (src/slam_llm/models/wavlm/WavLM.py:378-386, 575-580; the same graph in fairseq's HuBERT, models/slam_model.py:335-341):
 *   x[b, t, g*C + co] = h[b, t, g*C + co] + gelu(bias[g*C + co] + sum_j sum_ci W[g*C + co, ci, j] * h[b, t + j - taps/2, g*C + ci])
 * h, x: [B*T, ld] bf16 (x != h); w_packed: [groups][taps][C (co)][C padded to a multiple of 32 (ci, zero padded)] bf16 -- the
 * weight-normed conv weight re-packed tap-major; bias [groups*C] f32.  C = channels per group in {32, 48, 64, 80}
 * (slam_pos_conv_supported: d = 512 / 768 / 1024 / 1280 with 16 groups), taps <= 256.  No im2col buffer exists: the taps are LDS row
 * offsets into one input window per workgroup.
 * General form (the training path): out = residual + act(conv(h) + bias) with `pad` rows of left zero padding (forward: taps / 2);
 * bias nullable; act 1 = GELU, 0 = none; residual nullable = h itself (read from the input window); pre nullable: also receives
 * conv(h) + bias before the activation (kept for the backward).  The ADJOINT dL/dh = dL/dx + conv^T(dL/dconv) is the same launch
 * on dL/dconv with tap-reversed, channel-transposed weights, pad = taps - 1 - taps / 2, act 0, no bias, residual = dL/dx
 * (unfrozen-encoder training, models/slam_model.py:110-113). */
int slam_pos_conv_supported(int64_t channels_per_group, int64_t taps);
int slam_pos_conv_fwd(const void* h, int64_t ldh, const void* w_packed, const float* bias, void* x, int64_t ldx, void* pre,
                      int64_t ldpre, const void* residual, int64_t ldr, int64_t B, int64_t T, int64_t groups,
                      int64_t channels_per_group, int64_t taps, int64_t pad, int act, void* stream);
/* adjoint of slam_conv1d_im2col without padding (kernel k, stride s): dx[b, t, c] = sum of dcols[b, o, j*C + c] over t = s*o + j;
 * the conv feature extractor's dL/d(input) of the unfrozen HuBERT encoder (fairseq ConvFeatureExtractionModel backward). */
int slam_conv1d_col2im(const void* dcols, int64_t ldc, void* dx, int64_t B, int64_t Tin, int64_t C, int64_t k, int64_t stride,
                       void* stream);

/* ---- SwiGLU (LlamaMLP) : gate_up [M, 2F] = [gate | up] ---------------------------------------------- */
int slam_swiglu_fwd(const void* gate_up, int64_t ldgu, void* h, int64_t ldh, int64_t M, int64_t F,
                    void* stream);
/* interleaved != 0: gate_up is in the block layout written by slam_gemm_swiglu_bf16_nt (below); dgate_up is always [dgate | dup] */
int slam_swiglu_bwd(const void* gate_up, int64_t ldgu, const void* dh, int64_t lddh, void* dgate_up,
                    int64_t lddgu, int64_t M, int64_t F, int interleaved, void* stream);
/* gate|up product of LlamaMLP (transformers/models/llama/modeling_llama.py LlamaMLP.forward, reached from
 * src/slam_llm/models/slam_model.py:400) with act_fn(gate) * up in its epilogue: GU[M, N] = A[M, K] . B[N, K]^T and
 * H[M, N/2] = silu(gate) * up from ONE launch (the stand-alone slam_swiglu_fwd pass re-read GU from HBM).  B's rows are
 * INTERLEAVED in blocks of 64: rows [128 b, 128 b + 64) = gate_proj rows [64 b, 64 b + 64), rows [128 b + 64, 128 b + 128) =
 * the matching up_proj rows; GU comes out in the same block layout (slam_swiglu_bwd(..., interleaved = 1) reads it).  Served by
 * the 4-wave 256x256 kernel only: slam_gemm_swiglu_supported() returns 1 when the auto rule picks that kernel for the shape
 * (K > 2048, K % 64 == 0, N % 128 == 0, N >= 2048, operands within its 32-bit descriptors), else 0 -- then run
 * slam_gemm_bf16_nt + slam_swiglu_fwd.  silu is evaluated on the bf16-rounded gate / up, exactly like slam_swiglu_fwd. */
int slam_gemm_swiglu_supported(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb);
int slam_gemm_swiglu_bf16_nt(const void* A, int64_t lda, const void* B_interleaved, int64_t ldb, void* GU, int64_t ldgu,
                             void* H, int64_t ldh, int64_t M, int64_t N, int64_t K, void* stream);

/* ---- projector ReLU backward (projector.py:25) and LoRA weight packing (peft Linear: scaling = alpha/r) ---
 * relu_bwd: dh *= (h > 0) in place.  lora_pack_b: dst[row,j] = dstT[j,row] = bf16(scale*B[row,j]), B f32 [rows,r]:
 * writes the adapter into the K-extension columns of the fused weight [W | s*B] and of its transpose. */
int slam_relu_bwd(void* dh, int64_t lddh, const void* h, int64_t ldh, int64_t M, int64_t N, void* stream);
/* out[n] (+)= sum_m x[m,n] (bias gradient of nn.Linear, projector.py:24-26), fixed summation order */
int slam_colsum_bf16(const void* x, int64_t ldx, float* out, int64_t M, int64_t N, int accumulate, void* stream);
int slam_lora_pack_b(const float* B, float scale, void* dst, int64_t ld_dst, void* dstT, int64_t ld_dstT,
                     int64_t rows, int64_t r, void* stream);

/* LoRA gradients (peft Linear backward): out[r*ld_r + c*ld_c] (+)= alpha * sum_m S[m,r] * X[m,c];
 * S [M,R] bf16 (R in 8|16|32|64), X [M,C] bf16, out fp32; workspace: slam_skinny_gram_workspace_bytes(M,R,C). */
int64_t slam_skinny_gram_workspace_bytes(int64_t M, int64_t R, int64_t C);
int slam_skinny_gram(const void* S, int64_t lds, const void* X, int64_t ldx, float* out, int64_t out_ld_r,
                     int64_t out_ld_c, int64_t M, int64_t R, int64_t C, float alpha, int accumulate, float drop_p,
                     uint64_t seed, uint64_t offset, float* workspace, void* stream);

/* LoRA first hop u[M, R] = lora_dropout(x)[M, K] . A[R, K]^T (peft 0.6.0 Linear.forward: lora_A(lora_dropout(x))), R <= 64
 * = all adapters sharing x stacked.  drop_p > 0 applies the counter-based mask of slam_dropout_bf16 (same seed / offset /
 * element index -> same mask) in registers; slam_skinny_gram's drop_p does the same to X for dA = du^T dropout(x).
 * Rpad >= R (round 5): columns [R, Rpad) of U -- the zero padding of the K-extension the frozen GEMM runs over -- are written as
 * zeros by the same launch (Rpad = R: none). */
int slam_lora_a_fwd(const void* X, int64_t ldx, const void* A, int64_t lda, void* U, int64_t ldu, int64_t M, int64_t R,
                    int64_t Rpad, int64_t K, float drop_p, uint64_t seed, uint64_t offset, void* stream);

/* LoRA backward, second hop under lora_dropout (peft 0.6.0 Linear: lora_B(lora_A(lora_dropout(x)))): dx[M, K] += mask . (du[M, R] . At[K, R]^T) / (1 - p)
 * in ONE pass over dx -- du = dL/d(lora_A output) (R = 32 | 64 columns incl. zero padding), At = the adapters' A matrices stacked and transposed
 * ([K, R]), mask = slam_dropout_bf16's at index offset + m * K + k (what the forward's slam_lora_a_fwd applied to x).  Bit-identical to
 * slam_gemm_bf16_nt into a scratch buffer + slam_dropout_bf16(accumulate). */
int slam_lora_hop_dropout(const void* DU, int64_t lddu, const void* At, int64_t ldat, void* DX, int64_t lddx, int64_t M, int64_t K,
                          int64_t R, float drop_p, uint64_t seed, uint64_t offset, void* stream);

/* ---- embed + audio splice (src/slam_llm/models/slam_model.py:370-392) and its backward --------------
 * input_ids int64 [B,T] (-1 -> 0 in place), modality_mask uint8 [B,T], enc = projector output [B,Ta,ldenc],
 * out [B*T, ldo]; spans int32 [B,2] (start,len) is produced by fwd and consumed by bwd. No host sync. */
int slam_embed_splice_fwd(int64_t* input_ids, const uint8_t* modality_mask, const void* embed_table,
                          int64_t vocab, const void* enc, int64_t ldenc, void* out, int64_t ldo,
                          int32_t* spans, int64_t B, int64_t T, int64_t Ta, int64_t d, void* stream);
int slam_embed_splice_bwd(const int32_t* spans, const void* dX, int64_t lddx, void* denc, int64_t ldde,
                          int64_t B, int64_t T, int64_t Ta, int64_t d, void* stream);

/* ---- loss + accuracy (HF ForCausalLMLoss; slam_model.py:402-405; utils/metric.py:3-19) --------------
 * targets[b*T+t] = labels[b,t+1] (or -1 when ignored / t = T-1); n_valid = #valid targets (device int). */
int slam_ce_targets(const int64_t* labels, int32_t* targets, int32_t* n_valid, int64_t B, int64_t T,
                    int64_t ignore_index, void* stream);
/* device-side selection of the labelled rows (replaces a host read-back of the label count + a torch argsort; the reference runs
 * lm_head and CrossEntropyLoss(ignore_index=-100) over every row, src/slam_llm/models/slam_model.py:400-405): rows[i] / tsel[i] =
 * index / target of the i-th row with targets >= 0 (row order), -1 beyond the count up to `cap`; inv[r] = position of row r in rows or
 * -1; count[0] = number of labelled rows (a count above cap means the caller's bound was too small). */
int slam_label_rows(const int32_t* targets, int64_t M, int64_t cap, int32_t* rows, int32_t* tsel, int32_t* inv, int32_t* count,
                    void* stream);
/* per row: loss, argmax==target; when write_grad, logits are overwritten by dlogits = (softmax-onehot)/n_valid */
int slam_ce_fwd_bwd(void* logits, int64_t ld, const int32_t* targets, const int32_t* n_valid,
                    float* row_loss, int32_t* row_correct, int64_t rows, int64_t V, int write_grad,
                    void* stream);
/* out2[0] = mean loss over valid targets, out2[1] = accuracy */
int slam_ce_finalize(const float* row_loss, const int32_t* row_correct, const int32_t* n_valid,
                     int64_t rows, float* out2, void* stream);

/* ---- optimizer (torch.optim.AdamW, src/slam_llm/pipeline/finetune.py:247-251) ------------------------
 * one fused pass over the flat trainable buffer; also refreshes the bf16 compute copy (nullable). */
int slam_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* param_bf16,
                    int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                    int64_t step, float grad_scale, void* stream);

/* the same update with lr, 1 - beta1^step, sqrt(1 - beta2^step) read from device memory (hyper[0..2]): for a training step captured
 * in a hipGraph, whose kernel arguments are frozen at capture (the LambdaLR value of pipeline/finetune.py:253-260 changes every step) */
int slam_adamw_hyper(float lr, float beta1, float beta2, int64_t step, float* out3_host);   /* the three words, as slam_adamw_step forms them */
int slam_adamw_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* param_bf16, int64_t n,
                        const float* hyper, float beta1, float beta2, float eps, float weight_decay, float grad_scale, void* stream);

/* AnyPrecisionAdamW (src/slam_llm/policies/anyprecision_optimizer.py:73-178, selected at pipeline/finetune.py:237-245): bf16
 * momentum / variance (and optional bf16 Kahan compensation: non-null), every tensor op of the reference rounded to its tensor's
 * dtype in the reference's order.  The scalars are the reference's float32 values (its `step` is a float32 tensor): decay =
 * 1 - lr*wd, denom_correction = sqrt(1 - beta2^t), neg_step_size = -lr / (1 - beta1^t).  params_are_bf16 = 1 emulates the
 * reference's pure_bf16 parameters inside the fp32 master buffer (values stay bf16-representable). */
int slam_adamw_anyprecision_step(float* param, const float* grad, void* exp_avg_bf16, void* exp_avg_sq_bf16,
                                 void* compensation_bf16, void* param_bf16, int64_t n, float decay, int use_decay,
                                 float beta1, float one_minus_beta1, float beta2, float one_minus_beta2,
                                 float denom_correction, float eps, float neg_step_size, int params_are_bf16, void* stream);
int slam_cast_f32_to_bf16(const float* in, void* out, int64_t n, void* stream);
/* counter-based dropout (peft lora_dropout): out (+)= keep(seed, offset + m*N + n) ? x/(1-p) : 0.
 * The mask is a pure function of (seed, offset, index): backward recomputes it (same call on the incoming gradient). */
int slam_dropout_bf16(const void* x, int64_t ldx, void* out, int64_t ldo, int64_t M, int64_t N, float p, uint64_t seed,
                      uint64_t offset, int accumulate, void* stream);
/* a += b (bf16 residual-gradient merge of the post-LN Q-Former blocks) */
int slam_add_bf16(void* a, int64_t lda, const void* b, int64_t ldb, int64_t M, int64_t N, void* stream);
int slam_cast_bf16_to_f32(const void* in, float* out, int64_t n, int accumulate, void* stream);

/* ---- decode (generate) --------------------------------------------------------------------------------------
 * Replaces, for `slam_model.generate` (src/slam_llm/models/slam_model.py:409-456), what HF runs per generated token:
 * LlamaDecoderLayer with a DynamicCache (transformers/models/llama/modeling_llama.py), nn.Linear on a few rows, and
 * the per-step `_reorder_cache` / `reorder_cache(beam_idx)` of beam search (transformers/generation/utils.py). */

/* y[M<=64, N+N2] = x[M,K] . [W (N rows, ldb); W2 (N2 rows, ldb2)]^T (+ residual), HBM-bound weight streaming (every weight
 * byte read once, K split over waves / workgroups, fixed-order reduction -> bit-reproducible).  W2 (nullable) stacks a
 * second row block under W -- the decode path appends the LoRA A matrices so that u = x A^T comes out of the same
 * launch.  swiglu != 0: W = [gate (N/2 rows); up (N/2 rows)] and the output is [M, N/2] = silu(gate) * up (HF LlamaMLP).
 * splits: 0 automatic, > 0 forced cross-workgroup split, < 0 forced in-workgroup split (M <= 16).
 * workspace: slam_gemm_skinny_workspace_bytes(M, N+N2, K, splits, swiglu) bytes (0 for the in-workgroup path). */
int64_t slam_gemm_skinny_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t splits, int swiglu);
int slam_gemm_skinny_bf16_nt(const void* A, int64_t lda, const void* B, int64_t ldb, const void* B2, int64_t ldb2,
                             int64_t N2, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, const void* residual,
                             int64_t ldr, int out_dtype, int swiglu, void* workspace, int64_t workspace_bytes,
                             int64_t splits, void* stream);

/* One decode step of self-attention for R = B*beams hypotheses, one launch per layer:
 *   new token: q|k|v = qkv[r, :(Hq+2Hkv)*D] (+ LoRA delta u . lora_b^T with u = qkv[r, (Hq+2Hkv)*D : +lora_r], peft's
 *   `base(x) + scale*B(A(x))`), RoPE (HF apply_rotary_pos_emb, position = positions[r]) on q and k, then
 *   k -> k_gen[r, j], v -> v_gen[r, j], ancestors[r, j] = r with j = generated tokens cached so far (*gen_count_dev if
 *   non-null -- graph-capturable -- else gen_count);
 *   attention of the new q over [prompt KV of item r / beams, slots prompt_start[item] .. Tp) | generated slots 0 .. j],
 *   slot i of hypothesis r being read from physical row ancestors[r, i].
 * k/v_prompt [B, Tp, Hkv*D], k/v_gen [R, G, Hkv*D] (G <= 1024), O [R, Hq*D] bf16.  Replaces HF's eager attention
 * over a cache that beam search re-orders with index_select every step. */
int slam_attn_decode(const void* qkv, int64_t ld, const void* lora_b, int64_t ldlb, int64_t lora_r,
                     const float* cos_table, const float* sin_table, const int32_t* positions, const void* k_prompt,
                     const void* v_prompt, const int32_t* prompt_start, void* k_gen, void* v_gen, int32_t* ancestors,
                     const int32_t* gen_count_dev, int64_t gen_count, void* O, int64_t ldo, int64_t R, int64_t beams,
                     int64_t Tp, int64_t G, int64_t Hq, int64_t Hkv, int64_t D, float scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SLAM_HIP_H */
