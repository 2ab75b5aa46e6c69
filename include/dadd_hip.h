/*
 * dadd_hip.h — C ABI of libdadd_hip.so: the MI355X (gfx950) kernels behind the DADD /
 * IP-Adapter DDIM sampler.
 *
 * The reference (umutdundar99/progressive-stable-diffusion) is pure Python with no FFI; its
 * hot path runs through PyTorch / diffusers operators.  Each entry point below replaces the
 * operator sequence cited next to it (paths relative to the reference repo).  A reference-side
 * binding would be a ctypes stub per function.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless marked host
 *   - activations: fp16, NHWC (== token-major [B, H*W, C]); latents / eps / frames: fp32 NCHW
 *   - weights: fp16, [Cout][ky][kx][Cin] (K contiguous); biases and norm affine: fp32
 *   - `stream` is a hipStream_t passed as void*; no entry point allocates, frees or
 *     synchronises, so all of them may be captured into a hipGraph
 *   - the library borrows memory, it never owns or frees caller buffers
 *   - return value: DADD_OK or a negative DADD_E* code; dadd_last_error() gives the text
 */
#ifndef DADD_HIP_H
#define DADD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DADD_OK 0
#define DADD_EINVAL (-1) /* shape / alignment contract violated (Python raises ValueError) */
#define DADD_EHIP (-2)   /* a HIP runtime call failed (Python raises RuntimeError) */
#define DADD_ESTATE (-3) /* capture / profiling state misuse */

/* epilogue flags of dadd_conv_igemm_f16 */
#define DADD_EPI_BIAS 1     /* + bias[n] */
#define DADD_EPI_ROWVEC 2   /* + rowvec[b][n]   (time-embedding projection of the sample) */
#define DADD_EPI_RESIDUAL 4 /* + residual[m][n] */
#define DADD_EPI_GEGLU 8    /* out[m][n/2] = hidden * gelu(gate); weight rows pre-interleaved */
#define DADD_EPI_LNFOLD 128 /* a LayerNorm over the K (= channel) axis of x folded into this linear: w carries gamma,
                              bias the composed (w beta + b), ln_c1[n] = sum_k w[n][k]; the kernel derives the row
                              mean / rstd from the A fragments it reads anyway: out = rstd (acc - mu c1) + bias */
#define DADD_EPI_LNSTAT 4096 /* the epilogue also writes the LayerNorm ROW partials of its OUTPUT (sum, sum of squares of the
                               rounded fp16 values over each block of tile_n/2 columns) into ln_stats_out [ln_parts][M][2],
                               ln_parts = N / (tile_n/2): the linear that consumes LayerNorm(out) takes them through
                               ln_stats_in and needs neither a LayerNorm launch nor statistics of its own */
#define DADD_PRE_GN 8192      /* GroupNorm (32 groups) of the INPUT applied on the way in: x is the un-normalised tensor,
                                gn_in_ws its chunk partials [B][gn_in_nchunk][32][2] (a producer's DADD_EPI_GNSTAT), gn_in_gamma /
                                gn_in_beta the affine; the 3x3 halo kernel normalises the halo in LDS (its loader waves, one
                                16-byte piece per lane and tap) — no GroupNorm launch, no normalised copy.  3x3 / stride 1 on
                                64-, 32- or 16-wide maps with 128x160 tiles, one source, Cin <= 1152 / 2048 / 2432 (what fits beside the
                                halo in LDS) */
#define DADD_PRE_GN_SILU 16384 /* ... followed by SiLU (ResnetBlock2D.norm1/2 + nonlinearity) */
#define DADD_EPI_GNSTAT 2048 /* the epilogue also writes the GroupNorm chunk partials of its OUTPUT (32 groups) into
                               gn_ws [B][gn_nchunk][32][2] (sum, sum of squares per row block of tile_m/2 rows): the
                               consuming dadd_groupnorm_f16 then skips its statistics pass (ws_chunks = gn_nchunk) */
#define DADD_EPI_GNAPPLY 32768 /* split-K launches on small maps (Ho*Wo * N/32 * 2 bytes <= 16 KiB per (sample, group) slab, finish
                                kernel, no GEGLU / GNSTAT): the finish kernel owns one (sample, group) per block, so it also writes
                                GroupNorm(out) (+ SiLU with DADD_EPI_GNAPPLY_SILU) into gn_out [M][N] with gn_out_gamma / _beta /
                                _eps - the next conv's input - instead of a GroupNorm launch over `out`.  `out` itself is still
                                written (residual / skip consumers).  Same arithmetic, in the same order, as the finish kernel
                                followed by dadd_groupnorm_f16's single-launch path: bit-identical results */
#define DADD_EPI_GNAPPLY_SILU 65536
#define DADD_EPI_QUICKGELU 256 /* x * sigmoid(1.702 x) after bias (CLIP MLP, transformers quick_gelu) */
#define DADD_EPI_GELU 512      /* exact-form GELU after bias (nn.GELU of the resampler / purifier MLPs) */
#define DADD_EPI_SIGMOID 1024  /* sigmoid after bias (FeaturePurifier gate) */
#define DADD_EPI_ACT_MASK (256 | 512 | 1024)
#define DADD_TUNE_SHALLOW 16 /* tuning: keep one K tile in flight instead of two (A/B measurements) */
#define DADD_TUNE_NODMA 32   /* tuning: register-staged kernel instead of the LDS-DMA ring kernel */
#define DADD_TUNE_PERSIST 64 /* tuning: LDS-DMA ring kept running over several output tiles per workgroup */

/* cross-attention modes of dadd_tri_xattn_f16 */
#define DADD_XATTN_SPLIT 0    /* triple pathway, independent softmaxes */
#define DADD_XATTN_BASELINE 1 /* one joint softmax over [AOE|image] */

const char* dadd_last_error(void);
int dadd_version(void);
/* one-time per-process setup on the current device (kernel attributes); call before any launch
 * and outside stream capture */
int dadd_init(void);
/* out[0]=CU count, out[1]=LDS bytes/CU, out[2]=clock kHz, out[3]=gcnArch as int (950) */
int dadd_device_info(int device, int64_t out[4]);

/* ---- implicit-GEMM convolution / linear on MFMA -------------------------------------------
 * out[m][n] = epi( sum_{tap,c} x[b, oy*stride+ky-pad, ox*stride+kx-pad, c] * w[n][tap][c] )
 * m = (b*Ho+oy)*Wo+ox.  taps = 1 (linear / conv1x1) or 9 (conv3x3).  `ups` = 1 reads x through
 * a nearest 2x upsample (the virtual input is 2Hi x 2Wi).  Channels [0,C1) come from x,
 * [C1,C1+C2) from x2 (skip-concat without materialising torch.cat).
 * Replaces: F.conv2d / nn.Linear inside diffusers ResnetBlock2D, Transformer2DModel,
 * Downsample2D, Upsample2D, FeedForward(GEGLU) and the to_q/to_k/to_v/to_out Linears that the
 * reference calls at src/models/unet/unet.py:140-144, src/models/vae/vae.py:88,112 and
 * src/models/attention_processor_routing_gates.py:123,133-137,161-162,183.
 * Contract: (C1+C2) % 64 == 0, C1 % 64 == 0, N % 8 == 0, 16-byte aligned pointers.
 * tile_n 64 / tile_m 64 select the small LDS-DMA tiles (short GEMMs of the 16x16 / 8x8 maps).
 * splitk > 1 needs `partial` (fp32, splitk*M*N); the slabs are combined by the last-arriving slice of
 * each tile when `counters` is given, else by a finish kernel launched by the same call. */
typedef struct {
  const void* x;
  const void* x2;
  const void* w;
  void* out;
  float* partial;
  const float* bias;
  const float* rowvec;
  const void* residual;
  int32_t B, Hi, Wi, C1, C2, Ho, Wo, N;
  int32_t taps, stride, ups, pad;
  int32_t ldo, ldr, ld_rowvec;
  int32_t splitk, flags, tile_n; /* tile_n: 64, 128 or 160 (0 = choose) */
  int32_t tile_m;                /* 64 or 128 (0 = 128): rows of the output tile */
  int32_t* counters;             /* split-K tickets: >= #output tiles ints, zero between launches; with
                                    them the slabs are combined inside the launch (NULL: finish kernel) */
  const float* ln_c1;            /* DADD_EPI_LNFOLD: N floats (see the flag); replaces nn.LayerNorm + nn.Linear of
                                    BasicTransformerBlock.norm1/2/3 -> attn1.to_q|k|v / attn2.to_q / ff.net.0.proj */
  float ln_eps;
  float* gn_ws;                  /* DADD_EPI_GNSTAT (see the flag) */
  int32_t gn_nchunk, gn_cg;      /*   chunks per sample = Ho*Wo / (tile_m/2); channels per group = N / 32 */
  float* ln_stats_out;           /* DADD_EPI_LNSTAT (see the flag): [ln_parts_out][M][2] floats */
  const float* ln_stats_in;      /* with DADD_EPI_LNFOLD: row partials [ln_parts_in][M][2] of x written by the GEMM that
                                    produced x (its DADD_EPI_LNSTAT); NULL: the kernel sums the rows itself */
  int32_t ln_parts_out, ln_parts_in;
  const float* gn_in_ws;         /* DADD_PRE_GN (see the flag) */
  const float* gn_in_gamma;
  const float* gn_in_beta;
  int32_t gn_in_nchunk;
  float gn_in_eps;
  const float* gn_in_ws2;        /* DADD_PRE_GN over the concatenation [x | x2]: the chunk partials of x2 (its own 32 groups over
                                    C2 channels); needs (C1 + C2) / 32 to be a multiple of C1 / 32 and of C2 / 32 and to divide C1 */
  int32_t gn_in_nchunk2;
  void* gn_out;                  /* DADD_EPI_GNAPPLY (see the flag): fp16 [M][N] */
  const float* gn_out_gamma;
  const float* gn_out_beta;
  float gn_out_eps;
} dadd_igemm_desc;
int dadd_conv_igemm_f16(const dadd_igemm_desc* d, void* stream);

/* ---- the two thin-channel convolutions at the UNet / VAE ends ------------------------------
 * conv_in : x fp16 NHWC with 8 stored channels (4 or 3 real) -> fp16 NHWC Cout.
 * conv_out: fp16 NHWC C -> fp32 NCHW Cout<=4; mode 1 also applies clamp(-1,1),(x+1)/2,clamp(0,1)
 * (= _latents_to_images tail, src/pipelines/inference/inference_pipeline_ip.py:484-486); mode 2 clamps to
 * [-30, 20] (the logvar half of the VAE encoder moments, diffusers DiagonalGaussianDistribution). */
int dadd_conv3x3_cin8_f16(const void* x, const void* w, const float* bias, void* out, int B, int H,
                          int W, int Cout, void* stream);
/* conv_in of the UNet in one launch: fp32 NCHW latents (C <= 4 channels, rounded to fp16 in registers exactly as
 * dadd_pack_nchw_f32_to_nhwc8_f16 with scale 1 does) -> fp16 NHWC Cout; w is the [Cout][9][8] layout of conv_cin8.
 * Replaces UNet2DConditionModel.conv_in behind OrdinalUNet.forward (src/models/unet/unet.py:140-144).
 * gn_ws != NULL (Cout == 320, H*W a multiple of 256, gn_nchunk == H*W/256): also writes the GroupNorm chunk partials
 * [B][gn_nchunk][32][2] (sum, sum of squares of the rounded outputs, 256 pixels per chunk) that dadd_groupnorm_f16
 * (ws_chunks), DADD_PRE_GN and dadd_tf_head_f16 consume - the first ResNet's norm1 needs no statistics pass. */
int dadd_conv_in_nchw_f16(const float* x_nchw, const void* w, const float* bias, void* out, int B, int C,
                          int H, int W, int Cout, float* gn_ws, int gn_nchunk, void* stream);
int dadd_conv3x3_cout4_f16(const void* x, const void* w, const float* bias, float* out_nchw, int B,
                           int H, int W, int C, int Cout, int mode, void* stream);
/* conv_out of the UNet fused with the deterministic DDIM update of the sampler (no CFG): eps = conv3x3(x) + bias is
 * never stored; latents (fp32 NCHW, B x Cout x H x W) are updated in place with coef = {sqrt(a_t), sqrt(1-a_t),
 * sqrt(a_prev) (< 0: last step, return x0), sqrt(1-a_prev)} read from device memory — the arithmetic of
 * dadd_ddim_update_f32, operation for operation (inference_pipeline_ip.py:434-456). */
int dadd_conv_out_ddim_f16(const void* x, const void* w, const float* bias, float* latents, const float* coef,
                           int B, int H, int W, int C, int Cout, void* stream);

/* fp32 NCHW (C real channels, C<=8) -> fp16 NHWC with 8 channels (zero padded), times `scale`;
 * optional CxC matrix + bias applied per pixel first (AutoencoderKL.post_quant_conv 1x1).
 * Replaces latents / latent_scale (inference_pipeline_ip.py:476) + layout change. */
int dadd_pack_nchw_f32_to_nhwc8_f16(const float* x, void* out, int B, int C, int H, int W,
                                    float scale, const float* mat, const float* vec, void* stream);

/* out = (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise) * scale over n fp32 elements:
 * AutoencoderKL.encode(x).latent_dist.sample() * latent_scale (src/models/vae/vae.py:71-88,
 * src/models/diffusion_module_ip.py:410-411) with the noise supplied by the caller. */
int dadd_gaussian_sample_f32(const float* mean, const float* logvar, const float* noise, float scale,
                             float* out, int64_t n, void* stream);

/* Decoded frames fp32 NCHW in [0,1] -> uint8 NHWC (v * 255 truncated = ``.mul(255).to(torch.uint8)`` of
 * _tensor_to_bmp / _save_sequence, src/pipelines/inference/inference_pipeline_ip_data_augment.py:136-140,
 * inference_pipeline_ip.py:489-510): 4x fewer bytes over PCIe to the writers and over xGMI in the frame gather. */
int dadd_frames_to_u8(const float* frames_nchw, void* out_nhwc_u8, int B, int H, int W, void* stream);

/* ---- training-step forward (loss evaluation; src/models/diffusion_module_ip.py:392-462) ----------------------
 * x_t = sqrt(ab[t_b]) x0 + sqrt(1 - ab[t_b]) noise (``_q_sample`` :299-303), per-sample int64 timesteps. */
int dadd_q_sample_f32(const float* x0, const float* noise, const int64_t* t, const float* alphas_cumprod,
                      float* out, int B, int64_t per_sample, void* stream);
/* out[b] = mean_i (pred[b][i] - target[b][i])^2  (F.mse_loss(reduction="none").mean(dim=(1,2,3)), :440-441). */
int dadd_mse_rows_f32(const float* pred, const float* target, float* out, int B, int64_t per_sample,
                      void* stream);

/* ---- normalisation -------------------------------------------------------------------------
 * GroupNorm over a (virtually concatenated) NHWC tensor, optional SiLU, writes the concatenated
 * normalised tensor.  `ws` = fp32 scratch of B*DADD_GN_MAX_CHUNKS*groups*2 floats; with ws_chunks > 0 it already
 * holds [B][ws_chunks][groups][2] partials written by the producing GEMM's epilogue (DADD_EPI_GNSTAT; x2 == NULL)
 * and the statistics pass is skipped; with ws_chunks > 128 (VAE maps) `ws` needs B*64*groups*2 more floats behind the
 * partials: a small launch folds them to 64 chunks per sample first.
 * Replaces nn.GroupNorm(+F.silu) in ResnetBlock2D / Transformer2DModel / VAE (+ torch.cat). */
#define DADD_GN_MAX_CHUNKS 256
int dadd_groupnorm_f16(const void* x1, int C1, const void* x2, int C2, const float* gamma,
                       const float* beta, void* out, float* ws, int B, int HW, int groups,
                       float eps, int silu, int ws_chunks, void* stream);
/* LayerNorm over the last dim of [M][C] fp16 (BasicTransformerBlock.norm1/2/3). */
int dadd_layernorm_f16(const void* x, const float* gamma, const float* beta, void* out, int M,
                       int C, float eps, void* stream);

/* ---- attention -----------------------------------------------------------------------------
 * Flash self-attention, softmax(QK^T/sqrt(d))V, never materialising the scores.
 * q/k/v rows: token (b,n) at ptr + (b*N+n)*ld + head*d.  d in {40,80,160,512}.
 * Replaces diffusers AttnProcessor2_0 / F.scaled_dot_product_attention on attn1
 * (src/models/attention_processor_routing_gates.py:286) and the VAE mid-block attention. */
int dadd_self_attn_f16(const void* q, const void* k, const void* v, void* out, int B, int N,
                       int heads, int d, int ld_qkv, int ld_out, void* stream);
/* The same kernel with separate query / key-value sequences (Nq x Nk) and strides: d in {40,64,80,96,160,512}.
 * Replaces the SDPA inside transformers' CLIPAttention (src/models/image_encoder.py:34-88, 257 tokens, 16 x 64) and
 * nn.MultiheadAttention(768, 8) of the Resampler (image_encoder.py:152-228, 16 latents x 257 tokens) and of the
 * FeaturePurifier (src/models/feature_purifier.py:49-53,83-87, 16 x 16 tokens). */
int dadd_attn_f16(const void* q, const void* k, const void* v, void* out, int B, int Nq, int Nk, int heads,
                  int d, int ld_q, int ld_kv, int ld_out, void* stream);

/* Fused DADD cross-attention.  q: [B][N][C]; kv: [B][T][ld_kv] holds the step-invariant
 * projections of the conditioning tokens: columns [0,C)=to_k, [C,2C)=to_v, [2C,3C)=to_k_dis,
 * [3C,4C)=to_v_dis (baseline: only the first 2C, T=32).
 * SPLIT (T=48): z = anat_gate*softmax(q K_anat^T/sqrt d) V_anat          tokens [16,32), to_k/to_v
 *                 + dis_gate *softmax(q K_dis^T /sqrt d) V_dis           tokens [0,16),  to_*_dis
 *                 + lambda   *softmax(q K_dlt^T /sqrt d) V_dlt (iff lambda != 0)  tokens [32,48)
 * three independent softmaxes; gates are read from device memory (gates[0]=anat, gates[1]=dis), and so is
 * lambda when `lambda_dev` is non-null (it then overrides `lambda`): the reference reads `delta_scale` per call
 * (routing_gates.py:160), so one captured graph must serve a whole lambda sweep; lambda == 0 skips the delta
 * pathway inside the kernel (its tokens are never read), as :160,177-178 does.
 * Replaces SplitInjectionAttentionProcessor.__call__ lines 142-181 and
 * OrdinalIPAttnProcessor2_0.__call__ lines 92-122 (src/models/attention_processor_*.py). */
int dadd_tri_xattn_f16(const void* q, const void* kv, void* out, const float* gates, float lambda,
                       const float* lambda_dev, int mode, int B, int N, int heads, int d, int T, int ld_kv,
                       void* stream);

/* The whole attn2 block of the routing-gates processor as one launch.  x: LayerNorm output [B*HW][C];
 * mcat [B][384][C]: row (h*3+p)*16+t = log2(e)/sqrt(d) * K_p[b,t,h,:] . W_q[h*d:(h+1)*d, :]   (h < 8 heads,
 * p = 0 anatomy (tokens 16..31, to_k), 1 disease (tokens 0..15, to_k_dis), 2 delta (tokens 32..47, to_k_dis));
 * vw [B][C][384]: column (h*3+p)*16+t = gate_p * W_o[:, h*d:(h+1)*d] . V_p[b,t,h,:]  (gate_2 = lambda);
 * out = softmax16(x mcat^T) vw^T + bias + residual, 24 independent 16-wide softmaxes per token.
 * H*W % 128 == 0, C % 320 == 0.  Replaces to_q + SplitInjectionAttentionProcessor.__call__:142-181 + to_out
 * (src/models/attention_processor_routing_gates.py:118-190) — exact algebra, other rounding points.
 * ln_stats_out (or NULL): LayerNorm row partials of `out`, [C / 80][B*HW][2] floats, as DADD_EPI_LNSTAT writes them
 * (the GEGLU projection behind norm3 then takes them through ln_stats_in).
 * ln_stats_in (or NULL): norm2 folded into the score GEMM — x is then the UN-normalised hidden state, ln_stats_in its
 * row partials [ln_parts_in][B*HW][2] from the producer (DADD_EPI_LNSTAT), mcat carries gamma (mcat[b][n][c] * gamma_c),
 * ln_c1[b][n] = sum_c of those fp16 values, ln_d[b][n] = sum_c mcat0[b][n][c] * beta_c:
 * S = rstd_m (x mcat^T - mu_m c1) + d (BasicTransformerBlock.norm2 -> attn2.to_q, same algebra as DADD_EPI_LNFOLD). */
int dadd_attn2_fused_f16(const void* x, const void* mcat, const void* vw, const float* bias,
                         const void* residual, void* out, float* ln_stats_out, const float* ln_stats_in,
                         int ln_parts_in, const float* ln_c1, const float* ln_d, float ln_eps, int B, int HW, int C,
                         void* stream);

/* The tail of one transformer block at a 320-channel site as ONE launch per 64-token row block (csrc/ffn_block.hip):
 *   out = proj_out( h4 ) + bp + xres,   h4 = ff.net.2( GEGLU( ff.net.0.proj( LayerNorm3(x) ) ) ) + b2 + x
 * Replaces norm3 / ff.net.0.proj (GEGLU) / ff.net.2 / proj_out of diffusers' BasicTransformerBlock + Transformer2DModel
 * (the UNet body behind /root/reference/src/models/unet/unet.py:140-144).
 * x, xres, out: [M][320] fp16 (M = whole samples of HW tokens, HW % 64 == 0);  stream: dadd_ffn_block_bytes() bytes of
 * pre-swizzled weight pieces in consumption order (engine.pack_ffn_stream);  b1: GEGLU bias in piece order (2560);
 * gn_ws (optional): GroupNorm chunk partials of `out`, [M/HW][gn_nchunk = HW/32][32][2] fp32 (sums of the rounded values). */
int dadd_ffn_block_f16(const void* x, const void* stream, const float* ln_g, const float* ln_b, float ln_eps,
                       const float* b1, const float* b2, const float* bp, const void* xres, void* out, float* gn_ws,
                       int gn_nchunk, int M, int HW, int C, void* stream_handle);
int dadd_ffn_block_bytes(void);

/* The head of one transformer block at a 320-channel site as ONE launch per 64-token row block (csrc/tf_head.hip):
 *   hs = proj_in( GroupNorm(x) ) + bp;   qkv = LayerNorm1(hs) [Wq | Wk | Wv]^T
 * Replaces norm / proj_in / norm1 / attn1.to_q,k,v of diffusers' Transformer2DModel + BasicTransformerBlock (the UNet
 * body behind /root/reference/src/models/unet/unet.py:140-144).  x, hs: [M][320] fp16, qkv: [M][960]
 * (q | k | v blocks of 320 columns); gn_ws: GroupNorm chunk partials of x written by its producer,
 * [M/HW][gn_nchunk][32][2] fp32; stream: dadd_tf_head_bytes() bytes of pre-swizzled weight pieces
 * (engine.pack_head_stream). */
int dadd_tf_head_f16(const void* x, const void* stream, const float* gn_ws, int gn_nchunk, const float* gn_g,
                     const float* gn_b, float gn_eps, const float* bp, const float* ln_g, const float* ln_b, float ln_eps,
                     void* hs, void* qkv, int M, int HW, int C, void* stream_handle);
int dadd_tf_head_bytes(void);
/* diagnostics: per-workgroup cycle stamps (16 x uint64 each) of the following dadd_tf_head_f16 launches; NULL = off */
int dadd_tf_head_debug(void* buf);

/* ---- sampler glue --------------------------------------------------------------------------
 * Sinusoidal timestep features (flip_sin_to_cos, shift 0): out[m][0:dim/2]=cos, [dim/2:]=sin. */
int dadd_timestep_features_f32(const int64_t* t, float* out, int M, int dim, void* stream);
/* out[m][n] = act_out( sum_k act_in(x[m][k]) * w[n][k] + bias[n] ), fp32 rows, fp16 weights,
 * M small (time-embedding MLP, per-resblock time_emb_proj; with w_f32 = 1 fp32 weights: the AOE projector
 * MLP, src/models/ordinal_embedder.py:107-127).  act: 0 none, 1 SiLU, 2 GELU (exact). */
int dadd_linear_rows_f32(const float* x, const void* w, const float* bias, float* out, int M, int K,
                         int N, int act_in, int act_out, int w_f32, void* stream);
/* Per-step prologue run from inside the captured graph: row = step[0];
 * cur_rows[b][:] = table[row][:] for b < B; cur_coef[0:4] = coef[row][0:4]; then step[0] += 1.
 * step: TWO int32 (row, arrival ticket of the kernel's blocks — zero between launches). */
int dadd_begin_step(const float* table, float* cur_rows, int B, int ncols, const float* coef,
                    float* cur_coef, int32_t* step, void* stream);
/* DDIM update (eta = 0), op-for-op as inference_pipeline_ip.py:430-456:
 * eps = eps_u + g*(eps_c-eps_u) when eps_u != NULL; x0 = clamp((x - c1*eps)/c0, -4, 4);
 * x = last ? x0 : c2*x0 + c3*eps   with coef = {sqrt(ab_t), sqrt(1-ab_t), sqrt(ab_prev),
 * sqrt(1-ab_prev)} and last encoded as c2 < 0.  `guidance_dev` (device float, optional) overrides `guidance`. */
int dadd_ddim_update_f32(float* x, const float* eps_c, const float* eps_u, float guidance,
                         const float* guidance_dev, const float* coef, int64_t n, void* stream);

/* ---- conditioning front-end (once per batch): the pieces around the GEMM / attention kernels ---------------
 * CLIP patch rows: pixel_values [B][3][H][W] fp32 -> [B][1 + (H/P)(W/P)][Kp] fp16, row 0 of a sample zero (class
 * token slot), patch elements in conv-weight order (c, ky, kx), zero padded to Kp (multiple of 64): the operand of
 * the patch-embedding GEMM (CLIPVisionEmbeddings.patch_embedding, src/models/image_encoder.py:34-42). */
int dadd_clip_patch_rows_f16(const float* pixels, void* out, int B, int H, int W, int patch, int Kp, void* stream);
/* AOE class interpolation (src/models/ordinal_embedder.py:129-182): out[b][:] = lerp over table = base + cumsum(deltas)
 * at label[b] clamped to [0, classes-1]; fp32. */
int dadd_aoe_interp_f32(const float* labels, const float* base, const float* deltas, float* out, int B, int D,
                        int classes, void* stream);
/* FeaturePurifier tail (src/models/feature_purifier.py:88-95): out = LayerNorm(img - gate * dis); fp16 rows in,
 * fp32 rows out; gate already passed through the sigmoid. */
int dadd_purifier_tail_f16(const void* img, const void* dis, const void* gate, const float* gamma,
                           const float* beta, float* out, int M, int C, float eps, void* stream);

/* Weight prefetch on a side branch: dadd_prefetch reads `bytes` at `ptr` on an internal stream that forks from `stream`
 * at the call (inside a stream capture: a parallel branch of the graph, beside the kernels that precede the consumer), so
 * that a later kernel finds the weights in the Infinity Cache instead of in HBM; dadd_prefetch_join makes `stream` wait for
 * the branch - call it before dadd_graph_end and before anything that must not overlap.  No reference counterpart: the
 * reference leaves weight residency to the cache hierarchy (nn.Module parameters behind OrdinalUNet.forward,
 * src/models/unet/unet.py:140-144). */
int dadd_prefetch(const void* ptr, int64_t bytes, void* stream);
int dadd_prefetch_join(void* stream);

/* ---- hipGraph capture of the step loop ----------------------------------------------------- */
int dadd_graph_begin(void* stream);
int dadd_graph_end(void* stream, void** graph_exec_out);
int dadd_graph_launch(void* graph_exec, void* stream);
int dadd_graph_destroy(void* graph_exec);

/* ---- per-kernel timing of eager launches (bench.py roofline, scripts/step_profile.py) -----------
 * Between dadd_prof_begin() and dadd_prof_end() every kernel the library launches (eager launches only,
 * never during stream capture) is issued with a start/stop event pair attached to its dispatch packet
 * (hipExtLaunchKernelGGL): the pair reads the kernel's own begin/end timestamps, the clock rocprofv3's
 * kernel trace reads, so the time excludes queue gaps.  dadd_prof_end synchronises and returns the number of
 * recorded launches; dadd_prof_record(i) gives launch i in issue order: *name = kernel name as rocprofv3
 * prints it (static string), out[0] = milliseconds, out[1] = algorithmic flop (2*M*N*K for the matrix
 * kernels, 0 otherwise), out[2] = algorithmic bytes (compulsory HBM reads + writes). */
int dadd_prof_begin(void);
int dadd_prof_end(int* n_records);
int dadd_prof_record(int i, const char** name, double out[3]);

#ifdef __cplusplus
}
#endif
#endif /* DADD_HIP_H */
