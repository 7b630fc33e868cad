/*
 * include/hqq_hip.h — C ABI of libhqq_hip.so, the MI355X (gfx950) implementation of HQQ's two hot paths.
 *
 * Boundary rules:
 *   - plain pointers and sizes only; no torch / ATen types cross this ABI.
 *   - the caller owns every buffer (inputs, outputs, workspace); the library never allocates or
 *     frees device memory, keeps no pointer after a call returns and holds no per-process mode:
 *     whatever selects a kernel variant is a per-call `opts` bit (HQQ_OPT_*); no environment
 *     variable is read.
 *   - every function enqueues on the given hipStream_t (passed as void*; NULL = legacy default
 *     stream) and returns immediately; there is no host synchronisation inside.
 *   - return value: 0 on success; >0 a hipError_t from the launch; <0 an argument error
 *     (HQQ_ERR_*).  hqq_hip_last_error() gives a thread-local message.  Nothing throws.
 *   - device pointers must be 16-byte aligned and dense (contiguous).
 *
 * Each entry point cites the reference interface it replaces (mobiusml/hqq v0.2.8.post1):
 * the pybind module `hqq_aten` (hqq/kernels/hqq_aten_cuda.cpp:57-73) and the PyTorch code of
 * hqq/core/{bitpack,quantize,optimize}.py that HQQBackend.PYTORCH runs.
 */
#ifndef HQQ_HIP_H
#define HQQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HQQ_HIP_ABI_VERSION 8

/* element types of activations / meta / outputs ("compute_dtype" in the reference) */
enum { HQQ_F32 = 0, HQQ_F16 = 1, HQQ_BF16 = 2, HQQ_U8 = 3 };

/* argument errors */
enum {
  HQQ_ERR_NBITS = -1,      /* nbits not in {8,4,3,2,1}                                   */
  HQQ_ERR_SHAPE = -2,      /* sizes inconsistent with the packing / group size            */
  HQQ_ERR_DTYPE = -3,      /* dtype code not supported by this entry point                */
  HQQ_ERR_UNSUPPORTED = -4,/* valid HQQ configuration this kernel does not cover (caller decides what to do) */
  HQQ_ERR_WORKSPACE = -5,  /* workspace too small / NULL                                  */
  HQQ_ERR_ALIGN = -6       /* pointer not 16-byte aligned                                 */
};

int hqq_hip_abi_version(void);
const char* hqq_hip_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * BitPack  — hqq/core/bitpack.py:14-144 ; hqq_aten.unpack_{8,4,3,2,1}bit_* (hqq_aten_cuda.cpp:57-73,
 * kernels hqq_aten_cuda_kernel.cu:81-106,159-188,244-276,337-374).
 * Layout: `per` row-slabs of the unpacked [rows, cols] matrix share one packed element, slab 0
 * most significant: per = 1/2/4/8 for 8/4/2/1-bit into uint8, per = 10 for 3-bit into int32 with
 * rows zero padded to 10*ceil(rows/10).
 * ------------------------------------------------------------------------------------------- */
/* rows of the packed tensor for `rows` unpacked rows; HQQ_ERR_SHAPE if rows % per != 0 (torch raises there) */
int64_t hqq_hip_packed_rows(int nbits, int64_t rows);

/* U [rows, cols] uint8 (in_dtype HQQ_U8) or float32 holding integers (HQQ_F32, the solver's W_q)
 * -> out [packed_rows, cols] uint8 / int32(3-bit).  BitPack.pack_* */
int hqq_hip_pack(int nbits, const void* U, int in_dtype, int64_t rows, int64_t cols, void* out, void* stream);

/* packed [packed_rows, cols] -> out [per*packed_rows, cols] of out_dtype (HQQ_U8/F16/BF16/F32).
 * BitPack.unpack_*(W_q, dtype) ; hqq_aten.unpack_* */
int hqq_hip_unpack(int nbits, const void* packed, int64_t packed_rows, int64_t cols, void* out, int out_dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Quantizer.dequantize — hqq/core/quantize.py:183-199 ; hqq_aten.dequantize(W_q, scale, zero, N, K,
 * group_size, nbits, axis, packing) (hqq_aten_cuda.cpp:32-54, axis=0 only there; both axes here).
 *   out[N,K] = ((unpack(Wq)[:N*K/gs] - zero) * scale).reshape(N,K), two roundings in `dtype`.
 * scale/zero: [N*K/group_size] elements of `dtype`.  group_size = elements per (scale,zero).
 * axis=1: unpacked matrix is [N*K/gs, gs];  axis=0: [gs, N*K/gs].
 * ------------------------------------------------------------------------------------------- */
int hqq_hip_dequantize(int nbits, const void* Wq, const void* scale, const void* zero, void* out,
                       int64_t N, int64_t K, int64_t group_size, int axis, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * HQQLinear.forward (quantize.py:880-898 forward_pytorch / matmul; patching.py:82-86) fused:
 *   y[M,N] = x[M,K] @ dequantize(Wq)^T (+ bias[N]),  axis=1 layout, dequantised weights bit-identical
 *   to hqq_hip_dequantize, fp32 accumulation, one rounding to `dtype` (+ one for the bias add).
 * hqq_hip_gemv : small M (decode), HBM-bandwidth bound (MFMA only as a free dot-product unit).  1 <= M <= HQQ_GEMV_MAX_M
 * hqq_hip_gemm : large M (prefill), MFMA f16/bf16.                          any M >= 1
 * hqq_hip_forward picks one of the two by M.
 * Covered by hqq_hip_gemv: nbits in {8,4,2,1} with N % (8/nbits) == 0, group_size % 16 == 0, K % group_size == 0, fp16,
 * M <= 16 (bf16: nbits 4/2, M <= 4; up to HQQ_GEMV_MAX_M_SKINNY = 64 rows for fp16 and bf16, nbits 8/4/2, group_size 64,
 * K % 256 == 0, K >= 512: the weight-streaming skinny-GEMM kernel); nbits 3 with group_size 64, fp16, M <= 4.
 * Workspace: launches of 5..64 rows that split K, and 3-bit launches of >= 19 MB of packed weights, park fp32 partial sums (and
 * arrival counters) in a caller-owned workspace of hqq_hip_gemv_workspace_bytes(...) bytes (0 = this call needs none; then
 * workspace may be NULL).  Contract: 16-byte aligned device memory, ZERO when first used — the caller clears it once when
 * allocating it; every call leaves the counter area zero again — and not shared by calls that may run concurrently.  A larger
 * workspace than asked for is fine: one buffer sized for the largest launch serves a whole model.
 * hqq_hip_gemm: fp16 / bf16, nbits in {8,4,2} with group_size 64, K % 128 == 0 (the pipelined kernel: all operands by LDS-DMA, K split
 * across workgroups until the chip is full, fp32 partial tiles parked in the workspace and summed in split order by a second launch;
 * any M); fp16, nbits in {4,2}, K % 64 == 0, group_size % 16 == 0 (the output-tile kernels: the other group sizes, any M).
 * Anything else -> HQQ_ERR_UNSUPPORTED (the caller may compose hqq_hip_dequantize + its own GEMM).
 * ------------------------------------------------------------------------------------------- */
#define HQQ_GEMV_MAX_M 16
#define HQQ_GEMV_MAX_M_SKINNY 64   /* fp16 / bf16, 8-/4-/2-bit, group_size 64, K % 256 == 0, K >= 512: the skinny-GEMM kernel */
#define HQQ_GEMV_MAX_GROUP 4
/* per-call options */
#define HQQ_OPT_FACTORED       1u   /* decode arithmetic: the group affine map is factored out of the dot product and applied in fp32
                                       (no per-weight fp16 rounding: NOT the reference's weights; results differ from it by less than its
                                       own weight-rounding noise).  M <= 8.  Default (bit clear): every weight is rebuilt as
                                       round16(round16(q - z) * s), bit-identical to hqq_hip_dequantize / Quantizer.dequantize. */
#define HQQ_OPT_META_SCALABLE  2u   /* the caller asserts hqq_hip_meta_check() returned 0 failing groups for EVERY layer of the call: the
                                       exact rebuild may then use its three-op form (same bits, ~25 % less VALU work).  fp16. */
#define HQQ_OPT_GEMV3_ROWWISE  4u   /* 3-bit decode: force the row-per-wave kernel (tests / tuning) */
#define HQQ_OPT_GEMV3_SLABS    8u   /* 3-bit decode: force the slab-sharing kernel (needs workspace) */
#define HQQ_OPT_GEMM_REGTILE  16u   /* prefill: the register-tile variant of the fused GEMM */
#define HQQ_OPT_GEMM_CLASSIC  32u   /* fused GEMM: the plain output-tile kernels for every M (tests / tuning; default: the pipelined split-K
                                       kernel up to 1024 rows) */
#define HQQ_OPT_SKINNY_KS(n) ((uint32_t)(n) << 24)   /* 5..64 rows, and the split-K fused GEMM: force n K-splits (tuning; 0 = built-in rule) */
#define HQQ_OPT_GEMM_NARROW  64u   /* pipelined fused GEMM: force 4 waves per workgroup (64 packed rows per tile) — tuning */
#define HQQ_OPT_GEMM_WIDE   128u   /* pipelined fused GEMM: force 8 waves per workgroup (128 packed rows per tile) — tuning */
#define HQQ_OPT_GEMM_NOHYBRID 256u  /* pipelined fused GEMM: never split only the last round of tiles (tuning) */
#define HQQ_OPT_SKINNY_WIDE 512u  /* 5..64 rows: force the 64-packed-row tile for launches the 32-row tile would serve (tests / tuning) */
#define HQQ_OPT_W3S        1024u  /* nbits = 3: Wq is the 3-bit STREAM layout written by hqq_hip_w3s_pack (below), not the reference container */
#define HQQ_OPT_ALL (2047u | (255u << 24))
/* Which groups of a layer can NOT take the three-op exact rebuild: (zero, scale) pairs for which zero * 2^-J is inexact in fp16,
 * |zero| > 2^15 or scale * 2^J overflows (J = 9 - bit offset of the row's slab).  Writes the count to *fail_count (device memory,
 * uint32; the call clears it first).  Run once per layer when it is prepared; pass HQQ_OPT_META_SCALABLE only if it came out 0.
 * scale / zero as in hqq_hip_dequantize, axis = 1, fp16.  (No reference counterpart: the reference always does two torch ops.) */
int hqq_hip_meta_check(int nbits, const void* scale, const void* zero, int64_t N, int64_t K, int64_t group_size, int dtype,
                       uint32_t* fail_count, void* stream);
/* ---------------------------------------------------------------------------------------------
 * The 3-bit STREAM layout (ABI 5; csrc/w3s.h).  BitPack.pack_3bit_32 (hqq/core/bitpack.py:69-91) ORs ten row slabs of the level
 * matrix into one int32 with a slab height that is not a multiple of a row's groups: a word mixes ten unrelated output rows.  Like
 * every optimised backend of the reference (hqq/backends/torchao.py:202-241, marlin.py:74-123: re-layout when a layer is patched),
 * HQQLinearHIP converts the container ONCE into [N/2, K/16, 3] uint32 — packed row p = output rows p and p + N/2, 12 bytes = 16 k of
 * both, exactly 3 bits per level — which the decode / GEMM kernels stream like a 4-bit layer (pass HQQ_OPT_W3S with nbits = 3).
 * scale / zero are unchanged.  hqq_hip_w3s_unpack restores the reference's container bit for bit (state_dict(), dequantize()).
 * Needs group_size 64 layers: N % 2 == 0, K % 64 == 0.  Wq_ref: [ceil(N K / 640), 64] int32; w3s: N K 3 / 8 bytes.
 * hqq_hip_w3s_meta_check: hqq_hip_meta_check for a layer in this layout (every group: zero 2^-9 exact, scale 2^9 finite), fp16.
 * ------------------------------------------------------------------------------------------- */
int hqq_hip_w3s_pack(const void* Wq_ref, void* w3s_out, int64_t N, int64_t K, void* stream);
int hqq_hip_w3s_unpack(const void* w3s, void* Wq_ref_out, int64_t N, int64_t K, void* stream);
int hqq_hip_w3s_meta_check(const void* scale, const void* zero, int64_t N, int64_t K, uint32_t* fail_count, void* stream);
size_t hqq_hip_gemv_workspace_bytes(int nbits, int n_layers, const int64_t* N, int64_t M, int64_t K, int64_t group_size, int dtype, uint32_t opts);
int hqq_hip_gemv(int nbits, const void* x, const void* Wq, const void* scale, const void* zero, const void* bias,
                 void* y, int64_t M, int64_t N, int64_t K, int64_t group_size, int dtype, uint32_t opts, void* workspace, size_t workspace_bytes,
                 void* stream);
/* Horizontal fusion of up to HQQ_GEMV_MAX_GROUP layers that consume the SAME activation rows x[M,K] (q/k/v, gate/up,
 * experts of one token ...): one launch streams all their packed rows; layer i writes y[i][M, N[i]].  Every per-layer
 * argument is a host array of n_layers device pointers / sizes (read during the call, not kept); bias may be NULL or
 * hold NULL entries.  All layers share K, group_size, nbits and dtype.  hqq_hip_gemv is the n_layers = 1 case.
 * (The reference has no counterpart: HQQLinear.forward is per layer, quantize.py:880-898.) */
int hqq_hip_gemv_grouped(int nbits, int n_layers, const void* x, const void* const* Wq, const void* const* scale,
                         const void* const* zero, const void* const* bias, void* const* y, const int64_t* N,
                         int64_t M, int64_t K, int64_t group_size, int dtype, uint32_t opts, void* workspace, size_t workspace_bytes,
                         void* stream);
/* ---------------------------------------------------------------------------------------------
 * One exchange point of a column-sharded decode step (ABI 4; csrc/exchange.hip), one activation row, without a collective library.
 * The reference has no multi-GPU path for HQQLinear.forward (quantize.py:880-898); the shard is SURVEY.md section 8e's: rank r holds the
 * packed-row block r of every layer and computes, per slab s, the output columns s N/per + [r n', (r + 1) n'), n' = N / (per P).
 * This call stores the rank's slices y_loc[j] ([1, N_loc[j]], local slab-major order) straight into EVERY rank's full row of layer j
 * at those columns — peers' rows are device pointers the caller obtained with hipIpcOpenMemHandle (stores travel over xGMI) — writes
 * the exchange's generation (how often this point has been used: counted on the device, so it counts under graph replay too) into this rank's
 * flag word of every rank's flag block and waits until all `world` flag words of its own block have reached it.  Flags only grow: one that
 * arrives after a wait gave up cannot satisfy the next use of the point.  When the
 * kernel has finished, this rank's full rows are complete and in the reference's column order; the next kernel in stream order may
 * read them.  Capturable.  One launch, `world` workgroups.
 *   full    [world * n_layers] pointers: full[p * n_layers + j] = rank p's [1, N_loc[j] * world] row of layer j (p == rank: local)
 *   flags   [world] pointers: rank p's flag block of THIS point: HQQ_EXCHANGE_MAX_RANKS + 1 uint32 words (a flag word per rank, then the
 *           rank's own launch-ticket word), all zero before the first use and only reset collectively (every rank, between two barriers);
 *           fine-grained / uncached device memory is the right kind for them and for the rows (peers write while a local kernel polls)
 *   status  one local uint32: a wait that gives up after spin_limit polls (0 = 4 Mi, seconds) writes 1 + the missing rank there
 *           and returns — outputs of that exchange undefined, reported, never a hang; sticky until the caller clears it
 * Re-use rule: consecutive exchanges on a stream must alternate between at least two points (flag block + rows); a decoder block
 * has four.  Every rank must issue the same sequence of points.  dtype F16 / BF16; nbits picks `per` (3-bit shards: per = 1).
 * ------------------------------------------------------------------------------------------- */
#define HQQ_EXCHANGE_MAX_RANKS 16
#define HQQ_EXCHANGE_MAX_ROWS 64   /* ABI 6: M activation rows per exchange (a decode batch): y_loc[j] is [M, N_loc[j]], the full row sets [M, N_loc[j] * world]; row m of a
                                      rank's slab run lands at the same columns of the peers' row m (strided slab writes), so a batch needs no un-permute either */
int hqq_hip_exchange(int n_layers, const void* const* y_loc, const int64_t* N_loc, int64_t M, int nbits, int dtype, int world, int rank,
                     void* const* full, void* const* flags, void* status, uint32_t spin_limit, void* stream);
/* ---------------------------------------------------------------------------------------------
 * The steps either side of the GEMVs in a decode step (ABI 5; csrc/block.hip).  The reference's headline is the
 * tok/s of its generate loop (hqq/utils/generation_hf.py:117-540, Readme.md:153), whose decoder block around HQQLinear.forward is HF's
 * eager code: ~25 small kernels per block.  These three restate the HF modules' arithmetic rounding for rounding (transformers
 * models/llama/modeling_llama.py: LlamaRMSNorm.forward, apply_rotary_pos_emb, LlamaMLP.forward; cache_utils.StaticLayer.update), fp16 or bf16 (T below; bf16 ops = float arithmetic + one rounding per op, as torch evaluates them):
 *   hqq_hip_add_rmsnorm  h[rows, H] += delta (if delta != NULL: the residual add, one rounding), then xn = weight * T(float(h) * rsqrt(mean(h^2) + eps))
 *   hqq_hip_rope_cache   q_out[n_heads, hd] = (q * cos) + (rotate_half(q) * sin); the same for k, written with v into the caches
 *                        [n_kv_heads, cache_len, hd] at position *pos_dev (device memory: the call is graph-replay safe)
 *   hqq_hip_silu_mul     out[n] = T(silu(gate)) * up
 * ------------------------------------------------------------------------------------------- */
int hqq_hip_add_rmsnorm(void* h, const void* delta, const void* weight, float eps, void* xn_out, int64_t rows, int64_t H, int dtype, void* stream);
int hqq_hip_rope_cache(const void* q, const void* k, const void* v, const void* cos, const void* sin, const int64_t* pos_dev, void* q_out, void* k_cache,
                       void* v_cache, int64_t n_heads, int64_t n_kv_heads, int64_t head_dim, int64_t cache_len, int dtype, void* stream);
int hqq_hip_silu_mul(const void* gate, const void* up, void* out, int64_t n, int dtype, void* stream);
/* The per-token work either side of the decoder blocks (ABI 7; hqq/utils/generation_hf.py:405-540: embedding lookup, the rotary table's row, the causal mask of one query in
 * front; argmax, token hand-over, position increment behind) as ONE launch each — copies and compares only, bit-identical to the torch ops they replace:
 *   hqq_hip_token_prologue  h[H] = embed[*tok_dev]; cos / sin [head_dim] = cos_tab / sin_tab [L, head_dim] row *pos_dev (tables NULL: skipped);
 *                           mask[L] = i <= *pos_dev ? 0 : -inf (mask NULL: skipped).  A token / position outside the tables is CLAMPED to the last row (torch's index ops would raise; the caller validates).
 *   hqq_hip_argmax_advance  *next_tok_dev = the FIRST index of the largest of logits[n], a NaN counting as the largest (torch.argmax's order); *tok_dev = the same (NULL: skipped); *pos_dev += 1 (NULL: skipped) */
int hqq_hip_token_prologue(const int64_t* tok_dev, const int64_t* pos_dev, const void* embed, int64_t vocab, int64_t H, const void* cos_tab, const void* sin_tab, int64_t L,
                           int64_t head_dim, void* h, void* cos, void* sin, void* mask, int dtype, void* stream);
int hqq_hip_argmax_advance(const void* logits, int64_t n, int dtype, int64_t* next_tok_dev, int64_t* tok_dev, int64_t* pos_dev, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The decoder block's launches with those steps FOLDED IN (ABI 6; csrc/gemv_block.hip): hqq_hip_gemv_grouped for ONE activation row whose
 * prologue / epilogue does what hqq_hip_add_rmsnorm / hqq_hip_silu_mul / the residual add did as launches of their own — LlamaDecoderLayer.forward
 * (transformers models/llama/modeling_llama.py) around HQQLinear.forward (hqq/core/quantize.py:880-898) becomes 4 launches + rotary / attention:
 *   HQQ_BLOCK_NORM                   x is the RESIDUAL STREAM h [1, K]; every workgroup computes norm_weight * T(float(h) * rsqrt(mean(h^2) + eps))
 *                                    itself (LlamaRMSNorm, hqq_hip_add_rmsnorm's roundings; fp32 sum of squares in a fixed order) and contracts the
 *                                    layers with that; y[i] [1, N[i]] as hqq_hip_gemv_grouped writes them (q|k|v)
 *   HQQ_BLOCK_NORM | HQQ_BLOCK_SILU  one layer in the PAIRED layout — the level matrix of gate on top of the level matrix of up, packed as ONE layer of
 *                                    N = 2 x intermediate rows (hqq_amd.ops.pair_layers), so that a packed row holds gate row n and up row n —:
 *                                    y[0] [1, N / 2] = T(silu(T(gate[n]))) * T(up[n]) (LlamaMLP.forward; hqq_hip_silu_mul's roundings); gate / up are not stored
 *   HQQ_BLOCK_RESID                  one layer, x its input as usual; y[0] is the residual stream: h[n] = h[n] + T(result[n]) (o_proj, down_proj)
 * fp16 / bf16; nbits 4 / 2, or 3 with HQQ_OPT_W3S; group_size 64; exact weights (HQQ_OPT_META_SCALABLE honoured); K <= 8192 for the NORM forms.
 * No bias (the Llama linears have none), no workspace.  Same streaming loop, launch geometry and weights as hqq_hip_gemv.
 * ------------------------------------------------------------------------------------------- */
#define HQQ_BLOCK_NORM  1u
#define HQQ_BLOCK_RESID 2u
#define HQQ_BLOCK_SILU  4u
#define HQQ_BLOCK_ROPE  8u
/*   HQQ_BLOCK_NORM | HQQ_BLOCK_ROPE  the q | k | v group (n_layers = 3) with hqq_hip_rope_cache in its epilogue.  q and k in the ROTARY-PAIRED row order
 *                                    (hqq_amd.ops.rotary_pair_layout: element i < head_dim / 2 of head h is row h head_dim / 2 + i, its partner i + head_dim / 2
 *                                    row N / 2 + h head_dim / 2 + i — BitPack's row slabs then hold both in one packed row), v as it is.  y[0] = q_out
 *                                    [n_heads, head_dim] (rotated, natural order), y[1] / y[2] = the key / value caches [n_kv_heads, cache_len, head_dim]: the
 *                                    rotated key and the value are written at position *rope->pos (device memory: graph-replay safe; outside the cache: nothing).
 *                                    apply_rotary_pos_emb / StaticLayer.update, rounding for rounding as hqq_hip_rope_cache.  rope: NULL without the flag. */
typedef struct {
  const void* cos;       /* [head_dim] of the position, the compute dtype */
  const void* sin;
  const int64_t* pos;    /* device memory */
  int64_t head_dim, cache_len;
} hqq_rope_t;
int hqq_hip_gemv_block(int nbits, int n_layers, const void* x, const void* norm_weight, float eps, const void* const* Wq, const void* const* scale,
                       const void* const* zero, void* const* y, const int64_t* N, int64_t K, int64_t group_size, int dtype, uint32_t opts, uint32_t flags,
                       const hqq_rope_t* rope, void* stream);

/* Decode attention for ONE query per head over a static KV cache (opt-in: hqq_amd.utils.llama_fused.FusedLlamaStep(attention="hip")).
 * Replaces, in the reference's generate loop (hqq/utils/generation_hf.py:117-540), HF's call of F.scaled_dot_product_attention for a decode step:
 *   out[h, :] = softmax_j(q[h, :] . k_cache[h / (n_heads / n_kv_heads), j, :] * scaling) . v_cache[..., j, :]   over j = 0 .. pos_dev[0]
 * fp32 scores, softmax and accumulation, one rounding of the output: WITHIN ROUNDING of SDPA's result, not bit-identical to it (its flash
 * kernel blocks the keys and rounds the probabilities to the tensors' dtype) — which is why the default decode step keeps HF's attention function.
 * q [n_heads, head_dim] (rotary already applied), k_cache / v_cache [n_kv_heads, cache_len, head_dim], out [n_heads, head_dim]; fp16 / bf16;
 * head_dim 64 / 128 / 256; cache_len <= 30000; pos_dev: the query's position in device memory (graph-replay safe). */
int hqq_hip_attn_decode(const void* q, const void* k_cache, const void* v_cache, const int64_t* pos_dev, void* out, int64_t n_heads, int64_t n_kv_heads,
                        int64_t head_dim, int64_t cache_len, float scaling, int dtype, int64_t splits, void* workspace, size_t workspace_bytes, void* stream);
/* splits: 1 = one workgroup per query head (caches of up to ~1000 positions: 3.5-5.5 us); > 1 (at most 64): every head's visible keys are shared
 * out over `splits` workgroups whose (max, sum, output) records a second small launch merges in split order (long caches: 48 -> 19 us at 4096
 * positions with 8) — `workspace` then holds hqq_hip_attn_decode_workspace_bytes(n_heads, head_dim, splits) bytes (caller-owned, no initialisation). */
size_t hqq_hip_attn_decode_workspace_bytes(int64_t n_heads, int64_t head_dim, int64_t splits);
/* The same with hqq_hip_rope_cache folded in: q / k / v are the RAW projections ([n_heads, head_dim], [n_kv_heads, head_dim] twice), cos / sin
 * [head_dim]; every workgroup rotates its query and its KV head's new key (hqq_hip_rope_cache's arithmetic, rounding for rounding), takes the new
 * key / value from on-chip memory for position pos and reads the cache only below it; the new key / value are written to the cache (by one workgroup
 * per KV head) for the following steps: the cache ends up bit-identical to what hqq_hip_rope_cache writes. */
int hqq_hip_rope_attn_decode(const void* q, const void* k, const void* v, const void* cos, const void* sin, const int64_t* pos_dev, void* k_cache, void* v_cache,
                             void* out, int64_t n_heads, int64_t n_kv_heads, int64_t head_dim, int64_t cache_len, float scaling, int dtype, int64_t splits,
                             void* workspace, size_t workspace_bytes, void* stream);

/* workspace of hqq_hip_forward / hqq_hip_gemm for one layer at M rows (0 = none needed, workspace may be NULL): the decode kernels'
 * (hqq_hip_gemv_workspace_bytes) up to HQQ_GEMV_MAX_M_SKINNY rows, the split-K fused GEMM's fp32 partial tiles beyond.  Same contract. */
size_t hqq_hip_forward_workspace_bytes(int nbits, int64_t M, int64_t N, int64_t K, int64_t group_size, int dtype, uint32_t opts);
/* introspection (host arithmetic only): how the pipelined fused GEMM would run this shape — out8 = {waves per workgroup, tokens per tile,
 * feature tiles, token tiles, K splits, steps per split, tiles that run unsplit before the split ones, workgroups}; HQQ_ERR_UNSUPPORTED
 * when another kernel serves the shape */
int hqq_hip_gemm_plan(int nbits, int64_t M, int64_t N, int64_t K, int64_t group_size, int dtype, uint32_t opts, int* out8);
size_t hqq_hip_gemm_workspace_bytes(int nbits, int64_t M, int64_t N, int64_t K, int64_t group_size, int dtype, uint32_t opts);   /* hqq_hip_gemm called directly, any M */
/* 1 when, for this shape, the fused kernels behind hqq_hip_forward are measured faster on MI355X than hqq_hip_dequantize + a library
 * GEMM on the result (the caller's alternative for M > HQQ_GEMV_MAX_M_SKINNY), else 0: a speed hint, never a correctness matter. */
int hqq_hip_forward_prefers_fused(int nbits, int64_t M, int64_t N, int64_t K, int64_t group_size, int dtype);
int hqq_hip_gemm(int nbits, const void* x, const void* Wq, const void* scale, const void* zero, const void* bias,
                 void* y, int64_t M, int64_t N, int64_t K, int64_t group_size, int dtype, uint32_t opts, void* workspace, size_t workspace_bytes,
                 void* stream);
/* hqq_hip_gemm for a GROUP of 1..HQQ_GEMV_MAX_GROUP layers that read the same x[M, K] (q | k | v, gate | up of a decoder block — hqq/utils/patching.py:82-86 calls them
 * one by one): ONE launch of the pipelined fused GEMM over the layers' concatenated feature tiles (+ one split-K reduce), y[i][M, N[i]] per layer (ABI 8).
 * Every layer must be one hqq_hip_gemm serves on the pipelined kernel (hqq_hip_gemm_grouped_covers = 1: fp16 / bf16, nbits 8 / 4 / 2 or the 3-bit stream
 * layout with HQQ_OPT_W3S, group_size 64, K % 128 == 0, (N / per) % 4 == 0); opts (incl. HQQ_OPT_META_SCALABLE) apply to the whole group.  The K split is chosen
 * for the group's total width, so a row may differ in the last bit from the same layer launched alone (another association of the same fp32 sums).
 * Workspace: hqq_hip_gemm_grouped_workspace_bytes (same contract as hqq_hip_forward's). */
int hqq_hip_gemm_grouped_covers(int nbits, int n_layers, const int64_t* N, int64_t M, int64_t K, int64_t group_size, int dtype, uint32_t opts);
size_t hqq_hip_gemm_grouped_workspace_bytes(int nbits, int n_layers, const int64_t* N, int64_t M, int64_t K, int64_t group_size, int dtype, uint32_t opts);
int hqq_hip_gemm_grouped(int nbits, int n_layers, const void* x, const void* const* Wq, const void* const* scale, const void* const* zero, const void* const* bias,
                         void* const* y, const int64_t* N, int64_t M, int64_t K, int64_t group_size, int dtype, uint32_t opts, void* workspace, size_t workspace_bytes,
                         void* stream);
/* HQQLinear.matmul on already dequantised weights (quantize.py:880-882: torch.matmul(x, W.t())): y[M,N] = x[M,K] . Wd[N,K]^T (+ bias[N]),
 * fp16 / bf16, fp32 accumulation, one rounding (+ one for the bias add).  The GEMM half of the long-prompt route: hqq_hip_dequantize rebuilds
 * a layer's weights once, this contracts them with any number of tokens (csrc/gemm_dense.hip: 256 x 256 x 64 tiles, all operands by LDS-DMA,
 * four phases per K tile).  K % 64 == 0, N % 4 == 0; no workspace. */
int hqq_hip_gemm_dense(const void* x, const void* Wd, const void* bias, void* y, int64_t M, int64_t N, int64_t K, int dtype, void* stream);
/* hqq_hip_gemv for M it covers, hqq_hip_gemm otherwise; workspace: hqq_hip_forward_workspace_bytes with the same M */
int hqq_hip_forward(int nbits, const void* x, const void* Wq, const void* scale, const void* zero, const void* bias,
                    void* y, int64_t M, int64_t N, int64_t K, int64_t group_size, int dtype, uint32_t opts, void* workspace, size_t workspace_bytes,
                    void* stream);

/* ---------------------------------------------------------------------------------------------
 * Quantizer.quantize + optimize_weights_proximal_legacy + BitPack.pack_* in one call
 * (quantize.py:75-180, optimize.py:96-108, 201-255), axis=1, channel_wise=True.
 *   W          [N*K] of w_dtype (F32/F16/BF16); promoted to float32 (`tensor.float()`, quantize.py:102)
 *   max_v      round(2^nbits - 1)  (quantize.py:121);  pack_bits the container width {8,4,3,2,1}
 *   Wq_out     packed weights, layout of hqq_hip_pack
 *   scale_out  [N*K/gs] float32 = 1/scale (quantize.py:154) ; zero_out [N*K/gs] float32
 *   info_out   int32[2] on the device: {iterations run, stop iteration index}  (may be NULL)
 * The solver runs in float32 — the reference's CPU precision (optimize.py:231); packed levels, zero and scale equal the
 * reference's CPU path bit for bit (the reference's GPU path solves in fp16 and differs from its own CPU result).
 * ------------------------------------------------------------------------------------------- */
size_t hqq_hip_quantize_workspace_bytes(int64_t numel, int64_t group_size, int iters);
int hqq_hip_quantize(const void* W, int w_dtype, int64_t numel, int64_t group_size, int max_v, int pack_bits,
                     int round_zero, int optimize, int iters, float beta, float lp_norm,
                     void* Wq_out, float* scale_out, float* zero_out, int32_t* info_out,
                     void* workspace, size_t workspace_bytes, void* stream);

/* The same with axis=0 (quantize.py:104-116): W is viewed as [group_size, numel/group_size] and every COLUMN is a group (min/max,
 * scale, zero and the solver's mean run down the rows).  Wq_out: the packed [packed_rows(group_size), numel/group_size] tensor
 * (row slabs of the [group_size, C] level matrix share a byte / word, as BitPack.pack_* packs it); scale_out / zero_out
 * [numel/group_size] float32.  Same workspace size as hqq_hip_quantize. */
int hqq_hip_quantize_axis0(const void* W, int w_dtype, int64_t numel, int64_t group_size, int max_v, int pack_bits,
                           int round_zero, int optimize, int iters, float beta, float lp_norm,
                           void* Wq_out, float* scale_out, float* zero_out, int32_t* info_out,
                           void* workspace, size_t workspace_bytes, void* stream);

/* optimize_weights_proximal_legacy called on its own (optimize.py:208-255; `Quantizer.optimize_weights`): the same solver, started from the
 * CALLER's scale and zero instead of the group's min / max.  W viewed as [numel / group_size, group_size] (axis 1: a group per row) or
 * [group_size, numel / group_size] (axis 0: a group per column); scale_in / zero_in one float32 per group (scale as the quantiser uses it,
 * NOT inverted).  levels_out: the final W_q = clamp(rint(W * scale + zero), 0, max_v) as uint8 in W's view; zero_out: the solved zero per
 * group (scale is returned unchanged by the reference).  iters = 1 is one optimize_weights_proximal_legacy_step (optimize.py:201-206):
 * zero_out is then that step's new zero-point.  workspace: hqq_hip_quantize_workspace_bytes(numel, group_size, iters) + 4 bytes
 * per group. */
int hqq_hip_optimize(const void* W, int w_dtype, int64_t numel, int64_t group_size, int axis, int max_v, const float* scale_in, const float* zero_in,
                     int iters, float beta, float lp_norm, void* levels_out, float* zero_out, int32_t* info_out,
                     void* workspace, size_t workspace_bytes, void* stream);

/* channel_wise=False (quantize.py:114-116, 146): one scale and one zero for the WHOLE [rows, cols] tensor from its min and max, no
 * solver; the levels are packed in the tensor's own shape — Wq_out [packed_rows(rows), cols].  scale_out / zero_out: one float32 each
 * (scale already inverted, quantize.py:154).  cols % 8 == 0, rows * cols < 2^31.  workspace: HQQ_QUANTIZE_TENSOR_WS_BYTES. */
#define HQQ_QUANTIZE_TENSOR_WS_BYTES 16384
int hqq_hip_quantize_tensor(const void* W, int w_dtype, int64_t rows, int64_t cols, int max_v, int pack_bits, int round_zero,
                            void* Wq_out, float* scale_out, float* zero_out, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HQQ_HIP_H */
