// Shared declarations for the MI355X (gfx950) MASR inference kernels.
// Everything here is internal to libmasr_hip.so; the public C ABI is include/masr_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lm_scorer.h"

// MASR_EXPERIMENTS = 1 (MASR_BUILD_EXPERIMENTS=1 at build time): the measured-and-rejected kernels of earlier rounds are compiled
// in and reachable through their masr_debug_set keys (ffn_dual.hip, ffn_coop.hip, ffn_x3.hip, gemm_bf16x3.hip, attn_chain_kernel,
// the head stage on d_ff-split launches).  The default build holds the product kernels only; those keys then fail loudly.
#ifndef MASR_EXPERIMENTS
#define MASR_EXPERIMENTS 0
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace masr {

// Opt-in to more than 64 KB of dynamic LDS for a kernel, checked and per device: `st` is the call site's table of the sizes
// already granted on each device ordinal (one process may drive several GPUs).  A failing hipFuncSetAttribute is not silent: it is
// recorded with note_launch_error() and the C ABI entry point that issued the launch returns it (engine.hip, LAUNCHCHK).
struct LdsAttr {
    size_t granted[32] = {};
};
void ensure_dynamic_lds(const void* fn, size_t bytes, LdsAttr& st);
void note_launch_error(const char* what, hipError_t e);

#ifdef __HIPCC__
// Sum over the 64 lanes of a wave, result in every lane.  DPP row shifts / row broadcasts (a scan whose last lane holds the
// total) + one readlane: ~10 VALU instructions.  __shfl_xor lowers to ds_bpermute_b32, i.e. six dependent LDS round trips
// (~600 cycles) per sum -- measurable in the LayerNorm prologues of the latency-bound streaming kernels.
// (sequence, frame) of row `row0 + lr` of a [b][T] row space, from (b0, t0) = divmod(row0, T) computed once per block: a carry
// loop instead of an integer division per row (a division by a run-time divisor is ~40 instructions; the row-local stages did
// 16-32 of them per lane)
struct SeqRow {
    int b, t;
};
__device__ __forceinline__ SeqRow seq_row(int b0, int t0, int T, int lr) {
    SeqRow r{b0, t0 + lr};
    while (r.t >= T) {
        r.t -= T;
        ++r.b;
    }
    return r;
}

__device__ __forceinline__ float wave_sum_dpp(float v) {
#define MASR_DPP_F(x, ctrl, rows) \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), (ctrl), (rows), 0xf, false))
    v += MASR_DPP_F(v, 0x111, 0xf);   // row_shr:1
    v += MASR_DPP_F(v, 0x112, 0xf);   // row_shr:2
    v += MASR_DPP_F(v, 0x114, 0xf);   // row_shr:4
    v += MASR_DPP_F(v, 0x118, 0xf);   // row_shr:8
    v += MASR_DPP_F(v, 0x142, 0xa);   // row_bcast:15 -> rows 1, 3
    v += MASR_DPP_F(v, 0x143, 0xc);   // row_bcast:31 -> rows 2, 3
#undef MASR_DPP_F
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
#endif


// ---- GEMM: C[M,N] = epilogue(A[M,K] * W[N,K]^T) on v_mfma_f32_32x32x2_f32 ----------------
enum GemmAct { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2 };
enum GemmAMode { A_PLAIN = 0, A_CONV2 = 1 };
enum GemmEpi { EPI_STD = 0, EPI_GLU = 1, EPI_SPLITK = 2 };

struct GemmArgs {
    const float* A;      // [M, lda] row-major (A_PLAIN) or conv1 activations [B,T1,F1,C] (A_CONV2)
    const float* W;      // [N, K] row-major (K contiguous) -- torch Linear.weight layout
    const float* bias;   // [N] or nullptr
    float* C;            // [M, ldc]
    const float* R;      // residual [M, ldr] or nullptr (may alias C)
    const int* lens;     // optional per-sequence feature lengths for row masking (see mask_tp)
    int M, N, K;
    int lda, ldc, ldr;
    int act;             // GemmAct (EPI_STD only)
    float alpha;         // out = R + alpha * act(acc + bias)
    int mask_tp;         // >0: row r -> (b = r / mask_tp, t = r % mask_tp); rows with 4*t >= lens[b] give act(..)=0
    int bias_after_alpha; // 1: out = R + alpha*act(acc) + bias  (Squeezeformer input_proj: scaling precedes the Linear)
    // A_CONV2 geometry: row m -> (b, t2, f2) with m = (b*T2 + t2)*F2 + f2 ; k -> (kh, kw*C + c)
    int T1, F1, T2, F2, Cc;
    int nsplit, ksplit;  // EPI_SPLITK: number of K ranges (gridDim.y) and BK-slabs per range
    // skip_rps > 0 (with lens): row r belongs to sequence r / skip_rps at frame (r % skip_rps) / skip_div; a tile whose rows all lie in
    // ONE sequence at frames with 4 * frame >= lens[sequence] (padding) is not computed -- its output rows keep what they held
    int skip_rps, skip_div;
};

void launch_gemm(const GemmArgs& a, int amode, int epi, hipStream_t s);
void set_conv2_mid_fill(int pct);     // diagnostics (masr_debug_set key 33)
void set_gemm_waves(int n);          // diagnostics (masr_debug_set key 17): waves per workgroup of the large conv2 launch, 8 (default) or 4
// deep-K, few-row GEMM: split K over workgroups into `partial` [nsplit][M][N], then reduce + epilogue into a.C
void launch_gemm_splitk(const GemmArgs& a, float* partial, int nsplit, hipStream_t s, int amode = A_PLAIN);
// exploratory split-bf16 variant (gemm_bf16x3.hip; masr_debug_set key 20): standard epilogue only; false = not taken
#if MASR_EXPERIMENTS
bool launch_gemm_bf16x3(const GemmArgs& a, int amode, hipStream_t s);
void set_gemm_bf16x3_waves(int n);
// fused split-bf16 FFN (ffn_x3.hip): weights packed once per FFN (hi / lo pieces in fragment order)
size_t ffn_x3_packed_elems(int dff);
void set_ffn_x3_rotation(int on);
void launch_pack_ffn_x3(const float* w1, const float* w2, unsigned short* p1, unsigned short* p2, int dff, hipStream_t s);
bool launch_ffn_x3(float* x, const float* lnw, const float* lnb, const unsigned short* p1, const float* b1,
                   const unsigned short* p2, const float* b2, int M, int dff, float eps, float scale, hipStream_t s);
#else
inline bool launch_gemm_bf16x3(const GemmArgs&, int, hipStream_t) { return false; }
inline void set_gemm_bf16x3_waves(int) {}
inline size_t ffn_x3_packed_elems(int) { return 0; }
inline void set_ffn_x3_rotation(int) {}
inline void launch_pack_ffn_x3(const float*, const float*, unsigned short*, unsigned short*, int, hipStream_t) {}
inline bool launch_ffn_x3(float*, const float*, const float*, const unsigned short*, const float*, const unsigned short*, const float*,
                          int, int, float, float, hipStream_t) { return false; }
#endif

// ---- elementwise / reductions ------------------------------------------------------------
// LayerNorm over rows of width 256.  If seq_t > 0 the output row is remapped to
// b*(seq_t+pad)+pad+t (conv-module padded layout) and rows with 4*t >= lens[b] are zeroed.
void launch_layernorm(const float* x, const float* w, const float* b, float* y, int M, float eps,
                      int seq_t, int pad, const int* lens, hipStream_t s);
// CMVN + Conv2d(1->C,3x3,s2) + ReLU, output channels-last [B,T1,F1,C]
void set_conv1_nt(int on);           // diagnostics (masr_debug_set key 18): 0 = conv1 writes its output with plain stores
void launch_conv1(const float* feats, const float* mean, const float* istd, const float* w9c, const float* bias,
                  float* out, int B, int T, int F, int C, hipStream_t s);
// depthwise causal conv (k taps) + LayerNorm(C=256) + SiLU on padded layout [nseq, pad+Tq, 256] -> [nseq*Tq, 256]
void launch_dwconv_ln_silu(const float* g, const float* wkc, const float* bias, const float* lnw, const float* lnb,
                           float* out, int nseq, int Tq, int ktaps, float eps, hipStream_t s, const float* gconst = nullptr);
// same with an eval-mode BatchNorm folded into a per-channel scale/shift instead of the LayerNorm
void launch_glu_const(const float* bias512, float* out256, hipStream_t s);
void launch_queue_spin(long long ticks, hipStream_t s);      // holds the stream's hardware queue for `ticks` x 10 ns
void launch_queue_nop(hipStream_t s);
void launch_dwconv_bn_silu(const float* g, const float* wkc, const float* bias, const float* scale, const float* shift,
                           float* out, int nseq, int Tq, int ktaps, hipStream_t s, const float* gconst = nullptr);
// Efficient-Conformer stride layer: depthwise causal conv with stride 2 (+ LayerNorm + SiLU) on the padded layout
// [nseq][ktaps-1 + Tin][256] -> [nseq * ceil(Tin/2), 256]; and the AvgPool1d(2, ceil_mode) residual path
void launch_dwconv_stride2_ln_silu(const float* g, const float* wkc, const float* bias, const float* lnw,
                                   const float* lnb, float* out, int nseq, int Tin, int ktaps, float eps, hipStream_t s);
void launch_avgpool2(const float* x, float* out, int B, int T, hipStream_t s);
// Squeezeformer TimeReductionLayer1D depthwise part: k=5, stride 2, padding 3, pad-masked input -> [B, ceil(T/2), 256]
void launch_time_reduce_dw(const float* x, const float* w5c, const float* bias, const int* lens, float* out, int B, int T,
                           int mstride, hipStream_t s);
// Squeezeformer recover: x[b,t,:] = saved[b,t,:] + y[b, t/2, :]
void launch_recover_add(const float* saved, const float* y, float* x, int B, int T, int L, hipStream_t s);
// softmax over V per row (in place) + argmax / max prob
void launch_softmax_argmax(float* logits, int M, int V, int ldv, int write_probs, int* idx, float* maxp, hipStream_t s);
// CTC collapse per utterance
void launch_ctc_collapse(const int* idx, const float* maxp, const int* nframes, int B, int Tp, int blank,
                         int* tokens, int* ntok, float* score, hipStream_t s);
struct PoolSeg {            // masr_pool_step: one contiguous run of rows to copy (rows of `width` floats)
    long src, dst;
    int n, pad_;
};
void launch_ctc_collapse_hist(const int* idx, const float* maxp, const int* nframes, const int* in_rows, int ld_in, int B, int Tp,
                              int blank, int* rows, hipStream_t s);
void launch_copy_segments(const float* src, float* dst, const float* src2, float* dst2, const PoolSeg* seg, int nseg,
                          int max_rows, int width, hipStream_t s);
void launch_ctc_collapse_rows(const int* idx, const float* maxp, const int* nframes, int B, int Tp, int blank, int* rows,
                              hipStream_t s);
void launch_topk_prune(const float* probs, int M, int V, int top_n, float cutoff_prob, int* out_idx, float* out_logp,
                       int* out_cnt, int blank, float* out_blank_lp, hipStream_t s);
void launch_argmax_rows(const float* probs, int M, int V, int* idx, float* maxp, hipStream_t s);
void launch_frame_counts(const int* nsamp, int B, int* nfr, int* nenc, int halve, hipStream_t s);
void launch_export_att(const float* cache, float* out, int L, int H, int cap, int t, int dk, hipStream_t s, int rate = 1,
                       int shift = 0);   // rate 2: half-rate cache, entry (j + shift) / 2 (the reference repeat-interleaves)
struct PlaneCopy { const float* src_k; const float* src_v; float* dst_k; float* dst_v; };
void launch_kv_append_planar(const PlaneCopy* pc, int n, int Tq, hipStream_t s);
void launch_export_cnn_layer(const float* cache, float* out, int used, int total, int d, hipStream_t s);
void launch_export_att_planar(const float* kplane, const float* vplane, float* out, int H, int t, int dk, hipStream_t s);
void launch_affine_rows(const float* x, const float* w, const float* b, float* y, int M, int seq_t, int pad, hipStream_t s);
void launch_export_cnn(const float* cache, float* out, int L, int pad, int d, hipStream_t s);

// ---- row-block GEMM for the K = 256 projections (rowgemm.hip) -------------------------------------
struct AttSeq;
enum RowGemmPro { RG_PRO_PLAIN = 0, RG_PRO_LN = 1, RG_PRO_LN_PAD = 2, RG_PRO_AFFINE = 3,
                  RG_PRO_HIST = 4, RG_PRO_DWCONV = 5 };      // small-M kernel only (rowgemm_small.hip): fused streaming conv-module fronts
enum RowGemmEpi { RG_EPI_STORE = 0, RG_EPI_RESID = 1, RG_EPI_GLU = 2, RG_EPI_CTC = 3, RG_EPI_CHAIN = 4 };
struct RowGemmArgs {
    const float* A;       // [rows, lda] source rows (K = 256)
    const float* lnw;     // LayerNorm weight / bias (PRO_LN*)
    const float* lnb;
    const float* W;       // [N, 256]
    const float* Wp;      // optional: the same weights in fragment order (launch_pack_rows_pc, N rounded up to 256); the offline CHAIN /
                          // CTC launches then read them with buffer loads straight into MFMA operand registers (no LDS slab staging)
    const float* bias;    // [N]
    float* C;             // output [M, ldc]
    const float* R;       // residual [M, ldr] (EPI_RESID; may alias C)
    float* R2;            // EPI_CHAIN: where the updated residual stream x + out-proj goes (may alias R); C = GLU output
    const int* lens;      // per-sequence feature lengths (pad masks), or nullptr
    int* out_idx;         // EPI_CTC: per-row argmax
    float* out_maxp;      // EPI_CTC: per-row softmax probability of the argmax
    int M, N;             // output rows / weight rows
    int lda, ldc, ldr;
    float eps, alpha;
    int seq_t, pad;       // PRO_LN_PAD: output rows index [nseq][pad + seq_t], source rows [nseq][seq_t]
    int mask_tp;          // EPI_RESID: >0 -> row r = (b, t) with t = r % mask_tp masked when mstride*t >= lens[b]
    int mstride;          // feature frames per encoder frame for the pad masks (4; 8 after Squeezeformer time reduction)
    int out_seq_t;        // >0: output row (b, t) is stored at b*(out_seq_t + out_pad_tot) + out_pad_l + t
    int out_pad_l, out_pad_tot;
    int plane_cols;       // EPI_STORE: >0 -> column c goes to plane c / plane_cols (planar q | k | v buffers)
    long plane_stride;    //            floats between planes
    int a_seq_t, a_seq_stride;   // PRO_PLAIN: >0 -> source row of (b, t) = b * a_seq_stride + t, with (b, t) = divmod(row, a_seq_t)
    float* const* cache_rd;      // PRO_HIST: per-stream cnn cache (read half / written half of the double buffer); A = x rows,
    float* const* cache_wr;      //           seq_t new rows + pad history rows per stream, M = n * (seq_t + pad)
    int hist_affine;             //           1: new rows = lnw * x + lnb (Squeezeformer, convolution.py:109-110) instead of LayerNorm
    const float* dw_w;           // PRO_DWCONV: depthwise weights [pad + 1][256] and bias [256]; A = GLU rows in the padded layout
    const float* dw_b;           //             [n][pad + seq_t][256], lnw / lnb = the conv module's LayerNorm, M = n * seq_t
    const float* gconst;         //             offline causal conv: the pad history rows of every sequence are not materialised -- they
                                 //             all hold this constant row glu(bias) [256] (nullptr: read them like any other row)
    const AttSeq* kv_seqs;       // EPI_STORE, small-M kernel: columns >= 256 (k | v of the fused QKV projection) of row (b, t) =
    int kv_tq;                   //   divmod(row, kv_tq) go to stream b's key/value cache instead of C (replaces launch_kv_append)
};
bool launch_rowgemm(const RowGemmArgs& a, int pro, int epi, hipStream_t s);   // false: not applicable, nothing launched (HIST / DWCONV)
bool launch_rowgemm_small(const RowGemmArgs& a, int pro, int epi, hipStream_t s);   // rowgemm_small.hip; false = not applicable
void set_rowgemm_small_blocks(int n);                                                // tuning (masr_debug_set key 12)
void set_rowgemm_small(int on);                                                      // diagnostics (masr_debug_set key 6)

// Fused FFN block, in place: x <- x + scale * (W2 . silu(W1 . LN(x) + b1) + b2)   (ffn_pc.hip; launched through launch_ffn_fused in ffn_reduce.hip)
// partial/nsplit: split-d_ff mode for small M (streaming).  post: LayerNorm that follows the block in the layer; it is fused into
// the split-mode reduction (return value 1), otherwise the caller runs it (return value 0)
struct FfnPostLn {
    const float* lnw;
    const float* lnb;
    float* y;          // destination rows (may alias x)
    float eps;
};
// tail: a row-local stage appended to the full (non-split) kernel: out[M, N] = LayerNorm(x_new; lnw, lnb) . W[N, 256]^T + bias
// (the fused QKV projection that follows the first macaron FFN).  Return value 2 = done by the kernel.
struct FfnTail {
    const float* lnw;
    const float* lnb;
    const float* W;
    const float* bias;
    float* out;
    int N, ldo;
    const float* pre_lnw;      // optional: x <- LayerNorm(x; pre_lnw, pre_lnb) first, written back (the previous layer's norm_final,
    const float* pre_lnb;      // encoder.py:160-161, riding on this launch instead of its own)
    // planar output (the Efficient Conformer's grouped attention: q | k | v planes [nseq][seq_t + pad_t][256], the time padding
    // rows stay zero): plane_stride > 0 -> column tile c / 256 goes to plane c / 256, row (b, t) = divmod(row, seq_t) to row
    // b * (seq_t + pad_t) + t of that plane (ffn_pc.hip only)
    long plane_stride;
    int seq_t, pad_t;
};
// head: the rest of the conv module in front of the block, on the workgroup's own 32 rows, before the block's LayerNorm:
//   x <- x + mask(pointwise_conv2(SiLU(LayerNorm(depthwise_conv(glu)))))       (convolution.py:120-131, encoder.py:137-148)
// the depthwise conv reads the GLU rows of the padded layout [nseq][pad + seq_t][256] (ktaps - 1 = pad history rows in front,
// gconst: constant history rows that are not materialised).  Return value bit 2 (4) = done by the kernel.
struct FfnHead {
    const float* glu;          // nullptr: no head stage
    const float* dw_w;         // [ktaps][256]
    const float* dw_b;
    const float* lnw;          // the conv module's LayerNorm
    const float* lnb;
    const float* gconst;       // [256] or nullptr
    const float* W;            // pointwise_conv2 [256, 256]
    const float* bias;
    const int* lens;           // feature lengths for the pad mask, or nullptr
    int seq_t, ktaps, mstride;
    float* xout;               // d_ff-split launches (few rows) only: where the rows updated by the head stage go -- every d_ff
                               // slice of a row block runs the head on the SAME old rows of x, so x itself must stay untouched
                               // until the split reduction, which then reads xout and writes x
};
int launch_ffn_fused(float* x, const float* lnw, const float* lnb, const float* w1, const float* b1, const float* w2,
                     const float* b2, int M, int dff, float eps, float scale, int affine_prologue, float* partial,
                     int nsplit, hipStream_t s, const FfnPostLn* post = nullptr, const FfnTail* tail = nullptr,
                     const FfnHead* head = nullptr, bool packed = false);
void launch_pack_ffn_pc(const float* w1, const float* w2, float* p1, float* p2, int dff, hipStream_t s);
void launch_pack_rows_pc(const float* w, float* p, int N, hipStream_t s, int n_src = -1);   // weights [n_src, 256] -> N % 256 == 0 packed rows (rows >= n_src zero)
// sqz_layer.hip: one Squeezeformer (post-LN) half-layer per launch on the workgroup's own 32 rows --
//   stage 0 "mid": x <- LN_a(x + att . head_w^T + head_b);  x <- LN_b(x + FFN(ffn_s * x + ffn_b));
//                  glu_out <- GLU(pw1(mask(tail_s * x + tail_sb)))        (tail_n = 512, padded layout of the depthwise conv)
//   stage 1 "end": x <- LN_a(x + mask(pw2(SiLU(BN(dwconv(glu))))));  out <- LN_b(x + FFN(ffn_s * x + ffn_b));
//                  tail_out <- Wqkv . (tail_s * out + tail_sb) + bqkv     (tail_n = 768; tail_w == nullptr: no tail)
// head_w / tail_w: packed rows (launch_pack_rows_pc), w1 / w2: packed FFN weights (launch_pack_ffn_pc)
struct SqzStageArgs {
    float* x;                  // residual stream [M, 256], updated in place
    float* out;                // where the stage's last LayerNorm goes (x, or the encoder output for the last layer)
    const float* att;          // stage 0: attention output rows [M, 256]
    const float* head_w;       // stage 0: Wo; stage 1: pointwise_conv2
    const float* head_b;
    const float *ln_a_w, *ln_a_b, *ln_b_w, *ln_b_b;
    const float *ffn_s, *ffn_b, *w1, *b1, *w2, *b2;
    const float *tail_w, *tail_b, *tail_s, *tail_sb;
    float* tail_out;           // stage 1: qkv [M, 768]
    float* glu_out;            // stage 0: [nseq][glu_pad_tot + seq_t][256], real rows at glu_pad_l
    const float* glu;          // stage 1: the same buffer
    const float *dw_w, *dw_b, *bn_scale, *bn_shift, *gconst;
    const float* gpad;         // stage 1 with skip_pad, symmetric conv: glu(pointwise_conv1 bias) [256], the GLU row of a padded frame
    const int* lens;           // feature lengths for the pad mask, or nullptr
    int M, dff, seq_t, mstride, ktaps, tail_n, glu_pad_l, glu_pad_tot;
    float eps;
    int skip_pad;              // 1 (with lens): a row block of padded frames only is not computed (x, glu, qkv rows keep what they held)
};
bool launch_sqz_stage(const SqzStageArgs& a, int stage, hipStream_t s);
// ffn_dual.hip: the same block with two independent accumulator chains per wave (chunks of 256 hidden units); p1 / p2 from
// launch_pack_ffn_dual, tail->W from launch_pack_rows_dual (N = 768), head->W from launch_pack_rows_pc.
// Returns 0 / 2 (tail done) / 4 (head done), -1 when the sizes are not covered
#if MASR_EXPERIMENTS
int launch_ffn_dual(float* x, const float* lnw, const float* lnb, const float* p1, const float* b1, const float* p2,
                    const float* b2, int M, int dff, float eps, float scale, int affine_prologue, hipStream_t s,
                    const FfnTail* tail, const FfnHead* head);
void launch_pack_ffn_dual(const float* w1, const float* w2, float* p1, float* p2, int dff, hipStream_t s);
void launch_pack_rows_dual(const float* w, float* p, hipStream_t s);
#else
inline int launch_ffn_dual(float*, const float*, const float*, const float*, const float*, const float*, const float*, int, int, float,
                           float, int, hipStream_t, const FfnTail*, const FfnHead*) { return -1; }
inline void launch_pack_ffn_dual(const float*, const float*, float*, float*, int, hipStream_t) {}
inline void launch_pack_rows_dual(const float*, float*, hipStream_t) {}
#endif

// one-chunk FFN slices for few rows (ffn_coop.hip): all eight waves on both products; w1p = launch_pack_ffn_coop_w1's copy, w2p =
// ffn_pc.hip's packed W2, partial [dff / 128][M][256]
#if MASR_EXPERIMENTS
void launch_pack_ffn_coop_w1(const float* w1, float* p, int dff, hipStream_t s);
void launch_ffn_coop(const float* x, const float* lnw, const float* lnb, const float* w1, const float* b1, const float* w2, int M,
                     int dff, float eps, int affine, float* partial, hipStream_t s);
#else
inline void launch_pack_ffn_coop_w1(const float*, float*, int, hipStream_t) {}
inline void launch_ffn_coop(const float*, const float*, const float*, const float*, const float*, const float*, int, int, float, int,
                            float*, hipStream_t) {}
#endif
void launch_ffn_reduce(float* x, const float* partial, const float* b2, int M, int nsplit, float scale, hipStream_t s,
                       const FfnPostLn* post, const float* xin = nullptr);
void set_ffn_variant(int v);   // diagnostic ablations of the fused FFN kernel (0 = production)

// ---- CTC prefix beam search on the GPU (beam_gpu.hip) ---------------------------------------------
struct BeamGpuArgs {
    const int* cidx;       // [B * T_stride, K]
    const float* clp;      // [B * T_stride, K]
    const int* ccount;     // [B * T_stride]
    const float* blank_lp; // [B * T_stride] ln p(blank) of every frame: input of the decoder's min_cutoff rule (nullptr: rule off)
    const int* frames;     // [B] frames to consume per utterance (nullptr: T_stride)
    int T_stride, K, beam, blank, max_len;
    int* pool_parent;      // [B][pool_cap]
    int* pool_ch;          // [B][pool_cap]
    int pool_cap;
    int* state_i;          // [B][2 + 3 * beam]: n_live, pool_count, node[], ch[], lm m|oov[]     (persistent streams)
    unsigned long long* state_h;   // [B][3 * beam]: string identity of the prefix and of its parent, packed LM context
    float* state_f;        // [B][7 * beam]:     b[], nb[], score[], LM backoffs [4][]
    int init;              // 1: start from the empty prefix; 0: continue from state_*
    int* tokens;           // [B][max_len]
    int* len;              // [B]
    float* score;          // [B]
    long long* prof;       // optional [8] cycle counters of workgroup 0 (phase breakdown), or nullptr
    // external scorer (lm_scorer.h): a new prefix p + c adds alpha * ln P_LM(c | last words of p) + beta to its score
    int use_lm;
    LmView lm;             // table / known in the memory of the launch device
    float alpha, beta;
    int lm_cache;          // 1: one scorer probe per (distinct effective context, candidate) and frame; 0: one per (prefix, candidate)
    int narrow;            // 1: frames of <= 1024 extension entries run on the narrow step (beam_gpu.hip); 0: every frame on the wide step
};
// per-utterance search state in HBM: [3 * beam] u64 | [2 + 3 * beam] int | [7 * beam] float
inline size_t beam_state_bytes(int beam) { return (size_t)3 * beam * 8 + (size_t)(2 + 3 * beam) * 4 + (size_t)7 * beam * 4 + 8; }
inline void beam_state_carve(void* base, int B, int beam, unsigned long long** h, int** i, float** f) {
    *h = reinterpret_cast<unsigned long long*>(base);                       // [B][3 * beam]
    *i = reinterpret_cast<int*>(*h + (size_t)B * 3 * beam);                 // [B][2 + 3 * beam]
    *f = reinterpret_cast<float*>(*i + (size_t)B * (2 + 3 * (size_t)beam)); // [B][7 * beam]
}
size_t beam_gpu_lds_bytes(int beam, int K, bool use_lm);
int launch_beam_search(const BeamGpuArgs& a, int B, hipStream_t s);   // 1: sizes not supported

// ---- DeepSpeech2 (lstm.hip) --------------------------------------------------------------------
void launch_lstm_step(const float* gx, const float* whh, const float* h_prev, float* h_next, float* c, float* out,
                      const int* lens, int B, int T, int H, int step, int ndir, hipStream_t s);
void launch_layernorm_generic(const float* x, const float* w, const float* b, float* y, int M, int N, float eps,
                              hipStream_t s);
void launch_ds2_lens(const int* lens, int B, int Tq, int* out, hipStream_t s);

// ---- attention ---------------------------------------------------------------------------
struct AttSeq {            // one per sequence, device memory
    const float* q;        // first query row of this sequence (row stride q_stride)
    const float* k;        // first key row   (row stride kv_stride)
    const float* v;        // first value row (row stride kv_stride)
    float* out;            // first output row (row stride 256)
    int nq, nk;            // number of query rows / key rows
    int klen;              // keys j >= klen are masked (padding)
    int pos0;              // positional index of key 0
    int q_abs0;            // absolute index of query 0 (for the chunk mask)
    int pad_;
};
// attention + [out-projection + residual -> LayerNorm -> pointwise_conv1 -> GLU] of an offline Conformer layer as ONE launch
// (attention.hip attn_chain_kernel; d_model 256, 4 heads of 64): the arguments of launch_attention + those of the EPI_CHAIN rowgemm
struct AttnChainArgs {
    const AttSeq* seqs;        // [nseq]; q / k / v rows of sequence b, nq = nk = frames of the padded batch
    int nseq, nqb;             // nqb (query blocks of 32 per sequence) is filled in by the launcher
    int q_stride, kv_stride, chunk_size, pos_stride;
    const float* ptab;         // [max_pos, 256] positional keys of the layer
    const float* bias_u;       // [4][64]
    const float* bias_v;
    const float* Wp;           // [Wo; W_pw1] packed in fragment order (launch_pack_rows_pc, 768 rows)
    const float* bias;         // [768]  bo | b_pw1 (value | gate)
    const float* lnw;          // the conv module's LayerNorm
    const float* lnb;
    const float* R;            // residual stream in  [nseq * seq_t, 256]
    float* R2;                 // residual stream out (may alias R)
    float* C;                  // GLU output, padded layout [nseq][out_pad_tot + seq_t][256], real rows at out_pad_l
    const int* lens;           // feature lengths for the pad mask, or nullptr
    int seq_t, mstride, out_pad_l, out_pad_tot;
    float eps;
};
#if MASR_EXPERIMENTS
bool launch_attn_chain(const AttnChainArgs& a, int max_nq, hipStream_t s);
#else
inline bool launch_attn_chain(const AttnChainArgs&, int, hipStream_t) { return false; }
#endif
void launch_attention(const AttSeq* seqs, int nseq, int max_nq, int heads, int q_stride, int kv_stride,
                      const float* ptab /*[max_pos,256]*/, const float* bias_u, const float* bias_v,
                      int chunk_size, int pos_stride, hipStream_t s);
void set_attention_grouped_fold(int on);   // key 26: 0 = the two-wave two-term grouped attention kernel (A/B)
int rowgemm_small_blocks();                // row blocks below which the K-split projection kernel takes a launch
void set_attention_fewq_wgs(int n);        // key 28
void set_attention_fold(int on);     // diagnostics (masr_debug_set key 14): 0 = two-term score contraction in attention_kernel
void set_attention_fewq(int on);     // diagnostics (masr_debug_set key 7): 0 = always the query-tiled kernel
void launch_attention_grouped(const AttSeq* seqs, int nseq, int max_nq, int heads, int group, const float* ptab,
                              int t_true, const float* bias_u, const float* bias_v, hipStream_t s, int chunk_size = 0);
void launch_attseq_grouped(AttSeq* seqs, const float* q, const float* k, const float* v, float* out, const int* lens,
                           int B, int Tg, int group, int mstride, hipStream_t s);
void launch_kv_append(const AttSeq* seqs, const float* qkv, int n, int Tq, hipStream_t s);
void launch_cnn_cache_move(float* const* caches, float* lnpad, int n, int Tq, int pad, int dir, hipStream_t s);
// streaming conv-module front in one launch: history rows <- cache_rd, new rows <- LayerNorm / affine of x, cache_wr <- last pad rows
void launch_conv_hist(const float* x, const float* w, const float* b, float* const* cache_rd, float* const* cache_wr,
                      float* lnpad, int n, int Tq, int pad, int affine, float eps, hipStream_t s);
void launch_attseq_full(AttSeq* seqs, const float* qkv, float* out, const int* lens, int B, int Tp, int mstride,
                        hipStream_t s, int valid_only = 0);     // valid_only: nq = nk = klen (padded queries / keys are not touched)

// ---- features ------------------------------------------------------------------------------
size_t fbank_gain_scratch_floats(int B);   // size of gain_scratch ([B] gains + partial sums)
// use_db: 0 = no dB normalisation, 1 = gains computed on the device, 2 = gains supplied in gain_scratch[0 .. B)
// tables of the fbank front-end (engine.hip build_fbank_tables): povey window [400]; mel weights transposed [16 taps][80] (tap i of
// filter m = bin mel_lo[m] + i, zero past the filter's last bin); W512^k (k <= 256) for the real-split pass; twr4 [3 passes][3][64]:
// the twiddles of the radix-4 passes Ns = 4, 16, 64 per lane
struct FbankTables {
    const float *window, *melwt, *tw512, *twr4;
    const int* mel_lo;
};
void launch_fbank(const void* pcm, int sample_format /*0 int16, 1 float32*/, const int* nsamp, int B, int n_max,
                  int use_db, float target_db, const FbankTables& tb, float* feats, int T_max, float* gain_scratch,
                  int16_t* norm_out, hipStream_t s);
void launch_mean_square(const void* pcm, int sample_format, const int* nsamp, int B, int n_max, float* gain_scratch,
                        float* ms_out, hipStream_t s);
void launch_mfcc(const float* fbank, long rows, int n_ceps, const float* dct /*[80][n_ceps]*/, const float* lifter, float* out,
                 hipStream_t s);
void launch_linear_spec(const void* pcm, int sample_format, const int* nsamp, int B, int n_max, int use_db, const float* gain,
                        const double* win /*[320]*/, const double* tw /*[320][2]*/, double scale, float* feats /*[B][T_max][161]*/,
                        int T_max, hipStream_t s);
void launch_linear_frame_counts(const int* nsamp, int B, int* nfr, hipStream_t s);

}  // namespace masr
