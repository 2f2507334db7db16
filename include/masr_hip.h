/*
 * libmasr_hip.so -- C ABI of the MI355X (gfx950) MASR inference hot path.
 *
 * The reference (yeyupiaoling/MASR) has no FFI of its own: its boundary is three Python call
 * sites.  This header is the C-ABI a binding for that path would use; the
 * Python mirror of the reference classes lives in masr_amd/ and calls these functions via ctypes.
 *
 * Conventions
 *  - every function returns 0 on success, non-zero on error; masr_last_error() gives the message
 *  - all *_dev pointers are DEVICE pointers owned by the caller (e.g. torch tensors); the library
 *    never frees them, launches on the caller's hipStream_t (passed as void*) and never
 *    synchronises unless stated
 *  - host pointers are plain arrays read during the call
 *  - an engine is bound to one GPU and is not thread-safe (one engine per rank)
 *  - all arithmetic is fp32 ("f32" MFMA v_mfma_f32_32x32x2_f32), like the reference's CPU path
 */
#ifndef MASR_HIP_H
#define MASR_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct masr_engine masr_engine;

/* Model hyper-parameters = the YAML `encoder_conf` block (reference configs/conformer.yml:1-16)
 * + vocabulary size (masr/trainer.py:174-203 builds the model from exactly these). */
typedef struct masr_config {
    int32_t model_kind;      /* 0 = conformer                                                   */
    int32_t d_model;         /* encoder_conf.output_size        (256)                           */
    int32_t heads;           /* encoder_conf.attention_heads    (4)                             */
    int32_t d_ff;            /* encoder_conf.linear_units       (2048)                          */
    int32_t num_blocks;      /* encoder_conf.num_blocks         (12)                            */
    int32_t cnn_kernel;      /* encoder_conf.cnn_module_kernel  (15)                            */
    int32_t n_mels;          /* preprocess_conf.n_mels          (80)                            */
    int32_t vocab_size;      /* len(vocabulary.txt)                                             */
    int32_t causal;          /* `streaming: True` => causal conv + dynamic-chunk masks (model.py:37-42) */
    int32_t max_pos;         /* positional table length (reference max_len = 5000, embedding.py:14)  */
    int32_t device_id;
    int32_t reserved[5];     /* squeezeformer: [0] reduce_idx, [1] recover_idx; efficient_conformer: [0] stride layer, [1] grouped layers,
                                [2] group size; conformer: [0] = 1 -> cnn_module_norm: batch_norm (convolution.py:60-67; full-context only) */
} masr_config;

const char* masr_last_error(void);
int masr_version(void);

/* Lifecycle.  Replaces InferencePredictor.__init__'s torch.jit.load + .to(device)
 * (masr/infer_utils/inference_predictor.py:31-41). */
int masr_create(const masr_config* cfg, masr_engine** out);
void masr_destroy(masr_engine* e);

/* Weight upload keyed by the reference state_dict names (produced by
 * masr/trainer.py:308,684-689).  `host` is fp32, `shape` has `ndim` entries.  Names outside
 * `encoder.*` / `ctc.*` are ignored (attention decoder, unused by get_encoder_out*).
 * The optional name "__pos_table__" [max_pos, d_model] overrides the built-in sin/cos table
 * (conformer/embedding.py:31-37) so that it is bit-identical to torch's. */
int masr_load_tensor(masr_engine* e, const char* name, const float* host, const int64_t* shape, int32_t ndim);
/* Builds the derived device tensors (fused QKV, GLU-permuted pointwise_conv1, channels-last conv
 * weights, per-layer positional keys W_pos*PE) -- call once after all tensors are loaded. */
int masr_finalize(masr_engine* e, void* stream);

/* Feature front-end.  Replaces AudioFeaturizer.featurize
 * (masr/data_utils/featurizer/audio_featurizer.py:37-69,120-138) for a padded batch of 16 kHz
 * audio: per-utterance RMS-dB normalisation + int16 truncation + Kaldi fbank.  Works on an engine
 * without weights (tables are built by masr_create).
 *   samples_dev  [B, n_max] int16 PCM (sample_format 0; AudioSegment scales by 1/32768,
 *                audio.py:532-546) or float32 samples in [-1,1] scale (sample_format 1)
 *   n_samples_dev [B] int32
 *   feats_dev    [B, T_max, n_mels] f32, T_max = 1 + (n_max - 400) / 160; frames >= T_b are zero
 *   n_frames_dev [B] int32 (out, may be NULL)
 *   norm_pcm_dev [B, n_max] int16 (out, may be NULL): the normalised int16 samples (audio.py:549-574)
 *   gain_dev     [B] f32 (out, may be NULL): linear gain 10^(gain_dB/20) applied by normalize()
 *                (audio.py:256-264) -- lets the host mirror the reference's in-place mutation
 *   use_db_normalization  0 = off; 1 = gains computed on the device (every scalar step of rms_db / normalize / gain_db evaluated
 *                in double and rounded once to float32); 2 = gains SUPPLIED in gain_dev (in): the caller evaluated
 *                audio.py:287-304,519-529 with its own numpy on the mean square returned by masr_mean_square -- numpy's float32
 *                log10 / power are not correctly rounded and differ between hosts, so this is the mode that reproduces the
 *                reference's int16 samples bit for bit on the host it runs on.  Same meaning in masr_mfcc_batch / masr_linear_batch. */
int masr_fbank_batch(masr_engine* e, const void* samples_dev, int32_t sample_format, const int32_t* n_samples_dev,
                     int32_t B, int32_t n_max, int32_t use_db_normalization, float target_db, float* feats_dev,
                     int32_t* n_frames_dev, int16_t* norm_pcm_dev, float* gain_dev, void* stream);

/* mean_square_dev[b] = float32 np.mean(samples ** 2) of utterance b in numpy's own summation order (8192-element buffered
 * pairwise sums) -- the quantity AudioSegment.rms_db starts from (masr/data_utils/audio.py:519-529), bit-identical to numpy. */
int masr_mean_square(masr_engine* e, const void* samples_dev, int32_t sample_format, const int32_t* n_samples_dev, int32_t B,
                     int32_t n_max, float* mean_square_dev, void* stream);

/* MFCC front-end.  Replaces AudioFeaturizer._compute_mfcc (masr/data_utils/featurizer/audio_featurizer.py:98-117:
 * torchaudio.compliance.kaldi.mfcc(num_mel_bins=n_mels=80, num_ceps=n_mfcc, frame 25/10 ms, dither 0)) behind the same
 * AudioSegment handling as masr_fbank_batch (featurize(), :36-62).  mfcc_dev [B, T, n_ceps] f32, T = 1 + (n_max-400)/160. */
int masr_mfcc_batch(masr_engine* e, const void* samples_dev, int32_t sample_format, const int32_t* n_samples_dev, int32_t B,
                    int32_t n_max, int32_t use_db_normalization, float target_db, int32_t n_ceps, float* mfcc_dev,
                    int32_t* n_frames_dev, float* gain_dev, void* stream);
/* Linear log power spectrogram.  Replaces AudioFeaturizer._compute_linear (audio_featurizer.py:73-95; float64 arithmetic,
 * 20 ms Hann frames every 10 ms, 161 bins) on the float32 samples of the segment (:51-53).
 * feats_dev [B, T, 161] f32, T = (n_max-320)/160 + 1; n_frames_dev [B] = frames per utterance. */
int masr_linear_batch(masr_engine* e, const void* samples_dev, int32_t sample_format, const int32_t* n_samples_dev, int32_t B,
                      int32_t n_max, int32_t use_db_normalization, float target_db, float* feats_dev, int32_t* n_frames_dev,
                      float* gain_dev, void* stream);

/* Full-context encoder.  Replaces ConformerEncoder.forward called from
 * ConformerModel.get_encoder_out (masr/model_utils/conformer/model.py:152-167, encoder.py:305-346).
 *   feats_dev [B, T, n_mels] f32 (zero padded), feat_lens_dev [B] int32 (frames)
 *   decoding_chunk_size: -1 = full attention (get_encoder_out), >0 = chunk mask (encoder.forward(..., c, -1))
 *   enc_out_dev [B, T', d_model] f32, T' = ((T-1)/2-1)/2 */
int masr_encode_full(masr_engine* e, const float* feats_dev, const int32_t* feat_lens_dev, int32_t B, int32_t T,
                     int32_t decoding_chunk_size, float* enc_out_dev, void* stream);

/* CTC head.  Replaces CTCLoss.softmax (masr/model_utils/loss/ctc.py:62-70): probs = softmax(Linear).
 *   enc_dev [M, d_model] -> probs_dev [M, V] ; optional per-frame argmax / max-prob outputs */
int masr_ctc_probs(masr_engine* e, const float* enc_dev, int32_t M, float* probs_dev, int32_t* argmax_dev,
                   float* maxprob_dev, void* stream);
/* Same head without materialising probs for the caller: per-frame (argmax, max prob) only --
 * all that greedy_decoder (masr/decoders/ctc_greedy_decoder.py:20-21) reads from probs. */
int masr_ctc_greedy_frames(masr_engine* e, const float* enc_dev, int32_t M, int32_t* argmax_dev, float* maxprob_dev,
                           void* stream);
/* Best-path collapse + score.  Replaces the list logic of greedy_decoder
 * (ctc_greedy_decoder.py:20-30): tokens_dev [B, Tp] (-1 padded), n_tokens_dev [B],
 * score_dev [B] = fp32 sequential mean of the non-blank max-probs (caller multiplies by 100).
 * n_frames_dev [B] (may be NULL = all Tp frames, the reference's batch behaviour trainer.py:340). */
int masr_ctc_collapse(masr_engine* e, const int32_t* argmax_dev, const float* maxprob_dev, const int32_t* n_frames_dev,
                      int32_t B, int32_t Tp, int32_t blank, int32_t* tokens_dev, int32_t* n_tokens_dev,
                      float* score_dev, void* stream);
/* Stand-alone argmax/max-prob over host-provided probabilities already on the device
 * (greedy_decoder's np.argmax, ctc_greedy_decoder.py:20). */
int masr_argmax_rows(masr_engine* e, const float* probs_dev, int32_t M, int32_t V, int32_t* argmax_dev,
                     float* maxprob_dev, void* stream);

/* External n-gram language model scorer of the beam search.  Replaces the `Scorer` of paddlespeech_ctcdecoders (scorer.cpp on
 * KenLM; constructed in masr/decoders/beam_search_decoder.py:29-35, swig_wrapper.py:4-18; parameters alpha / beta of
 * configs/conformer.yml:74-88).  Character-based ARPA models up to order 5; `vocab_utf8` = the model's vocabulary tokens (the
 * LM words are matched to them by string; <s> / </s> get ids V / V + 1; n-grams with words the model cannot emit are dropped).
 * KenLM binaries are not parsed (returns non-zero, masr_lm_last_error()).  The table is uploaded to a GPU on first use there.
 *   masr_lm_cond_log_prob      ln P(ids[n-1] | ids[..n-2]) as Scorer::get_log_cond_prob scores the <s>-padded n-gram
 *                              (OOV_SCORE = -1000 as soon as one word is unknown to the LM)
 *   masr_lm_sentence_log_prob  Scorer::get_sent_log_prob: sum over the words and </s> */
typedef struct masr_lm masr_lm;
int masr_lm_load_arpa(const char* path, const char* const* vocab_utf8, int32_t V, masr_lm** out);
void masr_lm_destroy(masr_lm* lm);
const char* masr_lm_last_error(void);
int masr_lm_info(const masr_lm* lm, int32_t* max_order, int64_t* n_ngrams, int32_t* char_based, int64_t* skipped);
int masr_lm_cond_log_prob(const masr_lm* lm, const int32_t* ids, int32_t n, float* out);
int masr_lm_sentence_log_prob(const masr_lm* lm, const int32_t* ids, int32_t n, float* out);
/* Word-based models (an LM word longer than one character: Scorer::is_character_based() false, scorer.cpp load_lm): the words get
 * their own ids -- masr_lm_word_id (-1: not an LM word; the two scoring calls above then take word ids) -- and the scorer owns the
 * spelling dictionary of Scorer::fill_dictionary (every LM word spelled with vocabulary tokens + the space token);
 * masr_lm_dict_size = the number of words in it (Scorer::get_dict_size, printed by beam_search_decoder.py:38-42). */
int masr_lm_word_id(const masr_lm* lm, const char* word_utf8, int32_t* id);
int masr_lm_dict_size(const masr_lm* lm, int32_t* dict_size);

/* CTC prefix beam search.  Replaces BeamSearchDecoder.* -> paddlespeech_ctcdecoders
 * (masr/decoders/beam_search_decoder.py:45-96, swig_wrapper.py:35-121; third-party, un-vendored: parity
 * unpinned, external LM scorer not implemented = the alpha 0 path).
 * masr_ctc_topk (GPU): per-frame vocabulary pruning (cutoff_prob / cutoff_top_n, conformer.yml:74-88):
 *   idx_dev/logp_dev [M, top_n] candidates in descending probability (log(p + FLT_MIN)), count_dev [M].
 * masr_beam_* (host, like the reference's C++ thread pool): prefix search over those candidates.
 *   offline batch: masr_beam_search_batch over [B, T_stride, K] candidate arrays, frames_host [B] valid frames,
 *   tokens_host [B, max_len] out, len_host [B], score_host [B] = log probability of the best prefix;
 *   streaming: masr_beam_create / advance / result / reset / destroy (decode_chunk, reset_decoder). */
typedef struct masr_beam masr_beam;
int masr_ctc_topk(masr_engine* e, const float* probs_dev, int32_t M, int32_t V, int32_t top_n, float cutoff_prob,
                  int32_t* idx_dev, float* logp_dev, int32_t* count_dev, void* stream);
/* the same + blank_logp_dev [M] = ln p(blank) of every frame: the input of the decoder's pruning rule when an external scorer is
 * bound (ctc_beam_search_decoder.cpp: with a full beam, (prefix, c) is skipped once log p(c) + score(prefix) <
 * min_cutoff = score(worst live prefix) + ln p(blank) - max(0, beta)).  The *_lm searches below take that array
 * (blank_logp_*; NULL = rule off, every candidate of a frame is scored -- not what the reference decoder does). */
int masr_ctc_topk_blank(masr_engine* e, const float* probs_dev, int32_t M, int32_t V, int32_t top_n, float cutoff_prob,
                        int32_t blank, int32_t* idx_dev, float* logp_dev, int32_t* count_dev, float* blank_logp_dev,
                        void* stream);
int masr_beam_create(int32_t beam_size, int32_t blank, masr_beam** out);
void masr_beam_destroy(masr_beam* h);
int masr_beam_reset(masr_beam* h);
int masr_beam_advance(masr_beam* h, const int32_t* idx_host, const float* logp_host, const int32_t* count_host,
                      int32_t T, int32_t K);
int masr_beam_advance_lm(masr_beam* h, const int32_t* idx_host, const float* logp_host, const int32_t* count_host,
                         const float* blank_logp_host, int32_t T, int32_t K);
int masr_beam_result(masr_beam* h, int32_t* tokens_host, int32_t max_len, int32_t* len, float* score);
/* binds the external scorer (or NULL) and restarts the search: a prefix extended by a character adds
 * alpha * ln P_LM(character | last words of the prefix) + beta to its score (ctc_beam_search_decoder.cpp); the reported score is
 * the decoder's approx_ctc (scorer share removed again).  The *_lm variants of the batch searches take the same three values. */
int masr_beam_set_lm(masr_beam* h, const masr_lm* lm, float alpha, float beta);
int masr_beam_search_batch(const int32_t* idx_host, const float* logp_host, const int32_t* count_host,
                           const int32_t* frames_host, int32_t B, int32_t T_stride, int32_t K, int32_t beam_size,
                           int32_t blank, int32_t num_threads, int32_t* tokens_host, int32_t max_len, int32_t* len_host,
                           float* score_host);
int masr_beam_search_batch_lm(const int32_t* idx_host, const float* logp_host, const int32_t* count_host,
                              const int32_t* frames_host, int32_t B, int32_t T_stride, int32_t K, int32_t beam_size,
                              int32_t blank, int32_t num_threads, const masr_lm* lm, float alpha, float beta,
                              const float* blank_logp_host, int32_t* tokens_host, int32_t max_len, int32_t* len_host,
                              float* score_host);

/* CTC prefix beam search of a whole batch ON THE GPU (one workgroup per utterance; live prefixes, candidate scores and
 * the top-`beam_size` selection live in LDS, trie nodes of survivors in HBM).  Same inputs as masr_beam_search_batch but
 * DEVICE pointers (the outputs of masr_ctc_topk stay on the device), same outputs (token ids of the best prefix, its
 * length and log probability) as device arrays.  Replaces ctc_beam_search_decoding_batch of the third-party
 * paddlespeech_ctcdecoders (masr/decoders/swig_wrapper.py:67-103, beam_search_decoder.py:59-73), LM-free.
 * Limits: cutoff_top_n <= 64, beam_size <= 512, beam_size * cutoff_top_n * 6 B + tables <= 160 KB of LDS; returns non-zero (masr_last_error) beyond them -- use masr_beam_search_batch then. */
int masr_beam_search_gpu(masr_engine* e, const int32_t* idx_dev, const float* logp_dev, const int32_t* count_dev,
                         const int32_t* frames_dev, int32_t B, int32_t T_stride, int32_t K, int32_t beam_size,
                         int32_t blank, int32_t* tokens_dev, int32_t max_len, int32_t* len_dev, float* score_dev,
                         void* stream);
/* the same with the external scorer applied inside the kernel (LM table, known-word map in HBM; every live prefix carries its
 * packed last words, matched context length and backoff weights in LDS) */
int masr_beam_search_gpu_lm(masr_engine* e, const int32_t* idx_dev, const float* logp_dev, const int32_t* count_dev,
                            const int32_t* frames_dev, int32_t B, int32_t T_stride, int32_t K, int32_t beam_size,
                            int32_t blank, masr_lm* lm, float alpha, float beta, const float* blank_logp_dev,
                            int32_t* tokens_dev, int32_t max_len, int32_t* len_dev, float* score_dev, void* stream);

/* Streaming variant: a device-resident search per stream.  masr_gbeam_advance consumes the pruned candidates of the next T
 * frames (device arrays from masr_ctc_topk) and returns the best prefix so far (tokens/len/score device arrays); the
 * live prefixes and the trie nodes stay on the device between calls.  Replaces CTCBeamSearchDecoder.next / decode / reset
 * of the third-party module (masr/decoders/swig_wrapper.py:106-121, beam_search_decoder.py:75-96), LM-free.
 * max_frames bounds the number of frames of one utterance (trie node pool = max_frames * beam_size). */
int masr_gbeam_open(masr_engine* e, int32_t beam_size, int32_t blank, int32_t max_frames, int32_t* handle);
int masr_gbeam_advance(masr_engine* e, int32_t handle, const int32_t* idx_dev, const float* logp_dev,
                       const int32_t* count_dev, int32_t T, int32_t K, int32_t* tokens_dev, int32_t max_len,
                       int32_t* len_dev, float* score_dev, void* stream);
int masr_gbeam_advance_lm(masr_engine* e, int32_t handle, const int32_t* idx_dev, const float* logp_dev,
                          const int32_t* count_dev, const float* blank_logp_dev, int32_t T, int32_t K, int32_t* tokens_dev,
                          int32_t max_len, int32_t* len_dev, float* score_dev, void* stream);
int masr_gbeam_reset(masr_engine* e, int32_t handle);
int masr_gbeam_close(masr_engine* e, int32_t handle);
/* external scorer of a streaming search (between utterances only: after open or reset) */
int masr_gbeam_set_lm(masr_engine* e, int32_t handle, masr_lm* lm, float alpha, float beta);

/* Silero VAD network (the segmentation model of MASRPredictor.predict_long).  Replaces the onnxruntime session of the reference's
 * VADPredictor (masr/infer_utils/vad_predictor.py:36 InferenceSession(silero_vad.onnx), :83-104 session.run per window): the
 * weights are read out of the user's copy of that ONNX file by the host side (masr_amd/infer_utils/silero_vad.py) and handed
 * over tensor by tensor (names and shapes: csrc/silero.hip masr_vad_finalize), per sample rate (the file holds a 16 kHz and an
 * 8 kHz model).  masr_vad_forward runs B sequences x n_win consecutive windows of `window` samples (audio_dev [B, n_win * window]
 * float32, zero-padded by the caller like vad_predictor.py:126-127) from the LSTM state h_dev / c_dev [2, B, 64] (updated in
 * place: the reference's self._h / self._c) to probs_dev [B, n_win] -- n_win = 1 is one session.run, n_win = all windows of a
 * recording is the loop of get_speech_timestamps (:122-129) in two launches.  No CPU path: masr_vad_create fails without a GPU. */
typedef struct masr_vad masr_vad;
int masr_vad_create(int32_t device_id, masr_vad** out);
void masr_vad_destroy(masr_vad* v);
const char* masr_vad_last_error(void);
int masr_vad_load_tensor(masr_vad* v, int32_t sample_rate, const char* name, const float* data_host, int64_t n);
int masr_vad_finalize(masr_vad* v, int32_t sample_rate);
int masr_vad_forward(masr_vad* v, int32_t sample_rate, const float* audio_dev, int32_t B, int32_t n_win, int32_t window,
                     float* h_dev, float* c_dev, float* probs_dev, void* stream);

/* One call for the whole offline hot path (MASRPredictor.predict semantics, masr/predict.py:167-192,
 * batched like MASRTrainer.evaluate, trainer.py:632): PCM -> fbank -> encoder -> CTC greedy.
 * decode_all_frames != 0 reproduces the reference batch quirk of decoding padded frames. */
int masr_transcribe_batch(masr_engine* e, const int16_t* pcm_dev, const int32_t* n_samples_dev, int32_t B,
                          int32_t n_max, int32_t use_db_normalization, float target_db, int32_t decode_all_frames,
                          int32_t* tokens_dev, int32_t* n_tokens_dev, float* score_dev, void* stream);
/* The same pass as the facade drives it (MASRPredictor.predict / predict_batch, masr/predict.py:167-192): samples in either
 * format of masr_fbank_batch, use_db_normalization 0 / 1 / 2 with its meaning there (2: gains SUPPLIED in gain_dev [B], read
 * only -- the bit-exact int16 route of AudioSegment.normalize, audio.py:287-304), and the hypotheses as ONE packed int32 row per
 * utterance, rows_dev [B, T' + 2] = token ids (-1 padded) | token count | score bits (f32), so that a caller needs ONE copy
 * back per pass (T' = encoder frames of n_max samples, halved once more for the Efficient Conformer). */
int masr_transcribe_rows(masr_engine* e, const void* samples_dev, int32_t sample_format, const int32_t* n_samples_dev, int32_t B,
                         int32_t n_max, int32_t use_db_normalization, float target_db, const float* gain_dev,
                         int32_t decode_all_frames, int32_t* rows_dev, void* stream);

/* Serving pool: any number of concurrent predict_stream sessions on one engine, ONE call per step.  Replaces, per session, the
 * stream framing of MASRPredictor.predict_stream (masr/predict.py:237-343: carried-over samples re-normalised on every call,
 * :274-281; feature frames cached until a 67-frame decoding window is full, :283-306; 3 overlap frames kept, :329; the greedy
 * decoder's history, decoders/ctc_greedy_decoder.py:52-89) and, across sessions, the one-predictor-per-websocket loop of
 * infer_server.py:103-156 -- all sessions fed since the last step share one upload, one feature launch, lock-step chunk steps
 * (masr_encode_chunk), one collapse launch and one copy back.  Greedy decoding; a handle is the engine's stream id.
 *   masr_pool_create: feature_method 0 fbank / 1 mfcc (n_mfcc) / 2 linear (audio_featurizer.py:51-69), use_db_normalization and
 *                     target_db of preprocess_conf, max_frames_out as in masr_stream_open
 *   masr_pool_step:   the feeds since the last step, in order: feed_handle[k], feed_samples[k] (host memory: int16 PCM, format 0,
 *                     scaled by 1 / 2^15 like buf_to_float, data_utils/utils.py:382-411; or float32, format 1), feed_n[k] samples,
 *                     feed_is_end[k] (a session's flags are OR-ed).  gain_fn: evaluator of AudioSegment.normalize's scalar
 *                     expressions (audio.py:287-304,519-529) on the mean squares the device returns -- the python facade passes
 *                     its numpy evaluation, which is what makes the normalised samples bit-identical to the reference's on the
 *                     same host; NULL = libm (log10f / powf).  Returns non-zero to abort the step (gain beyond max_gain_db).
 *                     All feeds are validated (known handle, format, 0 <= feed_n <= 2^28, non-null samples) before any session
 *                     state changes: a rejected call commits nothing.
 *                     Out: the sessions of the step in first-fed order (handles_out[i], state_out[i] = 1 when the session
 *                     advanced by at least one window, 0 = the reference returns None; then, state -1, the sessions LEFT OUT of
 *                     the step because their stream has no room for the frames it would emit -- their feeds of this step are
 *                     dropped, their state is untouched, the other sessions advance), and for the advanced ones, in that
 *                     order, packed rows [row_width] = token ids (-1 padded) | count | score bits, in pinned host memory
 *                     (rows_host) and in HBM (rows_dev: what a multi-GPU front-end all-gathers); all out pointers stay valid
 *                     until the next step of this pool. */
typedef struct masr_pool masr_pool;
typedef int (*masr_gain_fn)(const float* mean_square, int32_t n, float target_db, float* gain_out, void* user);
int masr_pool_create(masr_engine* e, int32_t feature_method, int32_t n_mfcc, int32_t use_db_normalization, float target_db,
                     int32_t max_frames_out, masr_pool** out);
void masr_pool_destroy(masr_pool* p);
int masr_pool_open(masr_pool* p, int32_t* handle);
int masr_pool_close(masr_pool* p, int32_t handle);
int masr_pool_reset(masr_pool* p, int32_t handle);      /* MASRPredictor.reset_stream, predict.py:346-353 */
int masr_pool_step(masr_pool* p, int32_t n_feeds, const int32_t* feed_handle, const void* const* feed_samples,
                   const int64_t* feed_n, const int32_t* feed_format, const int32_t* feed_is_end, masr_gain_fn gain_fn,
                   void* gain_user, int32_t* n_sessions, const int32_t** handles_out, const int32_t** state_out,
                   const int32_t** rows_host, int32_t* row_width, int32_t** rows_dev, void* stream);
/* diagnostics: host time of masr_pool_step per phase, accumulated over `steps` calls (ms): assemble the samples | upload + mean
 * squares + wait | gains | features + frame bookkeeping | windows (lock-step chunk steps enqueued) | collapse + copy back + wait */
int masr_pool_profile(masr_pool* p, double* phase_ms, int64_t* steps, int32_t reset);
/* encoder frames that T feature frames give (Conv2dSubsampling4, subsampling.py:65-112; halved once more behind the Efficient
 * Conformer's stride layer) and the engine's geometry -- what a host needs to size the buffers above */
int masr_encoder_frames(masr_engine* e, int32_t feature_frames, int32_t* encoder_frames);
int masr_engine_info(masr_engine* e, int32_t* device_id, int32_t* n_mels, int32_t* vocab_size);

/* Streaming.  Replaces InferencePredictor.predict_chunk_conformer / reset_stream
 * (inference_predictor.py:80-102) -> ConformerEncoder.forward_chunk (encoder.py:348-420) with
 * required_cache_size < 0 (keep all history, predict.py:312-313).  Stream state (attention KV
 * cache, conv cache, offset) lives in the engine; `n` streams advance in lock-step per call.
 *   stream_ids [n] host int32, feats_dev [n, Tc, n_mels] (Tc <= 67), probs_dev [n, Tc', V] */
int masr_stream_open(masr_engine* e, int32_t max_frames_out, int32_t* stream_id);
int masr_stream_reset(masr_engine* e, int32_t stream_id);
int masr_stream_close(masr_engine* e, int32_t stream_id);
/* required_cache_size of forward_chunk for one stream (conformer/encoder.py:397-410): < 0 (default) every cached key stays
 * visible -- what MASRPredictor.predict_stream passes (predict.py:312-313); >= 0 a chunk attends over at most that many cached
 * keys, in frames of the encoder's input rate (Conformer, Squeezeformer: squeezeformer/encoder.py:292-297,338-347, and
 * Efficient-Conformer: efficient_conformer/encoder.py:323-336,365-381 -- their half-rate layers follow the reference's
 * next_cache_start // 2 trimming of the repeat-interleaved cache).  masr_stream_export_cache then returns the kept rows;
 * masr_stream_cache_len = their number (att_cache.size(2) of the reference after the last step). */
int masr_stream_set_history(masr_engine* e, int32_t stream_id, int32_t required_cache_size);
int masr_stream_offset(masr_engine* e, int32_t stream_id, int32_t* offset);
int masr_stream_cache_len(masr_engine* e, int32_t stream_id, int32_t* cache_len);
/* Subsampled frames the stream can still take before masr_encode_chunk refuses it (max_frames_out / the positional table's
 * max_len, conformer/embedding.py:48-50): lets a caller that advances many streams in one call leave a full one out instead of
 * failing the call. */
int masr_stream_room(masr_engine* e, int32_t stream_id, int32_t* frames_left);

/* Host side of the boundary: the utterances of one device pass -- B separate host arrays src[i] of n_samples[i] samples of
 * sample_bytes (2: int16 PCM, 4: float32) -- into ONE padded staging buffer dst [B rows of row_bytes], the rest of every row
 * zeroed: the padded batch the reference's collate_fn forms (masr/data_utils/collate_fn.py:6-34), built on the audio so that a
 * single transfer brings the pass to the device.  `threads` (1..16) library-owned worker threads take rows in turn; dst is
 * normally pinned memory.  No engine, no GPU. */
int masr_stage_rows(void* dst, int64_t row_bytes, const void* const* src, const int32_t* n_samples, int32_t B, int32_t sample_bytes,
                    int32_t threads);

/* Host-side resampling of one utterance to the model's rate.  Replaces resampy.resample(samples, sr, target, filter) behind
 * AudioSegment.resample (masr/data_utils/audio.py:306-317; resampy is third-party: its published band-limited sinc interpolation
 * is restated, parity unpinned).  x [n_orig] float32 host, ratio = sr_new / sr_orig, win / dwin [nwin] the right wing of the
 * windowed sinc (already scaled by ratio when ratio < 1) and its first differences, num_table samples per zero crossing,
 * y [n_out] with n_out = (int)(n_orig * ratio).  No engine, no GPU: an input-format step in front of the hot path. */
int masr_resample_f32(const float* x, int64_t n_orig, double ratio, const double* win, const double* dwin, int64_t nwin,
                      int32_t num_table, float* y, int64_t n_out);
int masr_encode_chunk(masr_engine* e, const int32_t* stream_ids, int32_t n, const float* feats_dev, int32_t Tc,
                      float* probs_dev, int32_t* argmax_dev, float* maxprob_dev, void* stream);
/* Read back a stream's caches in the reference layout (for parity tests):
 * att [L, H, t, 2*dk], cnn [L, 1, d, kernel-1] -- device pointers, t = current offset. */
int masr_stream_export_cache(masr_engine* e, int32_t stream_id, float* att_dev, float* cnn_dev, void* stream);

/* Single kernels, exposed for unit tests and profiling. */
int masr_op_layernorm(masr_engine* e, const float* x_dev, const float* w_dev, const float* b_dev, float* y_dev,
                      int32_t M, float eps, void* stream);
int masr_op_gemm(masr_engine* e, const float* a_dev, const float* w_dev, const float* bias_dev, const float* res_dev,
                 float* c_dev, int32_t M, int32_t N, int32_t K, int32_t act, float alpha, void* stream);

/* Side streams owned by the library: kind 0 / 1 = prefix searches of consecutive device passes, 2 = per-pass preparation (upload,
 * mean squares, gains), 3 = copies, 4 = the encoder passes of lane 1 (masr_select_lane).  One set per DEVICE, made with the first
 * masr_create on it and shared by every engine there; non-blocking, default priority.  The streams are CHOSEN BY PROBING which
 * hardware queue a candidate landed on (the runtime deals queues round-robin in creation order, so that depends on everything
 * the process created before): none of them shares the NULL stream's queue, kinds 0 and 1 are on different queues; kinds 0 and 4
 * are one stream (never used together), kinds 2 and 3 are one stream (short work only) -- see engine.hip.  The caller borrows them (torch.cuda.ExternalStream) and never destroys
 * them.  They take the place of the worker processes of the reference's batch decoder (masr/decoders/beam_search_decoder.py:59-73:
 * num_processes search workers beside the model's forward). */
#define MASR_SIDE_STREAMS 5
int masr_side_stream(masr_engine* e, int32_t kind, void** stream_out);

/* Lanes: an engine holds MASR_LANES sets of forward workspaces (activations, descriptor tables; ONE set of weights).  The calls
 * that follow masr_select_lane(e, k) use set k, so two offline passes of one engine can be in flight on two streams -- the
 * length-sorted device passes of a batch (the reference forms them one after the other: trainer.py:592-651 evaluates batch by
 * batch, predict.py:194-234 utterance by utterance) side by side: a pass whose row blocks do not fill the last round of a launch
 * leaves CUs to the other pass's kernels (BASELINE configs[2], two passes of 32: 17.0 -> 14.9 ms of encoder time).  Launches that
 * share a lane must be ordered by their stream, exactly as all launches of an engine had to be before; a set is allocated on its
 * lane's first use and sized by the largest pass it has seen.  Results do not depend on the lane.  Lane 0 is active after
 * masr_create; streaming sessions (masr_stream_*) keep their caches outside the lanes.  Not thread-safe, like every call. */
#define MASR_LANES 2
int masr_select_lane(masr_engine* e, int32_t lane);

/* Diagnostics: A/B switches of the kernels, for in-process measurements (tools/ *_ab.py); production = the defaults.
 *   1  fused-FFN variant (0 production, 1 without weight loads)      2  beam-search phase profile of workgroup 0
 *   5  1 = no out-proj + pw1 chain kernel                            6  0 = no K-split projection kernel
 *   7  0 = always the query-tiled attention kernel                   8  1 = no QKV tail stage on the first FFN
 *   9  1 = no conv-module head stage on the second FFN              12  row blocks below which the K-split projection kernel runs
 *  13  row blocks below which the FFN splits d_ff                   14  0 = two-term attention scores (no positional-key fold)
 *  15  0 = the offline embed projection never splits K              16  time every n-th matching launch (masr_profile_*)
 *  17  waves per workgroup of the offline conv2 launch (8 | 4)   18  0 = conv1 writes with plain instead of streaming stores
 *  19  timing experiment: every chunk-step layer on layer 0's weights
 *  21  waves per workgroup of the split-bf16 conv2 launch (8 | 4)   22  0 = no per-workgroup chunk rotation in the split-bf16 FFN
 *  23  0 = the full FFN launches stream their weights through the wave-private LDS slabs instead of reading the packed copies
 *      straight into registers (bit-identical either way)
 *  24  1 = the full FFN launches run two accumulator chains per wave (ffn_dual.hip) instead of one (ffn_pc.hip; bit-identical)
 *  25  0 = the offline out-proj + pw1 chain kernel and the CTC head stream their weights through LDS slabs instead of reading
 *      packed copies with buffer loads (bit-identical)
 *  20  1 = EXPLORATORY split-bf16 precision mode (not the reference's fp32 arithmetic, never the contract path): conv2, the embed
 *      projection and the other launches of the generic GEMM in the offline forward as a_hi*w_hi + a_hi*w_lo + a_lo*w_hi on
 *      the bf16 matrix pipe, fp32 accumulation (csrc/gemm_bf16x3.hip); 3 = also the FFN, unfused (slower than the fused fp32 FFN) */
int masr_debug_set(masr_engine* e, int32_t key, int32_t value);

/* Profiling: time every launch of one kernel class with HIP events on the launch stream.
 * kind: 0 none, 1 gemm (all), 2 ffn-w1 gemm, 3 conv2 gemm, 4 attention, 5 fbank.
 * masr_profile_read synchronises the events and returns total ms / launch count / flops since reset. */
int masr_profile_select(masr_engine* e, int32_t kind);
int masr_profile_read(masr_engine* e, double* total_ms, int64_t* launches, double* flops, int32_t reset);

#ifdef __cplusplus
}
#endif
#endif
