// CTC prefix beam search on the GPU (LM-free part of the ctc_beam_search decoder; reference call sites
// masr/decoders/beam_search_decoder.py:45-96 -> third-party paddlespeech_ctcdecoders, see beam_search.cpp for the
// host restatement of the same algorithm and the "parity unpinned" note).
//
// One workgroup (1024 threads) per utterance walks the frames sequentially; everything a step needs lives in LDS:
//   live prefixes  (<= beam):  64-bit identity of the prefix STRING (hash chain over its characters) and of its parent,
//                              back-pointer node in HBM, last character, log P(blank end), log P(non-blank end), score
//   candidates     (<= K):     the pruned vocabulary of the frame (topk_prune_kernel), descending probability
//   entry scores   (<= beam * K): every (prefix, character) extension, as order-preserving integer keys
// Step (== Beam::step of beam_search.cpp, reformulated without a pointer trie):
//   A  every live prefix p:  b_cur = lp(blank) + score(p);  nb_cur = lp(ch(p)) + nb_prev(p)        (repeat of its last char)
//   B  every pair (p, c != blank): add = c == ch(p) ? lp(c) + b_prev(p) : lp(c) + score(p).  If the child (p, c) is itself a
//      live prefix (p's list of live children: every step the live prefixes find their parent through an LDS hash of
//      the string identities -- a prefix that was dropped and is re-created later is the SAME prefix, as in the
//      reference's trie where dead nodes with live descendants are kept) the term is merged into that
//      prefix's nb_cur -- a node has one parent, so at most one such term per live prefix and the two-term log-sum-exp is
//      order independent -- otherwise it is a new entry.  Threads own (prefix, k-range): the prefix state is read once.
//   C  score = logsumexp(b_cur, nb_cur); keep the `beam` best entries.  A valid lower bound of the beam-th best score
//      (the beam-th best of the 1024 per-thread maxima over a strided sample, 3 radix passes on one value per thread)
//      discards ~95 % of the entries; an exact 4-pass radix select over the survivors gives the threshold; ordered
//      (deterministic) compaction.  Survivors that are new get trie nodes (parent pointer + character) appended to the
//      utterance's node pool in HBM -- only survivors ever get a node.
// With the external scorer bound the decoder's own pruning rule applies (ctc_beam_search_decoder.cpp: min_cutoff / full_beam):
// when the beam is full, a pair (p, c) is skipped -- blank, repeat and extension terms alike -- once
// log p(c) + score(p) < score(worst live prefix) + ln p(blank) - max(0, beta).  The reference walks the prefixes in descending
// score order and breaks at the first one that fails; the test is monotone in score(p), so it is the same set of pairs.
// Pairs that are cut cost no language-model lookups.
// Entries with score -inf are never revived except through their parent's extension, which re-creates them, so dropping
// them is equivalent to the pointer trie that keeps them.
// The best prefix is read back by walking parent pointers.
#include <math.h>

#include "common.h"

namespace masr {

static constexpr int BS_THREADS = 1024;
static constexpr int BS_WAVES = BS_THREADS / 64;
static constexpr int BS_KMAX = 64;

__device__ __forceinline__ float lse2(float x, float y) {
    if (x == -INFINITY) return y;
    if (y == -INFINITY) return x;
    const float m = fmaxf(x, y);
    return m + logf(expf(x - m) + expf(y - m));
}
__device__ __forceinline__ unsigned okey(float f) {       // larger float -> larger unsigned; never 0 for a real score
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ikey(unsigned k) {       // inverse of okey
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// identity of the prefix string: hash chain over its characters (64 bits; never 0).  The chain step is the scorer's lm_mix (two
// 32-bit multiplies; for a fixed character a bijection of the parent's identity): the step is also taken once per (prefix,
// candidate) pair of a parent with many live children in one version of round 6, where three 64-bit multiplies were the pair's cost
__device__ __forceinline__ unsigned long long str_hash(unsigned long long parent, int ch) {
    const unsigned long long z = lm_mix(parent, (unsigned long long)(ch + 2));
    return z ? z : 1ull;
}
static constexpr int BS_HASH = 1024;

// inclusive wave64 prefix sum with DPP row shifts / row broadcasts (6 VALU ops, no LDS crossbar traffic)
__device__ __forceinline__ int wave_scan_incl(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return v;
}

// wave-wide unsigned max / min, result in every lane (the scan's DPP pattern; lanes without a source keep their own value)
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#define BS_DPP_MAXU(ctrl, rows) v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, (ctrl), (rows), 0xf, false))
    BS_DPP_MAXU(0x111, 0xf); BS_DPP_MAXU(0x112, 0xf); BS_DPP_MAXU(0x114, 0xf); BS_DPP_MAXU(0x118, 0xf);
    BS_DPP_MAXU(0x142, 0xa); BS_DPP_MAXU(0x143, 0xc);
#undef BS_DPP_MAXU
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) { return ~wave_max_u32(~v); }

// exclusive block scan of one int per thread (packed counters); `wsum` must be a scratch row [BS_WAVES] that nobody
// else touches during this step (one barrier per scan); returns the exclusive prefix, total in `total`
__device__ __forceinline__ int block_scan(int v, int* wsum, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int incl = wave_scan_incl(v);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < BS_WAVES; ++w) {
        const int x = wsum[w];
        if (w < wave) base += x;
        tot += x;
    }
    total = tot;
    return base + incl - v;
}

// histogram update with wave aggregation: scores of one frame share their leading bytes, so whole waves usually hit
// one bin -- one atomic per wave instead of a 64-way serialised one
__device__ __forceinline__ void hist_add(int* hist, bool active, int bin) {
    const unsigned long long act = __ballot(active);
    if (!act) return;
    const int leader = __ffsll((long long)act) - 1;
    const int b0 = __builtin_amdgcn_readlane(bin, leader);
    const unsigned long long same = __ballot(active && bin == b0);
    if (same == act) {
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[b0], __popcll(act));
    } else if (active) {
        atomicAdd(&hist[bin], 1);
    }
}

// every wave: find the bin (scanning from 255 down) in which the cumulative count reaches `need`, and how many are
// still needed inside that bin
__device__ __forceinline__ void pick_bin(const int* hist, int need, int& bin, int& rem) {
    const int lane = threadIdx.x & 63;
    int c4[4], s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        c4[j] = hist[255 - (4 * lane + j)];
        s += c4[j];
    }
    const int incl = wave_scan_incl(s);
    const int excl = incl - s;
    const bool mine = excl < need && need <= incl;
    int b = 0, r = 0, acc = excl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (acc < need && need <= acc + c4[j]) {
            b = 255 - (4 * lane + j);
            r = need - acc;
        }
        acc += c4[j];
    }
    const unsigned long long m = __ballot(mine);
    const int leader = m ? __ffsll((long long)m) - 1 : 0;
    bin = __builtin_amdgcn_readlane(b, leader);
    rem = __builtin_amdgcn_readlane(r, leader);
}

// as pick_bin, and the number of keys in the histogram (its 256 bins together)
__device__ __forceinline__ void pick_bin_tot(const int* hist, int need, int& bin, int& rem, int& total, int& binc) {
    const int lane = threadIdx.x & 63;
    int c4[4], s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        c4[j] = hist[255 - (4 * lane + j)];
        s += c4[j];
    }
    const int incl = wave_scan_incl(s);
    const int excl = incl - s;
    const bool mine = excl < need && need <= incl;
    int b = 0, r = 0, bc = 0, acc = excl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (acc < need && need <= acc + c4[j]) {
            b = 255 - (4 * lane + j);
            r = need - acc;
            bc = c4[j];
        }
        acc += c4[j];
    }
    const unsigned long long m = __ballot(mine);
    const int leader = m ? __ffsll((long long)m) - 1 : 0;
    bin = __builtin_amdgcn_readlane(b, leader);
    rem = __builtin_amdgcn_readlane(r, leader);
    binc = __builtin_amdgcn_readlane(bc, leader);
    total = __builtin_amdgcn_readlane(incl, 63);
}

// ---- narrow frames ------------------------------------------------------------------------------------------------------------
// A frame whose extension table has at most NW_ENT entries (live prefixes x candidates of the frame: every frame in which a trained
// model's posterior leaves 1-3 candidates) is a few hundred items of work, and the 1024-thread step above spends its time on what
// every one of its 16 waves executes regardless of the entry count -- scans, bin picks, loop control: ~2 500 instructions per wave
// and frame at four waves per SIMD.  Such a frame is run by the first NW_WAVES waves only (one per SIMD), each thread owning a BLOCK
// of consecutive prefixes / entries (so that index order = thread order and the ordered compaction needs one scan), its keys in
// registers from the extension phase to the compaction (no entry table, no survivor list, no bound stage: one exact 4-pass radix
// select whose first pass also counts the finite keys); the other waves wait at the frame's NW_BARRIERS barriers.  Same arithmetic
// per (prefix, candidate) pair, same selection rule and the same order of the new live set as the wide step: identical transcripts
// and scores (masr_debug_set key 37 = 0 runs every frame on the wide step; tests/test_beam_search.py compares the two).
#ifndef BS_NARROW_WAVES
#define BS_NARROW_WAVES 8
#endif
static constexpr int NW_WAVES = BS_NARROW_WAVES;
static constexpr int NW_T = 64 * NW_WAVES;
static constexpr int NW_EPT = 8;                                    // entries per thread, consecutive
static constexpr int NW_ENT = NW_EPT * NW_T;
static constexpr int NW_PPT = (512 + NW_T - 1) / NW_T;              // live prefixes per thread, consecutive (beam <= 512)
static constexpr int NW_BARRIERS = 11;

// candidate index of character c in the frame's map (-1: not a candidate of this frame)
__device__ __forceinline__ int cand_of(const int* ckmap, int c) {
    unsigned h = ((unsigned)c * 0x9E3779B1u) >> 25;
    while (true) {
        const int v = ckmap[h];
        if (v == 0) return -1;
        if (((v - 1) >> 8) == c) return (v - 1) & 255;
        h = (h + 1) & 127;
    }
}

// NPT = extension entries per thread in the selection phase, strided (beam * K <= 1024 * NPT)
// ORD = 0: no external scorer; 3 | 5: a language model of order <= ORD is bound (bounds the probes per scored extension, which
//       live in registers while a batch of extensions is in flight)
template <int NPT, int ORD>
__global__ __launch_bounds__(BS_THREADS) void beam_search_kernel(BeamGpuArgs a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    constexpr bool use_lm = ORD > 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;      // (shadowed inside the frame loop, see there)
    const int u = blockIdx.x;
    const int beam = a.beam, K = a.K;
    // ---- LDS carve-up --------------------------------------------------------------------------------
    unsigned* ekeys = reinterpret_cast<unsigned*>(smem_raw);             // [beam * K] extension keys, index p * cnt + k
    unsigned long long* lv_hid = reinterpret_cast<unsigned long long*>(ekeys + ((beam * K + 1) & ~1));   // [2][beam] each
    unsigned long long* lv_phid = lv_hid + 2 * beam;
    unsigned long long* hkey = lv_phid + 2 * beam;                       // [BS_HASH] string identity -> live index
    int* hval = reinterpret_cast<int*>(hkey + BS_HASH);                  // [BS_HASH]
    int* lv_node = hval + BS_HASH;
    int* lv_ch = lv_node + 2 * beam;
    float* lv_b = reinterpret_cast<float*>(lv_ch + 2 * beam);
    float* lv_nb = lv_b + 2 * beam;
    float* lv_sc = lv_nb + 2 * beam;
    float* bcur = lv_sc + 2 * beam;                                     // [beam]
    float* rep = bcur + beam;                                           // [beam] repeat term of nb_cur
    float* ext = rep + beam;                                            // [beam] parent-extension term of nb_cur
    int* head = reinterpret_cast<int*>(ext + beam);                      // [beam] narrow step: selected extensions (prefix | candidate << 16), dense
    int* next = head + beam;                                             // [beam] ... and their keys (rounds 3-5: children lists)
    int* nch = next + beam;                                              // [beam] (spare)
    unsigned long long* lv_ctx = reinterpret_cast<unsigned long long*>(nch + beam);   // [2][beam] packed LM context (1024 + 16 beam words behind hkey: 8-byte aligned)
    float* lv_bo = reinterpret_cast<float*>(lv_ctx + 2 * beam);          // [2][beam][4] backoffs of the context's suffixes
    int* lv_m = reinterpret_cast<int*>(lv_bo + 8 * beam);                // [2][beam] m | oov << 8
    // (the scorer's 14 words per live prefix exist only when a language model is bound: an LM-free beam 500 x 40 fits without them)
    int* pu = lv_m + 2 * beam;                                           // [beam] row of the per-frame scorer table of the prefix's context, or -1
    // candidate arrays of a frame, TWO sets: a narrow frame stages the candidates of the next narrow frame while its own are still in use
    int* c_idx0 = use_lm ? pu + beam + (beam & 1) : reinterpret_cast<int*>(lv_ctx);   // [2][BS_KMAX]; bit 30 set = unknown to the language model
                                                                         // (+ beam & 1: the histogram area behind it holds 64-bit keys)
    float* c_lp0 = reinterpret_cast<float*>(c_idx0 + 2 * BS_KMAX);       // [2][BS_KMAX]
    float* c_uni0 = c_lp0 + 2 * BS_KMAX;                                 // [2][BS_KMAX] ln P_LM(candidate) (unigram), per frame
    float* c_ubo0 = c_uni0 + 2 * BS_KMAX;                                // [2][BS_KMAX] ln backoff of the candidate's unigram
    int* hist = reinterpret_cast<int*>(c_ubo0 + 2 * BS_KMAX);            // [7][256]: one per radix pass of a step
    int* wsum = hist + 7 * 256;                                          // [6 + NPT][BS_WAVES]: one scan row per scan of a step
    int* misc = wsum + (6 + NPT) * BS_WAVES;                                         // [8]: 0 blank_k, 1 sel_bin, 2 need
    unsigned short* slist = reinterpret_cast<unsigned short*>(misc + 8 + 32);  // [beam * K] surviving extension entries
    // wide step: which candidates of a prefix are live children (bit k of kidmask[p]: the child computes that extension term itself,
    // the (p, k) pair is skipped) and the frame's map character -> candidate index (open addressing, 128 slots, entry (c << 8 | k) + 1)
    unsigned long long* kidmask = reinterpret_cast<unsigned long long*>(
        reinterpret_cast<unsigned char*>(slist) + (((size_t)beam * K * 2 + 7) & ~(size_t)7));     // [beam]
    int* ckmap = reinterpret_cast<int*>(kidmask + beam);                                           // [128]
    // Scorer table of the frame (use_lm): ln P_LM(c | context) depends on the prefix only through its EFFECTIVE context -- the m most
    // recent words, m = length of the longest suffix that exists in the model -- so the live prefixes' contexts are deduplicated
    // (LDS hash in the histogram area, which the selection phases only need later and which is cleared again behind the extension
    // phase) and every (distinct context, candidate) pair is probed ONCE into lmtab (the survivor list's area, idle until the
    // selection): the 12 000 probes of a frame at beam 300 x 40 candidates become (#contexts x 40).  The values are the ones
    // lm_cond_desc returns for any prefix of that context: identical scores.  masr_debug key 32 = 0 probes per pair (A/B).
    unsigned long long* ckey = reinterpret_cast<unsigned long long*>(hist);       // [512] effective-context keys (0 = free)
    int* cuid = hist + 2 * 512;                                                   // [512] table row of the slot's context
    int* urep = cuid + 512;                                                       // [<= 256] a live prefix that has the context
    float* lmtab = reinterpret_cast<float*>(slist);                               // [ucap][cnt]
    const int ucap = min(256, (beam * K / 2) / max(K, 1));
    const bool lm_cache = use_lm && a.lm_cache != 0;

    int* pool_parent = a.pool_parent + (size_t)u * a.pool_cap;
    int* pool_ch = a.pool_ch + (size_t)u * a.pool_cap;
    int* st_i = a.state_i + (size_t)u * (2 + 3 * beam);
    float* st_f = a.state_f + (size_t)u * (7 * beam);
    unsigned long long* st_h = a.state_h + (size_t)u * (3 * beam);
    int n, pool_count, cur = 0;
    if (a.init) {
        n = 1;
        pool_count = 1;
        if (tid == 0) {
            pool_parent[0] = -1;
            pool_ch[0] = -1;
            lv_node[0] = 0; lv_ch[0] = -1; lv_hid[0] = 0x1234567887654321ull; lv_phid[0] = 0ull;
            lv_b[0] = 0.f; lv_nb[0] = -INFINITY; lv_sc[0] = 0.f;
            if (use_lm) {
                const LmState s0 = lm_state_of(a.lm, lm_root_ctx(a.lm));
                lv_ctx[0] = s0.ctx; lv_m[0] = s0.m | (s0.oov << 8);
                for (int j = 0; j < 4; ++j) lv_bo[j] = s0.bo[j];
            }
        }
    } else {
        n = st_i[0];
        pool_count = st_i[1];
        for (int i = tid; i < n; i += BS_THREADS) {
            lv_node[i] = st_i[2 + i]; lv_ch[i] = st_i[2 + beam + i];
            lv_hid[i] = st_h[i]; lv_phid[i] = st_h[beam + i];
            lv_b[i] = st_f[i]; lv_nb[i] = st_f[beam + i]; lv_sc[i] = st_f[2 * beam + i];
            if (use_lm) {
                lv_ctx[i] = st_h[2 * beam + i]; lv_m[i] = st_i[2 + 2 * beam + i];
                for (int j = 0; j < 4; ++j) lv_bo[4 * i + j] = st_f[(3 + j) * beam + i];
            }
        }
    }
    const unsigned NEG = okey(-INFINITY);
    const int T = a.frames ? min(a.frames[u], a.T_stride) : a.T_stride;
    const int G = BS_THREADS / beam;                  // threads per prefix in the extension phase (>= 2)
    const int my_p0 = tid / G, my_g0 = tid - my_p0 * G;
    // Candidates are fetched TWO frames ahead (nx2_*: registers of threads 0..K-1; the count in every thread), and the scorer's view
    // of a frame's candidates -- known-word flag, unigram entry -- ONE frame ahead (nx_kn, nx_u*: the first probe of the unigram;
    // a collision walks on when the frame is staged), so that no frame starts with a chain of dependent global loads.  The counts
    // come through the VECTOR memory path (vz: a zero the compiler cannot see): a scalar load would be drained by the
    // lgkmcnt(0) in front of every barrier.
    const size_t row0 = (size_t)u * a.T_stride;
    int vz;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
    int nx_cnt = 0, nx_c = 0, nx2_cnt = 0, nx2_c = 0, nx_kn = 0;
    float nx_lp = 0.f, nx_blp = 0.f, nx2_lp = 0.f, nx2_blp = 0.f, nx_up0 = 0.f, nx_ub0 = 0.f;
    unsigned long long nx_uk0 = 0ull;
    const bool cutting = use_lm && a.blank_lp != nullptr;       // the decoder's min_cutoff rule (needs ln p(blank) per frame)
    // frame f of this utterance -> (count, candidate of this lane, ln p(blank))
#define BS_LOAD_FRAME(f, cnt_, c_, lp_, blp_) do {                                                              \
        cnt_ = min(a.ccount[row0 + (f) + vz], K);                                                                \
        if (tid < K) { c_ = a.cidx[(row0 + (f)) * K + tid]; lp_ = a.clp[(row0 + (f)) * K + tid]; }              \
        if (cutting && tid == 0) blp_ = a.blank_lp[row0 + (f)];                                                 \
    } while (0)
    // first probe of the unigram of the candidate in nx_c and its known-word flag (wave 0; results are used one frame later)
#define BS_PROBE_UNIGRAM() do {                                                                                  \
        if (use_lm && wave == 0 && lane < nx_cnt && nx_c >= 0 && nx_c < a.lm.n_words) {                          \
            nx_kn = a.lm.known[nx_c];                                                                            \
            lm_load_entry(a.lm, lm_key(0ull, 0, nx_c) & a.lm.mask, &nx_uk0, &nx_up0, &nx_ub0);                   \
        }                                                                                                        \
    } while (0)
    // stage the frame in nx_* (wave 0): candidates, blank index, scorer flags / unigrams -> LDS; then advance the pipeline
#define BS_STAGE_FRAME(ci_, cl_, cu_, cb_, cntv_, tt_) do {                                                      \
        const int cnt_ = (cntv_);                                                                                \
        if (wave == 0) {                                                                                         \
            const bool isb = lane < cnt_ && nx_c == a.blank;                                                     \
            const unsigned long long bm = __ballot(isb);                                                         \
            ckmap[lane] = 0; ckmap[64 + lane] = 0;        /* character -> candidate index (one wave: LDS operations in order) */ \
            if (lane < cnt_) {                                                                                   \
                unsigned hq = ((unsigned)nx_c * 0x9E3779B1u) >> 25;                                              \
                while (atomicCAS(&ckmap[hq], 0, ((nx_c << 8) | lane) + 1) != 0) hq = (hq + 1) & 127;             \
            }                                                                                                    \
            if (lane < cnt_) {                                                                                   \
                /* the scorer's known-word flag of this candidate rides in bit 30 of its index: one lookup per frame and */ \
                /* candidate instead of one per (prefix, candidate) pair */                                      \
                const bool unk = use_lm && !(nx_c >= 0 && nx_c < a.lm.n_words && nx_kn);                         \
                (ci_)[lane] = nx_c | (unk ? (1 << 30) : 0);                                                      \
                (cl_)[lane] = nx_lp;                                                                             \
                if (use_lm) {                             /* the candidate's unigram: one lookup per frame and candidate */ \
                    float up = LM_OOV_SCORE, ub = 0.f;                                                           \
                    if (!unk) {                                                                                  \
                        const unsigned long long key = lm_key(0ull, 0, nx_c);                                    \
                        if (nx_uk0 == key) { up = nx_up0; ub = nx_ub0; }                                         \
                        else if (nx_uk0 != 0ull) lm_find(a.lm, key, &up, &ub);                                   \
                    }                                                                                            \
                    (cu_)[lane] = up;                                                                            \
                    (cb_)[lane] = ub;                                                                            \
                }                                                                                                \
            }                                                                                                    \
            if (lane == 0) {                                                                                     \
                misc[0] = bm ? __ffsll((long long)bm) - 1 : -1;                                                  \
                misc[1] = 0; misc[2] = (int)0xFFFFFFFFu;      /* wide step: max / min over the frame's finite keys */ \
                misc[3] = (int)0xFFFFFFFFu;                  /* min over the live prefixes' score keys */         \
                misc[4] = __float_as_int(nx_blp);                                                                \
                misc[5] = 0;                                 /* distinct effective scorer contexts of this frame */ \
            }                                                                                                    \
        }                                                                                                        \
        nx_cnt = nx2_cnt; nx_c = nx2_c; nx_lp = nx2_lp; nx_blp = nx2_blp;                                        \
        if ((tt_) + 1 < T) BS_PROBE_UNIGRAM();                                                                   \
        if ((tt_) + 2 < T) BS_LOAD_FRAME((tt_) + 2, nx2_cnt, nx2_c, nx2_lp, nx2_blp);                            \
    } while (0)
    if (T > 0) {
        BS_LOAD_FRAME(0, nx_cnt, nx_c, nx_lp, nx_blp);
        BS_PROBE_UNIGRAM();
        if (T > 1) BS_LOAD_FRAME(1, nx2_cnt, nx2_c, nx2_lp, nx2_blp);
    }
    const bool prof = a.prof && u == (int)gridDim.x - 1 && tid == 0;      // (the LAST workgroup: the longest utterance of a length-sorted pass)
    // phase counters of workgroup 0 live in LDS (16 x 64 bit behind misc): 0-4 wide step, 5 narrow frames, 6-13 narrow step
    unsigned long long* pcl = reinterpret_cast<unsigned long long*>(misc + 8);
    if (tid < 16) pcl[tid] = 0ull;
#define BS_TICK(i) do { if (prof) { const long long now_ = clock64(); pcl[i] += (unsigned long long)(now_ - tick_); tick_ = now_; } } while (0)
    long long tick_ = prof ? clock64() : 0;
    __syncthreads();

    bool prepared = false;
    int staged_cnt = 0, cpar = 0;
    for (int t = 0; t < T; ++t) {
        // The thread's indices are made opaque once per frame: every LDS address of the step is then recomputed from them (one or
        // two VALU instructions) instead of being hoisted out of the frame loop, kept alive across all its phases and SPILLED -- a
        // scratch reload waits with vmcnt(0), i.e. for every prefetch in flight as well (18-40 spilled VGPRs before this).
        int tid_ = (int)threadIdx.x, my_p_ = my_p0, my_g_ = my_g0;
        asm volatile("" : "+v"(tid_), "+v"(my_p_), "+v"(my_g_));
        const int tid = tid_, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int my_p = my_p_, my_g = my_g_;
        // `prepared`: the previous (narrow) frame already cleared this frame's tables and staged its candidates into the other set
        const int cnt = prepared ? staged_cnt : nx_cnt;
        int* c_idx = c_idx0 + cpar * BS_KMAX;
        float* c_lp = c_lp0 + cpar * BS_KMAX;
        float* c_uni = c_uni0 + cpar * BS_KMAX;
        float* c_ubo = c_ubo0 + cpar * BS_KMAX;
        const int o = cur * beam, o2 = (cur ^ 1) * beam;
        if ((a.narrow & 1) && n * cnt <= NW_ENT) {
            // ================= narrow frame: waves 0 .. NW_WAVES-1 work, the others wait ======================================
            // scorer-state table of the frame: next to ln P_LM(c | context) the table fill records m and the backoffs of the context
            // EXTENDED by c (lm_cond_next: they come out of the same probes), so a surviving extension takes its scorer state from
            // LDS instead of probing the model again (lm_state_of: 2 - 4 more dependent loads per new prefix, at the end of the
            // frame's critical path).  STW words per (context, candidate) in the wide step's entry-key area, idle in narrow frames.
            // ALL sixteen waves fill it (the waiting ones wake up between barriers 2 and 3): one probe round trip for up to 1024
            // (context, candidate) pairs.
            constexpr int STW = ORD > 1 ? ORD - 1 : 1;
            float* lmst = reinterpret_cast<float*>(ekeys);
            const int ucap_n = min(ucap, (beam * K) / (STW * max(cnt, 1)));
            auto fill_scorer_table = [&]() {
                const int nu = min(misc[5], ucap_n);
                for (int e = tid; e < nu * cnt; e += BS_THREADS) {
                    const int uq = e / cnt, k = e - uq * cnt;
                    const int craw = c_idx[k];
                    float v = LM_OOV_SCORE;
                    if (!(craw >> 30)) {
                        const int p = urep[uq];
                        LmState sp;
                        sp.ctx = lv_ctx[o + p];
                        sp.m = lv_m[o + p] & 255;
                        sp.oov = 0;
#pragma unroll
                        for (int q = 0; q < 4; ++q) sp.bo[q] = lv_bo[4 * (o + p) + q];
                        int m2;
                        float bo2[STW];
                        v = lm_cond_next<(ORD > 0 ? ORD : 1)>(a.lm, sp, craw, c_uni[k], c_ubo[k], &m2, bo2);
                        lmst[e * STW] = __int_as_float(m2);
#pragma unroll
                        for (int q = 1; q < STW; ++q) lmst[e * STW + q] = bo2[q];
                    }
                    lmtab[e] = v;
                }
            };
            if (wave >= NW_WAVES) {
                // (the waiting waves mirror the working waves' count pipeline: a frame is staged once, by THIS frame's N0 or by the
                //  tail of the previous narrow frame)
                if (!prepared) {
                    nx_cnt = nx2_cnt;
                    if (t + 2 < T) nx2_cnt = min(a.ccount[row0 + t + 2 + vz], K);
                    __syncthreads();                                                                          // (1)
                }
                __syncthreads();                                                                              // (2)
                if (lm_cache) fill_scorer_table();
                for (int b = 2; b < NW_BARRIERS; ++b) __syncthreads();
                n = misc[6];
                pool_count = misc[7];
                cur ^= 1;
                prepared = t + 1 < T && n * nx_cnt <= NW_ENT;
                if (prepared) {
                    staged_cnt = nx_cnt;
                    nx_cnt = nx2_cnt;
                    if (t + 3 < T) nx2_cnt = min(a.ccount[row0 + t + 3 + vz], K);
                    cpar ^= 1;
                }
                continue;
            }
            if (prof) ++pcl[5];
            // ---- N0. tables, candidates, prefetch ------------------------------------------------------------------------
            if (!prepared) {
                for (int i = tid; i < BS_HASH; i += NW_T) hkey[i] = 0ull;
                if (lm_cache)
                    for (int i = tid; i < 512; i += NW_T) ckey[i] = 0ull;
                BS_STAGE_FRAME(c_idx, c_lp, c_uni, c_ubo, cnt, t);
                __syncthreads();                                                                              // (1)
            }
            BS_TICK(6);
            // ---- N1. string identities -> hash, scorer contexts -> rows, worst live score ------------------------------------
            const int blank_k = misc[0];
            const int i0 = NW_PPT * tid;                  // this thread's live prefixes: i0 .. i0 + NW_PPT - 1
            unsigned long long my_ek[NW_PPT];
            bool owner[NW_PPT];
            int own_slot[NW_PPT];
#pragma unroll
            for (int j = 0; j < NW_PPT; ++j) { owner[j] = false; own_slot[j] = 0; }
            unsigned kmin = 0xFFFFFFFFu;
#pragma unroll
            for (int j = 0; j < NW_PPT; ++j) {
                const int i = i0 + j;
                my_ek[j] = 0ull;
                if (i < n) {
                    rep[i] = -INFINITY; ext[i] = -INFINITY; kidmask[i] = 0ull;       // (first touched behind barrier 2)
                    const unsigned long long key = lv_hid[o + i];
                    unsigned h = (unsigned)(key >> 40) & (BS_HASH - 1);
                    while (atomicCAS(&hkey[h], 0ull, key) != 0ull) h = (h + 1) & (BS_HASH - 1);
                    hval[h] = i;
                    kmin = min(kmin, okey(lv_sc[o + i]));
                    if (lm_cache) {
                        const int mo = lv_m[o + i], m = mo & 255;
                        const unsigned long long ctx = lv_ctx[o + i];
                        unsigned long long ek = 0ull;
                        if (!(mo >> 8) && m >= 1) {
                            if (m < 4) ek = (ctx & ((1ull << (16 * m)) - 1ull)) | ((unsigned long long)m << 60);
                            else if (!((ctx >> 48) == 0x1000ull || (ctx >> 48) == 0x2000ull || (ctx >> 48) == 0x3000ull)) ek = ctx;
                        }
                        my_ek[j] = ek;
                        if (ek != 0ull) {
                            unsigned hc = (unsigned)((ek * 0x9E3779B97F4A7C15ull) >> 55);        // 9 bits
                            while (true) {
                                const unsigned long long old = atomicCAS(&ckey[hc], 0ull, ek);
                                if (old == 0ull) {       // first prefix with this context: it owns the table row
                                    owner[j] = true;
                                    own_slot[j] = (int)hc;
                                    break;
                                }
                                if (old == ek) break;
                                hc = (hc + 1) & 511;
                            }
                        }
                    }
                }
            }
            if (lm_cache) {
                // table rows of the new contexts: ONE counter update per wave (the owners' ranks come from the ballot) instead of one
                // returning atomic per context on a single LDS word
#pragma unroll
                for (int j = 0; j < NW_PPT; ++j) {
                    const unsigned long long om = __ballot(owner[j]);
                    if (om) {
                        const int leader = __ffsll((long long)om) - 1;
                        int base = 0;
                        if (lane == leader) base = atomicAdd(&misc[5], __popcll(om));
                        base = __builtin_amdgcn_readlane(base, leader);
                        if (owner[j]) {
                            const int uid = base + __popcll(om & ((1ull << lane) - 1ull));
                            cuid[own_slot[j]] = uid < ucap_n ? uid : -1;
                            if (uid < ucap_n) urep[uid] = i0 + j;
                        }
                    }
                }
            }
            const bool full_beam = cutting && n == beam;
            if (full_beam) {
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) kmin = min(kmin, (unsigned)__shfl_xor((int)kmin, off, 64));
                if (lane == 0) atomicMin(reinterpret_cast<unsigned*>(&misc[3]), kmin);
            }
            __syncthreads();                                                                                  // (2)
            BS_TICK(7);
            // ---- N2. blank term, live children of live parents, scorer table ------------------------------------------------
            int parP[NW_PPT], kqP[NW_PPT];
#pragma unroll
            for (int j = 0; j < NW_PPT; ++j) { parP[j] = -1; kqP[j] = -1; }
            const float min_cut = full_beam ? (float)((double)ikey((unsigned)misc[3]) + (double)__int_as_float(misc[4]) -
                                                      fmax(0.0, (double)a.beta))
                                            : -INFINITY;
#pragma unroll
            for (int j = 0; j < NW_PPT; ++j) {
                const int i = i0 + j;
                if (i < n) {
                    {
                        const float lpb = blank_k >= 0 ? c_lp[blank_k] : -INFINITY, scb = lv_sc[o + i];
                        bcur[i] = blank_k >= 0 && !(lpb + scb < min_cut) ? lpb + scb : -INFINITY;
                    }
                    const unsigned long long key = lv_phid[o + i];
                    unsigned h = (unsigned)(key >> 40) & (BS_HASH - 1);
                    int par = -1;
                    while (key != 0ull) {
                        const unsigned long long hk = hkey[h];
                        if (hk == key) { par = hval[h]; break; }
                        if (hk == 0ull) break;
                        h = (h + 1) & (BS_HASH - 1);
                    }
                    if (par >= 0) {                      // my parent is live: am I one of its candidates in this frame?
                        const int kq = cand_of(ckmap, lv_ch[o + i]);
                        if (kq >= 0) {
                            atomicOr(&kidmask[par], 1ull << kq);
                            parP[j] = par;
                            kqP[j] = kq;
                        }
                    }
                    if (lm_cache) {
                        int row = -1;
                        if (my_ek[j] != 0ull) {
                            unsigned hc = (unsigned)((my_ek[j] * 0x9E3779B97F4A7C15ull) >> 55);
                            while (ckey[hc] != my_ek[j]) hc = (hc + 1) & 511;
                            row = cuid[hc];
                        }
                        pu[i] = row;
                    }
                }
            }
            if (lm_cache) fill_scorer_table();
            __syncthreads();                                                                                  // (3)
            BS_TICK(8);
            // ---- N3. extensions: NW_EPT consecutive entries e = p * cnt + k per thread, keys stay in registers ---------------
            const int nq = n * cnt;
            const int eptf = (nq + NW_T - 1) / NW_T;      // block of consecutive entries per thread in THIS frame: 1 .. NW_EPT
#pragma unroll
            for (int j = 0; j < NW_PPT; ++j)
                if (kqP[j] >= 0) {                        // a live child of a live parent takes the parent's extension term itself
                    const int p = parP[j], k = kqP[j];
                    const float scp = lv_sc[o + p], pbp = lv_b[o + p];
                    const int craw = c_idx[k];
                    const int c = craw & ~(1 << 30);
                    const float lp = c_lp[k];
                    float val = -INFINITY;
                    if (c != a.blank && !(lp + scp < min_cut)) {
                        if (c == lv_ch[o + p]) val = pbp > -INFINITY ? lp + pbp : -INFINITY;
                        else val = lp + scp;
                        if (use_lm && val > -INFINITY) {
                            const int mo = lv_m[o + p];
                            const int rowp = lm_cache ? pu[p] : -1;
                            float lmp;
                            if ((mo >> 8) || (craw >> 30)) lmp = LM_OOV_SCORE;
                            else if (rowp >= 0) lmp = lmtab[rowp * cnt + k];
                            else {
                                LmState spp;
                                spp.ctx = lv_ctx[o + p];
                                spp.m = mo & 255; spp.oov = 0;
#pragma unroll
                                for (int q = 0; q < 4; ++q) spp.bo[q] = lv_bo[4 * (o + p) + q];
                                lmp = lm_cond_desc<(ORD > 0 ? ORD : 1)>(a.lm, spp, c, c_uni[k]);
                            }
                            val += a.alpha * lmp + a.beta;
                        }
                    }
                    ext[i0 + j] = val;
                }
            unsigned keyE[NW_EPT];
            int peE[NW_EPT];                              // p | k << 16 of the entry
            {
                const int e0 = eptf * tid;
                int p = 0, k = 0;
                if (e0 < nq) {
                    p = (int)(((float)e0 + 0.5f) * (1.0f / (float)cnt));      // e0 / cnt (exact: e0 < 4096, cnt <= 64)
                    k = e0 - p * cnt;
                }
#pragma unroll
                for (int j = 0; j < NW_EPT; ++j) {
                    keyE[j] = 0u;
                    peE[j] = p | (k << 16);
                    if (j < eptf && e0 + j < nq) {
                        const float sc = lv_sc[o + p];
                        const int craw = c_idx[k];
                        const int c = craw & ~(1 << 30);
                        const float lp = c_lp[k];
                        float val = -INFINITY;
                        if (c != a.blank && !(lp + sc < min_cut)) {
                            const float pb = lv_b[o + p];
                            if (c == lv_ch[o + p]) {
                                rep[p] = lp + lv_nb[o + p];
                                val = pb > -INFINITY ? lp + pb : -INFINITY;
                            } else {
                                val = lp + sc;
                            }
                            if ((kidmask[p] >> k) & 1ull) {
                                val = -INFINITY;       // the child (p, c) is a live prefix: it took this term itself (above)
                            } else if (use_lm && val > -INFINITY) {
                                const int mo = lv_m[o + p];
                                const int my_row = lm_cache ? pu[p] : -1;
                                float lmp;
                                if ((mo >> 8) || (craw >> 30)) lmp = LM_OOV_SCORE;
                                else if (my_row >= 0) lmp = lmtab[my_row * cnt + k];
                                else {
                                    LmState sp;
                                    sp.ctx = lv_ctx[o + p];
                                    sp.m = mo & 255; sp.oov = 0;
#pragma unroll
                                    for (int q = 0; q < 4; ++q) sp.bo[q] = lv_bo[4 * (o + p) + q];
                                    lmp = lm_cond_desc<(ORD > 0 ? ORD : 1)>(a.lm, sp, c, c_uni[k]);
                                }
                                val += a.alpha * lmp + a.beta;
                            }
                        }
                        const unsigned kk = okey(val);
                        keyE[j] = kk > NEG ? kk : 0u;             // 0: no entry / -inf
                        if (++k == cnt) { k = 0; ++p; }
                    }
                }
            }
            for (int i = tid; i < 512; i += NW_T) reinterpret_cast<int2*>(hist)[i] = make_int2(0, 0);   // 4 tables for the select
            __syncthreads();                                                                                  // (4)
            BS_TICK(9);
            // ---- N4. the prefixes themselves -------------------------------------------------------------------------------------
            unsigned kxP[NW_PPT];
            float nbP[NW_PPT], scP[NW_PPT];
#pragma unroll
            for (int j = 0; j < NW_PPT; ++j) {
                const int i = i0 + j;
                kxP[j] = 0u; nbP[j] = -INFINITY; scP[j] = -INFINITY;
                if (i < n) {
                    nbP[j] = lse2(rep[i], ext[i]);
                    scP[j] = lse2(bcur[i], nbP[j]);
                    const unsigned kk = okey(scP[j]);
                    kxP[j] = kk > NEG ? kk : 0u;
                }
            }
            // ---- N5. exact radix select over the finite keys; pass 0 also counts them ------------------------------------------
            unsigned thr = NEG;
            int need_eq = 0;
            {
                // a pass whose picked bin is needed WHOLE ends the selection: every key at or above the bin stays, no tie to break
                // (the later passes are then bare barriers; with scores ~1e-2 apart that is the third pass, often the second)
                unsigned prefix = 0, mask = 0;
                int need = beam;
                bool open = true, exact = true;
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    const int shift = 24 - 8 * pass;
                    int* hp = hist + pass * 256;
                    if (open) {
#pragma unroll
                        for (int j = 0; j < NW_PPT; ++j) hist_add(hp, kxP[j] != 0 && (kxP[j] & mask) == prefix, (kxP[j] >> shift) & 255);
#pragma unroll
                        for (int j = 0; j < NW_EPT; ++j)
                            if (j < eptf) hist_add(hp, keyE[j] != 0 && (keyE[j] & mask) == prefix, (keyE[j] >> shift) & 255);
                    }
                    __syncthreads();                                                                          // (5) .. (8)
                    if (open) {
                        int bin, rem, total, binc;
                        pick_bin_tot(hp, need, bin, rem, total, binc);
                        if (pass == 0 && total <= beam) {             // nothing to drop: every finite key stays (thr = NEG)
                            open = false;
                            exact = false;
                        } else {
                            prefix |= (unsigned)bin << shift;
                            mask |= 255u << shift;
                            need = rem;
                            if (rem == binc && pass < 3) {
                                thr = prefix - 1u;
                                open = false;
                                exact = false;
                            }
                        }
                    }
                }
                if (exact) { thr = prefix; need_eq = need; }
            }
            BS_TICK(10);
            // ---- N6. ordered compaction: live prefixes first (by index), then the extensions (by entry index) ---------------
            int vP = 0, vE = 0;
#pragma unroll
            for (int j = 0; j < NW_PPT; ++j) vP += (kxP[j] > thr) | ((kxP[j] == thr && kxP[j] != 0) << 16);
#pragma unroll
            for (int j = 0; j < NW_EPT; ++j) vE += (keyE[j] > thr) | ((keyE[j] == thr && keyE[j] != 0) << 16);
            int exP, exE, totP = 0, totE = 0;
            {
                const int inP = wave_scan_incl(vP), inE = wave_scan_incl(vE);
                if (lane == 63) { wsum[wave] = inP; wsum[BS_WAVES + wave] = inE; }
                __syncthreads();                                                                              // (9)
                BS_TICK(11);
                int bP = 0, bE = 0;
#pragma unroll
                for (int w = 0; w < NW_WAVES; ++w) {
                    const int xp = wsum[w], xe = wsum[BS_WAVES + w];
                    if (w < wave) { bP += xp; bE += xe; }
                    totP += xp; totE += xe;
                }
                exP = bP + inP - vP;
                exE = bE + inE - vE;
            }
            const int eq_take_P = min(totP >> 16, need_eq);
            const int n_exist = (totP & 0xffff) + eq_take_P;
            const int need_eq_new = need_eq - eq_take_P;
            const int n_new = (totE & 0xffff) + min(totE >> 16, need_eq_new);
            {
                int gb = exP & 0xffff, eb = exP >> 16;
#pragma unroll
                for (int j = 0; j < NW_PPT; ++j) {
                    const unsigned kk = kxP[j];
                    const int i = i0 + j;
                    const bool gt = kk > thr, eq = kk == thr && kk != 0;
                    if (gt || (eq && eb < need_eq)) {
                        const int slot = gb + min(eb, need_eq);
                        lv_node[o2 + slot] = lv_node[o + i];
                        lv_hid[o2 + slot] = lv_hid[o + i];
                        lv_phid[o2 + slot] = lv_phid[o + i];
                        lv_ch[o2 + slot] = lv_ch[o + i];
                        lv_b[o2 + slot] = bcur[i];
                        lv_nb[o2 + slot] = nbP[j];
                        lv_sc[o2 + slot] = scP[j];
                        if (use_lm) {
                            lv_ctx[o2 + slot] = lv_ctx[o + i]; lv_m[o2 + slot] = lv_m[o + i];
#pragma unroll
                            for (int q = 0; q < 4; ++q) lv_bo[4 * (o2 + slot) + q] = lv_bo[4 * (o + i) + q];
                        }
                    }
                    gb += gt; eb += eq;
                }
            }
            {
                // the selected extensions, in entry order, as a dense list: the new prefixes are then built one per thread instead of
                // one divergent pass per block position
                int* sel_pe = head;                                     // (the children lists are dead behind the extension phase;
                unsigned* sel_key = reinterpret_cast<unsigned*>(next);  //  the select's histogram area is cleared for the next frame below)
                int gb = exE & 0xffff, eb = exE >> 16;
#pragma unroll
                for (int j = 0; j < NW_EPT; ++j) {
                    const unsigned kk = keyE[j];
                    const bool gt = kk > thr, eq = kk == thr && kk != 0;
                    if (gt || (eq && eb < need_eq_new)) {
                        const int r = gb + min(eb, need_eq_new);            // rank among the selected extensions
                        sel_pe[r] = peE[j];
                        sel_key[r] = kk;
                    }
                    gb += gt; eb += eq;
                }
                __syncthreads();                                                                              // (10)
                BS_TICK(12);
                for (int r = tid; r < n_new; r += NW_T) {
                    const unsigned kk = sel_key[r];
                    const int slot = n_exist + r, node = pool_count + r;
                    const int p = sel_pe[r] & 0xffff, k = sel_pe[r] >> 16;
                    const int c = c_idx[k] & ~(1 << 30);
                    const float add = ikey(kk);           // the entry's own score (acoustic term + scorer term), as ranked
                    if (node < a.pool_cap) {
                        pool_parent[node] = lv_node[o + p];
                        pool_ch[node] = c;
                    }
                    lv_node[o2 + slot] = node;
                    lv_hid[o2 + slot] = str_hash(lv_hid[o + p], c);
                    lv_phid[o2 + slot] = lv_hid[o + p];
                    lv_ch[o2 + slot] = c;
                    lv_b[o2 + slot] = -INFINITY;
                    lv_nb[o2 + slot] = add;
                    lv_sc[o2 + slot] = add;
                    if (use_lm) {
                        const int row = lm_cache ? pu[p] : -1;
                        if (row >= 0 && !(c_idx[k] >> 30)) {      // (a row: the parent's context has no unknown word)
                            const float* st = lmst + (row * cnt + k) * STW;
                            lv_ctx[o2 + slot] = lm_push(lv_ctx[o + p], c);
                            lv_m[o2 + slot] = __float_as_int(st[0]);
                            lv_bo[4 * (o2 + slot)] = c_ubo[k];
#pragma unroll
                            for (int q = 1; q < 4; ++q) lv_bo[4 * (o2 + slot) + q] = q < STW ? st[q] : 0.f;
                        } else {
                            const LmState sn = lm_state_of(a.lm, lm_push(lv_ctx[o + p], c));
                            lv_ctx[o2 + slot] = sn.ctx; lv_m[o2 + slot] = sn.m | (sn.oov << 8);
#pragma unroll
                            for (int q = 0; q < 4; ++q) lv_bo[4 * (o2 + slot) + q] = sn.bo[q];
                        }
                    }
                }
            }
            pool_count += n_new;
            n = n_exist + n_new;
            cur ^= 1;
            if (tid == 0) { misc[6] = n; misc[7] = pool_count; }
            // the next frame is narrow too: its tables are cleared and its candidates staged HERE (into the other candidate set;
            // the string hash, the context hash and the select's tables are dead by now), so it starts at N1 -- one barrier and
            // the staging's latency less per frame
            prepared = t + 1 < T && n * nx_cnt <= NW_ENT;
            if (prepared) {
                for (int i = tid; i < BS_HASH; i += NW_T) hkey[i] = 0ull;
                if (lm_cache)
                    for (int i = tid; i < 512; i += NW_T) ckey[i] = 0ull;
                staged_cnt = nx_cnt;
                cpar ^= 1;
                BS_STAGE_FRAME(c_idx0 + cpar * BS_KMAX, c_lp0 + cpar * BS_KMAX, c_uni0 + cpar * BS_KMAX, c_ubo0 + cpar * BS_KMAX,
                               staged_cnt, t + 1);
            }
            __syncthreads();                                                                                  // (11)
            BS_TICK(13);
            continue;
        }
        // ---- 0. candidates -> LDS, clear tables; prefetch the next frame -------------------------------------
        for (int i = tid; i < 7 * 256; i += BS_THREADS) hist[i] = 0;
        if (tid < n) { rep[tid] = -INFINITY; ext[tid] = -INFINITY; kidmask[tid] = 0ull; }
        if (tid < BS_HASH) hkey[tid] = 0ull;
        BS_STAGE_FRAME(c_idx, c_lp, c_uni, c_ubo, cnt, t);
        prepared = false;
        __syncthreads();
        // ---- 1. live children lists; blank term ------------------------------------------------------------------
        const int blank_k = misc[0];
        if (tid < n) {
            const unsigned long long key = lv_hid[o + tid];
            unsigned h = (unsigned)(key >> 40) & (BS_HASH - 1);
            while (atomicCAS(&hkey[h], 0ull, key) != 0ull) h = (h + 1) & (BS_HASH - 1);
            hval[h] = tid;
        }
        unsigned long long my_ek = 0ull;                 // effective scorer context of live prefix tid (0: not cached)
        if (lm_cache && tid < n) {
            const int mo = lv_m[o + tid], m = mo & 255;
            const unsigned long long ctx = lv_ctx[o + tid];
            if (!(mo >> 8) && m >= 1) {                  // (an OOV context scores a constant, m = 0 the unigram: no probes either way)
                if (m < 4) my_ek = (ctx & ((1ull << (16 * m)) - 1ull)) | ((unsigned long long)m << 60);
                else if (!((ctx >> 48) == 0x1000ull || (ctx >> 48) == 0x2000ull || (ctx >> 48) == 0x3000ull)) my_ek = ctx;
            }
            if (my_ek != 0ull) {
                unsigned h = (unsigned)((my_ek * 0x9E3779B97F4A7C15ull) >> 55);        // 9 bits
                while (true) {
                    const unsigned long long old = atomicCAS(&ckey[h], 0ull, my_ek);
                    if (old == 0ull) {                   // first prefix with this context: it owns the table row
                        const int uid = atomicAdd(&misc[5], 1);
                        cuid[h] = uid < ucap ? uid : -1;
                        if (uid < ucap) urep[uid] = tid;
                        break;
                    }
                    if (old == my_ek) break;
                    h = (h + 1) & 511;
                }
            }
        }
        const bool full_beam = cutting && n == beam;
        if (full_beam && tid < ((n + 63) & ~63)) {      // score of the worst live prefix: wave minimum, one LDS atomic per wave
            unsigned k = tid < n ? okey(lv_sc[o + tid]) : 0xFFFFFFFFu;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) k = min(k, (unsigned)__shfl_xor((int)k, off, 64));
            if (lane == 0) atomicMin(reinterpret_cast<unsigned*>(&misc[3]), k);
        }
        __syncthreads();
        // min_cutoff in the host search's arithmetic (double, rounded once)
        const float min_cut = full_beam ? (float)((double)ikey((unsigned)misc[3]) + (double)__int_as_float(misc[4]) -
                                                  fmax(0.0, (double)a.beta))
                                        : -INFINITY;
        int my_par = -1, my_kq = -1;                    // this live prefix as the child (my_par, candidate my_kq) of a live parent
        if (tid < n) {                                  // is my parent prefix live, and am I one of its candidates in this frame?
            {
                const float lpb = blank_k >= 0 ? c_lp[blank_k] : -INFINITY, scb = lv_sc[o + tid];
                bcur[tid] = blank_k >= 0 && !(lpb + scb < min_cut) ? lpb + scb : -INFINITY;
            }
            const unsigned long long key = lv_phid[o + tid];
            unsigned h = (unsigned)(key >> 40) & (BS_HASH - 1);
            int par = -1;
            while (key != 0ull) {
                const unsigned long long hk = hkey[h];
                if (hk == key) { par = hval[h]; break; }
                if (hk == 0ull) break;
                h = (h + 1) & (BS_HASH - 1);
            }
            if (par >= 0) {                              // my parent is live: am I one of its candidates in this frame?
                const int kq = cand_of(ckmap, lv_ch[o + tid]);
                if (kq >= 0) {
                    atomicOr(&kidmask[par], 1ull << kq);
                    my_par = par;
                    my_kq = kq;
                }
            }
        }
        if (lm_cache) {
            if (tid < n) {                               // my context's table row
                int row = -1;
                if (my_ek != 0ull) {
                    unsigned h = (unsigned)((my_ek * 0x9E3779B97F4A7C15ull) >> 55);
                    while (ckey[h] != my_ek) h = (h + 1) & 511;
                    row = cuid[h];
                }
                pu[tid] = row;
            }
            const int nu = min(misc[5], ucap);
            if (prof) { pcl[14] += (unsigned long long)misc[5]; pcl[15] += 1ull; }      // distinct scorer contexts of the wide frames / wide frames
            for (int e = tid; e < nu * cnt; e += BS_THREADS) {
                const int uq = e / cnt, k = e - uq * cnt;
                const int craw = c_idx[k];
                float v = LM_OOV_SCORE;
                if (!(craw >> 30)) {
                    const int p = urep[uq];
                    LmState sp;
                    sp.ctx = lv_ctx[o + p];
                    sp.m = lv_m[o + p] & 255;
                    sp.oov = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) sp.bo[j] = lv_bo[4 * (o + p) + j];
                    // (all suffix lengths probed at once: one memory round trip per table entry instead of up to ORD - 1 -- the
                    //  table fill is a chain of dependent loads in front of the frame's extension phase)
                    int m2;
                    float bo2[ORD > 1 ? ORD - 1 : 1];
                    v = lm_cond_next<(ORD > 0 ? ORD : 1)>(a.lm, sp, craw, c_uni[k], 0.f, &m2, bo2);
                }
                lmtab[e] = v;
            }
        }
        __syncthreads();
        BS_TICK(0);
        // ---- 2. extensions (p, c): G threads per prefix, each a strided set of candidates ------------------------
        const int nq = n * cnt;
        if (my_kq >= 0) {
            // A live prefix whose parent is live takes the parent's extension by its own last character ITSELF (the arithmetic of the
            // (parent, candidate) pair below, on the parent's state): the pair is then skipped through the parent's kidmask bit.
            // Rounds 3-5 let every pair search the parent's children list -- hundreds of siblings under flat posteriors.
            const int p = my_par, k = my_kq;
            const float scp = lv_sc[o + p], pbp = lv_b[o + p];
            const int craw = c_idx[k];
            const int c = craw & ~(1 << 30);
            const float lp = c_lp[k];
            float val = -INFINITY;
            if (c != a.blank && !(lp + scp < min_cut)) {
                if (c == lv_ch[o + p]) val = pbp > -INFINITY ? lp + pbp : -INFINITY;
                else val = lp + scp;
                if (use_lm && val > -INFINITY) {
                    const int mo = lv_m[o + p];
                    const int rowp = lm_cache ? pu[p] : -1;
                    float lmp;
                    if ((mo >> 8) || (craw >> 30)) lmp = LM_OOV_SCORE;
                    else if (rowp >= 0) lmp = lmtab[rowp * cnt + k];
                    else {
                        LmState spp;
                        spp.ctx = lv_ctx[o + p];
                        spp.m = mo & 255; spp.oov = 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) spp.bo[j] = lv_bo[4 * (o + p) + j];
                        lmp = lm_cond_desc<(ORD > 0 ? ORD : 1)>(a.lm, spp, c, c_uni[k]);
                    }
                    val += a.alpha * lmp + a.beta;
                }
            }
            ext[tid] = val;
        }
        if (my_p < n) {
            const float sc = lv_sc[o + my_p], pb = lv_b[o + my_p], pnb = lv_nb[o + my_p];
            const int ch = lv_ch[o + my_p];
            const unsigned long long kids = kidmask[my_p];
            LmState sp;
            if (use_lm) {
                sp.ctx = lv_ctx[o + my_p];
                const int mo = lv_m[o + my_p];
                sp.m = mo & 255; sp.oov = mo >> 8;
#pragma unroll
                for (int j = 0; j < 4; ++j) sp.bo[j] = lv_bo[4 * (o + my_p) + j];
            }
            const int my_row = lm_cache ? pu[my_p] : -1;
            for (int k = my_g; k < cnt; k += G) {
                const int craw = c_idx[k];
                const int c = craw & ~(1 << 30);
                const float lp = c_lp[k];
                float val = -INFINITY;
                if (c != a.blank && !(lp + sc < min_cut)) {
                    if (c == ch) {
                        rep[my_p] = lp + pnb;                       // only this k repeats the last character of p
                        val = pb > -INFINITY ? lp + pb : -INFINITY;
                    } else {
                        val = lp + sc;
                    }
                    if ((kids >> k) & 1ull) {
                        val = -INFINITY;                            // the child (p, c) is a live prefix: it took this term itself (above)
                    } else if (use_lm && val > -INFINITY) {
                        // the external scorer: every way into the prefix p + c carries the same alpha * ln P_LM + beta
                        const float lmp = (sp.oov || (craw >> 30)) ? LM_OOV_SCORE
                                          : my_row >= 0            ? lmtab[my_row * cnt + k]
                                                                   : lm_cond_desc<(ORD > 0 ? ORD : 1)>(a.lm, sp, c, c_uni[k]);
                        val += a.alpha * lmp + a.beta;
                    }
                }
                ekeys[my_p * cnt + k] = okey(val);
            }
        }
        __syncthreads();
        if (lm_cache) {                                  // the context hash lived in the histogram area: clear it for the selection
            for (int i = tid; i < 7 * 256; i += BS_THREADS) hist[i] = 0;      // (visible behind the block scan's barrier below)
        }
        BS_TICK(1);
        // ---- 3. the prefixes themselves; this thread's strided sample of the extension keys -------------------------
        unsigned kex = 0;
        float my_nb = -INFINITY, my_sc = -INFINITY;
        if (tid < n) {
            my_nb = lse2(rep[tid], ext[tid]);
            my_sc = lse2(bcur[tid], my_nb);
            kex = okey(my_sc);
        }
        unsigned keys[NPT];
        unsigned tmax = kex, tminw = kex - (NEG + 1u);                  // (tminw: min over the FINITE keys, as key - (NEG + 1): "no entry" and
        int fin = kex > NEG;                                            //  -inf keys wrap around to huge values and drop out of the minimum)
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int e = tid + i * BS_THREADS;
            keys[i] = e < nq ? ekeys[e] : 0u;                           // 0 = no entry
            tmax = max(tmax, keys[i]);
            tminw = min(tminw, keys[i] - (NEG + 1u));
            fin += keys[i] > NEG;
        }
        // The finite keys of a frame share their leading bytes (scores of one frame lie within a factor of two or so: sign,
        // exponent and often more are common): the workgroup's max and min ride on the count's barrier, and the radix passes over
        // the bytes in which they agree are skipped -- one of three bound passes and one of four select passes, nearly always
        {   // (one LDS atomic pair per wave into misc[1] / misc[2], reset by the frame's staging; two reads behind the barrier)
            const unsigned wmax = wave_max_u32(tmax), wminw = wave_min_u32(tminw);
            if (lane == 0) {
                atomicMax(reinterpret_cast<unsigned*>(&misc[1]), wmax);
                atomicMin(reinterpret_cast<unsigned*>(&misc[2]), wminw);
            }
        }
        int tot;
        block_scan(fin, wsum + 0 * BS_WAVES, tot);
        const unsigned kmaxb = (unsigned)misc[1] > NEG ? (unsigned)misc[1] : 0u;
        const unsigned kminb = (unsigned)misc[2] < 0u - (NEG + 1u) ? (unsigned)misc[2] + NEG + 1u : 0xFFFFFFFFu;      // (wrapped: no finite key)
        const int skip = (a.narrow & 2) ? 0 : kmaxb >= kminb ? ((kmaxb ^ kminb) == 0u ? 4 : (__clz((int)(kmaxb ^ kminb)) >> 3)) : 0;     // common leading bytes
        const unsigned cmask = skip == 0 ? 0u : skip >= 4 ? 0xFFFFFFFFu : 0xFFFFFFFFu << (32 - 8 * skip);
        BS_TICK(2);
        // ---- 4a. lower bound of the beam-th best key: the beam-th best of the per-thread maxima, to 24 bits ---------------
        unsigned low = NEG + 1;                                         // every finite entry
        const bool overflow = tot > beam;
        // (round 5 measured skipping the bound for frames of <= 1 024 / <= 4 096 extension entries -- what a sharp posterior gives:
        //  no gain / slower, 19.6 -> 19.6 / 21.0 us per frame at 3.7 candidates: the survivors' rows cost the compaction what the
        //  three bound passes cost the selection)
        if (overflow) {
            unsigned prefix = kmaxb & cmask, mask = cmask;
            int need = beam;
            for (int pass = min(skip, 3); pass < 3; ++pass) {
                const int shift = 24 - 8 * pass;
                int* hp = hist + pass * 256;
                hist_add(hp, (tmax & mask) == prefix, (tmax >> shift) & 255);
                __syncthreads();
                int bin, rem, total, binc;
                pick_bin_tot(hp, need, bin, rem, total, binc);
                prefix |= (unsigned)bin << shift;
                mask |= 255u << shift;
                need = rem;
                if (rem == binc) break;          // the picked bin is needed whole: exactly `beam` maxima are >= prefix already
            }
            low = max(prefix & 0xFFFFFF00u, NEG + 1);                   // >= beam entries are >= low (24 bits, as before)
        }
        // ---- 4b. survivors (extension entries >= low) -> ordered list, re-dealt one per thread ----------------------------
        int ns = 0;
#pragma unroll
        for (int i = 0; i < NPT; ++i) ns += keys[i] >= low;
        int nsurv;
        int pos = block_scan(ns, wsum + 1 * BS_WAVES, nsurv);
#pragma unroll
        for (int i = 0; i < NPT; ++i)
            if (keys[i] >= low) slist[pos++] = (unsigned short)(tid + i * BS_THREADS);
        __syncthreads();
        unsigned ke[NPT];
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            ke[i] = 0u;
            if (i * BS_THREADS < nsurv) {
                const int idx = tid + i * BS_THREADS;
                if (idx < nsurv) ke[i] = ekeys[slist[idx]];
            }
        }
        const unsigned kx = kex >= low ? kex : 0u;                       // live prefixes take part with their own score
        // ---- 4c. exact radix select among the survivors -------------------------------------------------------------------
        unsigned thr = NEG;        // select key > thr, plus `need_eq` of the keys == thr
        int need_eq = 0;
        if (overflow) {
            unsigned prefix = kmaxb & cmask, mask = cmask;
            int need = beam;
            for (int pass = skip; pass < 4; ++pass) {
                const int shift = 24 - 8 * pass;
                int* hp = hist + (3 + pass) * 256;
                hist_add(hp, kx != 0 && (kx & mask) == prefix, (kx >> shift) & 255);
#pragma unroll
                for (int i = 0; i < NPT; ++i)
                    if (i * BS_THREADS < nsurv) hist_add(hp, ke[i] != 0 && (ke[i] & mask) == prefix, (ke[i] >> shift) & 255);
                __syncthreads();
                int bin, rem, total, binc;
                pick_bin_tot(hp, need, bin, rem, total, binc);
                prefix |= (unsigned)bin << shift;
                mask |= 255u << shift;
                need = rem;
                if (rem == binc && pass < 3) {   // the picked bin is needed whole: every key at or above it stays, no tie to break
                    prefix -= 1u;
                    need = 0;
                    break;
                }
            }
            thr = prefix;
            need_eq = need;
        }
        BS_TICK(3);
        // ---- 5. ordered compaction into the other live buffer: live prefixes first (by index), then extensions ------
        int totE;
        const int exE = block_scan((kx > thr) | ((kx == thr && kx != 0) << 16), wsum + 2 * BS_WAVES, totE);
        const int gtE = totE & 0xffff, eqE = totE >> 16;
        const int eq_take_E = min(eqE, need_eq);
        const int n_exist = gtE + eq_take_E;
        int my_slot = -1;
        if (tid < n) {
            const int gb = exE & 0xffff, eb = exE >> 16;
            if (kx > thr || (kx == thr && kx != 0 && eb < need_eq)) my_slot = gb + min(eb, need_eq);
        }
        const int need_eq_new = need_eq - eq_take_E;
        // list order = index order of the re-dealt items: item idx = tid + i * 1024 -> scan row by row
        int n_new = 0;
        int gbase = 0, ebase = 0;                                       // selected / equal counts of the previous rows
        int row_ex[NPT];
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            row_ex[i] = 0;
            if (i * BS_THREADS < nsurv) {
                int totR;
                const int ex = block_scan((ke[i] > thr) | ((ke[i] == thr && ke[i] != 0) << 16), wsum + (3 + i) * BS_WAVES, totR);
                row_ex[i] = (gbase + (ex & 0xffff)) | ((ebase + (ex >> 16)) << 16);
                gbase += totR & 0xffff;
                ebase += totR >> 16;
            }
        }
        n_new = gbase + min(ebase, need_eq_new);
        if (my_slot >= 0) {
            lv_node[o2 + my_slot] = lv_node[o + tid];
            lv_hid[o2 + my_slot] = lv_hid[o + tid];
            lv_phid[o2 + my_slot] = lv_phid[o + tid];
            lv_ch[o2 + my_slot] = lv_ch[o + tid];
            lv_b[o2 + my_slot] = bcur[tid];
            lv_nb[o2 + my_slot] = my_nb;
            lv_sc[o2 + my_slot] = my_sc;
            if (use_lm) {
                lv_ctx[o2 + my_slot] = lv_ctx[o + tid]; lv_m[o2 + my_slot] = lv_m[o + tid];
#pragma unroll
                for (int j = 0; j < 4; ++j) lv_bo[4 * (o2 + my_slot) + j] = lv_bo[4 * (o + tid) + j];
            }
        }
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            if (i * BS_THREADS < nsurv) {
                const unsigned kk = ke[i];
                const int gb = row_ex[i] & 0xffff, eb = row_ex[i] >> 16;
                if (kk > thr || (kk == thr && kk != 0 && eb < need_eq_new)) {
                    const int r = gb + min(eb, need_eq_new);            // rank among the selected extensions
                    const int slot = n_exist + r, node = pool_count + r;
                    const int e = slist[tid + i * BS_THREADS];
                    const int p = e / cnt, k = e - p * cnt;
                    const int c = c_idx[k] & ~(1 << 30);
                    const float add = ikey(kk);           // the entry's own score (acoustic term + scorer term), as ranked
                    if (node < a.pool_cap) {
                        pool_parent[node] = lv_node[o + p];
                        pool_ch[node] = c;
                    }
                    lv_node[o2 + slot] = node;
                    lv_hid[o2 + slot] = str_hash(lv_hid[o + p], c);
                    lv_phid[o2 + slot] = lv_hid[o + p];
                    lv_ch[o2 + slot] = c;
                    lv_b[o2 + slot] = -INFINITY;
                    lv_nb[o2 + slot] = add;
                    lv_sc[o2 + slot] = add;
                    if (use_lm) {                         // the new prefix's scorer state: its words, matched suffixes, backoffs
                        const LmState sn = lm_state_of(a.lm, lm_push(lv_ctx[o + p], c));
                        lv_ctx[o2 + slot] = sn.ctx; lv_m[o2 + slot] = sn.m | (sn.oov << 8);
#pragma unroll
                        for (int j = 0; j < 4; ++j) lv_bo[4 * (o2 + slot) + j] = sn.bo[j];
                    }
                }
            }
        }
        pool_count += n_new;
        n = n_exist + n_new;
        cur ^= 1;
        __syncthreads();
        BS_TICK(4);
    }
    if (prof)
        for (int i = 0; i < 16; ++i) a.prof[i] = (long long)pcl[i];

    // ---- persist the live set (streams), pick the best prefix, walk the parent pointers ----------------------
    const int o = cur * beam;
    if (tid == 0) { st_i[0] = n; st_i[1] = pool_count; }
    for (int i = tid; i < n; i += BS_THREADS) {
        st_i[2 + i] = lv_node[o + i]; st_i[2 + beam + i] = lv_ch[o + i];
        st_h[i] = lv_hid[o + i]; st_h[beam + i] = lv_phid[o + i];
        st_f[i] = lv_b[o + i]; st_f[beam + i] = lv_nb[o + i]; st_f[2 * beam + i] = lv_sc[o + i];
        if (use_lm) {
            st_h[2 * beam + i] = lv_ctx[o + i]; st_i[2 + 2 * beam + i] = lv_m[o + i];
            for (int j = 0; j < 4; ++j) st_f[(3 + j) * beam + i] = lv_bo[4 * (o + i) + j];
        }
    }
    if (wave == 0) {
        float bs = -INFINITY;
        int bc = 0x7fffffff, bi = -1;
        for (int i = lane; i < n; i += 64) {
            const float s = lv_sc[o + i];
            const int c = lv_ch[o + i];
            if (bi < 0 || s > bs || (s == bs && c < bc)) { bs = s; bc = c; bi = i; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float os = __shfl_xor(bs, off, 64);
            const int oc = __shfl_xor(bc, off, 64), oi = __shfl_xor(bi, off, 64);
            if (oi >= 0 && (bi < 0 || os > bs || (os == bs && (oc < bc || (oc == bc && oi < bi))))) { bs = os; bc = oc; bi = oi; }
        }
        if (lane == 0 && bi < 0) {
            a.len[u] = 0;
            a.score[u] = -INFINITY;
        } else if (lane == 0) {
            int node = lv_node[o + bi], L = 0;
            for (int x = node; x > 0 && x < a.pool_cap; x = pool_parent[x]) ++L;
            a.len[u] = L;
            int pos = L - 1;
            for (int x = node; x > 0 && x < a.pool_cap; x = pool_parent[x], --pos)
                if (pos < a.max_len) a.tokens[(size_t)u * a.max_len + pos] = pool_ch[x];
            if (use_lm && L <= a.max_len) {
                // approx_ctc of the reference decoder: the scorer's share is taken out again -- score - |prefix| * beta
                // - alpha * ln P_LM(sentence), where the sentence probability also counts </s> (Scorer::get_sent_log_prob)
                unsigned long long ctx = lm_root_ctx(a.lm);
                float sent = L == 0 ? lm_cond(a.lm, lm_state_of(a.lm, ctx), a.lm.bos) : 0.f;
                for (int i = 0; i <= L; ++i) {
                    const int w = i < L ? a.tokens[(size_t)u * a.max_len + i] : a.lm.eos;
                    sent += lm_cond(a.lm, lm_state_of(a.lm, ctx), w);
                    ctx = lm_push(ctx, w);
                }
                bs = bs - (float)L * a.beta - a.alpha * sent;
            }
            a.score[u] = bs;
        }
    }
}

size_t beam_gpu_lds_bytes(int beam, int K, bool use_lm) {
    return (size_t)beam * K * 6 + (size_t)(27 + (use_lm ? 15 : 0)) * beam * 4 + 8 + BS_HASH * 12 + 8 * BS_KMAX * 4 + 7 * 256 * 4 +
           (6 + 32) * BS_WAVES * 4 + (8 + 32) * 4 + 128 + (size_t)beam * 8 + 128 * 4 + 8;
}

template <int NPT, int ORD>
static void launch_beam_t(const BeamGpuArgs& a, int B, size_t lds, hipStream_t s) {
    auto k = beam_search_kernel<NPT, ORD>;
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(k), lds, attr);
    hipLaunchKernelGGL(k, dim3(B), dim3(BS_THREADS), lds, s, a);
}
template <int ORD>
static int launch_beam_o(const BeamGpuArgs& a, int B, size_t lds, int per, hipStream_t s) {
    if (per <= 4) launch_beam_t<4, ORD>(a, B, lds, s);
    else if (per <= 12) launch_beam_t<12, ORD>(a, B, lds, s);
    else if (per <= 20) launch_beam_t<20, ORD>(a, B, lds, s);
    else if (per <= 32) launch_beam_t<32, ORD>(a, B, lds, s);
    else return 1;
    return 0;
}

int launch_beam_search(const BeamGpuArgs& a, int B, hipStream_t s) {
    if (B <= 0) return 0;
    if (a.K > BS_KMAX || a.beam > 512 || a.beam < 1 || a.pool_cap > (1 << 30)) return 1;
    if (a.use_lm && (a.lm.max_order < 1 || a.lm.max_order > 5)) return 1;
    const size_t lds = beam_gpu_lds_bytes(a.beam, a.K, a.use_lm != 0);
    if (lds > 160 * 1024) return 1;
    const int per = (a.beam * a.K + BS_THREADS - 1) / BS_THREADS;
    if (!a.use_lm) return launch_beam_o<0>(a, B, lds, per, s);
    return a.lm.max_order <= 3 ? launch_beam_o<3>(a, B, lds, per, s) : launch_beam_o<5>(a, B, lds, per, s);
}

}  // namespace masr
