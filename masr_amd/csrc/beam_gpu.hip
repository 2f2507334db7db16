// CTC prefix beam search on the GPU (LM-free part of the ctc_beam_search decoder; reference call sites
// masr/decoders/beam_search_decoder.py:45-96 -> third-party paddlespeech_ctcdecoders, see beam_search.cpp for the
// host restatement of the same algorithm and the "parity unpinned" note).
//
// One workgroup (512 threads) per utterance walks the frames sequentially; everything a step needs lives in LDS:
//   live prefixes  (<= beam):  trie node id, parent node id, last character, log P(blank end), log P(non-blank end), score
//   candidates     (<= K):     the pruned vocabulary of the frame (topk_prune_kernel), descending probability
//   entry scores   (<= beam * (K + 1)): the live prefixes themselves + every (prefix, character) extension
// Step (== Beam::step of beam_search.cpp, reformulated without a pointer trie):
//   A  every live prefix p:  b_cur = lp(blank) + score(p);  nb_cur = lp(ch(p)) + nb_prev(p)        (repeat of its last char)
//   B  every pair (p, c != blank): add = c == ch(p) ? lp(c) + b_prev(p) : lp(c) + score(p).  If the child (p, c) is itself a
//      live prefix (hash lookup keyed by (node(p), c)) the term is merged into its nb_cur -- a node has one parent, so at
//      most one such term per live prefix and the two-term log-sum-exp is order independent -- otherwise it is a new entry.
//   C  score = logsumexp(b_cur, nb_cur); keep the `beam` best entries: 4-pass radix select on the order-preserving integer
//      image of the scores, then an index-ordered compaction (deterministic).  Survivors that are new get trie nodes
//      (parent pointer + character) appended to the utterance's node pool in HBM -- only survivors ever get a node.
// Entries with score -inf are never revived except through their parent's extension, which re-creates them, so dropping
// them is equivalent to the pointer trie that keeps them.
// The best prefix is read back by walking parent pointers.
#include <math.h>

#include "common.h"

namespace masr {

static constexpr int BS_THREADS = 512;
static constexpr int BS_WAVES = BS_THREADS / 64;
static constexpr int BS_HASH = 1024;
static constexpr int BS_KMAX = 64;

__device__ __forceinline__ float lse2(float x, float y) {
    if (x == -INFINITY) return y;
    if (y == -INFINITY) return x;
    const float m = fmaxf(x, y);
    return m + logf(expf(x - m) + expf(y - m));
}
__device__ __forceinline__ unsigned okey(float f) {       // larger float -> larger unsigned
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ unsigned hash32(unsigned k) {
    k *= 0x9E3779B1u;
    return (k >> 22) & (BS_HASH - 1);
}


__global__ __launch_bounds__(BS_THREADS) void beam_search_kernel(BeamGpuArgs a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int u = blockIdx.x;
    const int beam = a.beam, K = a.K;
    const int Emax = beam * (K + 1);
    // ---- LDS carve-up --------------------------------------------------------------------------------
    unsigned* keys = reinterpret_cast<unsigned*>(smem_raw);              // [Emax]
    int* lv_node = reinterpret_cast<int*>(keys + Emax);                  // [2][beam] each
    int* lv_pnode = lv_node + 2 * beam;
    int* lv_ch = lv_pnode + 2 * beam;
    float* lv_b = reinterpret_cast<float*>(lv_ch + 2 * beam);
    float* lv_nb = lv_b + 2 * beam;
    float* lv_sc = lv_nb + 2 * beam;
    float* bcur = lv_sc + 2 * beam;                                     // [beam]
    float* nbcur = bcur + beam;                                         // [beam]
    unsigned* hkey = reinterpret_cast<unsigned*>(nbcur + beam);          // [BS_HASH]
    int* hval = reinterpret_cast<int*>(hkey + BS_HASH);                                          // [BS_HASH]
    int* c_idx = hval + BS_HASH;                                         // [BS_KMAX]
    float* c_lp = reinterpret_cast<float*>(c_idx + BS_KMAX);             // [BS_KMAX]
    int* hist = reinterpret_cast<int*>(c_lp + BS_KMAX);                  // [BS_WAVES][256]
    int* wsum = hist + BS_WAVES * 256;                                   // [BS_WAVES] scan scratch
    int* misc = wsum + BS_WAVES;                                         // [8]: 0 blank_k, 1 sel_bin, 2 need, 3 n_sel_exist

    int* pool_parent = a.pool_parent + (size_t)u * a.pool_cap;
    int* pool_ch = a.pool_ch + (size_t)u * a.pool_cap;
    int* st_i = a.state_i + (size_t)u * (2 + 3 * beam);
    float* st_f = a.state_f + (size_t)u * (3 * beam);

    int n, pool_count, cur = 0;
    if (a.init) {
        n = 1;
        pool_count = 1;
        if (tid == 0) {
            pool_parent[0] = -1;
            pool_ch[0] = -1;
            lv_node[0] = 0; lv_pnode[0] = -1; lv_ch[0] = -1;
            lv_b[0] = 0.f; lv_nb[0] = -INFINITY; lv_sc[0] = 0.f;
        }
    } else {
        n = st_i[0];
        pool_count = st_i[1];
        for (int i = tid; i < n; i += BS_THREADS) {
            lv_node[i] = st_i[2 + i]; lv_pnode[i] = st_i[2 + beam + i]; lv_ch[i] = st_i[2 + 2 * beam + i];
            lv_b[i] = st_f[i]; lv_nb[i] = st_f[beam + i]; lv_sc[i] = st_f[2 * beam + i];
        }
    }
    __syncthreads();

    const int T = a.frames ? min(a.frames[u], a.T_stride) : a.T_stride;
    for (int t = 0; t < T; ++t) {
        const size_t row = (size_t)u * a.T_stride + t;
        const int cnt = min(a.ccount[row], K);
        const int o = cur * beam, o2 = (cur ^ 1) * beam;
        // ---- 0. candidates, hash clear ----------------------------------------------------------------
        if (tid == 0) { misc[0] = -1; misc[3] = 0; }
        for (int i = tid; i < BS_HASH; i += BS_THREADS) hkey[i] = 0xffffffffu;
        __syncthreads();
        if (tid < cnt) {
            const int c = a.cidx[row * K + tid];
            c_idx[tid] = c;
            c_lp[tid] = a.clp[row * K + tid];
            if (c == a.blank) misc[0] = tid;
        }
        // ---- 1. hash of the live prefixes: (parent node, char) -> live index ----------------------------
        if (tid < n && lv_pnode[o + tid] >= 0) {
            const unsigned key = (unsigned)lv_pnode[o + tid] * 8192u + (unsigned)lv_ch[o + tid];
            unsigned h = hash32(key);
            while (atomicCAS(&hkey[h], 0xffffffffu, key) != 0xffffffffu) h = (h + 1) & (BS_HASH - 1);
            hval[h] = tid;
        }
        __syncthreads();
        // ---- 2. phase A: the prefixes themselves ------------------------------------------------------------
        const int blank_k = misc[0];
        if (tid < n) {
            const float sc = lv_sc[o + tid];
            bcur[tid] = blank_k >= 0 ? c_lp[blank_k] + sc : -INFINITY;
            float nb = -INFINITY;
            const int ch = lv_ch[o + tid];
            for (int k = 0; k < cnt; ++k)
                if (c_idx[k] == ch && ch != a.blank) nb = c_lp[k] + lv_nb[o + tid];
            nbcur[tid] = nb;
        }
        __syncthreads();
        // ---- 3. phase B: extensions ------------------------------------------------------------------------
        const int E = n + n * cnt;
        for (int e = tid; e < n * cnt; e += BS_THREADS) {
            const int p = e / cnt, k = e - p * cnt;
            const int c = c_idx[k];
            float val = -INFINITY;
            if (c != a.blank) {
                const float lp = c_lp[k];
                const float add = c == lv_ch[o + p] ? (lv_b[o + p] > -INFINITY ? lp + lv_b[o + p] : -INFINITY) : lp + lv_sc[o + p];
                const unsigned key = (unsigned)lv_node[o + p] * 8192u + (unsigned)c;
                unsigned h = hash32(key);
                int found = -1;
                while (true) {
                    const unsigned hk = hkey[h];
                    if (hk == key) { found = hval[h]; break; }
                    if (hk == 0xffffffffu) break;
                    h = (h + 1) & (BS_HASH - 1);
                }
                if (found >= 0) nbcur[found] = lse2(nbcur[found], add);   // unique writer: a node has one parent
                else val = add;
            }
            keys[n + e] = okey(val);
        }
        __syncthreads();
        if (tid < n) keys[tid] = okey(lse2(bcur[tid], nbcur[tid]));
        __syncthreads();
        // ---- 4. radix select of the beam-th best key ---------------------------------------------------------
        unsigned thr = 0;          // select key > thr, plus `need_eq` of the keys == thr
        int need_eq = 0;
        const unsigned NEG = okey(-INFINITY);
        int fin = 0;               // entries with a finite score; if they all fit, keep exactly those
        for (int e = tid; e < E; e += BS_THREADS) fin += keys[e] != NEG;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) fin += __shfl_xor(fin, off, 64);
        if (lane == 0) wsum[wave] = fin;
        __syncthreads();
        fin = 0;
        for (int w = 0; w < BS_WAVES; ++w) fin += wsum[w];
        __syncthreads();
        const bool take_all = fin <= beam;
        if (!take_all) {
            unsigned prefix = 0, mask = 0;
            int need = beam;
            for (int pass = 0; pass < 4; ++pass) {
                const int shift = 24 - 8 * pass;
                for (int i = tid; i < BS_WAVES * 256; i += BS_THREADS) hist[i] = 0;
                __syncthreads();
                for (int e = tid; e < E; e += BS_THREADS) {
                    const unsigned kk = keys[e];
                    if ((kk & mask) == prefix) atomicAdd(&hist[wave * 256 + ((kk >> shift) & 255)], 1);
                }
                __syncthreads();
                if (wave == 0) {
                    int c4[4], s = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int bin = 255 - (4 * lane + j);
                        int v = 0;
#pragma unroll
                        for (int w = 0; w < BS_WAVES; ++w) v += hist[w * 256 + bin];
                        c4[j] = v;
                        s += v;
                    }
                    int incl = s;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) {
                        const int v = __shfl_up(incl, off, 64);
                        if (lane >= off) incl += v;
                    }
                    int excl = incl - s;
                    if (excl < need && need <= incl) {
                        int acc = excl;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (acc < need && need <= acc + c4[j]) {
                                misc[1] = 255 - (4 * lane + j);
                                misc[2] = need - acc;
                            }
                            acc += c4[j];
                        }
                    }
                }
                __syncthreads();
                prefix |= (unsigned)misc[1] << shift;
                mask |= 255u << shift;
                need = misc[2];
                __syncthreads();
            }
            thr = prefix;
            need_eq = need;
        }
        // ---- 5. index-ordered compaction into the other live buffer -------------------------------------------
        const int per = (E + BS_THREADS - 1) / BS_THREADS;
        const int e0 = min(tid * per, E), e1 = min(e0 + per, E);
        int gt = 0, eq = 0;
        for (int e = e0; e < e1; ++e) {
            const unsigned kk = keys[e];
            if (take_all) gt += kk != NEG;
            else { gt += kk > thr; eq += kk == thr; }
        }
        int packed = gt | (eq << 16), incl = packed;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        int total = 0;
        for (int w = 0; w < BS_WAVES; ++w) total += wsum[w];
        const int excl = base + incl - packed;
        int gt_before = excl & 0xffff, eq_before = excl >> 16;
        const int n_new_total = take_all ? fin : beam;
        (void)total;
        // first pass: survivors that already are live prefixes (entries < n come first in index order)
        for (int e = e0; e < e1; ++e) {
            const unsigned kk = keys[e];
            bool sel;
            if (take_all) sel = kk != NEG;
            else sel = kk > thr || (kk == thr && eq_before < need_eq);
            const int slot = gt_before + (take_all ? 0 : min(eq_before, need_eq));
            if (take_all) gt_before += sel;
            else { gt_before += kk > thr; eq_before += kk == thr; }
            if (!sel) { keys[e] = 0xffffffffu; continue; }     // mark: not selected
            keys[e] = (unsigned)slot;                          // selected: remember the slot
            if (e < n) {
                lv_node[o2 + slot] = lv_node[o + e];
                lv_pnode[o2 + slot] = lv_pnode[o + e];
                lv_ch[o2 + slot] = lv_ch[o + e];
                lv_b[o2 + slot] = bcur[e];
                lv_nb[o2 + slot] = nbcur[e];
                lv_sc[o2 + slot] = lse2(bcur[e], nbcur[e]);
                atomicAdd(&misc[3], 1);
            }
        }
        __syncthreads();
        const int n_exist = misc[3];
        for (int e = max(e0, n); e < e1; ++e) {
            const unsigned slot = keys[e];
            if (slot == 0xffffffffu) continue;
            const int q = e - n;
            const int p = q / cnt, k = q - p * cnt;
            const int node = pool_count + ((int)slot - n_exist);
            const float add = c_idx[k] == lv_ch[o + p] ? c_lp[k] + lv_b[o + p] : c_lp[k] + lv_sc[o + p];
            if (node < a.pool_cap) {
                pool_parent[node] = lv_node[o + p];
                pool_ch[node] = c_idx[k];
            }
            lv_node[o2 + slot] = node;
            lv_pnode[o2 + slot] = lv_node[o + p];
            lv_ch[o2 + slot] = c_idx[k];
            lv_b[o2 + slot] = -INFINITY;
            lv_nb[o2 + slot] = add;
            lv_sc[o2 + slot] = add;
        }
        pool_count += n_new_total - n_exist;
        n = n_new_total;
        cur ^= 1;
        __syncthreads();
    }

    // ---- persist the live set (streams), pick the best prefix, walk the parent pointers ----------------------
    const int o = cur * beam;
    if (tid == 0) { st_i[0] = n; st_i[1] = pool_count; }
    for (int i = tid; i < n; i += BS_THREADS) {
        st_i[2 + i] = lv_node[o + i]; st_i[2 + beam + i] = lv_pnode[o + i]; st_i[2 + 2 * beam + i] = lv_ch[o + i];
        st_f[i] = lv_b[o + i]; st_f[beam + i] = lv_nb[o + i]; st_f[2 * beam + i] = lv_sc[o + i];
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
            a.score[u] = bs;
            int pos = L - 1;
            for (int x = node; x > 0 && x < a.pool_cap; x = pool_parent[x], --pos)
                if (pos < a.max_len) a.tokens[(size_t)u * a.max_len + pos] = pool_ch[x];
        }
    }
}

size_t beam_gpu_lds_bytes(int beam, int K) {
    const size_t Emax = (size_t)beam * (K + 1);
    return Emax * 4 + (size_t)12 * beam * 4 + (size_t)2 * beam * 4 + 2 * BS_HASH * 4 + 2 * BS_KMAX * 4 + BS_WAVES * 256 * 4 +
           BS_WAVES * 4 + 8 * 4 + 64;
}

int launch_beam_search(const BeamGpuArgs& a, int B, hipStream_t s) {
    if (B <= 0) return 0;
    if (a.K > BS_KMAX || a.beam > 512 || a.beam < 1 || a.pool_cap > 524000) return 1;
    const size_t lds = beam_gpu_lds_bytes(a.beam, a.K);
    if (lds > 160 * 1024) return 1;
    static size_t attr = 0;
    if (lds > attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(beam_search_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
        attr = lds;
    }
    hipLaunchKernelGGL(beam_search_kernel, dim3(B), dim3(BS_THREADS), lds, s, a);
    return 0;
}

}  // namespace masr
