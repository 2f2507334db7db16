// Silero VAD network on the GPU -- the segmentation model of MASRPredictor.predict_long.
//
// Reference: masr/infer_utils/vad_predictor.py:83-104 runs `silero_vad.onnx` (shipped next to it) through onnxruntime, one
// 512-sample window per session.run, inside get_speech_timestamps (:106-175) and stream_vad (:177-216).  This file is the network
// itself, written from the graph of that file (oracle/silero.py states it in numpy; masr_amd/utils/onnx_lite.py reads the
// weights out of the user's copy of the ONNX file): per window
//   reflect-pad 96 | STFT as 258 dot products of 256 taps per frame (hop 64) | magnitude, ln(1 + 2^20 m), adaptive normalisation
//   4 x [depthwise k=5 -> ReLU -> pointwise (+ projection | identity shortcut) -> ReLU -> 1x1 conv (stride) -> ReLU]  258->16->32->32->64
//   2 LSTM layers (hidden 64, ONNX gate order i, o, f, c) carried across windows | ReLU -> 64 -> 1 -> sigmoid | mean over steps
// Two kernels per call, whatever the number of windows:
//   silero_front_kernel   one workgroup per (window, sequence): everything up to the LSTM input.  Windows are independent there,
//                         so a whole recording goes through in one launch (the reference walks it window by window).
//   silero_lstm_kernel    one workgroup per sequence walks the windows in order.  512 threads: threads 0..255 own the gate rows of
//                         layer 1, 256..511 those of layer 2, each with its 64 + 64 weights in registers; layer 2 works on step
//                         s - 1 while layer 1 works on step s (they only meet in h1(s - 1)), two barriers per step.
// Plain VALU code: the network is 0.4 MFLOP per window and strictly sequential in its recurrent part.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/masr_hip.h"

namespace {

thread_local std::string g_vad_err;
int vfail(const std::string& m) {
    g_vad_err = m;
    return 1;
}
#define VHIP(expr)                                                                                           \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess) return vfail(std::string(#expr) + ": " + hipGetErrorString(_e));               \
    } while (0)

constexpr int NB = 129;          // STFT bins
constexpr int NC0 = 258;         // magnitude | normalised log spectrum
constexpr int FMAX = 24;         // frames of the longest window (1536 / 64)
constexpr int WMAX = 1536;

struct Block {                   // one encoder block (device pointers)
    const float *dw_w, *dw_b;    // [C][5], [C]
    const float *pw_wT, *pw_b;   // [C][O] (transposed), [O]
    const float *proj_wT, *proj_b;   // [C][O] or nullptr (identity shortcut)
    const float *out_wT, *out_b; // [O][O] transposed ([in][out]), [O]
    int C, O, stride;
};
struct Net {
    const float* basisT;         // [256][258]
    float filt[7];
    Block blk[4];
    const float *lstm_w[2], *lstm_r[2], *lstm_b[2];   // [256][64], [256][64], [256] (Wb + Rb)
    const float* dec_w;          // [64]
    float dec_b;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- front end: audio -> LSTM inputs -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void silero_front_kernel(Net net, const float* __restrict__ audio, int n_win, int window,
                                                           float* __restrict__ feats /* [B][n_win][T][64] */, int T) {
    __shared__ float xs[WMAX + 192];
    __shared__ float feat[NC0 * FMAX];           // [258][F]: magnitude | spect (then normalised)
    __shared__ float tmp[NC0 * FMAX];            // STFT output, then the depthwise output of a block
    __shared__ float ha[64 * FMAX], hb[64 * FMAX];
    __shared__ float fmean[FMAX + 6];
    __shared__ float mm;
    const int tid = threadIdx.x, win = blockIdx.x, b = blockIdx.y;
    const int F = window / 64;
    const float* x = audio + ((size_t)b * n_win + win) * window;
    for (int i = tid; i < window + 192; i += 256) {
        int s = i - 96;
        if (s < 0) s = -s;
        if (s >= window) s = 2 * (window - 1) - s;
        xs[i] = x[s];
    }
    __syncthreads();
    // STFT: row r of the basis against every frame
    for (int r = tid; r < NC0; r += 256) {
        float acc[FMAX];
#pragma unroll
        for (int f = 0; f < FMAX; ++f) acc[f] = 0.f;
        for (int k = 0; k < 256; ++k) {
            const float w = net.basisT[k * NC0 + r];
#pragma unroll
            for (int f = 0; f < FMAX; ++f)
                if (f < F) acc[f] = fmaf(w, xs[64 * f + k], acc[f]);
        }
#pragma unroll
        for (int f = 0; f < FMAX; ++f)
            if (f < F) tmp[r * F + f] = acc[f];
    }
    __syncthreads();
    for (int i = tid; i < NB * F; i += 256) {
        const float re = tmp[i], im = tmp[NB * F + i];
        const float m = sqrtf(re * re + im * im);
        feat[i] = m;
        feat[NB * F + i] = logf(1.0f + 1048576.0f * m);
    }
    __syncthreads();
    if (tid < F) {                                // mean of the log spectrum over the bins, per frame
        float s = 0.f;
        for (int bin = 0; bin < NB; ++bin) s += feat[(NB + bin) * F + tid];
        fmean[3 + tid] = s * (1.0f / NB);
    }
    __syncthreads();
    if (tid == 0) {                               // reflect-pad by 3, 7-tap smoothing, mean over the window
        for (int j = 0; j < 3; ++j) {
            fmean[2 - j] = fmean[3 + 1 + j];
            fmean[3 + F + j] = fmean[3 + F - 2 - j];
        }
        float tot = 0.f;
        for (int f = 0; f < F; ++f) {
            float s = 0.f;
            for (int k = 0; k < 7; ++k) s = fmaf(net.filt[k], fmean[f + k], s);
            tot += s;
        }
        mm = tot / (float)F;
    }
    __syncthreads();
    for (int i = tid; i < NB * F; i += 256) feat[NB * F + i] -= mm;
    __syncthreads();
    // the four blocks; `in` [C][Fin] -> `out` [O][Fout]
    // (`mid` is always ha and a block's output always hb: the output conv reads ha only, and by then the block's input -- feat for
    //  the first block, hb afterwards -- is dead)
    const float* in = feat;
    int Fin = F;
    for (int k = 0; k < 4; ++k) {
        const Block bl = net.blk[k];
        // depthwise k = 5, zero padding 2, ReLU
        for (int i = tid; i < bl.C * Fin; i += 256) {
            const int c = i / Fin, f = i - c * Fin;
            float s = bl.dw_b[c];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int ff = f + j - 2;
                if (ff >= 0 && ff < Fin) s = fmaf(bl.dw_w[c * 5 + j], in[c * Fin + ff], s);
            }
            tmp[i] = fmaxf(s, 0.f);
        }
        __syncthreads();
        // pointwise + shortcut, ReLU -> ha [O][Fin]  (thread = (frame, output channel): the transposed weights are read coalesced)
        for (int i = tid; i < bl.O * Fin; i += 256) {
            const int f = i / bl.O, o = i - f * bl.O;
            float s = bl.pw_b[o];
            for (int c = 0; c < bl.C; ++c) s = fmaf(bl.pw_wT[c * bl.O + o], tmp[c * Fin + f], s);
            float sc;
            if (bl.proj_wT) {
                sc = bl.proj_b[o];
                for (int c = 0; c < bl.C; ++c) sc = fmaf(bl.proj_wT[c * bl.O + o], in[c * Fin + f], sc);
            } else {
                sc = in[o * Fin + f];
            }
            ha[o * Fin + f] = fmaxf(s + sc, 0.f);
        }
        __syncthreads();
        // 1 x 1 convolution with stride, ReLU -> hb [O][Fout]
        const int Fout = (Fin + bl.stride - 1) / bl.stride;
        for (int i = tid; i < bl.O * Fout; i += 256) {
            const int f = i / bl.O, o = i - f * bl.O;
            float s = bl.out_b[o];
            for (int c = 0; c < bl.O; ++c) s = fmaf(bl.out_wT[c * bl.O + o], ha[c * Fin + f * bl.stride], s);
            hb[o * Fout + f] = fmaxf(s, 0.f);
        }
        __syncthreads();
        in = hb;
        Fin = Fout;
    }
    // in [64][T] -> feats [T][64]
    float* dst = feats + (((size_t)b * n_win + win) * T) * 64;
    for (int i = tid; i < 64 * T; i += 256) {
        const int t = i / 64, c = i - t * 64;
        dst[i] = in[c * T + t];
    }
}

// ---- recurrent part: two LSTM layers skewed by one step, decoder, mean over the steps of a window ----------------------------
__global__ __launch_bounds__(512) void silero_lstm_kernel(Net net, const float* __restrict__ feats, int n_win, int T,
                                                          float* __restrict__ h_io, float* __restrict__ c_io, int B,
                                                          float* __restrict__ probs) {
    __shared__ __align__(16) float xbuf[2][64];
    __shared__ __align__(16) float h1[64], h2[64];
    __shared__ float c1[64], c2[64];
    __shared__ float gates[2][256];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int layer = tid >> 8, j = tid & 255;
    const int S = n_win * T;
    const float* fx = feats + (size_t)b * S * 64;
    // state layout of the reference: h, c [2 layers][B][64]
    if (tid < 64) {
        h1[tid] = h_io[(0 * B + b) * 64 + tid];
        c1[tid] = c_io[(0 * B + b) * 64 + tid];
        h2[tid] = h_io[(1 * B + b) * 64 + tid];
        c2[tid] = c_io[(1 * B + b) * 64 + tid];
        if (S > 0) xbuf[0][tid] = fx[tid];
    }
    float w[64], r[64];
    {
        const float* wp = net.lstm_w[layer] + j * 64;
        const float* rp = net.lstm_r[layer] + j * 64;
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            w[k] = wp[k];
            r[k] = rp[k];
        }
    }
    const float bias = net.lstm_b[layer][j];
    const float dw = tid >= 256 && tid < 320 ? net.dec_w[tid - 256] : 0.f;
    float pacc = 0.f;            // decoder outputs of the current window (thread 256)
    __syncthreads();
    for (int s = 0; s <= S; ++s) {
        // phase A: gate pre-activations; layer 1 on step s, layer 2 on step s - 1 (both read h1 = h1(s - 1))
        const bool active = layer == 0 ? s < S : s >= 1;
        if (active) {
            const float* inp = layer == 0 ? xbuf[s & 1] : h1;
            const float* rec = layer == 0 ? h1 : h2;
            float a0 = bias, a1 = 0.f;
#pragma unroll
            for (int k = 0; k < 64; k += 4) {
                const float4 xv = *reinterpret_cast<const float4*>(inp + k);
                const float4 hv = *reinterpret_cast<const float4*>(rec + k);
                a0 = fmaf(w[k], xv.x, a0); a0 = fmaf(w[k + 1], xv.y, a0); a0 = fmaf(w[k + 2], xv.z, a0); a0 = fmaf(w[k + 3], xv.w, a0);
                a1 = fmaf(r[k], hv.x, a1); a1 = fmaf(r[k + 1], hv.y, a1); a1 = fmaf(r[k + 2], hv.z, a1); a1 = fmaf(r[k + 3], hv.w, a1);
            }
            gates[layer][j] = a0 + a1;
        }
        __syncthreads();
        // phase B: cell updates (ONNX gate order i, o, f, c); the next input row is fetched meanwhile
        if (tid < 64 && s < S) {
            const float gi = gates[0][tid], go = gates[0][64 + tid], gf = gates[0][128 + tid], gc = gates[0][192 + tid];
            const float c = sigmoidf_(gf) * c1[tid] + sigmoidf_(gi) * tanhf(gc);
            c1[tid] = c;
            h1[tid] = sigmoidf_(go) * tanhf(c);
        } else if (tid >= 256 && tid < 320 && s >= 1) {
            const int u = tid - 256;
            const float gi = gates[1][u], go = gates[1][64 + u], gf = gates[1][128 + u], gc = gates[1][192 + u];
            const float c = sigmoidf_(gf) * c2[u] + sigmoidf_(gi) * tanhf(gc);
            c2[u] = c;
            const float h = sigmoidf_(go) * tanhf(c);
            h2[u] = h;
            // decoder: ReLU -> 64 -> 1 -> sigmoid; threads 256..319 are exactly one wave
            float d = fmaxf(h, 0.f) * dw;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) d += __shfl_xor(d, off, 64);
            if (u == 0) {
                pacc += sigmoidf_(d + net.dec_b);
                const int step = s - 1;
                if ((step + 1) % T == 0) {
                    probs[(size_t)b * n_win + step / T] = pacc / (float)T;
                    pacc = 0.f;
                }
            }
        } else if (tid >= 320 && tid < 384 && s + 1 < S) {
            xbuf[(s + 1) & 1][tid - 320] = fx[(size_t)(s + 1) * 64 + tid - 320];
        }
        __syncthreads();
    }
    if (tid < 64) {
        h_io[(0 * B + b) * 64 + tid] = h1[tid];
        c_io[(0 * B + b) * 64 + tid] = c1[tid];
        h_io[(1 * B + b) * 64 + tid] = h2[tid];
        c_io[(1 * B + b) * 64 + tid] = c2[tid];
    }
}

}  // namespace

struct masr_vad {
    int device = 0;
    std::map<std::string, std::vector<float>> host[2];      // [0] 16 kHz model, [1] 8 kHz model: tensors by name until finalize
    float* blob[2] = {nullptr, nullptr};
    Net net[2];
    bool ready[2] = {false, false};
    float* feats = nullptr;
    size_t feats_cap = 0;
};

extern "C" {

const char* masr_vad_last_error(void) { return g_vad_err.c_str(); }

int masr_vad_create(int32_t device_id, masr_vad** out) {
    if (!out) return vfail("null argument");
    int n = 0;
    VHIP(hipGetDeviceCount(&n));
    if (device_id < 0 || device_id >= n) return vfail("masr_vad_create: no such GPU (there is no CPU path)");
    masr_vad* v = new masr_vad();
    v->device = device_id;
    *out = v;
    return 0;
}

void masr_vad_destroy(masr_vad* v) {
    if (!v) return;
    (void)hipSetDevice(v->device);
    for (int m = 0; m < 2; ++m)
        if (v->blob[m]) (void)hipFree(v->blob[m]);
    if (v->feats) (void)hipFree(v->feats);
    delete v;
}

int masr_vad_load_tensor(masr_vad* v, int32_t sample_rate, const char* name, const float* data_host, int64_t n) {
    if (!v || !name || !data_host || n <= 0) return vfail("null argument");
    if (sample_rate != 16000 && sample_rate != 8000) return vfail("the Silero file holds a 16 kHz and an 8 kHz model");
    v->host[sample_rate == 16000 ? 0 : 1][name].assign(data_host, data_host + n);
    return 0;
}

// tensors expected per model (names as masr_amd/infer_utils/silero_vad.py hands them over; sizes checked here):
//   basis [258,256]  norm_filter [7]  b{k}.dw.w [C,5]  b{k}.dw.b [C]  b{k}.pw.w [O,C]  b{k}.pw.b [O]  b{k}.proj.w [O,C]  b{k}.proj.b [O]
//   (k = 0, 1, 3)  b{k}.out.w [O,O]  b{k}.out.b [O]  b{k}.out.stride [1]  lstm{0,1}.W [256,64]  lstm{0,1}.R [256,64]
//   lstm{0,1}.b [256] (Wb + Rb)  dec.w [64]  dec.b [1]
int masr_vad_finalize(masr_vad* v, int32_t sample_rate) {
    if (!v) return vfail("null argument");
    if (sample_rate != 16000 && sample_rate != 8000) return vfail("sample_rate must be 16000 or 8000");
    const int m = sample_rate == 16000 ? 0 : 1;
    auto& H = v->host[m];
    VHIP(hipSetDevice(v->device));
    static const int CH[5] = {258, 16, 32, 32, 64};
    std::vector<float> blob;
    std::map<std::string, size_t> off;
    auto need = [&](const std::string& name, size_t n, bool optional = false) -> int {
        auto it = H.find(name);
        if (it == H.end()) return optional ? 2 : vfail("masr_vad_finalize: tensor " + name + " was not loaded");
        if (it->second.size() != n)
            return vfail("masr_vad_finalize: tensor " + name + " has " + std::to_string(it->second.size()) + " elements, expected " +
                         std::to_string(n));
        return 0;
    };
    auto put = [&](const std::string& key, const std::vector<float>& data) {
        while (blob.size() % 4) blob.push_back(0.f);            // 16-byte alignment of every tensor
        off[key] = blob.size();
        blob.insert(blob.end(), data.begin(), data.end());
    };
    auto transposed = [](const std::vector<float>& w, int rows, int cols) {   // [rows][cols] -> [cols][rows]
        std::vector<float> t((size_t)rows * cols);
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c) t[(size_t)c * rows + r] = w[(size_t)r * cols + c];
        return t;
    };
    if (need("basis", 258 * 256) || need("norm_filter", 7) || need("dec.w", 64) || need("dec.b", 1)) return 1;
    put("basisT", transposed(H["basis"], 258, 256));
    int strides[4];
    bool has_proj[4];
    for (int k = 0; k < 4; ++k) {
        const int C = CH[k], O = CH[k + 1];
        const std::string p = "b" + std::to_string(k) + ".";
        if (need(p + "dw.w", (size_t)C * 5) || need(p + "dw.b", C) || need(p + "pw.w", (size_t)O * C) || need(p + "pw.b", O) ||
            need(p + "out.w", (size_t)O * O) || need(p + "out.b", O) || need(p + "out.stride", 1))
            return 1;
        const int pr = need(p + "proj.w", (size_t)O * C, true);
        if (pr == 1) return 1;
        has_proj[k] = pr == 0;
        if (has_proj[k] && need(p + "proj.b", O)) return 1;
        if (!has_proj[k] && C != O) return vfail("masr_vad_finalize: block " + std::to_string(k) + " needs a projection (C != O)");
        strides[k] = (int)H[p + "out.stride"][0];
        if (strides[k] != 1 && strides[k] != 2) return vfail("masr_vad_finalize: unsupported stride");
        put(p + "dw.w", H[p + "dw.w"]);
        put(p + "dw.b", H[p + "dw.b"]);
        put(p + "pw.wT", transposed(H[p + "pw.w"], O, C));
        put(p + "pw.b", H[p + "pw.b"]);
        if (has_proj[k]) {
            put(p + "proj.wT", transposed(H[p + "proj.w"], O, C));
            put(p + "proj.b", H[p + "proj.b"]);
        }
        put(p + "out.wT", transposed(H[p + "out.w"], O, O));
        put(p + "out.b", H[p + "out.b"]);
    }
    for (int l = 0; l < 2; ++l) {
        const std::string p = "lstm" + std::to_string(l) + ".";
        if (need(p + "W", 256 * 64) || need(p + "R", 256 * 64) || need(p + "b", 256)) return 1;
        put(p + "W", H[p + "W"]);
        put(p + "R", H[p + "R"]);
        put(p + "b", H[p + "b"]);
    }
    put("dec.w", H["dec.w"]);
    if (v->blob[m]) VHIP(hipFree(v->blob[m]));
    v->blob[m] = nullptr;
    VHIP(hipMalloc((void**)&v->blob[m], blob.size() * sizeof(float)));
    VHIP(hipMemcpy(v->blob[m], blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice));
    Net& n = v->net[m];
    const float* base = v->blob[m];
    n.basisT = base + off["basisT"];
    for (int k = 0; k < 7; ++k) n.filt[k] = H["norm_filter"][k];
    for (int k = 0; k < 4; ++k) {
        const std::string p = "b" + std::to_string(k) + ".";
        Block& b = n.blk[k];
        b.C = CH[k];
        b.O = CH[k + 1];
        b.stride = strides[k];
        b.dw_w = base + off[p + "dw.w"];
        b.dw_b = base + off[p + "dw.b"];
        b.pw_wT = base + off[p + "pw.wT"];
        b.pw_b = base + off[p + "pw.b"];
        b.proj_wT = has_proj[k] ? base + off[p + "proj.wT"] : nullptr;
        b.proj_b = has_proj[k] ? base + off[p + "proj.b"] : nullptr;
        b.out_wT = base + off[p + "out.wT"];
        b.out_b = base + off[p + "out.b"];
    }
    for (int l = 0; l < 2; ++l) {
        const std::string p = "lstm" + std::to_string(l) + ".";
        n.lstm_w[l] = base + off[p + "W"];
        n.lstm_r[l] = base + off[p + "R"];
        n.lstm_b[l] = base + off[p + "b"];
    }
    n.dec_w = base + off["dec.w"];
    n.dec_b = H["dec.b"][0];
    v->ready[m] = true;
    H.clear();
    return 0;
}

int masr_vad_forward(masr_vad* v, int32_t sample_rate, const float* audio_dev, int32_t B, int32_t n_win, int32_t window,
                     float* h_dev, float* c_dev, float* probs_dev, void* stream) {
    if (!v || !audio_dev || !h_dev || !c_dev || !probs_dev) return vfail("null argument");
    if (sample_rate != 16000 && sample_rate != 8000) return vfail("Supported sampling rates: [8000, 16000]");
    const int m = sample_rate == 16000 ? 0 : 1;
    if (!v->ready[m]) return vfail("masr_vad_forward: the model of this sample rate was not loaded / finalized");
    if (B <= 0 || n_win <= 0) return 0;
    // any chunk length from sr / 31.25 samples (the graph's own lower bound: 512 at 16 kHz, 256 at 8 kHz) up to 1536; the STFT takes
    // floor(window / 64) frames, the samples behind the last full hop only enter through the right reflection, like the ONNX graph
    if (window < (sample_rate == 16000 ? 512 : 256) || window > WMAX)
        return vfail("window_size_samples must lie in [sample_rate / 31.25, 1536]");
    VHIP(hipSetDevice(v->device));
    const Net& n = v->net[m];
    int F = window / 64;
    for (int k = 0; k < 4; ++k) F = (F + n.blk[k].stride - 1) / n.blk[k].stride;
    const int T = F;
    const size_t need = (size_t)B * n_win * T * 64;
    if (need > v->feats_cap) {
        if (v->feats) VHIP(hipFree(v->feats));
        v->feats = nullptr;
        v->feats_cap = 0;
        VHIP(hipMalloc((void**)&v->feats, (need + need / 4) * sizeof(float)));
        v->feats_cap = need + need / 4;
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(silero_front_kernel, dim3(n_win, B), dim3(256), 0, s, n, audio_dev, n_win, window, v->feats, T);
    hipLaunchKernelGGL(silero_lstm_kernel, dim3(B), dim3(512), 0, s, n, v->feats, n_win, T, h_dev, c_dev, B, probs_dev);
    VHIP(hipGetLastError());
    return 0;
}

}  // extern "C"
