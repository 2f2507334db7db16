// Host-side band-limited sinc interpolation (J. O. Smith's algorithm as published in resampy: resampy/interpn.py
// _resample_loop, resampy/core.py resample) -- the resampler behind the reference's AudioSegment.resample
// (masr/data_utils/audio.py:306-317).  Same arithmetic, operation by operation, as the numpy form in
// masr_amd/data_utils/resample.py (which documents the algorithm and the filter table): double index arithmetic, weight =
// win[k] + eta * dwin[k] as a separate multiply and add, every tap added to the float accumulator through double.  An
// input-format step in front of the hot path; third-party algorithm, parity unpinned (resampy is absent from the image).
#include <cstdint>

#include "../../include/masr_hip.h"

#pragma STDC FP_CONTRACT OFF

extern "C" int masr_resample_f32(const float* x, int64_t n_orig, double ratio, const double* win, const double* dwin, int64_t nwin,
                                 int32_t num_table, float* y, int64_t n_out) {
    if (!x || !win || !dwin || !y || n_orig <= 0 || n_out < 0 || nwin <= 0 || num_table <= 0 || !(ratio > 0.0)) return 1;
    const double scale = ratio < 1.0 ? ratio : 1.0;
    const double time_increment = 1.0 / ratio;
    const int64_t index_step = (int64_t)(scale * (double)num_table);
    if (index_step <= 0) return 1;
    for (int64_t t = 0; t < n_out; ++t) {
        const double time_register = (double)t * time_increment;
        const int64_t n = (int64_t)time_register;
        if (n >= n_orig) return 1;
        float acc = 0.f;
        double frac = scale * (time_register - (double)n);
        double index_frac = frac * (double)num_table;
        int64_t offset = (int64_t)index_frac;
        double eta = index_frac - (double)offset;
        int64_t lim = (nwin - offset) / index_step;
        const int64_t i_max = n + 1 < lim ? n + 1 : lim;
        for (int64_t i = 0; i < i_max; ++i) {
            const int64_t k = offset + i * index_step;
            volatile double prod = eta * dwin[k];
            const double weight = win[k] + prod;
            volatile double term = weight * (double)x[n - i];
            acc = (float)((double)acc + term);
        }
        frac = scale - frac;
        index_frac = frac * (double)num_table;
        offset = (int64_t)index_frac;
        eta = index_frac - (double)offset;
        lim = (nwin - offset) / index_step;
        const int64_t k_max = n_orig - n - 1 < lim ? n_orig - n - 1 : lim;
        for (int64_t k2 = 0; k2 < k_max; ++k2) {
            const int64_t k = offset + k2 * index_step;
            volatile double prod = eta * dwin[k];
            const double weight = win[k] + prod;
            volatile double term = weight * (double)x[n + k2 + 1];
            acc = (float)((double)acc + term);
        }
        y[t] = acc;
    }
    return 0;
}
