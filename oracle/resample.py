"""TEST INFRASTRUCTURE ONLY.  Loop-form restatement of the published resampy algorithm behind the reference's
``AudioSegment.resample`` (masr/data_utils/audio.py:306-317 -> resampy.resample; resampy/interpn.py ``_resample_loop``,
resampy/core.py ``resample``; third-party, absent from the image: **parity unpinned**).  Pure Python loops, small inputs only;
the filter table is the product's regenerated one (masr_amd/data_utils/resample.py: the table is data, the loop is the oracle)."""
import numpy as np


def resample_loop(x, sr_orig, sr_new, win, num_table):
    x = np.asarray(x)
    ratio = float(sr_new) / sr_orig
    n_out = int(x.shape[0] * ratio)
    y = np.zeros(n_out, x.dtype)
    interp_win = ratio * win if ratio < 1 else win
    interp_delta = np.diff(interp_win, append=interp_win[-1])
    scale = min(1.0, ratio)
    time_increment = 1.0 / ratio
    t_out = np.arange(n_out) * time_increment
    index_step = int(scale * num_table)
    nwin, n_orig = interp_win.shape[0], x.shape[0]
    for t in range(n_out):
        time_register = t_out[t]
        n = int(time_register)
        frac = scale * (time_register - n)
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        i_max = min(n + 1, (nwin - offset) // index_step)
        for i in range(i_max):
            weight = interp_win[offset + i * index_step] + eta * interp_delta[offset + i * index_step]
            y[t] += weight * x[n - i]
        frac = scale - frac
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        k_max = min(n_orig - n - 1, (nwin - offset) // index_step)
        for k in range(k_max):
            weight = interp_win[offset + k * index_step] + eta * interp_delta[offset + k * index_step]
            y[t] += weight * x[n + k + 1]
    return y
