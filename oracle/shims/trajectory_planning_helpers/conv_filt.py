import numpy as np


def conv_filt(signal, filt_window, closed):
    """Centred moving average of odd width (tph conv_filt); unclosed signals keep their first/last half window."""
    if not filt_window % 2 == 1:
        raise RuntimeError("Window width of moving average filter must be odd!")
    w_window_half = int((filt_window - 1) / 2)
    if closed:
        signal_tmp = np.concatenate((signal[-w_window_half:], signal, signal[:w_window_half]), axis=0)
        signal_filt = np.convolve(signal_tmp, np.ones(filt_window) / float(filt_window),
                                  mode="same")[w_window_half:-w_window_half]
    else:
        signal_filt = np.copy(signal)
        signal_filt[w_window_half:-w_window_half] = \
            np.convolve(signal, np.ones(filt_window) / float(filt_window), mode="same")[w_window_half:-w_window_half]
    return signal_filt
