"""AudioToMelSpectrogramPreprocessor (nemo/collections/asr/modules/audio_preprocessing.py:56-254):
same constructor arguments; forward(input_signal, length) -> (features [B, n_mels, T_pad], feature lengths)."""
import torch
import torch.nn as nn

from nemo.collections.asr.parts.features import FilterbankFeatures


class AudioToMelSpectrogramPreprocessor(nn.Module):
    def __init__(self, sample_rate=16000, window_size=0.02, window_stride=0.01, n_window_size=None,
                 n_window_stride=None, window="hann", normalize="per_feature", n_fft=None, preemph=0.97, features=64,
                 lowfreq=0, highfreq=None, log=True, log_zero_guard_type="add", log_zero_guard_value=2 ** -24,
                 dither=1e-5, pad_to=16, frame_splicing=1, stft_exact_pad=False, stft_conv=False, pad_value=0,
                 mag_power=2.0):
        super().__init__()
        if window_size and n_window_size:
            raise ValueError(f"{self} received both window_size and n_window_size. Only one should be specified.")
        if window_stride and n_window_stride:
            raise ValueError(f"{self} received both window_stride and n_window_stride. Only one should be specified.")
        if window_size:
            n_window_size = int(window_size * sample_rate)
        if window_stride:
            n_window_stride = int(window_stride * sample_rate)
        self._sample_rate = sample_rate
        self.featurizer = FilterbankFeatures(
            sample_rate=sample_rate, n_window_size=n_window_size, n_window_stride=n_window_stride, window=window,
            normalize=normalize, n_fft=n_fft, preemph=preemph, nfilt=features, lowfreq=lowfreq, highfreq=highfreq,
            log=log, log_zero_guard_type=log_zero_guard_type, log_zero_guard_value=log_zero_guard_value,
            dither=dither, pad_to=pad_to, frame_splicing=frame_splicing, stft_exact_pad=stft_exact_pad,
            stft_conv=stft_conv, pad_value=pad_value, mag_power=mag_power)

    @torch.no_grad()
    def forward(self, input_signal, length):
        return self.get_features(input_signal, length)

    def get_features(self, input_signal, length):
        return self.featurizer(input_signal, length)

    @property
    def filter_banks(self):
        return self.featurizer.filter_banks
