"""FilterbankFeatures: waveform -> normalised log-mel (nemo/collections/asr/parts/features.py:192-397,
normalize_batch :53-67).  Host PyTorch implementation for CPU / calibration use; on an MI355X the same
maths runs in the HIP front-end kernels (qasr_frontend_mel) when the model is in engine mode."""
import math

import torch
import torch.nn as nn

from qasr.melbank import mel_filterbank

CONSTANT = 1e-5


def normalize_batch(x, seq_len, normalize_type):
    if normalize_type == "per_feature":
        if int(seq_len.min()) <= 1:
            raise ValueError("normalize_batch with `per_feature` normalize_type received a tensor of length 1. "
                             "This will result in torch.std() returning nan")
        mean = torch.stack([x[i, :, :int(seq_len[i])].mean(dim=1) for i in range(x.shape[0])])
        std = torch.stack([x[i, :, :int(seq_len[i])].std(dim=1) for i in range(x.shape[0])]) + CONSTANT
        return (x - mean.unsqueeze(2)) / std.unsqueeze(2)
    if normalize_type == "all_features":
        mean = torch.stack([x[i, :, :int(seq_len[i])].mean() for i in range(x.shape[0])])
        std = torch.stack([x[i, :, :int(seq_len[i])].std() for i in range(x.shape[0])]) + CONSTANT
        return (x - mean.view(-1, 1, 1)) / std.view(-1, 1, 1)
    return x


class FilterbankFeatures(nn.Module):
    def __init__(self, sample_rate=16000, n_window_size=320, n_window_stride=160, window="hann",
                 normalize="per_feature", n_fft=None, preemph=0.97, nfilt=64, lowfreq=0, highfreq=None, log=True,
                 log_zero_guard_type="add", log_zero_guard_value=2 ** -24, dither=CONSTANT, pad_to=16,
                 max_duration=16.7, frame_splicing=1, stft_exact_pad=False, stft_conv=False, pad_value=0,
                 mag_power=2.0):
        super().__init__()
        if not (isinstance(n_window_size, int) and isinstance(n_window_stride, int) and n_window_size > 0
                and n_window_stride > 0):
            raise ValueError(f"{self} got an invalid value for either n_window_size or n_window_stride. "
                             "Both must be positive ints.")
        if stft_conv or stft_exact_pad or frame_splicing != 1 or log_zero_guard_type not in ('add', 'clamp'):
            raise NotImplementedError('only the torch.stft / centre-padded configuration of QuartzNet & Jasper')
        self.win_length, self.hop_length = n_window_size, n_window_stride
        self.n_fft = n_fft or 2 ** math.ceil(math.log2(self.win_length))
        wins = {'hann': torch.hann_window, 'hamming': torch.hamming_window, 'blackman': torch.blackman_window,
                'bartlett': torch.bartlett_window}
        self.register_buffer("window", wins[window](self.win_length, periodic=False) if window in wins else
                             torch.ones(self.win_length))
        self.normalize, self.log, self.dither, self.nfilt = normalize, log, dither, nfilt
        self.preemph, self.pad_to, self.pad_value, self.mag_power = preemph, pad_to, pad_value, mag_power
        self.log_zero_guard_type, self.log_zero_guard_value = log_zero_guard_type, log_zero_guard_value
        highfreq = highfreq or sample_rate / 2
        self.register_buffer("fb", torch.from_numpy(mel_filterbank(sample_rate, self.n_fft, nfilt, lowfreq, highfreq))
                             .unsqueeze(0))
        max_length = self.get_seq_len(torch.tensor(max_duration * sample_rate, dtype=torch.float))
        self.max_length = max_length + (pad_to - (max_length % pad_to) if pad_to > 0 else 0)

    def get_seq_len(self, seq_len):
        return torch.ceil(seq_len / self.hop_length).to(dtype=torch.long)

    @property
    def filter_banks(self):
        return self.fb

    def _guard(self, x):
        v = self.log_zero_guard_value
        if isinstance(v, str):
            v = {'tiny': torch.finfo(x.dtype).tiny, 'eps': torch.finfo(x.dtype).eps}[v]
        return v

    @torch.no_grad()
    def forward(self, x, seq_len):
        seq_len = self.get_seq_len(seq_len.float())
        if self.dither > 0:
            x = x + self.dither * torch.randn_like(x)
        if self.preemph is not None:
            x = torch.cat((x[:, :1], x[:, 1:] - self.preemph * x[:, :-1]), dim=1)
        spec = torch.stft(x.float(), n_fft=self.n_fft, hop_length=self.hop_length, win_length=self.win_length,
                          center=True, window=self.window.to(dtype=torch.float), return_complex=True)
        x = torch.sqrt(spec.real.pow(2) + spec.imag.pow(2))
        if self.mag_power != 1.0:
            x = x.pow(self.mag_power)
        x = torch.matmul(self.fb.to(x.dtype), x)
        if self.log:
            g = self._guard(x)
            x = torch.log(x + g) if self.log_zero_guard_type == "add" else torch.log(torch.clamp(x, min=g))
        if self.normalize:
            x = normalize_batch(x, seq_len, normalize_type=self.normalize)
        keep = torch.arange(x.size(-1), device=x.device).unsqueeze(0) < seq_len.unsqueeze(1)
        x = torch.where(keep.unsqueeze(1), x, torch.full_like(x, float(self.pad_value)))
        if self.pad_to == "max":
            x = nn.functional.pad(x, (0, self.max_length - x.size(-1)), value=self.pad_value)
        elif self.pad_to > 0 and x.size(-1) % self.pad_to:
            x = nn.functional.pad(x, (0, self.pad_to - x.size(-1) % self.pad_to), value=self.pad_value)
        return x, seq_len
