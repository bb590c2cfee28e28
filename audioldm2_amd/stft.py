"""STFT / mel front-end on MI355X: drop-in for `audioldm2.utilities.audio.stft.STFT` and
`TacotronSTFT` (utilities/audio/stft.py:15-178) — same constructor arguments, same buffers
(`forward_basis`, `mel_basis`), same return values and layouts:
    STFT.transform(x [B, T])                 -> (magnitude [B, F, frames], phase [B, F, frames])
    TacotronSTFT.mel_spectrogram(y [B, T])   -> (log-mel [B, n_mel, frames], magnitudes, phases, energy)
Like the reference (stft.py:72: `.cpu()`), results are returned on the host.

Execution: reflect-pad kernel -> the DFT-basis convolution as an implicit GEMM whose A rows are the
overlapping frames of the padded signal (pixel pitch = hop < K = n_fft; nothing is unfolded in HBM)
-> magnitude/phase kernel -> mel projection GEMM with the log-clamp epilogue.

The reference takes `librosa.filters.mel` / `librosa.util.pad_center` from librosa==0.9.2, which is
not vendored; `mel_filterbank` below restates the published Slaney definition (htk=False,
norm='slaney', the 0.9.2 defaults of the positional call at stft.py:145-147).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
from scipy.signal import get_window

from . import ops
from .ops import ACT_LOGCLAMP


def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax) -> np.ndarray:
    """Slaney-scale, Slaney-normalised triangular filterbank, float32 [n_mels, n_fft//2 + 1]."""
    if fmax is None:
        fmax = sr / 2.0
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    w = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float64)
    for i in range(n_mels):
        w[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    w *= (2.0 / (mel_f[2: n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


class STFT(nn.Module):
    def __init__(self, filter_length, hop_length, win_length, window="hann"):
        super().__init__()
        self.filter_length, self.hop_length, self.win_length, self.window = filter_length, hop_length, win_length, window
        fourier_basis = np.fft.fft(np.eye(filter_length))
        cutoff = int(filter_length / 2 + 1)
        fourier_basis = np.vstack([np.real(fourier_basis[:cutoff, :]), np.imag(fourier_basis[:cutoff, :])])
        forward_basis = torch.FloatTensor(fourier_basis[:, None, :])
        if window is not None:
            assert filter_length >= win_length
            fft_window = get_window(window, win_length, fftbins=True)
            lpad = (filter_length - win_length) // 2  # librosa.util.pad_center
            fft_window = np.pad(fft_window, (lpad, filter_length - win_length - lpad))
            forward_basis *= torch.from_numpy(fft_window).float()
        self.register_buffer("forward_basis", forward_basis.float())
        self._pk = None

    def _transform_dev(self, x: torch.Tensor):
        """x [B, T] on the GPU -> (mag [B*frames, ldm] zero-padded rows, phase [B*frames, F], frames)."""
        if self._pk is None:
            self._pk = ops.pack_conv(self.forward_basis[:, 0, :])  # linear layout [N = 2F, K = n_fft]
        assert self.hop_length % 4 == 0 and self.filter_length % 4 == 0
        B, T = x.shape
        sig = ops.reflect_pad_1d(x.float().contiguous(), self.filter_length // 2)
        frames = (T + 2 * (self.filter_length // 2) - self.filter_length) // self.hop_length + 1
        spec = ops.frames_gemm(sig, frames, self.hop_length, self._pk)  # [B, frames, 2F]
        F = self.filter_length // 2 + 1
        ldm = (F + 3) // 4 * 4
        mag, phase = ops.mag_phase(spec.view(B * frames, 2 * F), F, ldm)
        return mag, phase, frames

    @torch.no_grad()
    def transform(self, input_data):
        """stft.py:52-81"""
        B = input_data.size(0)
        F = self.filter_length // 2 + 1
        mag, phase, frames = self._transform_dev(input_data.to("cuda"))
        magc = ops.nhwc_to_nchw(mag.view(B, frames, 1, -1)).view(B, -1, frames)[:, :F]
        phc = ops.nhwc_to_nchw(phase.view(B, frames, 1, F)).view(B, F, frames)
        return magc.cpu(), phc.cpu()


class TacotronSTFT(nn.Module):
    def __init__(self, filter_length, hop_length, win_length, n_mel_channels, sampling_rate, mel_fmin, mel_fmax):
        super().__init__()
        self.n_mel_channels = n_mel_channels
        self.sampling_rate = sampling_rate
        self.stft_fn = STFT(filter_length, hop_length, win_length)
        mel_basis = torch.from_numpy(mel_filterbank(sampling_rate, filter_length, n_mel_channels, mel_fmin, mel_fmax)).float()
        self.register_buffer("mel_basis", mel_basis)
        self._pk = None

    @torch.no_grad()
    def mel_spectrogram(self, y, normalize_fun=torch.log):
        """stft.py:159-178 (normalize_fun must be torch.log: the clamp+log is a GEMM epilogue)."""
        assert normalize_fun is torch.log
        assert torch.min(y.data) >= -1, torch.min(y.data)
        assert torch.max(y.data) <= 1, torch.max(y.data)
        B = y.size(0)
        F = self.stft_fn.filter_length // 2 + 1
        mag, phase, frames = self.stft_fn._transform_dev(y.to("cuda"))
        ldm = mag.shape[1]
        if self._pk is None:
            mb = torch.zeros(self.n_mel_channels, ldm)
            mb[:, :F] = self.mel_basis.detach().cpu()
            self._pk = ops.pack_conv(mb)
        mel = ops.linear(mag, self._pk, act=ACT_LOGCLAMP, act_slope=1e-5)  # [B*frames, n_mel]
        energy = ops.row_l2norm(mag, F).view(B, frames)
        melc = ops.nhwc_to_nchw(mel.view(B, frames, 1, -1)).view(B, -1, frames)
        magc = ops.nhwc_to_nchw(mag.view(B, frames, 1, ldm)).view(B, ldm, frames)[:, :F]
        phc = ops.nhwc_to_nchw(phase.view(B, frames, 1, F)).view(B, F, frames)
        return melc.cpu(), magc.cpu(), phc.cpu(), energy.cpu()
