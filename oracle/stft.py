"""Oracle: STFT / mel front-end restated on the CPU (numpy bases + torch conv1d).  TEST INFRA ONLY.

  STFT.__init__ / transform        utilities/audio/stft.py:18-50 / 52-81
  TacotronSTFT.mel_spectrogram     stft.py:159-178
  dynamic_range_compression        utilities/audio/audio_processing.py:85-91
Third-party arithmetic (librosa==0.9.2, not vendored, not installed here — parity UNPINNED for it):
  librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with 0.9.2 defaults htk=False, norm='slaney'
  (call site stft.py:145-147) and librosa.util.pad_center (stft.py:42) are restated below from their
  published definitions (Slaney auditory-toolbox mel scale: linear below 1 kHz at 200/3 Hz per mel,
  log above with step ln(6.4)/27; triangular filters normalised by 2/(f[m+2]-f[m])).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F
from scipy.signal import get_window


def hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    with np.errstate(divide="ignore"):
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax) -> np.ndarray:
    """librosa 0.9.2 `filters.mel(..., htk=False, norm='slaney')` -> float32 [n_mels, n_fft//2+1]."""
    if fmax is None:
        fmax = sr / 2.0
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2: n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights.astype(np.float32)


def stft_forward_basis(filter_length: int, win_length: int, window: str = "hann") -> np.ndarray:
    """stft.py:26-46: rows [real(0..F-1) ; imag(0..F-1)] of fft(eye(N)), times the periodic window
    centre-padded to filter_length.  float32 [2F, N]."""
    fourier_basis = np.fft.fft(np.eye(filter_length))
    cutoff = int(filter_length / 2 + 1)
    fb = np.vstack([np.real(fourier_basis[:cutoff, :]), np.imag(fourier_basis[:cutoff, :])])
    basis = torch.FloatTensor(fb)
    win = get_window(window, win_length, fftbins=True)
    lpad = (filter_length - win_length) // 2  # librosa.util.pad_center
    win = np.pad(win, (lpad, filter_length - win_length - lpad))
    basis = basis * torch.from_numpy(win).float()
    return basis.numpy()


@torch.no_grad()
def stft_transform(x: torch.Tensor, filter_length: int, hop_length: int, win_length: int):
    """stft.py:52-81: x [B, T] -> (magnitude [B, F, frames], phase [B, F, frames])."""
    basis = torch.from_numpy(stft_forward_basis(filter_length, win_length))[:, None, :]
    xp = F.pad(x[:, None, None, :], (filter_length // 2, filter_length // 2, 0, 0), mode="reflect")[:, 0]
    ft = F.conv1d(xp, basis, stride=hop_length, padding=0)
    cutoff = filter_length // 2 + 1
    re, im = ft[:, :cutoff], ft[:, cutoff:]
    return torch.sqrt(re ** 2 + im ** 2), torch.atan2(im, re)


@torch.no_grad()
def mel_spectrogram(x: torch.Tensor, filter_length=1024, hop_length=160, win_length=1024, n_mel=64,
                    sampling_rate=16000, mel_fmin=0, mel_fmax=8000):
    """stft.py:159-178: returns (log-mel [B, n_mel, frames], magnitudes, phases, energy)."""
    assert float(x.min()) >= -1 and float(x.max()) <= 1
    mag, phase = stft_transform(x, filter_length, hop_length, win_length)
    mel_basis = torch.from_numpy(mel_filterbank(sampling_rate, filter_length, n_mel, mel_fmin, mel_fmax))
    mel = torch.matmul(mel_basis, mag)
    mel = torch.log(torch.clamp(mel, min=1e-5))  # audio_processing.py:85-91 (C = 1)
    energy = torch.norm(mag, dim=1)
    return mel, mag, phase, energy
