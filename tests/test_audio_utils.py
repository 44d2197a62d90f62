"""CPU: utils/audio.py - the librosa calls of reference utils/audio.py:7-87 restated with NumPy / SciPy.  librosa is not
installable here and the reference holds no vector for these calls: parity is UNPINNED, the tests check the published
properties of the algorithms."""
import numpy as np
import pytest

from decode_tonal_langauge_amd.utils import audio as au


def test_mel_filterbank_properties():
    sr, n_fft, n_mels = 24414, 2048, 80
    fb = au.mel_filterbank(sr, n_fft, n_mels)
    assert fb.shape == (n_mels, 1 + n_fft // 2) and (fb >= 0).all()
    hz = np.linspace(0, sr / 2, 1 + n_fft // 2)
    centres = (fb * hz).sum(1) / fb.sum(1)
    assert (np.diff(centres) > 0).all()                                   # bands ordered in frequency
    # Slaney normalisation: every triangle has unit area in Hz (integrated on the FFT grid)
    area = fb.sum(1) * (hz[1] - hz[0])
    assert np.allclose(area[5:], 1.0, rtol=0.12)
    # the scale is linear below 1 kHz (equal spacing of the low centres) and logarithmic above
    low = centres[centres < 900]
    assert np.allclose(np.diff(low), np.diff(low)[0], rtol=0.05)
    high = centres[centres > 2000]
    assert np.allclose(np.diff(np.log(high)), np.diff(np.log(high))[0], rtol=0.05)
    # the mel <-> Hz maps invert each other and 1 kHz is 15 mel
    f = np.array([0.0, 440.0, 1000.0, 5000.0])
    assert np.allclose(au._mel_to_hz(au._hz_to_mel(f)), f) and abs(float(au._hz_to_mel(1000.0)) - 15.0) < 1e-12


def test_audio_to_mel_shape_db_reference_and_tone_band():
    sr = 16000
    t = np.arange(sr) / sr
    f0 = 1000.0
    y = 0.5 * np.sin(2 * np.pi * f0 * t)
    kw = {"n_fft": 1024, "hop_length": 256, "n_mels": 40}
    mel = au.audio_to_mel(y, sr, mel_kwargs=kw)
    n_frames = 1 + sr // 256
    assert mel.dtype == np.float32 and mel.shape == (40 * n_frames,)
    m = mel.reshape(40, n_frames)
    assert abs(float(m.max())) < 1e-5 and float(m.min()) >= -80.0 - 1e-4  # ref=np.max -> peak 0 dB, top_db 80 floor
    fb = au.mel_filterbank(sr, 1024, 40)
    hz = np.linspace(0, sr / 2, 513)
    centres = (fb * hz).sum(1) / fb.sum(1)
    assert abs(int(np.argmax(m[:, n_frames // 2])) - int(np.argmin(np.abs(centres - f0)))) <= 1
    lin = au.audio_to_mel(y, sr, mel_in_db=False, mel_kwargs=kw).reshape(40, n_frames)
    assert np.allclose(au.power_to_db(lin, ref=np.max), m, atol=1e-3)
    with pytest.raises(ValueError, match="1D"):
        au.audio_to_mel(np.zeros((2, 100)), sr, mel_kwargs=kw)
    with pytest.raises(TypeError):
        au.audio_to_mel(y, sr, mel_kwargs={"bogus": 1})


def test_stft_istft_round_trip_and_db_power_inverse():
    rng = np.random.default_rng(0)
    y = rng.standard_normal(8000)
    S = au.stft(y, n_fft=512, hop_length=128)
    back = au.istft(S, hop_length=128, length=len(y))
    assert np.allclose(back, y, atol=1e-10)
    p = rng.random((10, 7)) + 1e-3
    assert np.allclose(au.db_to_power(au.power_to_db(p, ref=1.0, top_db=None)), p)
    assert np.allclose(au.db_to_power(np.array([0.0, -10.0]), ref=1e-4), [1e-4, 1e-5])


def test_mel_to_audio_reconstructs_the_spectrum_of_a_tone():
    sr = 8000
    t = np.arange(2 * sr) / sr
    y = 0.3 * np.sin(2 * np.pi * 440.0 * t) + 0.2 * np.sin(2 * np.pi * 1320.0 * t)
    kw = {"n_fft": 512, "hop_length": 128, "n_mels": 64}
    mel = au.audio_to_mel(y, sr, mel_in_db=False, mel_kwargs=kw)
    wave = au.mel_to_audio(mel, 64, audio_sampling_rate=sr, mel_in_db=False, n_fft=512, hop_length=128, length=len(y))
    assert wave.shape == y.shape and wave.dtype == np.float32 and np.isfinite(wave).all()
    spec = lambda v: np.abs(np.fft.rfft(v * np.hanning(len(v))))
    a, b = spec(y), spec(wave)
    f = np.fft.rfftfreq(len(y), 1 / sr)
    top = f[np.argsort(b)[-200:]]
    assert (np.abs(top - 440).min() < 15) and (np.abs(top - 1320).min() < 30)      # both partials come back
    mel2 = au.audio_to_mel(wave, sr, mel_in_db=False, mel_kwargs=kw)
    err = np.linalg.norm(np.sqrt(mel2) - np.sqrt(mel)) / np.linalg.norm(np.sqrt(mel))
    assert err < 0.35                                                              # spectral convergence of Griffin-Lim
    # dB input with the reference's convention (ref = 1e-4)
    db = au.power_to_db(mel.reshape(64, -1), ref=1e-4, top_db=None).reshape(-1)
    w2 = au.mel_to_audio(db, 64, audio_sampling_rate=sr, n_fft=512, hop_length=128, length=len(y))
    assert np.allclose(w2, wave, atol=1e-4)


def test_mel_to_audio_single_frame_is_empty_like_a_centred_istft():
    """One mel frame (the synthetic configurations: output_dim = n_mels = 80) inverts to hop * (frames - 1) = 0 samples
    with a centred STFT - as in librosa; two frames give one hop.  train_synthesizer skips empty waves."""
    rng = np.random.default_rng(3)
    assert au.mel_to_audio(-40.0 + 30.0 * rng.random(80), 80, n_fft=512, hop_length=128, n_iter=2).size == 0
    two = au.mel_to_audio(-40.0 + 30.0 * rng.random(160), 80, n_fft=512, hop_length=128, n_iter=2)
    assert two.shape == (128,) and np.isfinite(two).all()
