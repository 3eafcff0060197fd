#!/usr/bin/env python3
"""Decoded samples of the two audio files BASELINE.json names (configs[1]: audio/speech_74.wav, configs[3]: audio/stim312_wind.wav) as
bench / test INPUT fixtures: tests/golden/audio_speech_74.npz, tests/golden/audio_stim312_wind.npz (int16 samples + sampling rate; data,
not source).  Container-only: reads /root/reference/audio; the GPU box has the .npz files only.
    python tools/make_audio_fixtures.py
"""
import os
import numpy as np
from scipy.io import wavfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name in ('speech_74', 'stim312_wind'):
    fs, x = wavfile.read('/root/reference/audio/%s.wav' % name)
    assert x.dtype == np.int16 and x.ndim == 1
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'audio_%s.npz' % name), samples=x, fs=np.int32(fs))
    print(name, fs, x.shape)
