

def test_load_wav_accepts_every_pcm_width(tmp_path):
    """jatts_amd.spkemb.load_wav (torchaudio.load's contract for PCM wavs): 8 / 16 / 24 / 32-bit files all come back as float32 in
    [-1, 1) with the same values to their own quantisation (only 16-bit was accepted)."""
    import wave

    import numpy as np
    from jatts_amd.spkemb import load_wav
    t = np.arange(400) / 16000.0
    y = 0.6 * np.sin(2 * np.pi * 440 * t)
    for width, scale, tol in ((1, 128.0, 1.0 / 128), (2, 32768.0, 1.0 / 32768), (3, 8388608.0, 1e-6), (4, 2147483648.0, 1e-6)):
        q = np.round(y * (scale - 1)).astype(np.int64)
        if width == 1:
            raw = (q + 128).astype(np.uint8).tobytes()
        elif width == 3:
            raw = b"".join(int(v).to_bytes(3, "little", signed=True) for v in q)
        else:
            raw = q.astype({2: "<i2", 4: "<i4"}[width]).tobytes()
        p = tmp_path / f"w{width}.wav"
        with wave.open(str(p), "wb") as w:
            w.setnchannels(1)
            w.setsampwidth(width)
            w.setframerate(16000)
            w.writeframes(raw)
        x, sr = load_wav(str(p))
        assert sr == 16000 and x.dtype.is_floating_point and x.numel() == 400
        assert float(np.abs(x.numpy() - y).max()) <= 2 * tol, width
