"""CPU oracle for the jatts stage-4 hot path (TEST INFRASTRUCTURE ONLY).

Everything under ``oracle/`` is a CPU restatement of the reference algorithm
used as the *checker* for the HIP path.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The product package ``jatts_amd`` never imports ``oracle`` and has
no CPU fallback: it raises if the HIP extension is missing.

Pinning status (see DESIGN.md "Oracle"):
  * FastSpeech2 text2mel (fs2_oracle.py, length_regulator.py): PINNED against
    golden vectors produced by importing the real reference from
    /root/reference (tests/golden/make_golden.py, fixtures in tests/golden/).
  * HiFi-GAN generator (hifigan_oracle.py): the arithmetic lives in the
    third-party, un-vendored, unpinned `parallel-wavegan` dependency (reference
    setup.cfg:17, call sites jatts/vocoder/vocoder.py:13,41,43,64), absent here:
    UNPINNED against that package itself; restated from the published HiFi-GAN
    v1 architecture and PINNED ON AN INDEPENDENT IMPLEMENTATION of it (Hugging
    Face transformers' FastSpeech2ConformerHifiGan run here on the same weights
    and mels: tests/golden/hifigan_xcheck.npz, agreement to 1e-15 in fp64) for
    the 22.05 kHz V1 strides; the odd-stride padding convention stays recalled.
    Vocoder.decode's normalisation (vocoder.py:56-61) IS pinned.
"""
