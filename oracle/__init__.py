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
  * HiFi-GAN generator (hifigan_oracle.py): PARITY UNPINNED — the arithmetic
    lives in the third-party, un-vendored, unpinned `parallel-wavegan`
    dependency (reference setup.cfg:17, call sites jatts/vocoder/vocoder.py:
    13,41,43,64); restated from the published HiFi-GAN v1 architecture.
    Vocoder.decode's normalisation (vocoder.py:56-61) IS pinned.
"""
