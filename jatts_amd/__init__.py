"""jatts_amd — MI355X-native (gfx950) implementation of the jatts stage-4 hot path:
text2mel ``model.inference`` + HiFi-GAN ``Vocoder.decode`` behind the reference's own
Python interface, with all arithmetic in hand-written HIP kernels (libjatts_hip.so)."""
__version__ = "0.1.0"
