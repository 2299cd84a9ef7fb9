from .hifigan import HiFiGANGenerator  # noqa: F401
from .vocoder import Vocoder  # noqa: F401
