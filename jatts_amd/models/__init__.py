"""Registry mirror of ``jatts.models`` (reference models/__init__.py:1-9):
``getattr(jatts_amd.models, config["model_type"])(**config["model_params"])``."""
from .fastspeech2 import FastSpeech2  # noqa: F401
from .vits import VITS  # noqa: F401
from .matchatts import MatchaTTS, MatchaTTS_MAS  # noqa: F401
