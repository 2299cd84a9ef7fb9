"""``Vocoder`` — drop-in for jatts.vocoder.vocoder.Vocoder (vocoder.py:16-67).

Same constructor (checkpoint, config, stats, device, trg_stats=None, take_norm_feat=True)
and ``decode(c) -> (y, sampling_rate)`` contract; the generator is jatts_amd's HIP
HiFi-GAN instead of parallel_wavegan's.  The de-normalise / re-normalise step of
``decode`` (vocoder.py:56-61) is folded into ONE per-channel affine executed by the
``jatts_affine_cast`` kernel:  (c*s_t + m_t - m_v)/s_v = c*(s_t/s_v) + (m_t - m_v)/s_v.

For convenience (no network / no h5py here) ``checkpoint`` may also be a state_dict,
``config`` a dict and ``stats`` a dict {"mean","scale"} or an .npz path.
"""
import logging
import time

import numpy as np
import torch
import yaml

from .. import hip
from .hifigan import HiFiGANGenerator


def read_stats(stats):
    if isinstance(stats, dict):
        return np.asarray(stats["mean"], dtype=np.float32), np.asarray(stats["scale"], dtype=np.float32)
    if str(stats).endswith(".npz"):
        z = np.load(stats)
        return z["mean"].astype(np.float32), z["scale"].astype(np.float32)
    try:
        import h5py
    except ImportError as e:  # same failure mode as jatts.utils.read_hdf5 without h5py
        raise ImportError("reading .h5 stats needs h5py; pass a dict or .npz instead") from e
    with h5py.File(stats, "r") as f:
        return f["mean"][()].astype(np.float32), f["scale"][()].astype(np.float32)


class Vocoder(object):
    def __init__(self, checkpoint, config, stats, device, trg_stats=None, take_norm_feat=True):
        self.device = torch.device(device)
        if take_norm_feat:
            assert trg_stats is not None, "trg_stats must be given if take_norm_feat=True"
            self.trg_stats = {
                "mean": torch.tensor(np.asarray(trg_stats["mean"]), dtype=torch.float).to(self.device),
                "scale": torch.tensor(np.asarray(trg_stats["scale"]), dtype=torch.float).to(self.device),
            }
        self.take_norm_feat = take_norm_feat
        if isinstance(config, dict):
            self.config = config
        else:
            with open(config) as f:
                self.config = yaml.load(f, Loader=yaml.Loader)
        gtype = self.config.get("generator_type", "HiFiGANGenerator")
        if gtype != "HiFiGANGenerator":
            raise NotImplementedError(f"generator_type {gtype}: only HiFiGANGenerator is on the HIP path")
        self.model = HiFiGANGenerator(**self.config.get("generator_params", {}))
        if isinstance(checkpoint, dict):
            sd = checkpoint
        else:
            sd = torch.load(checkpoint, map_location="cpu")
            sd = sd["model"]["generator"] if "model" in sd else sd
        self.model.load_state_dict(sd)
        logging.info("Loaded vocoder parameters.")
        self.model.remove_weight_norm()
        self.model = self.model.eval().to(self.device)
        mean, scale = read_stats(stats)
        self.stats = {"mean": torch.tensor(mean, dtype=torch.float).to(self.device),
                      "scale": torch.tensor(scale, dtype=torch.float).to(self.device)}
        # one fused affine for decode(): prepared once (tiny host-side constant folding)
        if take_norm_feat:
            self._scale = (self.trg_stats["scale"] / self.stats["scale"]).contiguous()
            self._shift = ((self.trg_stats["mean"] - self.stats["mean"]) / self.stats["scale"]).contiguous()
        else:
            self._scale = (1.0 / self.stats["scale"]).contiguous()
            self._shift = (-self.stats["mean"] / self.stats["scale"]).contiguous()

    def set_precision(self, precision):
        self.model.set_precision(precision)
        return self

    @torch.no_grad()
    def decode_batch(self, rb, mel):
        """Packed ragged batch: mel f32 (rows, n_mels) -> packed waveform f32 (rows*hop,)."""
        return self.model.inference_batch(rb, mel.float().contiguous(), scale=self._scale, shift=self._shift)

    @torch.no_grad()
    def normalized(self, c):
        """The normalised features fed to the generator (for parity tests of vocoder.py:56-61)."""
        return hip.affine_cast(c.float().contiguous(), hip.F32, scale=self._scale, shift=self._shift)

    @torch.no_grad()
    def decode(self, c):
        c = c.to(self.device)
        start = time.time()
        rb = hip.RaggedBatch([c.shape[0]], self.device)
        y = self.decode_batch(rb, c).view(-1)
        rtf = (time.time() - start) / (len(y) / self.config["sampling_rate"])
        logging.info(f"Finished waveform generation. (RTF = {rtf:.03f}).")
        return y, self.config["sampling_rate"]
