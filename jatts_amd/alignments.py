"""Alignment learning on the HIP path: drop-in for ``jatts.modules.alignments`` (reference file
jatts/modules/alignments.py) -- ``AlignmentModule`` (:12-60) and ``viterbi_decode`` (:281-310, built on the numba
``_monotonic_alignment_search`` :63-93).  Forward only (the stage-4 neighbours of SURVEY 8(f).1): the reference runs the
search per utterance in a host loop with a device->host copy of every attention matrix; here the whole padded batch is one
launch (jatts_mas_viterbi) and nothing leaves the GPU.
"""
import torch

from . import hip
from .models._conformer import PackedConv


class AlignmentModule(torch.nn.Module):
    """Same constructor, parameter names and ``forward(text, feats, x_masks)`` contract as the reference class."""

    def __init__(self, adim, odim):
        super().__init__()
        self.adim, self.odim = adim, odim
        self.t_conv1 = torch.nn.Conv1d(adim, adim, kernel_size=3, padding=1)
        self.t_conv2 = torch.nn.Conv1d(adim, adim, kernel_size=1, padding=0)
        self.f_conv1 = torch.nn.Conv1d(odim, adim, kernel_size=3, padding=1)
        self.f_conv2 = torch.nn.Conv1d(adim, adim, kernel_size=3, padding=1)
        self.f_conv3 = torch.nn.Conv1d(adim, adim, kernel_size=1, padding=0)
        self._prep = None

    def _prepare(self):
        names = ("t_conv1", "t_conv2", "f_conv1", "f_conv2", "f_conv3")
        ver = tuple(p._version for n in names for p in (getattr(self, n).weight, getattr(self, n).bias))
        if self._prep is not None and self._prep.get("_ver") != ver:   # in-place parameter updates (an optimiser step) too
            self._prep = None
        if self._prep is None:
            dev = self.t_conv1.weight.device
            if dev.type != "cuda":
                raise hip._abi.JattsHipError("jatts_amd.AlignmentModule runs on the GPU only (no CPU fallback)")
            self._prep = {n: PackedConv(getattr(self, n).weight.detach(), getattr(self, n).bias.detach(), hip.F32, dev)
                          for n in names}
            self._prep["_ver"] = ver
        return self._prep

    def _apply(self, fn, *a, **k):
        self._prep = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._prep = None
        return super().load_state_dict(*a, **k)

    @torch.no_grad()
    def forward_ragged(self, rb_t, text, rb_f, feats):
        """Packed rows: text f32 (token rows, adim), feats f32 (frame rows, odim) -> log_p f32 (frame rows, ld)."""
        P = self._prepare()

        def conv(rb, x, name, relu):
            p = P[name]
            c_in = p.c_in
            if x.shape[1] != c_in:   # zero-pad channels to the packed width
                xp = torch.zeros(x.shape[0], c_in, dtype=torch.float32, device=x.device)
                xp[:, : x.shape[1]] = x
                x = xp
            return hip.conv1d(rb, x.contiguous(), p.w, c_in, p.n_out, p.k, dtype=hip.F32, bias=p.b,
                              act=hip.ACT_RELU if relu else hip.ACT_NONE)
        t = conv(rb_t, conv(rb_t, text.float(), "t_conv1", True), "t_conv2", False)
        f = conv(rb_f, conv(rb_f, conv(rb_f, feats.float(), "f_conv1", True), "f_conv2", True), "f_conv3", False)
        return hip.alignment_logp(rb_f, rb_t, f, t, self.adim)

    @torch.no_grad()
    def forward(self, text, feats, x_masks=None, feats_lengths=None):
        """text (B, T_text, adim), feats (B, T_feats, odim), x_masks (B, T_text) True = pad -> (B, T_feats, T_text).
        The reference scores every feats row, padded or not, and its k=3 convolutions read across the padding
        (SURVEY 8a note N1); pass ``feats_lengths`` (additive argument) to keep each utterance equal to its B=1 result:
        rows past the length are then returned as 0 (they are never used by viterbi_decode)."""
        B, Tt, _ = text.shape
        Tf = feats.shape[1]
        tl = [Tt] * B if x_masks is None else (~x_masks).sum(1).tolist()
        fl = [Tf] * B if feats_lengths is None else [int(v) for v in feats_lengths]
        rb_t, rb_f = hip.RaggedBatch(tl, text.device), hip.RaggedBatch(fl, text.device)
        tp = torch.cat([text[b, : tl[b]] for b in range(B)])
        fp = torch.cat([feats[b, : fl[b]] for b in range(B)])
        lp = self.forward_ragged(rb_t, tp, rb_f, fp)
        out = torch.zeros((B, Tf, Tt), dtype=torch.float32, device=text.device)
        w = min(lp.shape[1], Tt)
        o = 0
        for b in range(B):
            out[b, : fl[b], :] = float("-inf")
            out[b, : fl[b], :w] = lp[o:o + fl[b], :w]
            o += fl[b]
        return out


def pack_alignment_convs(sd, device, prefix="alignment_module."):
    """The five convolutions of AlignmentModule (alignments.py:19-25) out of a model state_dict, packed for jatts_conv1d (f32)."""
    return {n: PackedConv(sd[f"{prefix}{n}.weight"], sd[f"{prefix}{n}.bias"], hip.F32, device)
            for n in ("t_conv1", "t_conv2", "f_conv1", "f_conv2", "f_conv3")}


@torch.no_grad()
def padded_alignment(convs, hs, ys, ilens, olens, adim):
    """AlignmentModule.forward + viterbi_decode on a PADDED batch, as the models' forward() passes call them (alignments.py:26-60,
    281-310): hs f32 (B*Tm, adim) padded text encodings, ys f32 (B, To, odim) padded features.  The module's k=3 convolutions
    read across the padding (as in the reference), the softmax runs over each utterance's valid tokens (x_masks -> -inf), every
    frame row is scored.  -> (log_p_attn (B, To, Tm) with -inf at padded tokens, ds (B, Tm) float, bin_loss)."""
    dev = hs.device
    B, To, od = ys.shape
    Tm = hs.shape[0] // B

    def conv(rb, x, n, relu):
        pc = convs[n]
        if x.shape[1] != pc.c_in:
            x = hip.affine_cast(x, hip.F32, ldy=pc.c_in)
        return hip.conv1d(rb, x, pc.w, pc.c_in, pc.n_out, pc.k, dtype=hip.F32, bias=pc.b, act=hip.ACT_RELU if relu else hip.ACT_NONE)
    rbt, rbf = hip.RaggedBatch([Tm] * B, dev), hip.RaggedBatch([To] * B, dev)
    tf = conv(rbt, conv(rbt, hs, "t_conv1", True), "t_conv2", False)
    ff = conv(rbf, conv(rbf, conv(rbf, ys.reshape(B * To, od).contiguous(), "f_conv1", True), "f_conv2", True), "f_conv3", False)
    rbv = hip.RaggedBatch(ilens, dev)                              # valid tokens, packed (row selection: plumbing)
    sel = torch.cat([torch.arange(b * Tm, b * Tm + ilens[b], device=dev) for b in range(B)])
    lp3 = hip.alignment_logp(rbf, rbv, ff, tf.index_select(0, sel).contiguous(), adim).view(B, To, -1)
    log_p_attn = torch.full((B, To, Tm), float("-inf"), dtype=torch.float32, device=dev)
    for b in range(B):
        log_p_attn[b, :, : ilens[b]] = lp3[b, :, : ilens[b]]
    ds, bin_loss = viterbi_decode(log_p_attn, ilens, olens)
    return log_p_attn, ds, bin_loss


@torch.no_grad()
def viterbi_decode(log_p_attn, text_lengths, feats_lengths, k=None):
    """Reference signature (alignments.py:281): (B, T_feats, T_text) log-probabilities -> (ds (B, T_text) float, bin_loss)."""
    B, Tf, Tt = log_p_attn.shape
    dev = log_p_attn.device
    tl = [int(v) for v in text_lengths]
    fl = [int(v) for v in feats_lengths]
    rb_t, rb_f = hip.RaggedBatch(tl, dev), hip.RaggedBatch(fl, dev)
    ld = hip.round_up(Tt, 8)
    lp = torch.zeros(rb_f.total, ld, dtype=torch.float32, device=dev)
    o = 0
    for b in range(B):   # pack the frame rows (plumbing)
        lp[o:o + fl[b], :Tt] = log_p_attn[b, : fl[b]].float()
        o += fl[b]
    path, dur, score = hip.mas_viterbi(rb_f, rb_t, lp)
    ds = torch.zeros((B, Tt), device=dev)
    o = 0
    for b in range(B):
        ds[b, : tl[b]] = dur[o:o + tl[b]].float()
        o += tl[b]
    bin_loss = -(score / hip.h2d(fl, torch.float64, dev)).sum() / B
    return ds, bin_loss.float()


_SEL_CACHE = {}


def frame_token_indices(text_lengths, feats_lengths, Tt, Tf, device):
    """Row indices of the valid tokens / frames of a padded batch, as device int64 tensors (one host -> device copy each; build them
    BEFORE queueing GPU work: a mid-forward torch.tensor(..., device=...) stalls the host behind everything already queued)."""
    key = (tuple(int(n) for n in text_lengths), tuple(int(n) for n in feats_lengths), int(Tt), int(Tf), str(device))
    hit = _SEL_CACHE.get(key)          # constant per length bucket: building two 4 k / 25 k-element Python lists costs ms per step
    if hit is None:
        tsel = hip.h2d([b * Tt + i for b, n in enumerate(key[0]) for i in range(n)], torch.int64, device)
        fsel = hip.h2d([b * Tf + t for b, n in enumerate(key[1]) for t in range(n)], torch.int64, device)
        if len(_SEL_CACHE) >= 64:
            _SEL_CACHE.pop(next(iter(_SEL_CACHE)))
        _SEL_CACHE[key] = hit = (tsel, fsel)
    hip.keep(hit[0])          # (a graph being captured pins them: the cache evicts)
    hip.keep(hit[1])
    return hit


@torch.no_grad()
def viterbi_path(log_p_attn, text_lengths, feats_lengths, tsel=None, fsel=None):
    """Monotonic alignment search on a padded (B, T_feats, T_text) matrix -> (ds (B, T_text) float, path (B, T_feats) int64: the
    token index of every valid frame, 0 at padded frames).  One jatts_mas_viterbi launch plus three row gathers / scatters; used by
    the training forward, which needs the path itself for the binarisation loss (alignments.py:303-308)."""
    B, Tf, Tt = log_p_attn.shape
    dev = log_p_attn.device
    tl = [int(v) for v in text_lengths]
    fl = [int(v) for v in feats_lengths]
    if tsel is None or fsel is None:
        tsel, fsel = frame_token_indices(tl, fl, Tt, Tf, dev)
    rb_t, rb_f = hip.RaggedBatch(tl, dev), hip.RaggedBatch(fl, dev)
    ld = hip.round_up(Tt, 8)
    lp = torch.zeros(rb_f.total, ld, dtype=torch.float32, device=dev)
    lp[:, :Tt] = log_p_attn.reshape(B * Tf, Tt).index_select(0, fsel).float()
    path, dur, _ = hip.mas_viterbi(rb_f, rb_t, lp)
    ds = torch.zeros(B * Tt, device=dev).index_copy_(0, tsel, dur.float()).view(B, Tt)
    pth = torch.zeros(B * Tf, dtype=torch.int64, device=dev).index_copy_(0, fsel, path).view(B, Tf)
    return ds, pth
