"""Training-side pieces on the MI355X path -- the first slice of SURVEY §8 f.4 (reference
jatts/trainers/fastspeech2.py:24-100 `_train_step`, jatts/bin/tts_train.py:355-363 DistributedDataParallel).

What is here: (1) the FastSpeech2 criterion (`MelLoss`/`L1Loss`, `DurationPredictorLoss`, `PitchLoss`, `EnergyLoss` of
jatts/losses/) evaluated by HIP kernels on the dict that `FastSpeech2.forward()` returns, i.e. `_train_step` up to `gen_loss`;
(2) the backward of the hot path's work-horse op: `Conv1dFunction`, a torch.autograd.Function whose forward is jatts_conv1d and
whose backward is three HIP launches (dx = jatts_conv1d on flipped / transposed weights, dW = jatts_conv1d_wgrad, db =
jatts_col_sum), in f32; (3) `allreduce_gradients`: bucketed, flat gradient all-reduce over RCCL (what DDP does for the
reference), usable after any backward.  What is NOT here yet: backward passes of the attention / LayerNorm / GLU / GroupNorm
kernels and the train-mode behaviour of the models (dropout, batch-statistics BatchNorm) -- the models' forward() is the
eval-mode arithmetic.  No CPU fallback: CPU tensors raise.
"""
import torch
import torch.distributed as dist

from . import hip
from .models._conformer import PackedConv

L1, L2 = 0, 1


# ------------------------------------------------------------------------------------------ criterion
def _padded(rb_len, B):
    return hip.RaggedBatch([rb_len] * B, torch.device("cuda", torch.cuda.current_device()))


@torch.no_grad()
def fastspeech2_losses(ret, durations, pitch, energy, ilens, use_masking=True):
    """`_train_step`'s loss block (trainers/fastspeech2.py:62-84) on FastSpeech2.forward()'s return dict.
    -> dict(mel_loss, duration_loss, pitch_loss, energy_loss, loss), f32 scalars on the GPU.
    MelLoss = L1(before, ys) + L1(after, ys) over the frames t < olens (mean over selected elements, l1l2_loss.py:43-63);
    DurationPredictorLoss = MSE(d_outs, log(ds + 1)) over tokens t < ilens; Pitch / EnergyLoss = MSE over t < ilens."""
    before, after, ys, olens = ret["before_outs"], ret["after_outs"], ret["ys"], ret["olens"]
    dev = before.device
    B, To, od = before.shape
    Tm = ret["d_outs"].shape[1]
    rbo, rbt = hip.RaggedBatch([To] * B, dev), hip.RaggedBatch([Tm] * B, dev)
    vo = olens.to(device=dev, dtype=torch.int32) if use_masking else None
    vi = ilens.to(device=dev, dtype=torch.int32) if use_masking else None
    n_o = float(int(olens.sum()) if use_masking else B * To) * od
    n_i = float(int(ilens.sum()) if use_masking else B * Tm)
    ys2 = ys.to(dev).float().reshape(B * To, od).contiguous()
    mel = hip.masked_loss(rbo, before.reshape(B * To, od), ys2, vo, L1, 1.0 / n_o)
    if after is not None:
        mel = mel + hip.masked_loss(rbo, after.reshape(B * To, od), ys2, vo, L1, 1.0 / n_o)
    flat = lambda t: t.to(dev).float().reshape(B * Tm, 1).contiguous()  # noqa: E731
    dur = hip.masked_loss(rbt, flat(ret["d_outs"]), flat(durations[:, :Tm]), vi, L2, 1.0 / n_i, log_offset=1.0)
    pit = hip.masked_loss(rbt, flat(ret["p_outs"]), flat(pitch[:, :Tm]), vi, L2, 1.0 / n_i)
    ene = hip.masked_loss(rbt, flat(ret["e_outs"]), flat(energy[:, :Tm]), vi, L2, 1.0 / n_i)
    return dict(mel_loss=mel, duration_loss=dur, pitch_loss=pit, energy_loss=ene, loss=mel + dur + pit + ene)


# ------------------------------------------------------------------------------------------ conv1d with a backward
class Conv1dFunction(torch.autograd.Function):
    """y = conv1d(x) on a packed ragged batch (rows, c_in) -> (rows, n_out), f32; "same"-style geometry via (dil, pad).
    forward: jatts_conv1d.  backward: dx = jatts_conv1d(dy, W'[c][n][k-1-tap], pad' = (k-1) dil - pad),
    dW = jatts_conv1d_wgrad(x, dy), db = jatts_col_sum(dy)."""

    @staticmethod
    def forward(ctx, x, weight, bias, rb, dil, pad):
        n_out, c_in, k = weight.shape
        pc = PackedConv(weight, bias, hip.F32, x.device)
        xin = x.contiguous() if c_in == pc.c_in else hip.affine_cast(x.contiguous(), hip.F32, ldy=pc.c_in)
        y = hip.conv1d(rb, xin, pc.w, pc.c_in, n_out, k, dtype=hip.F32, dil=dil, pad=pad, bias=pc.b)
        ctx.save_for_backward(x, weight)
        ctx.geom = (rb, dil, pad, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        rb, dil, pad, has_bias = ctx.geom
        n_out, c_in, k = weight.shape
        dy = dy.contiguous().float()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            pcb = PackedConv(weight.detach().permute(1, 0, 2).flip(2).contiguous(), None, hip.F32, dy.device)
            dyp = dy if n_out == pcb.c_in else hip.affine_cast(dy, hip.F32, ldy=pcb.c_in)
            dx = hip.conv1d(rb, dyp, pcb.w, pcb.c_in, c_in, k, dtype=hip.F32, dil=dil, pad=(k - 1) * dil - pad)
        if ctx.needs_input_grad[1]:
            dw = hip.conv1d_wgrad(rb, x.detach().contiguous().float(), dy, c_in, n_out, k, dil, pad)
        if has_bias and ctx.needs_input_grad[2]:
            db = hip.col_sum(dy)
        return dx, dw, db, None, None, None


class RaggedConv1d(torch.nn.Module):
    """torch.nn.Conv1d's parameters (weight (n_out, c_in, k), bias) applied to packed ragged rows through Conv1dFunction."""

    def __init__(self, c_in, n_out, k, dilation=1, padding=None, bias=True):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.empty(n_out, c_in, k))
        self.bias = torch.nn.Parameter(torch.zeros(n_out)) if bias else None
        torch.nn.init.xavier_uniform_(self.weight)
        self.dil = dilation
        self.pad = (k - 1) // 2 * dilation if padding is None else padding

    def forward(self, rb, x):
        return Conv1dFunction.apply(x, self.weight, self.bias, rb, self.dil, self.pad)


# ------------------------------------------------------------------------------------------ gradient exchange
def allreduce_gradients(params, group=None, bucket_bytes=64 << 20, average=True):
    """Sum (average) the .grad of ``params`` over the ranks: gradients are packed into flat buckets of ~bucket_bytes (one
    all-reduce each: 64 MiB buckets suit a ring over 7 xGMI links per GPU; FastSpeech2's 281 MB of f32 gradients = 5 collectives)
    and unpacked in place.  What DistributedDataParallel does for the reference (tts_train.py:355-363), minus the overlap with
    backward.  Works on any backend (RCCL on the GPUs, gloo in the CPU tests).  Returns the number of collectives issued."""
    world = dist.get_world_size(group)
    grads = [p.grad for p in params if p.grad is not None]
    if world == 1 or not grads:
        return 0
    n_coll, i = 0, 0
    while i < len(grads):
        j, size = i, 0
        while (j < len(grads) and (j == i or size + grads[j].numel() * grads[j].element_size() <= bucket_bytes)
               and grads[j].dtype == grads[i].dtype and grads[j].device == grads[i].device):
            size += grads[j].numel() * grads[j].element_size()
            j += 1
        flat = torch.cat([g.reshape(-1) for g in grads[i:j]])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat /= world
        o = 0
        for g in grads[i:j]:
            g.copy_(flat[o:o + g.numel()].view_as(g))
            o += g.numel()
        n_coll += 1
        i = j
    return n_coll
