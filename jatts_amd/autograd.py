"""torch.autograd.Function wrappers around the HIP forward / backward kernels of the training path.

torch.autograd is the tape (it orders the backward calls and sums fan-out gradients); every layer's arithmetic, forward and
backward, is a HIP kernel from csrc/train_ops.hip / training.hip / the MFMA conv -- except the dense [T x T] products of the
attention, which are plain batched GEMMs: jatts_bgemm (csrc/bgemm.hip, exact-f32 MFMA; round 4 -- they went to rocBLAS before; what still does
is the FLOAT64 backward of the alignment module's distance matrix, AlignLogProb.backward, in the MAS-phase trainers).  All tensors are f32 and packed
(rows, channels) row-major with a RaggedBatch describing the sequences; a padded batch is a RaggedBatch of equal lengths.
Each Function names the reference module whose autograd it stands for.  No CPU fallback.
"""
import torch

from . import hip
from .training import Conv1dFunction  # noqa: F401  (re-exported: the conv / linear op)

LN_EPS = 1e-12
BN_EPS = 1e-5


class LayerNorm(torch.autograd.Function):
    """modules/transformer/layer_norm.py:12-42 over the channel axis (eps 1e-12)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = x.contiguous()
        ctx.save_for_backward(x, gamma)
        ctx.eps = eps
        return hip.layernorm(x, gamma.detach().contiguous(), beta.detach().contiguous(), hip.F32, eps)

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        dx, dg, db = hip.layernorm_bwd(x, dy.contiguous(), gamma.detach().contiguous(), ctx.eps, ctx.needs_input_grad[0],
                                       ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        return dx, dg, db, None


class Act(torch.autograd.Function):
    """ReLU / tanh / Swish (modules/conformer/swish.py), element-wise."""

    @staticmethod
    def forward(ctx, x, mode):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.mode = mode
        return hip.act_fwd(x, mode)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return hip.act_bwd(x, dy.contiguous(), ctx.mode), None


class GLU(torch.autograd.Function):
    """F.glu over the channel halves (modules/conformer/convolution.py:66)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return hip.glu_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return hip.glu_bwd(x, dy.contiguous())


class GroupNorm(torch.autograd.Function):
    """torch.nn.GroupNorm(groups, C) on each sequence of the (padded) batch (matchatts/decoder.py:66-78)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rb, groups, eps):
        x = x.contiguous()
        y, mean, rstd = hip.groupnorm_fwd(rb, x, groups, gamma.detach().contiguous(), beta.detach().contiguous(), eps)
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.rb, ctx.groups = rb, groups
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        dx, dg, db = hip.groupnorm_bwd(ctx.rb, x, dy.contiguous(), ctx.groups, gamma.detach().contiguous(), mean, rstd,
                                       ctx.needs_input_grad[0], ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        return dx, dg, db, None, None, None


class SnakeBeta(torch.autograd.Function):
    """SnakeBeta activation with log-scale alpha / beta (matchatts/transformer.py:84-102), after its Linear."""

    @staticmethod
    def forward(ctx, x, alpha, beta):
        x = x.contiguous()
        ctx.save_for_backward(x, alpha, beta)
        return hip.snakebeta_fwd(x, alpha.detach().contiguous(), beta.detach().contiguous())

    @staticmethod
    def backward(ctx, dy):
        x, alpha, beta = ctx.saved_tensors
        return hip.snakebeta_bwd(x, dy.contiguous(), alpha.detach().contiguous(), beta.detach().contiguous())


class DepthwiseConv(torch.autograd.Function):
    """nn.Conv1d(C, C, K, padding=(K-1)//2, groups=C) (convolution.py:44-52) on packed rows; weight (C, 1, K)."""

    @staticmethod
    def forward(ctx, x, weight, bias, rb):
        x = x.contiguous()
        Cc, _, K = weight.shape
        ctx.save_for_backward(x, weight)
        ctx.rb, ctx.pad, ctx.has_bias = rb, (K - 1) // 2, bias is not None
        return hip.dwconv(rb, x, weight.detach().reshape(Cc, K).contiguous(), None if bias is None else bias.detach().contiguous(), ctx.pad)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        Cc, _, K = weight.shape
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = hip.dwconv(ctx.rb, dy, weight.detach().reshape(Cc, K).contiguous(), None, K - 1 - ctx.pad, flip=True)
        if ctx.needs_input_grad[1]:
            dw = hip.dwconv_wgrad(ctx.rb, x, dy, K, ctx.pad).view(Cc, 1, K)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = hip.col_sum(dy)
        return dx, dw, db, None


class BatchNormTrain(torch.autograd.Function):
    """nn.BatchNorm1d in train mode: statistics over every row of the (padded) batch, biased variance for the normalisation;
    running_mean / running_var (unbiased) are updated in place with ``momentum`` like torch does."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps):
        x = x.contiguous()
        n = x.shape[0]
        s, _ = hip.col_stats(x)
        mean = s / n
        _, q = hip.col_stats(x, shift=mean)           # second pass on centred values (no E[x^2] - E[x]^2 cancellation)
        var = q / n
        rstd = torch.rsqrt(var + eps)
        if running_mean is not None:
            with torch.no_grad():
                running_mean.mul_(1.0 - momentum).add_(mean, alpha=momentum)
                running_var.mul_(1.0 - momentum).add_(var * (n / max(n - 1, 1)), alpha=momentum)
        scale = gamma.detach() * rstd
        y = hip.affine_cast(x, hip.F32, scale=scale.contiguous(), shift=(beta.detach() - mean * scale).contiguous())
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        s_dy, s_dyx = hip.col_stats(x, y2=dy, shift=mean, mul=rstd)
        dx = hip.bn_bwd_apply(x, dy, mean, rstd, gamma.detach().contiguous(), s_dy, s_dyx) if ctx.needs_input_grad[0] else None
        return dx, s_dyx, s_dy, None, None, None, None


class Embedding(torch.autograd.Function):
    """nn.Embedding(padding_idx) followed by the x * sqrt(adim) of the positional encoding (encoder.py:133-137,
    positional_encoding.py:221-235): rows = table[ids] * scale."""

    @staticmethod
    def forward(ctx, ids, table, scale, padding_idx):
        ctx.save_for_backward(ids)
        ctx.scale, ctx.pad, ctx.n = scale, padding_idx, table.shape[0]
        return hip.embed_scale(ids, table.detach().contiguous(), scale)

    @staticmethod
    def backward(ctx, dy):
        (ids,) = ctx.saved_tensors
        return None, hip.index_add_rows(dy.contiguous(), ids, ctx.n, ctx.scale, ctx.pad), None, None


class LengthRegulate(torch.autograd.Function):
    """LengthRegulator.forward (length_regulator.py:70-97): repeat token rows by their durations, zero-pad to rb_out."""

    @staticmethod
    def forward(ctx, hs, rb_in, cum, rb_out):
        ctx.geom = (rb_in, cum, rb_out)
        return hip.lr_gather(rb_in, cum, rb_out, hs.contiguous())

    @staticmethod
    def backward(ctx, dy):
        rb_in, cum, rb_out = ctx.geom
        return hip.lr_segment_sum(rb_in, cum, rb_out, dy.contiguous()), None, None, None


class ShiftSoftmax(torch.autograd.Function):
    """softmax((matrix_ac + rel_shift(matrix_bd)) / sqrt(d_k)) with the key mask of
    LegacyRelPositionMultiHeadedAttention.forward / forward_attention (attention.py:63-93,142-206); ac, bd (B, H, T, T)."""

    @staticmethod
    def forward(ctx, ac, bd, lens, scale, mode=1):
        """mode 1: legacy rel_shift, bd (B, H, T, T); mode 2: RelPositionMultiHeadedAttention.rel_shift (attention.py:236-258),
        bd (B, H, T, 2T-1)."""
        p = hip.shift_softmax_fwd(ac.contiguous(), None if bd is None else bd.contiguous(), lens, scale, mode)
        ctx.save_for_backward(p)
        ctx.scale, ctx.has_bd, ctx.mode = scale, bd is not None, mode
        return p

    @staticmethod
    def backward(ctx, dp):
        (p,) = ctx.saved_tensors
        ds, dbd = hip.shift_softmax_bwd(p, dp.contiguous(), ctx.scale, ctx.has_bd and ctx.needs_input_grad[1], ctx.mode)
        return ds, dbd, None, None, None


class RowDot(torch.autograd.Function):
    """nn.Linear(C, 1) of the predictors (duration_predictor.py:76, variance_predictor.py:63): (rows, C) -> (rows,)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = x.contiguous()
        ctx.save_for_backward(x, weight)
        return hip.row_dot(x, weight.detach().reshape(-1).contiguous(), bias.detach().contiguous())

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = hip.outer_rows(dy, weight.detach().reshape(-1).contiguous()) if ctx.needs_input_grad[0] else None
        dw = hip.col_wsum(x, dy).view_as(weight) if ctx.needs_input_grad[1] else None
        db = hip.col_sum(dy.view(-1, 1)).reshape(1) if ctx.needs_input_grad[2] else None     # (own reduction: see AddBias)
        return dx, dw, db


class AddBias(torch.autograd.Function):
    """x (rows, C) + b (C,) -- pos_bias_u / pos_bias_v on the query rows (attention.py:190-195) -- with the bias gradient taken by
    jatts_col_sum.  As a broadcast torch add, autograd reduces the (B, H, T, d_k) gradient with at::reduce_kernel's multi-block form,
    whose block semaphores are zeroed by a memset; inside a captured step on this stack that memset does not replay reliably and the
    bias gradients came back as garbage (1e38) at the recipes' batch size.  SumAll is the same for a full sum."""

    @staticmethod
    def forward(ctx, x, b):
        return x + b

    @staticmethod
    def backward(ctx, dy):
        C_ = dy.shape[-1]
        db = hip.col_sum(dy.reshape(-1, C_).contiguous()) if ctx.needs_input_grad[1] else None
        return dy, db


class QKVSplit(torch.autograd.Function):
    """The fused Q|K|V projection (rows, 3 A) -> (q + pos_bias_u, q + pos_bias_v, k, v), each (B, H, T, d_k) contiguous, in one launch
    each way (jatts_qkv_split / _bwd; the bias gradients are column sums taken in the same backward pass).  Replaces two broadcast adds
    and three permute copies forward, four zero-filled slice gradients and three adds backward, per attention layer."""

    @staticmethod
    def forward(ctx, qkv, u, v, B, T, H):
        """u = v = None: plain head split -> (q, k, v)."""
        ctx.shapes = None if u is None else (u.shape, v.shape)
        if u is None:
            return tuple(hip.qkv_split(qkv.contiguous(), None, None, B, T, H))
        return tuple(hip.qkv_split(qkv.contiguous(), u.reshape(-1).contiguous(), v.reshape(-1).contiguous(), B, T, H))

    @staticmethod
    def backward(ctx, *grads):
        if ctx.shapes is None:
            dq, dk_, dvv = grads
            dqkv, _, _ = hip.qkv_split_bwd(dq.contiguous(), None, dk_.contiguous(), dvv.contiguous())
            return dqkv, None, None, None, None, None
        dqu, dqv, dk_, dvv = grads
        dqkv, du, dv = hip.qkv_split_bwd(dqu.contiguous(), dqv.contiguous(), dk_.contiguous(), dvv.contiguous())
        return dqkv, du.view(ctx.shapes[0]), dv.view(ctx.shapes[1]), None, None, None


class BMM(torch.autograd.Function):
    """Batched matmul of the attention products on the exact-f32 matrix pipe (jatts_bgemm; round 4: these were torch.matmul -> rocBLAS):
    c = a @ b (trans_b False: b (..., k, n)) or a @ b^T (trans_b True: b (..., n, k)); a is (O, I, m, k), b is (O, I, ., .) or (I, ., .) --
    shared over O, as the position projection p_h is over the batch; its gradient is then summed over O."""

    @staticmethod
    def forward(ctx, a, b, trans_b):
        ctx.trans_b = bool(trans_b)
        ctx.save_for_backward(a, b)
        return hip.bgemm(a, b, trans_b=ctx.trans_b)

    @staticmethod
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        dc = dc.contiguous()
        da = db = None
        if ctx.needs_input_grad[0]:
            da = hip.bgemm(dc, b, trans_b=not ctx.trans_b)                       # dc @ b^T   |   dc @ b
        if ctx.needs_input_grad[1]:
            db = hip.bgemm(dc, a, trans_a=True) if ctx.trans_b else hip.bgemm(a, dc, trans_a=True)   # dc^T @ a (n x k)  |  a^T @ dc (k x n)
            if b.dim() == 3:
                db = db.sum(0)
        return da, db, None


class SumAll(torch.autograd.Function):
    """x.sum() through jatts_col_sum (see AddBias)."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = x.shape
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        return hip.col_sum(x2).sum()

    @staticmethod
    def backward(ctx, dy):
        return dy.expand(ctx.shape)


class OuterRows(torch.autograd.Function):
    """nn.Conv1d(1, C, kernel_size=1) of the pitch / energy embeddings (fastspeech2.py:366-393): (rows,) -> (rows, C)."""

    @staticmethod
    def forward(ctx, v, weight, bias):
        v = v.contiguous()
        ctx.save_for_backward(v, weight)
        return hip.outer_rows(v, weight.detach().reshape(-1).contiguous(), bias.detach().contiguous())

    @staticmethod
    def backward(ctx, dy):
        v, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dv = hip.row_dot(dy, weight.detach().reshape(-1).contiguous()) if ctx.needs_input_grad[0] else None
        dw = hip.col_wsum(dy, v).view_as(weight) if ctx.needs_input_grad[1] else None
        db = hip.col_sum(dy) if ctx.needs_input_grad[2] else None
        return dv, dw, db


class AddSeqVector(torch.autograd.Function):
    """hs + vec.unsqueeze(1): one vector per sequence added to all of its rows (speaker / sid embeddings,
    fastspeech2.py:589-597,750-753).  d_vec[b] = sum of d_hs over the rows of sequence b."""

    @staticmethod
    def forward(ctx, hs, vec, rb):
        ctx.rb = rb
        return hip.add_seq_vector(rb, hs.contiguous().clone(), vec.detach().contiguous())

    @staticmethod
    def backward(ctx, dy):
        rb = ctx.rb
        dvec = None
        if ctx.needs_input_grad[1]:
            dvec = hip.seq_sum(rb, dy.contiguous())
        return dy, dvec, None


class MaskRows(torch.autograd.Function):
    """x * non_pad_mask on a padded batch (variance_predictor.py:81-83, duration_predictor.py:93-96)."""

    @staticmethod
    def forward(ctx, x, rb, valid):
        ctx.geom = (rb, valid)
        y = x.contiguous().clone()
        return hip.zero_pad_rows(rb, y.view(rb.total, -1), valid).view_as(x)

    @staticmethod
    def backward(ctx, dy):
        rb, valid = ctx.geom
        d = dy.contiguous().clone()
        return hip.zero_pad_rows(rb, d.view(rb.total, -1), valid).view_as(dy), None, None


class MaskedLoss(torch.autograd.Function):
    """scale * sum over valid rows of |a - b| (kind 0) or (a - b)^2 (kind 1) -- the masked_select + mean of
    losses/l1l2_loss.py:43-63, duration_predictor_loss.py, variance_predictor_loss.py with scale = 1 / #selected."""

    @staticmethod
    def forward(ctx, pred, target, rb, valid, kind, scale, log_offset):
        pred = pred.contiguous()
        ctx.save_for_backward(pred, target)
        ctx.geom = (rb, valid, kind, scale, log_offset)
        return hip.masked_loss(rb, pred, target, valid, kind, scale, log_offset=log_offset)

    @staticmethod
    def backward(ctx, up):
        pred, target = ctx.saved_tensors
        rb, valid, kind, scale, log_offset = ctx.geom
        up = up.contiguous().float().reshape(1)
        return hip.masked_loss_bwd(rb, pred, target, valid, kind, scale, upstream=up, log_offset=log_offset), None, None, None, None, None, None


class Dropout(torch.autograd.Function):
    """Inverted dropout with a counter-based mask (seed, element index): the backward regenerates the mask."""

    @staticmethod
    def forward(ctx, x, p, seed, seed_dev=None):
        ctx.p, ctx.seed, ctx.seed_dev = p, seed, seed_dev
        return hip.dropout(x.contiguous(), p, seed, seed_dev)

    @staticmethod
    def backward(ctx, dy):
        return hip.dropout(dy.contiguous(), ctx.p, ctx.seed, ctx.seed_dev), None, None, None


class ActDropout(torch.autograd.Function):
    """dropout(act(x)) -- the feed-forward module's ReLU -> Dropout (multi_layer_conv.py:52-63) -- as one launch each way; the same
    counter-based mask as Dropout (bit-identical to Act followed by Dropout)."""

    @staticmethod
    def forward(ctx, x, mode, p, seed, seed_dev=None):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.cfg = (mode, p, seed, seed_dev)
        return hip.act_dropout(x, mode, p, seed, seed_dev)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        mode, p, seed, seed_dev = ctx.cfg
        return hip.act_dropout(x, mode, p, seed, seed_dev, dy=dy.contiguous()), None, None, None, None


class ResidualDropAdd(torch.autograd.Function):
    """x + alpha * dropout(h) -- a conformer layer's residual connection (encoder_layer.py:100-170) as one launch forward and one
    backward (d_x = dy is passed through, d_h = alpha * mask(dy) / (1 - p) regenerates the mask)."""

    @staticmethod
    def forward(ctx, x, h, alpha, p, seed, seed_dev=None):
        ctx.alpha, ctx.p, ctx.seed, ctx.seed_dev = alpha, p, seed, seed_dev
        return hip.dropout_add(h.contiguous(), x.contiguous(), p, alpha, seed, seed_dev)

    @staticmethod
    def backward(ctx, dy):
        dh = hip.dropout_add(dy.contiguous(), None, ctx.p, ctx.alpha, ctx.seed, ctx.seed_dev) if ctx.needs_input_grad[1] else None
        return (dy if ctx.needs_input_grad[0] else None), dh, None, None, None, None


class ForwardSum(torch.autograd.Function):
    """ForwardSumLoss.forward after the prior has been added (losses/forward_sum_loss.py:58-78): mean over the batch of the
    per-utterance CTC losses; the gradient is the one torch's ctc_loss backward produces."""

    @staticmethod
    def forward(ctx, log_p, ilens, olens, log_blank):
        nll, grad = hip.ctc_forward_sum(log_p.contiguous(), ilens, olens, log_blank, want_grad=True, grad_scale=1.0 / log_p.shape[0])
        ctx.save_for_backward(grad)
        return nll.sum() / log_p.shape[0]

    @staticmethod
    def backward(ctx, up):
        (grad,) = ctx.saved_tensors
        return grad * up, None, None, None


class AlignLogProb(torch.autograd.Function):
    """AlignmentModule.forward after its convolutions (alignments.py:50-60): log_softmax over the valid text tokens of
    -||feats_i - text_j||_2, -inf at padded tokens.  ff (B*To, A) frame features, tf (B*Tm, A) padded token features.
    forward: jatts_alignment_logp.  backward: log-softmax backward, then with w = -d_score / dist,
    d_ff_i = (sum_j w_ij) f_i - sum_j w_ij t_j and d_tf_j = (sum_i w_ij) t_j - sum_i w_ij f_i: two batched GEMMs (rocBLAS) and
    element-wise ops on the (B, To, Tm) matrices (f64 for the distance recomputed in its GEMM form)."""

    @staticmethod
    def forward(ctx, ff, tf, B, ilens, tsel=None, valid=None):
        """tsel (rows of the valid tokens in tf) / valid ((B, Tm) bool token mask): optional, precomputed by the caller so that no
        host -> device copy happens in the middle of the forward."""
        dev = ff.device
        A_ = ff.shape[1]
        To, Tm = ff.shape[0] // B, tf.shape[0] // B
        rbf, rbv = hip.RaggedBatch([To] * B, dev), hip.RaggedBatch(ilens, dev)
        if tsel is None:
            tsel = hip.h2d([b * Tm + i for b in range(B) for i in range(ilens[b])], torch.int64, dev)
        if valid is None:
            valid = torch.arange(Tm, device=dev).unsqueeze(0) < hip.h2d(ilens, torch.int64, dev).unsqueeze(1)       # (B, Tm)
        lp3 = hip.alignment_logp(rbf, rbv, ff.contiguous(), tf.index_select(0, tsel).contiguous(), A_).view(B, To, -1)
        lp = torch.full((B, To, Tm), float("-inf"), dtype=torch.float32, device=dev)
        n = min(Tm, lp3.shape[2])
        lp[:, :, :n] = lp3[:, :, :n]
        lp = lp.masked_fill(~valid.unsqueeze(1), float("-inf"))
        ctx.save_for_backward(ff, tf, lp, valid)
        ctx.B = B
        return lp

    @staticmethod
    def backward(ctx, dlp):
        ff, tf, lp, valid = ctx.saved_tensors
        B = ctx.B
        To, Tm, A_ = ff.shape[0] // B, tf.shape[0] // B, ff.shape[1]
        vm = valid.unsqueeze(1)
        g = dlp.double().masked_fill(~vm, 0.0)
        dscore = g - torch.exp(lp.double()) * g.sum(-1, keepdim=True)
        F_, T_ = ff.view(B, To, A_).double(), tf.view(B, Tm, A_).double()
        d2 = (F_ * F_).sum(-1).unsqueeze(2) + (T_ * T_).sum(-1).unsqueeze(1) - 2.0 * torch.matmul(F_, T_.transpose(1, 2))
        w = (-dscore / torch.sqrt(d2.clamp_min(1e-24))).masked_fill(~vm, 0.0)
        dF = w.sum(-1, keepdim=True) * F_ - torch.matmul(w, T_)
        # sum_i w_ij rides along as one more GEMM column: torch's strided multi-block reduction keeps its block semaphores zero with a
        # memset, which did not replay reliably inside a captured step on this stack (garbage text-side gradients at B 32 x 768 frames)
        wF = torch.matmul(w.transpose(1, 2), torch.cat([F_, torch.ones(B, To, 1, dtype=F_.dtype, device=F_.device)], dim=-1))
        dT = wF[..., A_:] * T_ - wF[..., :A_]
        return dF.reshape(B * To, A_).float(), dT.reshape(B * Tm, A_).float(), None, None, None, None


class WeightNorm(torch.autograd.Function):
    """torch.nn.utils.weight_norm (dim 0): w = g v / ||v||, one launch each way (torch spells it as norm, div, mul and ~10 kernels back;
    a VITS step has ~100 weight-normalised convolutions)."""

    @staticmethod
    def forward(ctx, g, v):
        g, v = g.contiguous(), v.contiguous()
        w, inv = hip.weight_norm_fwd(v, g)
        ctx.save_for_backward(g, v, inv)
        return w

    @staticmethod
    def backward(ctx, dw):
        g, v, inv = ctx.saved_tensors
        dv, dg = hip.weight_norm_bwd(v, g, inv, dw.contiguous())
        return dg, dv


class SplitAdd(torch.autograd.Function):
    """WaveNet ResidualBlock tail (vits/wavenet/residual_block.py:158-167): (o = [res | skip part], h, skip) -> (h + res, skip + skip
    part) in one pass; the backward is one concat instead of two zero-filled slice gradients and their sum."""

    @staticmethod
    def forward(ctx, o, h, skip):
        ctx.has_skip = skip is not None
        return hip.split_add(o.contiguous(), h.contiguous(), skip.contiguous() if skip is not None else None)

    @staticmethod
    def backward(ctx, dh, ds):
        ref = dh if dh is not None else ds
        rows, dim = ref.shape
        do = hip.concat2(dh.contiguous() if dh is not None else None, ds.contiguous() if ds is not None else None, rows, dim, ref.device)
        return do, dh, (ds if ctx.has_skip else None)


class Gate(torch.autograd.Function):
    """WaveNet gated activation tanh(a) * sigmoid(b) on [a | b] (vits/wavenet/residual_block.py:150-156)."""

    @staticmethod
    def forward(ctx, x, rb):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return hip.gated_tanh_sigmoid(rb, x, None, x.shape[1] // 2, hip.F32)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return hip.gate_bwd(x, dy.contiguous()), None
