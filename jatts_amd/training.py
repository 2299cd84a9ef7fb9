"""Training-side pieces on the MI355X path -- the first slice of SURVEY §8 f.4 (reference
jatts/trainers/fastspeech2.py:24-100 `_train_step`, jatts/bin/tts_train.py:355-363 DistributedDataParallel).

What is here: (1) the FastSpeech2 criterion (`MelLoss`/`L1Loss`, `DurationPredictorLoss`, `PitchLoss`, `EnergyLoss` of
jatts/losses/) evaluated by HIP kernels on the dict that `FastSpeech2.forward()` returns, i.e. `_train_step` up to `gen_loss`;
(2) the backward of the hot path's work-horse op: `Conv1dFunction`, a torch.autograd.Function whose forward is jatts_conv1d and
whose backward is three HIP launches (dx = jatts_conv1d on flipped / transposed weights, dW = jatts_conv1d_wgrad, db =
jatts_col_sum), in f32; (3) `allreduce_gradients`: bucketed, flat gradient all-reduce over RCCL (what DDP does for the
reference), usable after any backward; (4) `FastSpeech2Trainer`: the whole `_train_step` -- FastSpeech2.forward() in train()
mode (models/fastspeech2_train.py on the HIP forward / backward pairs of jatts_amd/autograd.py), the criterion, backward, the
gradient all-reduce, clip_grad_norm_ + Adam as HIP kernels, the reference's WarmupLR schedule.  What is NOT here yet: the
training paths of Matcha-TTS / VITS, f16 training.  No CPU fallback: CPU tensors raise.
"""
import contextlib

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
# precision="fp32_split" of the trainers (round 4): the forward and data-gradient convs of every Conv1dFunction on split-precision MFMA operands
# (JATTS_F32S, csrc/conv1d_split.h); weight gradients, normalisations, attention products and the optimiser stay exact f32.  Module-level
# switch set around a step by the trainer (split_convs()); a captured graph bakes in whatever was on during its capture.
SPLIT_CONVS = [False]


@contextlib.contextmanager
def split_convs(on=True):
    prev, SPLIT_CONVS[0] = SPLIT_CONVS[0], bool(on)
    try:
        yield
    finally:
        SPLIT_CONVS[0] = prev


class Conv1dFunction(torch.autograd.Function):
    """y = conv1d(x) on a packed ragged batch (rows, c_in) -> (rows, n_out), f32; "same"-style geometry via (dil, pad).
    forward: jatts_conv1d.  backward: dx = jatts_conv1d(dy, W'[c][n][k-1-tap], pad' = (k-1) dil - pad),
    dW, db = jatts_conv1d_wgrad(x, dy) (the bias gradient falls out of the staged dy tiles)."""

    @staticmethod
    def forward(ctx, x, weight, bias, rb, dil, pad):
        n_out, c_in, k = weight.shape
        split = SPLIT_CONVS[0] and (k - 1) * dil <= 32        # (beyond the split kernel's tiles: the exact-f32 kernel)
        if split:
            wp, winv, c_pad = hip.pack_conv_weight_split_dev(weight.detach())
        else:
            (wp, c_pad), winv = hip.pack_conv_weight_dev(weight.detach(), hip.F32), None
        xin = x.contiguous() if c_in == c_pad else hip.affine_cast(x.contiguous(), hip.F32, ldy=c_pad)
        y = hip.conv1d(rb, xin, wp, c_pad, n_out, k, dtype=hip.F32S if split else hip.F32, dil=dil, pad=pad,
                       bias=None if bias is None else bias.detach().contiguous(), w_inv=winv, out_f32=True)
        ctx.save_for_backward(x, weight)
        ctx.geom = (rb, dil, pad, bias is not None, split)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        rb, dil, pad, has_bias, split = ctx.geom
        n_out, c_in, k = weight.shape
        dy = dy.contiguous().float()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if split:      # the data gradient is a conv like the forward: same split arithmetic (the WEIGHT gradient below stays exact f32)
                wp, winv, c_pad = hip.pack_conv_weight_split_dev(weight.detach(), dgrad=True)
            else:
                (wp, c_pad), winv = hip.pack_conv_weight_dev(weight.detach(), hip.F32, dgrad=True), None       # W'[c][n][k-1-tap], packed in one launch
            dyp = dy if n_out == c_pad else hip.affine_cast(dy, hip.F32, ldy=c_pad)
            dx = hip.conv1d(rb, dyp, wp, c_pad, c_in, k, dtype=hip.F32S if split else hip.F32, dil=dil, pad=(k - 1) * dil - pad, w_inv=winv, out_f32=True)
        want_db = has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            dw = hip.conv1d_wgrad(rb, x.detach().contiguous().float(), dy, c_in, n_out, k, dil, pad, want_db=want_db)
            if want_db:
                dw, db = dw
        elif want_db:
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


def allreduce_flat(flat, group=None, bucket_bytes=64 << 20, average=True):
    """All-reduce a flat gradient buffer in place, bucket_bytes at a time (slices: no packing copies).  -> number of collectives."""
    world = dist.get_world_size(group)
    if world == 1:
        return 0
    per = max(1, bucket_bytes // flat.element_size())
    n = 0
    for o in range(0, flat.numel(), per):
        sl = flat[o:o + per]
        dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=group)
        if average:
            sl /= world
        n += 1
    return n


# ------------------------------------------------------------------------------------------ _train_step
def warmup_lr(base_lr, step_num, warmup_steps):
    """jatts/schedulers/warmup_lr.py:55-62: lr * warmup^0.5 * min(step^-0.5, step * warmup^-1.5), step_num counts from 1."""
    return base_lr * warmup_steps ** 0.5 * min(step_num ** -0.5, step_num * warmup_steps ** -1.5)


def scheduled_lr(kind, base_lr, step_num, **params):
    """Learning rate of optimiser step ``step_num`` (1-based) under the recipes' schedulers: "warmuplr" (fastspeech2.v1 / vits.v1:
    jatts/schedulers/warmup_lr.py), "steplr" (the Matcha recipes: torch.optim.lr_scheduler.StepLR(step_size, gamma), stepped once
    per optimiser step) or None / "none" (constant)."""
    kind = (kind or "none").lower()
    if kind == "warmuplr":
        w = params.get("warmup_steps", 4000)
        return warmup_lr(base_lr, step_num, w) if w else base_lr
    if kind == "steplr":
        return base_lr * params.get("gamma", 0.1) ** ((step_num - 1) // params["step_size"])
    if kind == "none":
        return base_lr
    raise ValueError(f"unknown scheduler {kind!r} (warmuplr, steplr, none)")


class FastSpeech2Trainer:
    """`FastSpeech2Trainer._train_step` (jatts/trainers/fastspeech2.py:24-100) on one GPU of a data-parallel job:
    forward (train mode) -> MelLoss + DurationPredictorLoss + PitchLoss + EnergyLoss -> backward -> gradient all-reduce over
    the ranks (when torch.distributed is initialised) -> clip_grad_norm_(grad_norm) -> Adam -> WarmupLR.
    The optimiser state is one flat f32 pair (m, v) per parameter; clip + Adam run as HIP kernels (jatts_sumsq / jatts_adam_step),
    the clip coefficient is read on the device (no host sync in the step besides the loss values the caller asks for)."""

    def __init__(self, model, lr=0.0008, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_norm=1.0, warmup_steps=4000, group=None,
                 bucket_bytes=64 << 20, overlap=True, gradient_accumulate_steps=1, scheduler="warmuplr", scheduler_params=None, capture_graph=False,
                 max_graphs=8, precision="fp32"):
        if precision not in ("fp32", "fp32_split"):
            raise ValueError("trainer precision: 'fp32' (exact-f32 MFMA, the reference's arithmetic) or 'fp32_split' (f32 tensors; the forward and "
                             "data-gradient convs on split f16 hi / lo MFMA operands, everything else exact f32)")
        self.precision = precision
        self.model, self.base_lr, self.betas, self.eps, self.wd = model, lr, betas, eps, weight_decay
        self.grad_norm, self.warmup_steps, self.group, self.bucket_bytes = grad_norm, warmup_steps, group, bucket_bytes
        self.overlap = overlap
        self.scheduler, self.scheduler_params = scheduler, dict(scheduler_params or {})      # e.g. "steplr", {"step_size": 10000, "gamma": 0.5}
        # trainers/base.py:64,135 / vits.py:113-121: `gradient_accumulate_steps` forward / backward passes (loss / G each) per optimiser
        # step; `steps` (and with it the loss schedules and WarmupLR) counts optimiser steps
        self.accumulate, self._micro = max(1, int(gradient_accumulate_steps)), 0
        model.train()   # (turns requires_grad on: the inference classes create frozen parameters)
        self.params = [p for p in model.parameters() if p.requires_grad]
        if not self.params or self.params[0].device.type != "cuda":
            raise hip._abi.JattsHipError("FastSpeech2Trainer needs the model on the GPU (no CPU fallback)")
        # one flat f32 buffer each for parameters, gradients and the two Adam moments; every parameter / .grad is a view into it:
        # zero_grad is one fill, the norm one reduction, Adam one launch, and the all-reduce runs on slices of the gradient
        # buffer with no packing copies (288 GB of HBM: the 4 x 281 MB of FastSpeech2 are nothing)
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat_p = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(n, dtype=torch.float32, device=dev)
        o = 0
        self._grad_views, self._grad_offsets = [], []
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                self.flat_p[o:o + k].copy_(p.data.reshape(-1))
                p.data = self.flat_p[o:o + k].view(p.shape)
                p.grad = self.flat_g[o:o + k].view(p.shape)
                self._grad_views.append(p.grad)
                self._grad_offsets.append(o)
                o += k
        self.steps = 0
        self.last_lr = None
        self._bad_ids = None
        # graph mode: the whole step (forward, losses, backward, clip + Adam) of a batch SIGNATURE (shapes + length tuples: a length bucket)
        # captured once as a hipGraph and replayed; the first step of a signature runs eagerly (it fills every per-lengths cache), the
        # second is captured.  See _graph_step.
        self.capture_graph = bool(capture_graph) and self._graph_capable
        self._graphs = {}
        self.max_graphs = max(1, int(max_graphs))     # captured graphs kept (each owns the activations of one step: GBs); least recently used evicted
        self._buckets = None

    # -- gradients into the flat buffer.  With .grad pre-set to views of flat_g, autograd's AccumulateGrad runs one `view += g` kernel per
    # parameter (260 of a FastSpeech2 step's launches, 650 of a VITS step's: 1.7-4.3 ms of GPU time).  Instead .grad is cleared before
    # backward -- autograd then just keeps each produced gradient tensor -- and ONE gather (ceil(n / 64) launches) copies them into their
    # slots; .grad goes back to the flat views afterwards, so callers see what they always saw.  The overlapped all-reduce keeps the view
    # form: its bucket hooks need each gradient in place the moment it is produced.
    def _detach_grads(self):
        for p in self.params:
            p.grad = None

    def _gather_grads(self, accumulate):
        grads, offs, keep = [], [], []
        for p, o in zip(self.params, self._grad_offsets):
            g = p.grad
            if g is None:
                continue
            if g.dtype != torch.float32 or not g.is_contiguous():
                g = g.float().contiguous()
                keep.append(g)
            grads.append(g)
            offs.append(o)
        hip.gather_grads(grads, offs, self.flat_g, accumulate)
        for p, v in zip(self.params, self._grad_views):
            p.grad = v

    # -- gradient all-reduce overlapped with backward (what DistributedDataParallel's reducer does for the reference): the flat gradient
    # buffer is cut into bucket_bytes slices; a post-accumulate hook on every parameter counts its bucket down and, when the last
    # gradient of a bucket has been accumulated, issues that slice's all-reduce asynchronously (RCCL orders it after the compute
    # stream's work so far and runs it on its own stream while backward continues).  Backward produces gradients roughly in reverse
    # parameter order, so the LAST slices go first.  train_step waits for the handles, launches whatever never fired, averages.
    def _setup_overlap(self):
        per = max(1, self.bucket_bytes // 4)
        n = self.flat_g.numel()
        edges = list(range(0, n, per)) + [n]
        self._buckets = [dict(lo=edges[i], hi=edges[i + 1], total=0, left=0, work=None) for i in range(len(edges) - 1)]
        o = 0
        for p in self.params:
            touched = range(o // per, (o + p.numel() - 1) // per + 1)     # a parameter may straddle bucket edges
            for b in touched:
                self._buckets[b]["total"] += 1

            def hook(_p, touched=touched):
                for b in touched:
                    bk = self._buckets[b]
                    bk["left"] -= 1
                    if bk["left"] == 0:
                        bk["work"] = dist.all_reduce(self.flat_g[bk["lo"]:bk["hi"]], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            p.register_post_accumulate_grad_hook(hook)
            o += p.numel()

    def _arm_overlap(self, live=True):
        for bk in self._buckets:
            bk["left"], bk["work"] = (bk["total"] if live else 1 << 60), None

    def _finish_overlap(self):
        world = dist.get_world_size(self.group)
        n = 0
        for bk in self._buckets:
            if bk["work"] is None:      # a parameter of this slice got no gradient this step: reduce the slice now
                bk["work"] = dist.all_reduce(self.flat_g[bk["lo"]:bk["hi"]], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            n += 1
        for bk in self._buckets:
            bk["work"].wait()
        self.flat_g /= world
        return n

    # -- checkpoints in the reference's format (jatts/trainers/base.py:85-124): {"model", "optimizer", "scheduler", "steps", "epochs"};
    # "optimizer" is a torch.optim.Adam state_dict (state indexed by the position in model.parameters()), so a checkpoint written here
    # resumes under the reference trainer and vice versa.
    def _views(self, flat):
        out, o = [], 0
        for p in self.params:
            out.append(flat[o:o + p.numel()].view(p.shape))
            o += p.numel()
        return out

    def _sched_params(self):
        sp = dict(self.scheduler_params)
        if (self.scheduler or "").lower() == "warmuplr":
            sp.setdefault("warmup_steps", self.warmup_steps)
        return sp

    def _seed(self):
        """Dropout stream of the micro-batch about to run: a function of (optimiser step, micro-batch index) -- `_Ctx` mixes the rank
        in -- so a resumed run continues the stream instead of replaying steps 1..N, and eval_step does not consume from it."""
        return self.steps * self.accumulate + self._micro + 1

    def state_dict(self, epochs=0):
        m, v = self._views(self.flat_m), self._views(self.flat_v)
        # torch steps the scheduler AFTER the optimiser: a checkpoint holds the lr of the NEXT step (StepLR is chainable and reads
        # group["lr"] back, so storing the lr just used would lose a decay at steps % step_size == 0)
        next_lr = scheduled_lr(self.scheduler, self.base_lr, self.steps + 1, **self._sched_params())
        opt = {"state": {}, "param_groups": [dict(lr=next_lr, betas=tuple(self.betas),
                                                  eps=self.eps, weight_decay=self.wd, amsgrad=False, maximize=False, foreach=None,
                                                  capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False,
                                                  initial_lr=self.base_lr, params=list(range(len(self.params))))]}
        if self.steps > 0:
            for i in range(len(self.params)):
                opt["state"][i] = {"step": torch.tensor(float(self.steps)), "exp_avg": m[i].detach().cpu().clone(),
                                   "exp_avg_sq": v[i].detach().cpu().clone()}
        sch = {"base_lrs": [self.base_lr], "last_epoch": self.steps, "_step_count": self.steps + 1, "_get_lr_called_within_step": False,
               "_last_lr": [next_lr], **self.scheduler_params}
        if (self.scheduler or "").lower() == "warmuplr":
            sch["warmup_steps"] = self.warmup_steps
        return {"model": {k: t.detach().cpu().clone() for k, t in self.model.state_dict().items()}, "optimizer": opt, "scheduler": sch,
                "steps": self.steps, "epochs": epochs}

    def save_checkpoint(self, checkpoint_path, epochs=0):
        import os
        d = os.path.dirname(checkpoint_path)
        if d and not os.path.exists(d):
            os.makedirs(d)
        torch.save(self.state_dict(epochs), checkpoint_path)

    def load_checkpoint(self, checkpoint_path, load_only_params=False):
        sd = torch.load(checkpoint_path, map_location="cpu") if isinstance(checkpoint_path, str) else checkpoint_path
        with torch.no_grad():      # copy INTO the flat views (load_state_dict copies in place: the views stay attached)
            self.model.load_state_dict(sd["model"])
        self.model._prep = None
        if load_only_params:
            return
        self.steps = int(sd["steps"])
        st = sd["optimizer"]["state"]
        if len(st) not in (0, len(self.params)):
            raise ValueError("optimizer state does not match the model's parameter list")
        with torch.no_grad():
            for i, (mv, vv) in enumerate(zip(self._views(self.flat_m), self._views(self.flat_v))):
                if i in st:
                    mv.copy_(st[i]["exp_avg"])
                    vv.copy_(st[i]["exp_avg_sq"])
        g = sd["optimizer"]["param_groups"][0]
        self.betas, self.eps, self.wd = tuple(g["betas"]), g["eps"], g["weight_decay"]
        self.base_lr = g.get("initial_lr", self.base_lr)
        if "scheduler" in sd and sd["scheduler"]:
            sch = sd["scheduler"]
            self.warmup_steps = sch.get("warmup_steps", self.warmup_steps)
            for k in ("step_size", "gamma"):           # a reference StepLR checkpoint carries its own schedule
                if k in sch:
                    self.scheduler_params[k] = sch[k]
            if "base_lrs" in sch and sch["base_lrs"]:
                self.base_lr = sch["base_lrs"][0]
        self._micro = 0
        self.last_lr = scheduled_lr(self.scheduler, self.base_lr, self.steps, **self._sched_params()) if self.steps > 0 else None

    def compute_losses(self, batch):
        """forward + criterion of trainers/fastspeech2.py:44-84 -> dict of differentiable scalars incl. "loss"."""
        from .models.fastspeech2_train import criterion, train_forward
        m = self.model
        ret = train_forward(m, batch["xs"], batch["ilens"], batch["ys"], batch["olens"], batch["durations"], batch["duration_lens"],
                            batch["pitch"], batch["pitch_lens"], batch["energys"], batch["energy_lens"], spembs=batch.get("spkembs"),
                            sids=batch.get("sids"), seed=self._seed())
        return criterion(ret, batch["durations"], batch["pitch"], batch["energys"], batch["ilens"])

    @torch.no_grad()
    def eval_step(self, batch):
        with split_convs(self.precision == "fp32_split"):
            return self._eval_step(batch)

    def _eval_step(self, batch):
        """`_eval_step` of the reference trainers (e.g. trainers/fastspeech2.py:150-215): the same forward + criterion in eval() mode
        (running-statistics BatchNorm, no dropout), no gradients, no update.  -> dict of loss tensors."""
        was = self.model.training
        self.model.eval()
        try:
            return {k: v.detach() for k, v in self.compute_losses(batch).items()}
        finally:
            self.model.train(was)

    _graph_capable = True      # a subclass whose step synchronises with the host switches it off
    SCALAR_RING = 8            # pinned staging slots for the per-step scalars of a replayed graph
    GRAD_NORM_SANITY = 1e12    # a replayed step whose (pre-clip) gradient norm is not finite or beyond this is treated as a broken capture

    def _signature(self, batch):
        sig = []
        for k in sorted(batch):
            v = batch[k]
            if v is None:
                continue
            if torch.is_tensor(v) and not v.is_cuda and v.dim() == 1 and v.dtype in (torch.int64, torch.int32):
                sig.append((k, tuple(int(x) for x in v.tolist())))          # length vectors: part of the bucket
            elif torch.is_tensor(v):
                sig.append((k, tuple(v.shape), str(v.dtype)))
            else:
                sig.append((k, repr(v)))
        return tuple(sig)

    def _graph_step(self, batch):
        """One optimiser step by graph replay.  Preconditions (checked where they can be): world size 1, no gradient accumulation, the
        length vectors (ilens / olens / *_lens) are CPU tensors, sum(durations) == olens per utterance."""
        dev = self.flat_p.device
        sig = self._signature(batch)
        st = self._graphs.get(sig)
        if st is not None and st.get("eager_only"):
            return self._train_step(batch)
        if st is None:                       # first sight of this bucket: eager step, remember the verified frame counts
            if "durations" in batch and batch["durations"] is not None:
                dsum = [int(v) for v in batch["durations"].sum(1).tolist()]
                if dsum != [int(v) for v in batch["olens"].tolist()]:
                    raise ValueError("graph mode needs sum(durations) == olens for every utterance")
            if len(self._graphs) >= 16 * max(1, self.max_graphs):      # signatures seen once and never again (unbucketed data): bounded
                for k in [k for k, v in self._graphs.items() if v.get("graph") is None and not v.get("eager_only")][: len(self._graphs) // 2]:   # (eager-only markers stay)
                    del self._graphs[k]
            self._graphs[sig] = {"graph": None}
            return self._train_step(batch)
        if st["graph"] is None:              # second sight: capture
            live = [(v.get("used", 0), k) for k, v in self._graphs.items() if v.get("graph") is not None]
            while len(live) >= self.max_graphs:               # a bucketed sampler can produce many signatures: bound the graphs' memory
                live.sort()
                _, k = live.pop(0)
                self._graphs[k] = {"graph": None}             # (its next sight captures again)
            st["in"] = {k: (v.to(dev).clone() if torch.is_tensor(v) and (v.is_cuda or v.dim() != 1 or v.dtype not in (torch.int64, torch.int32)) else v)
                        for k, v in batch.items()}
            st["seed"] = torch.zeros(1, dtype=torch.int64, device=dev)
            st["hyper"] = torch.zeros(7, dtype=torch.float32, device=dev)
            # per-step scalars travel through a RING of pinned staging buffers, each guarded by an event recorded behind its copies: the
            # host may run many steps ahead of the GPU (losses stay on the device), and one shared staging pair would let step N's
            # asynchronous copy read step N + 1's values
            st["ring"] = [dict(seed=torch.zeros(1, dtype=torch.int64).pin_memory(), hyper=torch.zeros(7, dtype=torch.float32).pin_memory(),
                               gn=torch.zeros(1, dtype=torch.float64).pin_memory(), step=0, ev=None) for _ in range(self.SCALAR_RING)]
            st["ring_pos"] = 0
            m = self.model
            m._seed_dev, m._static_olens = st["seed"], [int(v) for v in batch["olens"].tolist()]
            g = torch.cuda.CUDAGraph()
            # every cached device tensor the step is handed during the capture (length uploads, ragged geometry, positional tables, the
            # forward-sum prior, frame / token selectors) lives in bounded caches that evict: the graph reads them at their addresses on
            # every replay, so the graph's own record pins them
            st["keep"] = hip.keep_begin()
            try:
                torch.cuda.synchronize()
                with torch.cuda.graph(g):
                    hip.zero_pool_begin(dev)
                    self.flat_g.zero_()
                    self._detach_grads()
                    losses = self.compute_losses(st["in"])
                    losses["loss"].backward()
                    self._gather_grads(False)
                    ss = None
                    if self.grad_norm and self.grad_norm > 0:
                        ss = torch.zeros((), dtype=torch.float64, device=dev)
                        hip.sumsq(self.flat_g, ss)
                    hip.adam_step(self.flat_p, self.flat_g, self.flat_m, self.flat_v, 0.0, self.betas[0], self.betas[1], self.eps, self.wd, 1,
                                  grad_sumsq=ss, max_norm=self.grad_norm or 0.0, hyper_dev=st["hyper"])
                    st["out"] = {k: v.detach() for k, v in losses.items()}
                    if ss is not None:
                        st["out"]["grad_norm"] = ss.sqrt()
            except BaseException:
                # a failed capture must not be retried forever, nor leave the parameters without their .grad views: this signature runs eagerly
                self._graphs[sig] = {"graph": None, "eager_only": True}
                for p_, v_ in zip(self.params, self._grad_views):
                    p_.grad = v_
                raise
            finally:
                hip.keep_end()
                hip.zero_pool_end()
                m._seed_dev, m._static_olens = None, None
            st["graph"] = g
        # replay: refresh the static inputs and the per-step scalars (all stream-ordered copies; no host wait)
        if self._bad_ids is not None and self._bad_ids():       # out-of-range token ids of an EARLIER replay (zero rows, counted on the
            self._bad_ids = None                                # device by the embedding kernel inside the graph): IndexError, one step late
        # guard: the gradient norm of every replayed step comes back through the ring.  A finished slot that holds a non-finite
        # or absurd norm means either a replay that produced garbage (a reduction that does not survive capture on this stack, see _graph_step's
        # notes) or a genuinely diverged run; either way: refuse to go on silently -- the signature is marked eager-only and the call raises, a
        # few steps late, without a host sync per step, BEFORE this call has advanced the step counter / learning-rate schedule or touched
        # the static inputs.  A slot about to be recycled is waited for and checked too, so no step's norm goes unread.
        nxt = st["ring"][st["ring_pos"] % len(st["ring"])]
        for sl in st["ring"]:
            if sl["ev"] is None or not sl["step"]:
                continue
            if sl is nxt:
                sl["ev"].synchronize()       # this slot's copies of SCALAR_RING steps ago have run (normally long since)
            elif not sl["ev"].query():
                continue
            v = float(sl["gn"][0])
            sl["step"] = 0
            if not (v == v and abs(v) < self.GRAD_NORM_SANITY):
                self._graphs[sig] = {"graph": None, "eager_only": True}
                raise FloatingPointError(f"gradient norm {v} at trainer step {sl['at']} (graph replay): a diverged run or a capture that does not replay "
                                         "correctly -- replay of this batch signature is disabled, later steps of it run eagerly; the parameters are "
                                         "suspect: resume from the last checkpoint")
        for k, v in batch.items():
            if torch.is_tensor(v) and torch.is_tensor(st["in"].get(k)) and st["in"][k].is_cuda:
                st["in"][k].copy_(v, non_blocking=True)
        self.steps += 1
        lr = scheduled_lr(self.scheduler, self.base_lr, self.steps, **self._sched_params())
        self.last_lr = lr
        rank = dist.get_rank(self.group) if dist.is_available() and dist.is_initialized() else 0
        slot = st["ring"][st["ring_pos"] % len(st["ring"])]
        st["ring_pos"] += 1
        if slot["ev"] is not None:
            slot["ev"].synchronize()         # (already waited for by the guard when it held a step's norm)
        slot["seed"][0] = ((self.steps - 1) * self.accumulate + 1) * 4099 * 1000003 + rank * 1000003   # == _Ctx's (seed * 4099 + rank) * 1000003
        slot["hyper"].copy_(torch.tensor(hip.adam_hyper(lr, self.betas[0], self.betas[1], self.eps, self.wd, self.steps), dtype=torch.float32))
        st["seed"].copy_(slot["seed"], non_blocking=True)
        st["hyper"].copy_(slot["hyper"], non_blocking=True)
        st["graph"].replay()
        if "grad_norm" in st["out"]:
            slot["gn"].copy_(st["out"]["grad_norm"].reshape(1), non_blocking=True)
            slot["step"], slot["at"] = 1, self.steps
        slot["ev"] = torch.cuda.Event()
        slot["ev"].record()
        st["used"] = self.steps
        self.model._prep = None
        if self._bad_ids is None:
            self._bad_ids = hip.bad_ids_async(dev)
        # the graph's outputs are STATIC tensors that the next replay overwrites: hand out stream-ordered copies (one concatenation per dtype),
        # so that a caller who collects loss tensors and reads them later sees each step's own values
        out, by_dtype = {}, {}
        for k, v in st["out"].items():
            by_dtype.setdefault(v.dtype, []).append(k)
        for ks in by_dtype.values():
            vals = torch.stack([st["out"][k].reshape(()) for k in ks])
            for i, k in enumerate(ks):
                out[k] = vals[i]
        return out

    def train_step(self, batch):
        with split_convs(self.precision == "fp32_split"):
            return self._train_step_any(batch)

    def _train_step_any(self, batch):
        """batch: dict with the collater's keys (xs, ilens, ys, olens, durations, duration_lens, pitch, pitch_lens, energys,
        energy_lens).  -> dict of the loss tensors (on the GPU; .item() them only when logging)."""
        m = self.model
        m.train()
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1
        if self.capture_graph and not multi and self.accumulate == 1:
            sig_seen = self._graphs.get(self._signature(batch))
            if sig_seen is not None and not sig_seen.get("eager_only"):
                return self._graph_step(batch)          # capture / replay
            hip.zero_pool_begin(self.flat_p.device)
            try:
                return self._graph_step(batch)          # first sight: eager (inside an open zero pool, like every eager step)
            finally:
                hip.zero_pool_end()
        hip.zero_pool_begin(self.flat_p.device)     # the step's small zero-initialised accumulators: one fill instead of ~550
        try:
            return self._train_step(batch)
        finally:
            hip.zero_pool_end()

    def _train_step(self, batch):
        m = self.model
        if self._bad_ids is not None and self._bad_ids():       # out-of-range token ids of an EARLIER step (zero rows, counted on the
            self._bad_ids = None                                # device): raised here, one step late, instead of a host sync per step
        if self._micro == 0:
            self.flat_g.zero_()
        last = self._micro == self.accumulate - 1
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1
        in_place = multi and self.overlap        # gradients accumulated straight into flat_g's views (the bucket hooks need them there)
        if in_place:
            for p, v in zip(self.params, self._grad_views):      # (re-attach: a caller may have set .grad to None)
                p.grad = v
            if self._buckets is None:
                self._setup_overlap()
            self._arm_overlap(last)              # the exchange belongs to the last micro-batch's backward only
        else:
            self._detach_grads()
        losses = self.compute_losses(batch)
        (losses["loss"] / self.accumulate if self.accumulate > 1 else losses["loss"]).backward()
        if not in_place:
            self._gather_grads(self._micro > 0)
        self._micro += 1
        if not last:
            return {k: v.detach() for k, v in losses.items()}
        self._micro = 0
        if multi and self.overlap:
            self._finish_overlap()
        elif multi:
            allreduce_flat(self.flat_g, self.group, self.bucket_bytes)
        self.steps += 1
        lr = scheduled_lr(self.scheduler, self.base_lr, self.steps, **self._sched_params())
        self.last_lr = lr
        ss = None
        if self.grad_norm and self.grad_norm > 0:
            ss = torch.zeros((), dtype=torch.float64, device=self.flat_g.device)
            hip.sumsq(self.flat_g, ss)
        hip.adam_step(self.flat_p, self.flat_g, self.flat_m, self.flat_v, lr, self.betas[0], self.betas[1], self.eps, self.wd, self.steps,
                      grad_sumsq=ss, max_norm=self.grad_norm or 0.0)
        m._prep = None
        if self._bad_ids is None:
            self._bad_ids = hip.bad_ids_async(self.flat_p.device)
        losses = {k: v.detach() for k, v in losses.items()}
        if ss is not None:
            losses["grad_norm"] = ss.sqrt()
        return losses


class MatchaTTSTrainer(FastSpeech2Trainer):
    """`MatchaTTSTrainer._train_step` (jatts/trainers/matchatts.py:23-120) for the tts1 recipe (ground-truth durations; criterions
    CFMLoss + EncoderPriorLoss + DurationPredictorLoss, conf/matcha_tts.v1.prior.steplr.large.yaml): same flat-buffer optimiser,
    all-reduce and checkpoint layout as FastSpeech2Trainer; the duration loss joins once `steps > dp_train_start_steps`
    (trainers/matchatts.py:66-75).  ``cfm_t`` / ``cfm_noise`` in the batch inject the two random draws of CFM.compute_loss."""

    # graph mode: the CFM draws are device draws (graph-safe through torch's generator); the MAS model's search, its durations and the
    # Gaussian upsampling they drive all stay on the device, so its step captures too.  The loss schedule is part of the signature.

    def __init__(self, model, dp_train_start_steps=0, bin_loss_start_steps=0, lambda_align=2.0, **kw):
        super().__init__(model, **kw)
        self.dp_train_start_steps, self.bin_loss_start_steps, self.lambda_align = dp_train_start_steps, bin_loss_start_steps, lambda_align

    def _signature(self, batch):      # the loss schedule is part of the graph
        return super()._signature(batch) + (("schedule", self.steps > self.dp_train_start_steps, self.steps < self.dp_train_start_steps,
                                             self.steps > self.bin_loss_start_steps),)

    def compute_losses(self, batch):
        from .models.matchatts_train import criterion, train_forward
        m = self.model
        mas = m._MAS      # tts2 MatchaTTS_MAS: alignment module + MAS; + ForwardSumLoss / binarisation loss by schedule
        ret = train_forward(m, batch["xs"], batch["ilens"], batch["ys"], batch["olens"], batch.get("durations"), batch.get("duration_lens"),
                            spembs=batch.get("spkembs"), sids=batch.get("sids"), cfm_t=batch.get("cfm_t"), cfm_noise=batch.get("cfm_noise"),
                            seed=self._seed())
        return criterion(ret, batch.get("durations"), batch["ilens"], duration_loss=self.steps > self.dp_train_start_steps,
                         olens=batch["olens"], forward_sum=mas and self.steps < self.dp_train_start_steps,
                         bin_loss=mas and self.steps > self.bin_loss_start_steps, lambda_align=self.lambda_align)


class VITSTrainer(FastSpeech2Trainer):
    """`VITSTrainer._train_step` (jatts/trainers/vits.py:23-140) for the mel-VITS: lambda_mel x MelLoss + KLDivergenceLoss, the
    duration loss after `dp_train_start_steps`, lambda_align x ForwardSumLoss before it, lambda_align x the binarisation loss
    after `bin_loss_start_steps`; same flat-buffer optimiser / all-reduce / checkpoint layout.  ``post_noise`` in the batch
    injects the posterior encoder's random draw.  (gradient_accumulate_steps = 1.)"""

    def __init__(self, model, dp_train_start_steps=0, bin_loss_start_steps=0, lambda_align=2.0, lambda_mel=1.0, **kw):
        super().__init__(model, **kw)
        self.dp_train_start_steps, self.bin_loss_start_steps = dp_train_start_steps, bin_loss_start_steps
        self.lambda_align, self.lambda_mel = lambda_align, lambda_mel

    def _signature(self, batch):      # the loss schedule is part of the graph
        return super()._signature(batch) + (("schedule", self.steps > self.dp_train_start_steps, self.steps < self.dp_train_start_steps,
                                             self.steps > self.bin_loss_start_steps),)

    def compute_losses(self, batch):
        from .models.vits_train import criterion, train_forward
        m = self.model
        ret = train_forward(m, batch["xs"], batch["ilens"], batch["ys"], batch["olens"], batch["spkembs"], post_noise=batch.get("post_noise"),
                            seed=self._seed())
        return criterion(ret, batch["ilens"], batch["olens"], duration_loss=self.steps > self.dp_train_start_steps,
                         forward_sum=self.steps < self.dp_train_start_steps, bin_loss=self.steps > self.bin_loss_start_steps,
                         lambda_align=self.lambda_align, lambda_mel=self.lambda_mel)
