"""SURVEY §8 f.4, first slice: the FastSpeech2 criterion on forward()'s outputs (pinned on the reference's loss classes), the
backward of the conv op (against torch autograd in fp64 on the CPU) and one optimiser step through it."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import golden_state, load_golden, maxdiff, relerr
from jatts_amd.synthetic import FS2_SMALL

pytestmark = pytest.mark.gpu


def test_fastspeech2_losses_match_reference(cuda, lib):
    """forward() + criterion = `_train_step` up to gen_loss (trainers/fastspeech2.py:44-84), against the reference's own
    MelLoss / DurationPredictorLoss / PitchLoss / EnergyLoss evaluated on the reference's forward() (fs2_losses_small.npz)."""
    from jatts_amd.models import FastSpeech2
    from jatts_amd.training import fastspeech2_losses
    z, keys = load_golden("fs2_forward_small.npz")
    ref = np.load(__import__("os").path.join(__import__("helpers").GOLDEN, "fs2_losses_small.npz"))
    m = FastSpeech2(idim=20, **FS2_SMALL)
    m.load_state_dict(golden_state(keys, 0))
    m = m.to(cuda)
    t = lambda k: torch.tensor(z[k])  # noqa: E731
    il, ol = t("text_lengths"), t("feats_lengths")
    r = m(t("text"), il, t("feats"), ol, t("durations"), il, t("pitch"), il, t("energy"), il)
    got = fastspeech2_losses(r, t("durations"), t("pitch"), t("energy"), il)
    for k in ("mel_loss", "duration_loss", "pitch_loss", "energy_loss"):
        assert abs(float(got[k]) - float(ref[k])) <= 2e-4 * max(1.0, abs(float(ref[k]))), (k, float(got[k]), float(ref[k]))
    assert abs(float(got["loss"]) - sum(float(ref[k]) for k in ref.files)) <= 1e-3


@pytest.mark.parametrize("c_in,n_out,k,dil,lens", [(64, 96, 3, 1, [70, 5, 33]), (80, 64, 5, 2, [40, 41]), (128, 128, 1, 1, [129]),
                                                   (48, 20, 7, 3, [64, 30]), (192, 320, 5, 1, [300, 257, 64, 1]), (384, 200, 3, 1, [768, 500])])
def test_conv1d_backward_matches_autograd(cuda, lib, c_in, n_out, k, dil, lens):
    from jatts_amd import hip
    from jatts_amd.training import Conv1dFunction
    g = torch.Generator().manual_seed(c_in + n_out + k)
    R = sum(lens)
    x = torch.randn(R, c_in, generator=g)
    w = torch.randn(n_out, c_in, k, generator=g) / math.sqrt(c_in * k)
    b = torch.randn(n_out, generator=g) * 0.1
    gy = torch.randn(R, n_out, generator=g)
    pad = (k - 1) // 2 * dil
    # reference: torch autograd in fp64, one utterance at a time (zero padding at utterance boundaries)
    xr, wr, br = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    outs, o = [], 0
    for n in lens:
        outs.append(F.conv1d(xr[o:o + n].t().unsqueeze(0), wr, br, padding=pad, dilation=dil)[0].t())
        o += n
    yr = torch.cat(outs)
    yr.backward(gy.double())
    xd, wd, bd = x.to(cuda).requires_grad_(), w.to(cuda).requires_grad_(), b.to(cuda).requires_grad_()
    rb = hip.RaggedBatch(lens, cuda)
    y = Conv1dFunction.apply(xd, wd, bd, rb, dil, pad)
    y.backward(gy.to(cuda))
    assert relerr(y.detach(), yr.detach()) <= 2e-5
    assert relerr(xd.grad, xr.grad) <= 2e-5, relerr(xd.grad, xr.grad)
    assert relerr(wd.grad, wr.grad) <= 2e-5, relerr(wd.grad, wr.grad)
    assert relerr(bd.grad, br.grad) <= 2e-5


def test_one_optimiser_step_through_the_hip_conv(cuda, lib):
    """A two-layer ragged conv net trained for a few Adam steps on the HIP forward / backward: the loss goes down and the
    parameters follow the same trajectory as the torch reference model to f32 accuracy."""
    from jatts_amd import hip
    from jatts_amd.training import RaggedConv1d
    torch.manual_seed(0)
    lens = [50, 23, 64]
    rb = hip.RaggedBatch(lens, cuda)
    x = torch.randn(sum(lens), 64, device=cuda)
    tgt = torch.randn(sum(lens), 32, device=cuda)
    l1, l2 = RaggedConv1d(64, 128, 3).to(cuda), RaggedConv1d(128, 32, 5, dilation=2).to(cuda)
    r1, r2 = torch.nn.Conv1d(64, 128, 3, padding=1).to(cuda), torch.nn.Conv1d(128, 32, 5, padding=4, dilation=2).to(cuda)
    for a, b in ((l1, r1), (l2, r2)):
        b.weight.data.copy_(a.weight.data)
        b.bias.data.copy_(a.bias.data)
    opt = torch.optim.Adam(list(l1.parameters()) + list(l2.parameters()), lr=1e-2)
    opr = torch.optim.Adam(list(r1.parameters()) + list(r2.parameters()), lr=1e-2)

    def ref_forward():
        outs, o = [], 0
        for n in lens:
            h = torch.relu(r1(x[o:o + n].t().unsqueeze(0)))
            outs.append(r2(h)[0].t())
            o += n
        return torch.cat(outs)
    losses = []
    for _ in range(5):
        opt.zero_grad()
        y = l2(rb, torch.relu(l1(rb, x)))
        loss = ((y - tgt) ** 2).mean()
        loss.backward()
        opt.step()
        opr.zero_grad()
        lr_ = ((ref_forward() - tgt) ** 2).mean()
        lr_.backward()
        opr.step()
        losses.append(float(loss.detach()))
        assert abs(float(loss.detach()) - float(lr_.detach())) <= 1e-4 * max(1.0, float(lr_.detach()))
    assert losses[-1] < losses[0]
    assert maxdiff(l1.weight.data, r1.weight.data) <= 2e-4 and maxdiff(l2.weight.data, r2.weight.data) <= 2e-4


def _train_golden():
    import json
    z, keys = load_golden("fs2_train_small.npz")
    zi, _ = load_golden("fs2_forward_small.npz")
    cfg = json.loads(str(z["config"]))
    return z, zi, keys, cfg


@pytest.mark.parametrize("precision", ["fp32", "fp32_split"])
def test_fastspeech2_train_step_matches_reference(cuda, lib, precision):
    """(precision fp32_split, round 4: the forward / data-gradient convs on split f16 hi / lo MFMA operands -- held to the SAME tolerances.)
    One whole `_train_step` (trainers/fastspeech2.py:24-100) against the REAL reference run on the CPU
    (tests/golden/make_golden_train.py): train()-mode forward on the padded batch (batch-statistics BatchNorm, dropout 0),
    the four losses, EVERY parameter's gradient norm, full gradients of 22 parameters spread over all layer types, the total
    norm, the BatchNorm running statistics, and the parameters after clip_grad_norm_(1.0) + Adam under WarmupLR."""
    import json
    from jatts_amd.models import FastSpeech2
    from jatts_amd.training import FastSpeech2Trainer
    z, zi, keys, cfg = _train_golden()
    m = FastSpeech2(idim=20, **{**FS2_SMALL, **cfg})
    sd0 = golden_state(keys, 0)
    m.load_state_dict(sd0)
    m = m.to(cuda)
    t = lambda k: torch.tensor(zi[k])  # noqa: E731
    il, ol = t("text_lengths"), t("feats_lengths")
    batch = dict(xs=t("text"), ilens=il, ys=t("feats"), olens=ol, durations=t("durations"), duration_lens=il, pitch=t("pitch"),
                 pitch_lens=il, energys=t("energy"), energy_lens=il)
    tr = FastSpeech2Trainer(m, lr=0.0008, grad_norm=1.0, warmup_steps=4000, precision=precision)
    # forward + backward only first (gradients are consumed by the step)
    m.train()
    from jatts_amd import training as _tr
    from jatts_amd.models.fastspeech2_train import criterion
    _tr.SPLIT_CONVS[0] = precision == "fp32_split"          # (what the trainer sets around its own steps; reset at the end of the manual part)
    ret = m(batch["xs"], il, batch["ys"], ol, batch["durations"], il, batch["pitch"], il, batch["energys"], il)
    for k in ("before_outs", "after_outs", "d_outs", "p_outs", "e_outs"):
        assert relerr(ret[k].detach(), z["ref_" + k]) <= 2e-5, (k, relerr(ret[k].detach(), z["ref_" + k]))
    losses = criterion(ret, batch["durations"], batch["pitch"], batch["energys"], il)
    for k in ("mel_loss", "duration_loss", "pitch_loss", "energy_loss"):
        assert abs(float(losses[k]) - float(z[k])) <= 2e-5 * max(1.0, abs(float(z[k]))), (k, float(losses[k]), float(z[k]))
    losses["loss"].backward()
    _tr.SPLIT_CONVS[0] = False
    names = json.loads(str(z["grad_names"]))
    P = dict(m.named_parameters())
    worst = ("", 0.0)
    for n, ref_norm in zip(names, z["grad_norms"]):
        g = P[n].grad
        assert g is not None, n
        e = abs(float(g.norm()) - ref_norm) / max(ref_norm, 1e-3)   # (linear_k.bias and the conv biases in front of a batch-stat BatchNorm have exactly-zero true gradients: 1e-8 .. 1e-6 of rounding noise on both sides)
        worst = max(worst, (n, e), key=lambda v: v[1])
        assert e <= 2e-3, (n, float(g.norm()), ref_norm)
    for f in z.files:
        if f.startswith("grad:"):
            e = relerr(P[f[5:]].grad, z[f])
            assert e <= 2e-3, (f, e)
    tot = math.sqrt(sum(float(P[n].grad.double().pow(2).sum()) for n in names))
    assert abs(tot - float(z["total_grad_norm"])) <= 1e-3 * float(z["total_grad_norm"])
    B = dict(m.named_buffers())
    for f in z.files:
        if f.startswith("buf:"):
            assert relerr(B[f[4:]], z[f]) <= 2e-5, f
    # the whole step from the same starting point: parameters after clip + Adam + WarmupLR(step 1)
    m2 = FastSpeech2(idim=20, **{**FS2_SMALL, **cfg})
    m2.load_state_dict(sd0)
    m2 = m2.to(cuda)
    tr = FastSpeech2Trainer(m2, lr=0.0008, grad_norm=1.0, warmup_steps=4000, precision=precision)
    out = tr.train_step(batch)
    assert abs(tr.last_lr - float(z["lr_step1"])) <= 1e-12
    assert abs(float(out["grad_norm"]) - float(z["total_grad_norm"])) <= 1e-3 * float(z["total_grad_norm"])
    P2 = dict(m2.named_parameters())
    for f in z.files:
        if f.startswith("after:"):
            n = f[6:]
            before, after_ref, after = sd0[n].double(), torch.tensor(z[f]).double(), P2[n].detach().cpu().double()
            step_ref, step = after_ref - before, after - before
            # the first Adam step is -lr * g / (|g| + eps): compare the update itself, not the (1e7 x larger) parameter
            # (plus one f32 ulp of the parameter on either side: the update is 2e-7 on values of order 0.1)
            tol = 0.05 * tr.last_lr + 2.0 * float(before.abs().max()) * 2.0 ** -23
            assert float((step - step_ref).abs().max()) <= tol, (n, float((step - step_ref).abs().max()), tol)
            assert float(step.abs().max()) > 0.5 * tr.last_lr


def test_fastspeech2_training_reduces_the_loss(cuda, lib):
    """Twelve `_train_step`s with the v1 recipe's dropout rates ON (counter-based masks), lr without warm-up: the loss falls,
    everything stays finite, eval-mode forward() afterwards sees the updated weights."""
    from jatts_amd.models import FastSpeech2
    from jatts_amd.training import FastSpeech2Trainer
    z, zi, keys, cfg = _train_golden()
    m = FastSpeech2(idim=20, **{**FS2_SMALL, "stop_gradient_from_pitch_predictor": True, "use_masking": True})
    m.load_state_dict(golden_state(keys, 0))
    m = m.to(cuda)
    t = lambda k: torch.tensor(zi[k])  # noqa: E731
    il, ol = t("text_lengths"), t("feats_lengths")
    batch = dict(xs=t("text"), ilens=il, ys=t("feats"), olens=ol, durations=t("durations"), duration_lens=il, pitch=t("pitch"),
                 pitch_lens=il, energys=t("energy"), energy_lens=il)
    m.eval()
    y0 = m(batch["xs"], il, batch["ys"], ol, batch["durations"], il, batch["pitch"], il, batch["energys"], il)["after_outs"].clone()
    tr = FastSpeech2Trainer(m, lr=2e-3, grad_norm=1.0, warmup_steps=0)
    hist = [float(tr.train_step(batch)["loss"]) for _ in range(12)]
    assert all(math.isfinite(v) for v in hist)
    assert min(hist[-3:]) < 0.8 * hist[0], hist
    m.eval()
    y1 = m(batch["xs"], il, batch["ys"], ol, batch["durations"], il, batch["pitch"], il, batch["energys"], il)["after_outs"]
    assert maxdiff(y0, y1) > 1e-3 and bool(torch.isfinite(y1).all())


def test_trainer_checkpoint_is_the_reference_format_and_resumes(cuda, lib, tmp_path):
    """jatts/trainers/base.py:85-124: {"model", "optimizer", "scheduler", "steps", "epochs"} with a torch.optim.Adam state_dict
    (torch's own Adam must accept it, indexed like the reference's parameter order); a resumed trainer continues bit-for-bit."""
    import json
    from jatts_amd.models import FastSpeech2
    from jatts_amd.training import FastSpeech2Trainer
    z, zi, keys, cfg = _train_golden()
    t = lambda k: torch.tensor(zi[k])  # noqa: E731
    il, ol = t("text_lengths"), t("feats_lengths")
    batch = dict(xs=t("text"), ilens=il, ys=t("feats"), olens=ol, durations=t("durations"), duration_lens=il, pitch=t("pitch"),
                 pitch_lens=il, energys=t("energy"), energy_lens=il)

    def make():
        m = FastSpeech2(idim=20, **{**FS2_SMALL, **cfg})
        m.load_state_dict(golden_state(keys, 0))
        return m.to(cuda)
    a = FastSpeech2Trainer(make(), lr=1e-3, grad_norm=1.0, warmup_steps=10)
    assert [n for n, _ in a.model.named_parameters()] == json.loads(str(z["grad_names"]))   # the reference's parameter order
    for _ in range(2):
        a.train_step(batch)
    path = str(tmp_path / "ckpt" / "checkpoint-2steps.pkl")
    a.save_checkpoint(path, epochs=1)
    ck = torch.load(path, map_location="cpu")
    assert set(ck) == {"model", "optimizer", "scheduler", "steps", "epochs"} and ck["steps"] == 2
    ref_model = make()
    ref_model.train()
    topt = torch.optim.Adam(ref_model.parameters(), lr=1e-3)
    topt.load_state_dict(ck["optimizer"])                                                     # torch accepts the layout
    assert len(topt.state_dict()["state"]) == len(list(ref_model.parameters()))
    b = FastSpeech2Trainer(make(), lr=1e-3, grad_norm=1.0, warmup_steps=10)
    b.load_checkpoint(path)
    assert b.steps == 2
    la, lb = a.train_step(batch), b.train_step(batch)
    assert abs(float(la["loss"]) - float(lb["loss"])) <= 1e-6 * abs(float(la["loss"]))
    assert maxdiff(a.flat_p, b.flat_p) <= 5e-7 and abs(a.last_lr - b.last_lr) == 0.0      # (one f32 ulp at |p| ~ 1 is 1.2e-7: the atomics-based reductions)


def test_fastspeech2_speaker_conditioned_train_step_matches_reference(cuda, lib):
    """The same step with spk_embed_dim=16 ("add" integration: F.normalize -> projection) and 3 speaker ids (sid_emb), against the
    real reference (fs2_train_spk_small.npz): outputs, losses, every parameter's gradient norm, the two new parameters' gradients."""
    import json
    from jatts_amd.models import FastSpeech2
    from jatts_amd.models.fastspeech2_train import criterion
    _, zi, _, cfg = _train_golden()
    z, keys = load_golden("fs2_train_spk_small.npz")
    m = FastSpeech2(idim=20, **{**FS2_SMALL, **cfg, "spk_embed_dim": 16, "spks": 3})
    m.load_state_dict(golden_state(keys, 0))
    m = m.to(cuda).train()
    t = lambda k: torch.tensor(zi[k])  # noqa: E731
    il, ol = t("text_lengths"), t("feats_lengths")
    ret = m(t("text"), il, t("feats"), ol, t("durations"), il, t("pitch"), il, t("energy"), il, spembs=torch.tensor(z["spembs"]),
            sids=torch.tensor(z["sids"]))
    assert relerr(ret["after_outs"].detach(), z["ref_after_outs"]) <= 2e-5
    losses = criterion(ret, t("durations"), t("pitch"), t("energy"), il)
    for k in ("mel_loss", "duration_loss", "pitch_loss", "energy_loss"):
        assert abs(float(losses[k]) - float(z[k])) <= 2e-5 * max(1.0, abs(float(z[k]))), k
    losses["loss"].backward()
    P = dict(m.named_parameters())
    for n, ref_norm in zip(json.loads(str(z["grad_names"])), z["grad_norms"]):
        assert abs(float(P[n].grad.norm()) - ref_norm) / max(ref_norm, 1e-3) <= 2e-3, n
    for n in ("projection.weight", "sid_emb.weight"):
        assert relerr(P[n].grad, z["grad:" + n]) <= 2e-3, n


def test_matcha_tts1_train_step_matches_reference(cuda, lib):
    """One `_train_step` of jatts/trainers/matchatts.py:23-120 for the tts1 MatchaTTS against the REAL reference on the CPU
    (make_golden_train.py -> matcha_tts1_train_small.npz; train() mode, dropout 0, CFM draws injected, the diffusers attention
    stand-in of the forward goldens): the three losses, d_outs, every parameter's gradient norm, 12 full gradients (U-Net GroupNorm /
    SnakeBeta log-parameters / strided and transposed convs / time MLP / attention, encoder_proj, the text encoder)."""
    import json
    from jatts_amd.models import MatchaTTS
    from jatts_amd.models.matchatts_train import criterion
    z, keys = load_golden("matcha_tts1_train_small.npz")
    zi, _ = load_golden("matcha_tts1_forward_small.npz")
    cfg = json.loads(str(z["config"]))
    from jatts_amd.synthetic import matcha_golden_tweaks
    m = MatchaTTS(idim=20, **cfg)
    m.load_state_dict(matcha_golden_tweaks(golden_state(keys, 4)))
    m = m.to(cuda).train()
    t = lambda k: torch.tensor(zi[k])  # noqa: E731
    il, ol = t("text_lengths"), t("feats_lengths")
    ret = m(t("text"), il, t("feats"), ol, t("durations"), il, cfm_t=t("t"), cfm_noise=t("z"))
    assert relerr(ret["d_outs"].detach(), z["ref_d_outs"]) <= 2e-5
    losses = criterion(ret, t("durations"), il)
    for k in ("cfm_loss", "encoder_prior_loss", "duration_loss"):
        assert abs(float(losses[k]) - float(z[k])) <= 3e-5 * max(1.0, abs(float(z[k]))), (k, float(losses[k]), float(z[k]))
    losses["loss"].backward()
    P = dict(m.named_parameters())
    names = json.loads(str(z["grad_names"]))
    assert [n for n, _ in m.named_parameters()] == names
    for n, ref_norm in zip(names, z["grad_norms"]):
        assert P[n].grad is not None, n
        assert abs(float(P[n].grad.norm()) - ref_norm) / max(ref_norm, 1e-3) <= 3e-3, (n, float(P[n].grad.norm()), ref_norm)
    for f in z.files:
        if f.startswith("grad:"):
            assert relerr(P[f[5:]].grad, z[f]) <= 3e-3, (f, relerr(P[f[5:]].grad, z[f]))


def test_matcha_tts1_training_reduces_the_loss(cuda, lib):
    """MatchaTTSTrainer: ten steps with the recipe's dropout rates on and fixed CFM draws: the loss falls and stays finite; the
    duration loss joins after the first step (dp_train_start_steps = 0: `steps > 0`)."""
    import json
    from jatts_amd.models import MatchaTTS
    from jatts_amd.synthetic import matcha_golden_tweaks
    from jatts_amd.training import MatchaTTSTrainer
    z, keys = load_golden("matcha_tts1_train_small.npz")
    zi, _ = load_golden("matcha_tts1_forward_small.npz")
    cfg = {**json.loads(str(z["config"])), "transformer_enc_dropout_rate": 0.1, "decoder_dropout": 0.05}
    m = MatchaTTS(idim=20, **cfg)
    m.load_state_dict(matcha_golden_tweaks(golden_state(keys, 4)))
    m = m.to(cuda)
    t = lambda k: torch.tensor(zi[k])  # noqa: E731
    il = t("text_lengths")
    batch = dict(xs=t("text"), ilens=il, ys=t("feats"), olens=t("feats_lengths"), durations=t("durations"), duration_lens=il,
                 cfm_t=t("t"), cfm_noise=t("z"))
    tr = MatchaTTSTrainer(m, lr=1e-3, grad_norm=1.0, warmup_steps=0)
    out = [tr.train_step(batch) for _ in range(10)]
    assert "duration_loss" not in out[0] and "duration_loss" in out[1]
    cfm = [float(o["cfm_loss"]) + float(o["encoder_prior_loss"]) for o in out]
    assert all(math.isfinite(v) for v in cfm) and min(cfm[-3:]) < 0.9 * cfm[0], cfm


def test_matcha_tts1_graph_mode_replays_the_same_training(cuda, lib):
    """MatchaTTSTrainer(capture_graph=True) on the tts1 model: step 1 eager (no duration loss), step 2 eager again (the loss schedule
    is part of the signature: the duration loss joined), step 3 captured, later steps replayed -- same losses and gradients as an eager
    trainer on the same data and injected CFM draws, dropout on.  Without injected draws the graph draws them on the device each
    replay (torch's generator is graph-safe): losses differ from replay to replay and stay finite."""
    import json
    from jatts_amd.models import MatchaTTS
    from jatts_amd.synthetic import matcha_golden_tweaks
    from jatts_amd.training import MatchaTTSTrainer
    z, keys = load_golden("matcha_tts1_train_small.npz")
    zi, _ = load_golden("matcha_tts1_forward_small.npz")
    cfg = {**json.loads(str(z["config"])), "transformer_enc_dropout_rate": 0.1, "decoder_dropout": 0.05}

    def make():
        m = MatchaTTS(idim=20, **cfg)
        m.load_state_dict(matcha_golden_tweaks(golden_state(keys, 4)))
        return m.to(cuda)
    t = lambda k: torch.tensor(zi[k])  # noqa: E731
    il = t("text_lengths")
    batch = dict(xs=t("text"), ilens=il, ys=t("feats"), olens=t("feats_lengths"), durations=t("durations"), duration_lens=il,
                 cfm_t=t("t"), cfm_noise=t("z"))
    a = MatchaTTSTrainer(make(), lr=1e-3, grad_norm=1.0, warmup_steps=0)
    b = MatchaTTSTrainer(make(), lr=1e-3, grad_norm=1.0, warmup_steps=0, capture_graph=True)
    g = torch.Generator().manual_seed(1)
    for step in range(6):
        cur = dict(batch)
        if step >= 4:
            cur["cfm_noise"] = torch.randn(batch["cfm_noise"].shape, generator=g)
        la, lb = a.train_step(cur), b.train_step(cur)
        assert set(la) == set(lb)
        for k in la:
            tol = 2e-5 * (1 + 5 * step)
            assert abs(float(la[k]) - float(lb[k])) <= tol * max(1.0, abs(float(la[k]))), (step, k, float(la[k]), float(lb[k]))
        assert a.steps == b.steps == step + 1 and a.last_lr == b.last_lr
        assert maxdiff(a.flat_g, b.flat_g) <= 2e-5 * (1 + 5 * step) * max(1.0, float(a.flat_g.abs().max())), (step, maxdiff(a.flat_g, b.flat_g))
    assert len(b._graphs) == 2                                 # (no duration loss) eager only; (duration loss) captured
    assert sum(st["graph"] is not None for st in b._graphs.values()) == 1
    # device draws inside the graph
    free = {k: v for k, v in batch.items() if not k.startswith("cfm_")}
    c = MatchaTTSTrainer(make(), lr=1e-3, grad_norm=1.0, warmup_steps=0, capture_graph=True)
    vals = [float(c.train_step(free)["cfm_loss"]) for _ in range(6)]
    assert all(math.isfinite(v) for v in vals) and len({round(v, 6) for v in vals[3:]}) == 3, vals


def test_matcha_mas_train_step_matches_reference(cuda, lib):
    """MatchaTTS_MAS (tts2 recipe) with every loss term of jatts/trainers/matchatts.py:47-103 switched on at once -- CFM + prior +
    duration + 2 x ForwardSumLoss (beta-binomial prior, CTC) + 2 x binarisation -- against the REAL reference on the CPU
    (matcha_mas_train_small.npz): durations from the alignment search, the five losses, every parameter's gradient norm, six full
    gradients (alignment module, encoder_proj, text encoder, duration predictor)."""
    import json
    from jatts_amd.models import MatchaTTS_MAS
    from jatts_amd.models.matchatts_train import criterion
    from jatts_amd.synthetic import matcha_golden_tweaks
    z, keys = load_golden("matcha_mas_train_small.npz")
    zi, _ = load_golden("matcha_forward_small.npz")
    m = MatchaTTS_MAS(idim=20, **json.loads(str(z["config"])))
    m.load_state_dict(matcha_golden_tweaks(golden_state(keys, 3)))
    m = m.to(cuda).train()
    t = lambda k: torch.tensor(zi[k])  # noqa: E731
    il, ol = t("text_lengths"), t("feats_lengths")
    ret = m(t("text"), il, t("feats"), ol, cfm_t=t("t"), cfm_noise=t("z"))
    assert torch.equal(ret["ds"].cpu(), torch.tensor(z["ref_ds"]))
    losses = criterion(ret, None, il, duration_loss=True, olens=ol, forward_sum=True, bin_loss=True, lambda_align=2.0)
    for k in ("cfm_loss", "encoder_prior_loss", "duration_loss", "forward_sum_loss", "bin_loss"):
        assert abs(float(losses[k].detach()) - float(z[k])) <= 3e-5 * max(1.0, abs(float(z[k]))), (k, float(losses[k].detach()), float(z[k]))
    losses["loss"].backward()
    P = dict(m.named_parameters())
    names = json.loads(str(z["grad_names"]))
    assert [n for n, _ in m.named_parameters()] == names
    for n, ref_norm in zip(names, z["grad_norms"]):
        assert P[n].grad is not None, n
        assert abs(float(P[n].grad.norm()) - ref_norm) / max(ref_norm, 1e-3) <= 3e-3, (n, float(P[n].grad.norm()), ref_norm)
    for f in z.files:
        if f.startswith("grad:"):
            assert relerr(P[f[5:]].grad, z[f]) <= 3e-3, (f, relerr(P[f[5:]].grad, z[f]))


def test_matcha_mas_trainer_schedule(cuda, lib):
    """MatchaTTSTrainer on MatchaTTS_MAS with the recipe's schedule shape (dp_train_start_steps 3, bin_loss_start_steps 5): forward-sum
    loss first, the duration loss after step 3, the binarisation loss after step 5; all finite."""
    import json
    from jatts_amd.models import MatchaTTS_MAS
    from jatts_amd.synthetic import matcha_golden_tweaks
    from jatts_amd.training import MatchaTTSTrainer
    z, keys = load_golden("matcha_mas_train_small.npz")
    zi, _ = load_golden("matcha_forward_small.npz")
    m = MatchaTTS_MAS(idim=20, **json.loads(str(z["config"])))
    m.load_state_dict(matcha_golden_tweaks(golden_state(keys, 3)))
    m = m.to(cuda)
    t = lambda k: torch.tensor(zi[k])  # noqa: E731
    batch = dict(xs=t("text"), ilens=t("text_lengths"), ys=t("feats"), olens=t("feats_lengths"), cfm_t=t("t"), cfm_noise=t("z"))
    tr = MatchaTTSTrainer(m, dp_train_start_steps=3, bin_loss_start_steps=5, lr=5e-4, grad_norm=1.0, warmup_steps=0)
    out = [tr.train_step(batch) for _ in range(7)]
    assert "forward_sum_loss" in out[0] and "duration_loss" not in out[0] and "bin_loss" not in out[0]
    assert "forward_sum_loss" not in out[4] and "duration_loss" in out[4] and "bin_loss" not in out[4]
    assert "bin_loss" in out[6] and "duration_loss" in out[6]
    assert all(math.isfinite(float(o["loss"])) for o in out)
    assert float(out[2]["forward_sum_loss"]) < float(out[0]["forward_sum_loss"])


def test_mas_trainers_graph_mode(cuda, lib):
    """capture_graph=True on the two MAS trainers (MatchaTTS_MAS, VITS): the alignment search, its durations and everything downstream
    stay on the device, so the step captures; the loss schedule (forward-sum -> duration -> binarisation) changes the signature, each
    phase is captured on its second step.  Same losses as the eager trainer on injected draws, dropout off in these small configs."""
    import json
    from jatts_amd.models import VITS, MatchaTTS_MAS
    from jatts_amd.synthetic import matcha_golden_tweaks
    from jatts_amd.training import MatchaTTSTrainer, VITSTrainer
    t_ = lambda z, k: torch.tensor(z[k])  # noqa: E731
    z, keys = load_golden("matcha_mas_train_small.npz")
    zi, _ = load_golden("matcha_forward_small.npz")

    def make_m():
        m = MatchaTTS_MAS(idim=20, **json.loads(str(z["config"])))
        m.load_state_dict(matcha_golden_tweaks(golden_state(keys, 3)))
        return m.to(cuda)
    mb = dict(xs=t_(zi, "text"), ilens=t_(zi, "text_lengths"), ys=t_(zi, "feats"), olens=t_(zi, "feats_lengths"), cfm_t=t_(zi, "t"),
              cfm_noise=t_(zi, "z"))
    zv, keys_v = load_golden("vits_train_small.npz")
    zvi, _ = load_golden("vits_forward_small.npz")

    def make_v():
        m = VITS(idim=20, **json.loads(str(zv["config"])))
        m.load_state_dict(golden_state(keys_v, 2))
        return m.to(cuda)
    vb = dict(xs=t_(zvi, "text"), ilens=t_(zvi, "text_lengths"), ys=t_(zvi, "feats"), olens=t_(zvi, "feats_lengths"), spkembs=t_(zvi, "spembs"),
              post_noise=t_(zvi, "noise"))
    for cls, make, batch in ((MatchaTTSTrainer, make_m, mb), (VITSTrainer, make_v, vb)):
        kw = dict(dp_train_start_steps=3, bin_loss_start_steps=6, lr=2e-4, grad_norm=1.0, warmup_steps=0)
        a, b = cls(make(), **kw), cls(make(), capture_graph=True, **kw)
        assert b.capture_graph
        for step in range(11):
            la, lb = a.train_step(batch), b.train_step(batch)
            assert set(la) == set(lb), (step, sorted(la), sorted(lb))
            for k in la:
                tol = 5e-5 * (1 + 5 * step)
                assert abs(float(la[k]) - float(lb[k])) <= tol * max(1.0, abs(float(la[k]))), (cls.__name__, step, k, float(la[k]), float(lb[k]))
        # phases: steps 0-2 (forward-sum), 3 (none of the three), 4-6 (duration), 7.. (duration + binarisation)
        assert len(b._graphs) == 4
        assert sum(st["graph"] is not None for st in b._graphs.values()) == 3      # the one-step phase never reaches its capture


def test_vits_train_step_matches_reference(cuda, lib):
    """mel-VITS with every loss term of jatts/trainers/vits.py:47-110 on at once -- mel + KL + duration + 2 x ForwardSumLoss + 2 x
    binarisation -- against the REAL reference on the CPU (vits_train_small.npz): MAS durations, the five losses, every parameter's
    gradient norm, ten full gradients (weight-normalised WaveNet g / v, flow projection, text-encoder statistics projection, the
    new-style relative-position parameters of both conformers, feat_out, the speaker projection)."""
    import json
    from jatts_amd.models import VITS
    from jatts_amd.models.vits_train import criterion
    z, keys = load_golden("vits_train_small.npz")
    zi, _ = load_golden("vits_forward_small.npz")
    m = VITS(idim=20, **json.loads(str(z["config"])))
    m.load_state_dict(golden_state(keys, 2))
    m = m.to(cuda).train()
    t = lambda k: torch.tensor(zi[k])  # noqa: E731
    il, ol = t("text_lengths"), t("feats_lengths")
    ret = m(t("text"), il, t("feats"), ol, spembs=t("spembs"), post_noise=t("noise"))
    assert torch.equal(ret["ds"].cpu(), torch.tensor(z["ref_ds"]))
    losses = criterion(ret, il, ol, duration_loss=True, forward_sum=True, bin_loss=True, lambda_align=2.0)
    for k in ("mel_loss", "kl_loss", "duration_loss", "forward_sum_loss", "bin_loss"):
        assert abs(float(losses[k].detach()) - float(z[k])) <= 3e-5 * max(1.0, abs(float(z[k]))), (k, float(losses[k].detach()), float(z[k]))
    losses["loss"].backward()
    P = dict(m.named_parameters())
    names = json.loads(str(z["grad_names"]))
    assert [n for n, _ in m.named_parameters()] == names
    floor = 1e-5 * float(np.sqrt((z["grad_norms"] ** 2).sum()))   # exactly-zero true gradients (conv biases in front of a batch-stat
    for n, ref_norm in zip(names, z["grad_norms"]):                # BatchNorm, linear_k.bias) carry rounding noise of the global scale
        assert P[n].grad is not None, n
        assert abs(float(P[n].grad.norm()) - ref_norm) / max(ref_norm, floor, 1e-3) <= 3e-3, (n, float(P[n].grad.norm()), ref_norm)
    for f in z.files:
        if f.startswith("grad:"):
            assert relerr(P[f[5:]].grad, z[f]) <= 3e-3, (f, relerr(P[f[5:]].grad, z[f]))


def test_vits_trainer_steps(cuda, lib):
    """VITSTrainer with the recipe's schedule shape: forward-sum first, duration after dp_train_start_steps, finite, mel loss falling."""
    import json
    from jatts_amd.models import VITS
    from jatts_amd.training import VITSTrainer
    z, keys = load_golden("vits_train_small.npz")
    zi, _ = load_golden("vits_forward_small.npz")
    m = VITS(idim=20, **{**json.loads(str(z["config"])), "text_encoder_dropout_rate": 0.1, "transformer_dec_dropout_rate": 0.1})
    m.load_state_dict(golden_state(keys, 2))
    m = m.to(cuda)
    t = lambda k: torch.tensor(zi[k])  # noqa: E731
    batch = dict(xs=t("text"), ilens=t("text_lengths"), ys=t("feats"), olens=t("feats_lengths"), spkembs=t("spembs"), post_noise=t("noise"))
    tr = VITSTrainer(m, dp_train_start_steps=3, bin_loss_start_steps=4, lr=1e-3, grad_norm=1.0, warmup_steps=0)
    out = [tr.train_step(batch) for _ in range(8)]
    assert "forward_sum_loss" in out[0] and "duration_loss" not in out[0]
    assert "duration_loss" in out[4] and "forward_sum_loss" not in out[4] and "bin_loss" in out[5]
    assert all(math.isfinite(float(o["loss"])) for o in out)
    assert float(out[-1]["mel_loss"]) < float(out[0]["mel_loss"])


@pytest.mark.parametrize("n,c,k", [(96, 80, 3), (1, 256, 1), (384, 1, 1), (64, 64, 5), (33, 17, 4)])
def test_pack_conv_weight_kernel_matches_the_host_packing(cuda, lib, n, c, k):
    from jatts_amd import hip
    g = torch.Generator().manual_seed(n + c + k)
    w = torch.randn(n, c, k, generator=g).to(cuda)
    for dt in (hip.F32, hip.F16):
        got, c_pad = hip.pack_conv_weight_dev(w, dt)
        assert c_pad == hip.round_up(c, 64) and torch.equal(got, hip.pack_conv_weight(w, dt))
        got, c_pad = hip.pack_conv_weight_dev(w, dt, dgrad=True)
        assert c_pad == hip.round_up(n, 64) and torch.equal(got, hip.pack_conv_weight(w.permute(1, 0, 2).flip(2).contiguous(), dt))


@pytest.mark.parametrize("n,c,k", [(96, 80, 3), (1, 256, 1), (384, 1, 1), (64, 64, 5), (33, 17, 4)])
def test_split_weight_packer_on_the_device_equals_the_host_packing(cuda, lib, n, c, k):
    """jatts_pack_conv_weight_split (per-row maxima + pack, two launches: the training step re-packs every weight twice per step) is bit-identical
    to hip.pack_conv_weight_split -- packed halves and inverse scales -- for the weight itself and for the data-gradient operand; rows of
    zeros and padding rows take scale 1."""
    from jatts_amd import hip
    g = torch.Generator().manual_seed(n + c + k)
    w = (torch.randn(n, c, k, generator=g) * torch.pow(10.0, torch.rand(n, 1, 1, generator=g) * 6 - 4)).to(cuda)
    if n > 2:
        w[1] = 0.0
    got, inv, c_pad = hip.pack_conv_weight_split_dev(w)
    ref, rinv = hip.pack_conv_weight_split(w, 64)
    assert c_pad == hip.round_up(c, 64) and torch.equal(inv, rinv) and torch.equal(got, ref)
    got, inv, c_pad = hip.pack_conv_weight_split_dev(w, dgrad=True)
    ref, rinv = hip.pack_conv_weight_split(w.permute(1, 0, 2).flip(2).contiguous(), 64)
    assert c_pad == hip.round_up(n, 64) and torch.equal(inv, rinv) and torch.equal(got, ref)


def test_fastspeech2_train_step_nine_tap_embeddings(cuda, lib):
    """The constructor-default 9-tap pitch / energy embedding convolutions (Conv1d(1, adim, 9): the MFMA conv with its single input
    channel zero-padded) and no stop-gradient on the pitch predictor, against the real reference (fs2_train_embed9_small.npz)."""
    import json
    from jatts_amd.models import FastSpeech2
    from jatts_amd.models.fastspeech2_train import criterion
    _, zi, _, _ = _train_golden()
    z, keys = load_golden("fs2_train_embed9_small.npz")
    m = FastSpeech2(idim=20, **{**FS2_SMALL, **json.loads(str(z["config"]))})
    m.load_state_dict(golden_state(keys, 0))
    m = m.to(cuda).train()
    t = lambda k: torch.tensor(zi[k])  # noqa: E731
    il, ol = t("text_lengths"), t("feats_lengths")
    ret = m(t("text"), il, t("feats"), ol, t("durations"), il, t("pitch"), il, t("energy"), il)
    losses = criterion(ret, t("durations"), t("pitch"), t("energy"), il)
    for k in ("mel_loss", "duration_loss", "pitch_loss", "energy_loss"):
        assert abs(float(losses[k].detach()) - float(z[k])) <= 2e-5 * max(1.0, abs(float(z[k]))), k
    losses["loss"].backward()
    P = dict(m.named_parameters())
    for n, ref_norm in zip(json.loads(str(z["grad_names"])), z["grad_norms"]):
        assert abs(float(P[n].grad.norm()) - ref_norm) / max(ref_norm, 1e-3) <= 2e-3, (n, float(P[n].grad.norm()), ref_norm)
    assert relerr(P["pitch_embed.0.weight"].grad, z["grad:pitch_embed.0.weight"]) <= 2e-3


def test_gradient_accumulation_equals_one_big_step(cuda, lib):
    """trainers/base.py:135 / vits.py:113-121: with gradient_accumulate_steps = 2, two calls on the same micro-batch (each loss / 2)
    make ONE optimiser step equal to a plain step on that batch; `steps` counts optimiser steps."""
    from jatts_amd.models import FastSpeech2
    from jatts_amd.training import FastSpeech2Trainer
    z, zi, keys, cfg = _train_golden()
    t = lambda k: torch.tensor(zi[k])  # noqa: E731
    il, ol = t("text_lengths"), t("feats_lengths")
    batch = dict(xs=t("text"), ilens=il, ys=t("feats"), olens=ol, durations=t("durations"), duration_lens=il, pitch=t("pitch"),
                 pitch_lens=il, energys=t("energy"), energy_lens=il)

    def make():
        m = FastSpeech2(idim=20, **{**FS2_SMALL, **cfg})
        m.load_state_dict(golden_state(keys, 0))
        return m.to(cuda)
    a = FastSpeech2Trainer(make(), lr=1e-3, grad_norm=1.0, warmup_steps=0, gradient_accumulate_steps=2)
    b = FastSpeech2Trainer(make(), lr=1e-3, grad_norm=1.0, warmup_steps=0)
    o1 = a.train_step(batch)
    assert a.steps == 0 and "grad_norm" not in o1 and maxdiff(a.flat_p, b.flat_p) == 0.0      # nothing applied yet
    o2 = a.train_step(batch)
    ob = b.train_step(batch)
    assert a.steps == 1 and b.steps == 1
    assert abs(float(o2["grad_norm"]) - float(ob["grad_norm"])) <= 1e-4 * float(ob["grad_norm"])
    assert maxdiff(a.flat_p, b.flat_p) <= 2e-6          # first Adam step ~ lr * sign(g): equal up to gradient rounding noise near g = 0


def test_eval_step_equals_the_reference_eval_losses(cuda, lib):
    """Trainer.eval_step = forward + criterion in eval() mode without gradients: on the forward() golden batch it must give the losses
    the reference's loss classes give on the reference's eval-mode forward() (fs2_losses_small.npz), leave the parameters, the
    BatchNorm running statistics and the step counter alone, and return to train mode."""
    import os
    from helpers import GOLDEN
    from jatts_amd.models import FastSpeech2
    from jatts_amd.training import FastSpeech2Trainer
    z, keys = load_golden("fs2_forward_small.npz")
    ref = np.load(os.path.join(GOLDEN, "fs2_losses_small.npz"))
    m = FastSpeech2(idim=20, **FS2_SMALL)
    m.load_state_dict(golden_state(keys, 0))
    m = m.to(cuda)
    t = lambda k: torch.tensor(z[k])  # noqa: E731
    il, ol = t("text_lengths"), t("feats_lengths")
    batch = dict(xs=t("text"), ilens=il, ys=t("feats"), olens=ol, durations=t("durations"), duration_lens=il, pitch=t("pitch"),
                 pitch_lens=il, energys=t("energy"), energy_lens=il)
    tr = FastSpeech2Trainer(m, lr=1e-3, warmup_steps=0)
    p0 = tr.flat_p.clone()
    rm0 = dict(m.named_buffers())["postnet.postnet.0.1.running_mean"].clone()
    out = tr.eval_step(batch)
    for k in ("mel_loss", "duration_loss", "pitch_loss", "energy_loss"):
        assert abs(float(out[k]) - float(ref[k])) <= 2e-4 * max(1.0, abs(float(ref[k]))), (k, float(out[k]), float(ref[k]))
    assert m.training and tr.steps == 0 and torch.equal(tr.flat_p, p0)
    assert torch.equal(dict(m.named_buffers())["postnet.postnet.0.1.running_mean"], rm0)


def test_steplr_checkpoint_resumes_under_torch_with_the_decay(cuda, lib, tmp_path):
    """A checkpoint written at steps == step_size must hold the lr of the NEXT step: torch's StepLR is chainable and reads
    group["lr"] back, so a resumed reference trainer (Adam + StepLR, scheduler.step() after optimizer.step()) applies the decayed
    lr at step step_size + 1 exactly like this trainer does.  Also: a StepLR checkpoint's step_size / gamma are restored."""
    from jatts_amd.models import FastSpeech2
    from jatts_amd.training import FastSpeech2Trainer, scheduled_lr
    z, zi, keys, cfg = _train_golden()
    t = lambda k: torch.tensor(zi[k])  # noqa: E731
    il, ol = t("text_lengths"), t("feats_lengths")
    batch = dict(xs=t("text"), ilens=il, ys=t("feats"), olens=ol, durations=t("durations"), duration_lens=il, pitch=t("pitch"),
                 pitch_lens=il, energys=t("energy"), energy_lens=il)

    def make():
        m = FastSpeech2(idim=20, **{**FS2_SMALL, **cfg})
        m.load_state_dict(golden_state(keys, 0))
        return m.to(cuda)
    a = FastSpeech2Trainer(make(), lr=1e-3, grad_norm=1.0, scheduler="steplr", scheduler_params={"step_size": 2, "gamma": 0.5})
    for _ in range(2):
        a.train_step(batch)
    assert a.last_lr == 1e-3
    path = str(tmp_path / "checkpoint-2steps.pkl")
    a.save_checkpoint(path)
    ck = torch.load(path, map_location="cpu")
    # the reference's resume: torch Adam + StepLR load the two state_dicts (trainers/base.py:110-124)
    ref = make()
    ref.train()
    topt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    tsch = torch.optim.lr_scheduler.StepLR(topt, step_size=2, gamma=0.5)
    topt.load_state_dict(ck["optimizer"])
    tsch.load_state_dict(ck["scheduler"])
    lrs = []
    for _ in range(3):                      # steps 3, 4, 5 under torch
        lrs.append(topt.param_groups[0]["lr"])
        topt.step()
        tsch.step()
    ours = [scheduled_lr("steplr", 1e-3, s, step_size=2, gamma=0.5) for s in (3, 4, 5)]
    assert lrs == pytest.approx(ours, rel=1e-12) and ours[0] == 5e-4 and ours[2] == 2.5e-4
    a.train_step(batch)
    assert a.last_lr == 5e-4
    b = FastSpeech2Trainer(make(), lr=1e-3, grad_norm=1.0, scheduler="steplr", scheduler_params={"step_size": 1000, "gamma": 0.9})
    b.load_checkpoint(path)                 # the checkpoint's own schedule wins
    assert b.scheduler_params["step_size"] == 2 and b.scheduler_params["gamma"] == 0.5
    b.train_step(batch)
    assert b.last_lr == 5e-4 and maxdiff(a.flat_p, b.flat_p) <= 1e-7


def test_resume_continues_the_dropout_stream(cuda, lib, tmp_path):
    """Dropout masks are keyed on (optimiser step, micro-batch, rank), not on a call counter: a resumed run draws the masks of step
    N + 1 (not those of step 1 again), eval_step in between does not shift them, and so it stays bit-identical to the uninterrupted run."""
    from jatts_amd.models import FastSpeech2
    from jatts_amd.training import FastSpeech2Trainer
    z, zi, keys, cfg = _train_golden()
    t = lambda k: torch.tensor(zi[k])  # noqa: E731
    il, ol = t("text_lengths"), t("feats_lengths")
    batch = dict(xs=t("text"), ilens=il, ys=t("feats"), olens=ol, durations=t("durations"), duration_lens=il, pitch=t("pitch"),
                 pitch_lens=il, energys=t("energy"), energy_lens=il)

    def make():           # the recipe's dropout rates (FS2_SMALL's defaults), NOT the golden's zeros
        m = FastSpeech2(idim=20, **{**FS2_SMALL, "stop_gradient_from_pitch_predictor": True, "use_masking": True})
        m.load_state_dict(golden_state(keys, 0))
        return m.to(cuda)
    a = FastSpeech2Trainer(make(), lr=1e-3, grad_norm=1.0, warmup_steps=0)
    l1 = float(a.train_step(batch)["loss"])
    a.train_step(batch)
    path = str(tmp_path / "c.pkl")
    a.save_checkpoint(path)
    a.eval_step(batch)
    l3 = float(a.train_step(batch)["loss"])
    b = FastSpeech2Trainer(make(), lr=1e-3, grad_norm=1.0, warmup_steps=0)
    b.load_checkpoint(path)
    l3b = float(b.train_step(batch)["loss"])
    assert abs(l3b - l3) <= 1e-6 * abs(l3) and maxdiff(a.flat_p, b.flat_p) <= 1e-7
    c = FastSpeech2Trainer(make(), lr=1e-3, grad_norm=1.0, warmup_steps=0)
    assert abs(float(c.train_step(batch)["loss"]) - l1) <= 1e-6 * abs(l1)      # same step number -> same masks
    c.load_checkpoint(path)
    c.steps = 0                                           # the old behaviour: masks of step 1 on the step-3 weights
    assert abs(float(c.train_step(batch)["loss"]) - l3) > 1e-5 * abs(l3)


def test_out_of_range_token_ids_raise_like_nn_embedding(cuda, lib):
    """Every embed_scale launch bounds-checks: an id outside the table gives a zero row, never a read outside the
    table, and surfaces as IndexError -- at the forward's own host sync for FastSpeech2 (train and eval forward()), one step late
    through the trainer's stream-ordered snapshot for the sync-free Matcha / VITS steps."""
    from jatts_amd import hip
    from jatts_amd.models import FastSpeech2
    from jatts_amd.training import FastSpeech2Trainer
    table = torch.randn(7, 16, device=cuda)
    ids = torch.tensor([1, 7, -3, 6], device=cuda)
    out = hip.embed_scale(ids, table, 2.0)
    assert torch.equal(out[0], table[1] * 2.0) and torch.equal(out[3], table[6] * 2.0) and float(out[1:3].abs().max()) == 0.0
    with pytest.raises(IndexError):
        hip.check_bad_ids(cuda)
    hip.check_bad_ids(cuda)                                  # counter was reset
    hip.embed_scale(ids, table, 2.0)
    resolve = hip.bad_ids_async(cuda)
    with pytest.raises(IndexError):
        resolve(wait=True)
    hip.check_bad_ids(cuda)                                  # the snapshot took the count with it

    z, zi, keys, cfg = _train_golden()
    t = lambda k: torch.tensor(zi[k])  # noqa: E731
    il, ol = t("text_lengths"), t("feats_lengths")
    batch = dict(xs=t("text"), ilens=il, ys=t("feats"), olens=ol, durations=t("durations"), duration_lens=il, pitch=t("pitch"),
                 pitch_lens=il, energys=t("energy"), energy_lens=il)
    m = FastSpeech2(idim=20, **{**FS2_SMALL, **cfg})
    m.load_state_dict(golden_state(keys, 0))
    m = m.to(cuda)
    tr = FastSpeech2Trainer(m, lr=1e-3, grad_norm=1.0, warmup_steps=0)
    tr.train_step(batch)
    bad = dict(batch, xs=batch["xs"].clone())
    bad["xs"][0, 0] = 20                                     # == idim: one past the table
    with pytest.raises(IndexError):
        tr.train_step(bad)
    m.eval()
    with pytest.raises(IndexError):
        m(bad["xs"], il, bad["ys"], ol, bad["durations"], il, bad["pitch"], il, bad["energys"], il)
    tr.train_step(batch)                                     # and the trainer carries on with a good batch
    # graph mode: the replayed embedding kernel counts on the device, the trainer's stream-ordered snapshot raises one step late
    m2 = FastSpeech2(idim=20, **{**FS2_SMALL, **cfg})
    m2.load_state_dict(golden_state(keys, 0))
    tg = FastSpeech2Trainer(m2.to(cuda), lr=1e-3, grad_norm=1.0, warmup_steps=0, capture_graph=True)
    for _ in range(3):
        tg.train_step(batch)                                 # eager, capture + replay, replay
    tg.train_step(bad)                                       # replay with a bad id: zero row, counted
    torch.cuda.synchronize()
    with pytest.raises(IndexError):
        tg.train_step(batch)
        torch.cuda.synchronize()
        tg.train_step(batch)
    torch.cuda.synchronize()
    tg.train_step(batch)


def test_graph_mode_replays_the_same_training(cuda, lib):
    """FastSpeech2Trainer(capture_graph=True): the first step of a batch signature (shapes + length tuples) runs eagerly, the second is
    captured as ONE graph (forward, losses, backward, clip + Adam), later ones replay it with the inputs, the dropout base seed and the
    Adam / lr scalars refreshed on the device.  Against an eager trainer on the same data, dropout ON: same losses and parameters step
    by step (to the rounding of the atomics-based reductions), also when the DATA changes under the same signature."""
    from jatts_amd.models import FastSpeech2
    from jatts_amd.training import FastSpeech2Trainer
    z, zi, keys, cfg = _train_golden()
    t = lambda k: torch.tensor(zi[k])  # noqa: E731
    il, ol = t("text_lengths"), t("feats_lengths")
    batch = dict(xs=t("text"), ilens=il, ys=t("feats"), olens=ol, durations=t("durations"), duration_lens=il, pitch=t("pitch"),
                 pitch_lens=il, energys=t("energy"), energy_lens=il)

    def make():           # the recipe's dropout rates (FS2_SMALL's defaults)
        m = FastSpeech2(idim=20, **{**FS2_SMALL, "stop_gradient_from_pitch_predictor": True, "use_masking": True})
        m.load_state_dict(golden_state(keys, 0))
        return m.to(cuda)
    a = FastSpeech2Trainer(make(), lr=1e-3, grad_norm=1.0, warmup_steps=10)
    b = FastSpeech2Trainer(make(), lr=1e-3, grad_norm=1.0, warmup_steps=10, capture_graph=True)
    g = torch.Generator().manual_seed(0)
    for step in range(6):
        cur = dict(batch)
        if step >= 3:                                        # new data, same signature: the static inputs must be refreshed
            cur["ys"] = batch["ys"] + 0.1 * torch.randn(batch["ys"].shape, generator=g)
            cur["pitch"] = batch["pitch"] + 0.1 * torch.randn(batch["pitch"].shape, generator=g)
        la, lb = a.train_step(cur), b.train_step(cur)
        for k in ("loss", "mel_loss", "duration_loss", "pitch_loss", "energy_loss", "grad_norm"):
            # (two eager runs drift apart the same way: Adam amplifies the rounding noise of near-zero gradients, see below)
            tol = 2e-5 * (1 + 5 * step)
            assert abs(float(la[k]) - float(lb[k])) <= tol * max(1.0, abs(float(la[k]))), (step, k, float(la[k]), float(lb[k]))
        assert a.steps == b.steps == step + 1 and a.last_lr == b.last_lr
        assert maxdiff(a.flat_g, b.flat_g) <= 2e-5 * (1 + 5 * step), (step, maxdiff(a.flat_g, b.flat_g))   # same gradients (to the atomics' rounding)
        # parameters: Adam turns a gradient that is pure rounding noise (the depthwise-conv bias in front of BatchNorm: exactly zero in
        # exact arithmetic) into +-lr steps, in eager mode from run to run as well -- compare where the gradient carries signal
        o = 0
        for p_ in a.params:
            k = p_.numel()
            if float(a.flat_g[o:o + k].abs().max()) > 1e-4:
                assert maxdiff(a.flat_p[o:o + k], b.flat_p[o:o + k]) <= 5e-6 * (1 + 5 * step), (step, o)
            o += k
    (st,) = b._graphs.values()
    assert st["graph"] is not None                           # steps 2.. were replays of one captured graph
    # running BatchNorm statistics follow too (buffers are updated inside the graph)
    for (n1, b1), (_, b2) in zip(a.model.named_buffers(), b.model.named_buffers()):
        if b1.dtype.is_floating_point:
            assert maxdiff(b1, b2) <= 2e-4 * max(1.0, float(b1.abs().max())), n1
    # a second signature (one utterance fewer) with max_graphs = 1: capturing it evicts the first graph, which is re-captured on demand
    b.max_graphs = 1
    small = {k: (v[:-1] if torch.is_tensor(v) else v) for k, v in batch.items()}
    for _ in range(3):
        out = b.train_step(small)
    assert sum(v.get("graph") is not None for v in b._graphs.values()) == 1 and math.isfinite(float(out["loss"]))
    for _ in range(2):
        out = b.train_step(batch)
    assert sum(v.get("graph") is not None for v in b._graphs.values()) == 1 and math.isfinite(float(out["loss"]))


def test_captured_graph_pins_its_cached_inputs_and_stages_scalars_per_step(cuda, lib):
    """(1) a captured step reads its length uploads / ragged geometry / positional tables at baked-in addresses, while the caches
    that own those tensors are bounded and evict -- the graph's record (st["keep"]) must pin them: flood the caches until every entry of the
    capture is gone, free the allocator's cache, allocate over the freed blocks, replay; (2) the per-step scalars (dropout seed, Adam
    step size) go through a ring of pinned buffers guarded by events: queue several replays WITHOUT synchronising in between and still get
    the eager trainer's training, bit for bit (every reduction of the step is fixed-order)."""
    import gc
    from jatts_amd import hip
    from jatts_amd.models import FastSpeech2
    from jatts_amd.models import fastspeech2_train as ft
    from jatts_amd.training import FastSpeech2Trainer
    z, zi, keys, cfg = _train_golden()
    t = lambda k: torch.tensor(zi[k])  # noqa: E731
    il, ol = t("text_lengths"), t("feats_lengths")
    batch = dict(xs=t("text"), ilens=il, ys=t("feats"), olens=ol, durations=t("durations"), duration_lens=il, pitch=t("pitch"),
                 pitch_lens=il, energys=t("energy"), energy_lens=il)

    def make():
        m = FastSpeech2(idim=20, **{**FS2_SMALL, "stop_gradient_from_pitch_predictor": True, "use_masking": True})
        m.load_state_dict(golden_state(keys, 0))
        return m.to(cuda)
    a = FastSpeech2Trainer(make(), lr=1e-3, grad_norm=1.0, warmup_steps=10)
    b = FastSpeech2Trainer(make(), lr=1e-3, grad_norm=1.0, warmup_steps=10, capture_graph=True)
    for _ in range(2):                       # eager first sight, capture
        a.train_step(batch), b.train_step(batch)
    (st,) = b._graphs.values()
    assert st["graph"] is not None and len(st["keep"]) > 0
    # evict everything the capture was handed
    for i in range(700):
        hip.RaggedBatch([3 + i, 5], cuda)
    hip._H2D_CACHE.clear()
    hip._H2D_BYTES[0] = 0
    hip._GEOM_CACHE.clear()
    ft._POS_TABLES.clear()
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    junk = [torch.full((1 << 16,), float("nan"), device=cuda) for _ in range(64)]     # lands on whatever was freed
    # several replays queued back to back (no host synchronisation in between), then compare with the eager trainer
    la = [a.train_step(batch) for _ in range(6)]
    lb = [b.train_step(batch) for _ in range(6)]
    torch.cuda.synchronize()
    del junk
    for s, (x, y) in enumerate(zip(la, lb)):
        for k in ("loss", "mel_loss", "duration_loss", "pitch_loss", "energy_loss", "grad_norm"):
            assert float(x[k]) == float(y[k]), (s, k, float(x[k]), float(y[k]))
    assert torch.equal(a.flat_p, b.flat_p)


def test_graph_replay_guard_trips_on_an_absurd_gradient_norm(cuda, lib):
    """a capture that replays garbage must not train on silently.  Every replayed step's gradient norm travels back through
    the pinned ring (no host synchronisation per step); a finished slot beyond GRAD_NORM_SANITY raises, the signature falls back to eager.
    Provoked here by lowering the threshold below any real norm."""
    from jatts_amd.models import FastSpeech2
    from jatts_amd.training import FastSpeech2Trainer
    z, zi, keys, cfg = _train_golden()
    t = lambda k: torch.tensor(zi[k])  # noqa: E731
    il, ol = t("text_lengths"), t("feats_lengths")
    batch = dict(xs=t("text"), ilens=il, ys=t("feats"), olens=ol, durations=t("durations"), duration_lens=il, pitch=t("pitch"),
                 pitch_lens=il, energys=t("energy"), energy_lens=il)
    m = FastSpeech2(idim=20, **{**FS2_SMALL, "stop_gradient_from_pitch_predictor": True, "use_masking": True})
    m.load_state_dict(golden_state(keys, 0))
    tr = FastSpeech2Trainer(m.to(cuda), lr=1e-3, grad_norm=1.0, warmup_steps=10, capture_graph=True)
    for _ in range(3):                      # eager, capture, one replay: fine at the real threshold
        out = tr.train_step(batch)
    torch.cuda.synchronize()
    assert math.isfinite(float(out["loss"]))
    tr.GRAD_NORM_SANITY = 1e-30
    with pytest.raises(FloatingPointError):
        for _ in range(4):
            tr.train_step(batch)
            torch.cuda.synchronize()
    (st,) = tr._graphs.values()
    assert st.get("eager_only")
    tr.GRAD_NORM_SANITY = 1e12
    assert math.isfinite(float(tr.train_step(batch)["loss"]))      # the signature keeps training, eagerly
