"""GPU parity of the whole path: VQVAE2 forward / backward vs the CPU oracle, and one
optimisation step of every trainer vs the golden vectors produced by the REFERENCE's
trainer classes (tests/golden/make_golden.py).

North-star bar: losses and converted features within 1e-3 relative; VQ indices
bit-exact at the kernel boundary (tests/test_gpu_ops.py).  End to end the conv stacks
sum in a different order than MKL, so an index can only flip where two codes are
equidistant to ~1e-6: the step tests require >= 99.9 % identical indices and check the
decoded features, which is the stronger statement."""
import numpy as np
import pytest
import torch

from tests.helpers import REPO, STEP_CASES, compare_losses, fill_models, make_batch, run_golden_case, state_summary
from crank_amd.utils import load_yaml

pytestmark = pytest.mark.gpu

# scenarios whose codebook is still the reference's tiny uniform init (no EMA): every code is nearly equidistant from
# every frame, so code choices after an update flip on 1e-7 differences and the decoded features with them
CHAOTIC_POST = {"vqvae_noema"}


def _hip_factories():
    from crank_amd.bin.train import get_model
    from crank_amd.net.trainer.utils import get_criterion, get_optimizer, get_scheduler

    return (lambda conf, n, scaler=None: get_model(conf, n, "cuda", scaler=scaler), get_optimizer, lambda conf: get_criterion(conf, "cuda"),
            get_scheduler)


def _relmax(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-12)


@pytest.mark.parametrize("tag", list(STEP_CASES))
def test_step_matches_reference_goldens_bf16x3(tag):
    """Every loss of every step and the post-step parameters against the goldens of the reference's own trainers (1e-3);
    the post-step decoded features and code indices too, except where the scenario's code choices are chaotic."""
    from crank_amd import ops

    ops.set_precision("bf16x3")
    try:
        losses, models, trainer, fx, post = run_golden_case(tag, *_hip_factories(), device="cuda")
        torch.cuda.synchronize()
        bad = compare_losses(losses, fx, rtol=1e-3, atol=1e-5)
        print(tag, {k: round(v, 6) for k, v in losses[0].items() if v})
        assert not bad, bad
        dec = post["decoded"].cpu().numpy()
        err = _relmax(dec, fx["post_decoded"])
        same = [(post["qidx"][i].cpu().numpy() == fx[f"post_qidx{i}"]).mean() for i in range(2)]
        print(tag, "decoded rel err", err, "qidx identical fractions", same)
        if tag not in CHAOTIC_POST:
            assert err < 1e-3
            assert min(same) >= 0.999
        worst = ("", 0.0)
        for k, v in state_summary(models).items():
            ref = fx[k]
            e = np.abs(v - ref).max() / (np.abs(ref).max() + 1e-6)
            if e > worst[1]:
                worst = (k, e)
        print(tag, "worst post-step parameter summary", worst)
        assert worst[1] < 5e-3, worst
    finally:
        ops.set_precision("bf16")


@pytest.mark.parametrize("tag", list(STEP_CASES))
def test_step_bf16x3_forward_only_mode(tag):
    """"bf16x3f": forward passes in bf16x3, backward passes in plain bf16 (the planes the precise forward saved, read by
    the plain data- and weight-gradient kernels).  On given parameters every loss VALUE is the bf16x3 one (1e-3 of the
    reference, step 0); after optimizer updates the bf16 gradients show (Adam's first steps move every weight by
    lr * sign(gradient)), so later steps are only held to the plain-bf16 bound."""
    from crank_amd import ops

    ops.set_precision("bf16x3f")
    try:
        losses, models, trainer, fx, post = run_golden_case(tag, *_hip_factories(), device="cuda")
        torch.cuda.synchronize()
        bad = compare_losses(losses[:1], fx, rtol=1e-3, atol=1e-5)
        assert not bad, bad
        later = compare_losses(losses, fx, rtol=1e-3, atol=1e-5)
        print(tag, "bf16x3f: losses of later steps outside 1e-3:", later)
        bad = compare_losses(losses, fx, rtol=1e-1, atol=1e-3)
        assert not bad, bad
    finally:
        ops.set_precision("bf16")


def test_step_bf16_fast_mode_is_close():
    """The throughput mode (plain bf16 operands, fp32 accumulate) on the same scenario:
    losses within 3 % of the fp32 reference (bf16 has 8 mantissa bits)."""
    from crank_amd import ops

    ops.set_precision("bf16")
    losses, models, trainer, fx, post = run_golden_case("vqvae", *_hip_factories(), device="cuda")
    # step 0 is a pure forward comparison; step 1 follows one Adam update, whose first
    # step moves every weight by lr*sign(grad): bf16 noise on tiny gradients is amplified
    bad = compare_losses(losses[:1], fx, rtol=3e-2, atol=1e-3)
    print({k: round(v, 5) for k, v in losses[0].items() if v})
    assert not bad, bad
    bad = compare_losses(losses, fx, rtol=1e-1, atol=1e-3)
    assert not bad, bad
    # after two updates on this scenario (the reference's randn / zero EMA init blows the
    # codebook up) a bf16-sized perturbation flips some code choices, so
    # decoded features are compared only through the code agreement
    same = [(post["qidx"][i].cpu().numpy() == fx[f"post_qidx{i}"]).mean() for i in range(2)]
    print("bf16 qidx agreement after 2 steps (informational: codebook collapse makes it chaotic)", same)
    assert torch.isfinite(post["decoded"]).all()


def _oracle_factories():
    from oracle import modules as om

    return (lambda conf, n, scaler=None: om.get_model(conf, n, scaler), om.get_optimizer, om.get_criterion, lambda conf, opt: None)


@pytest.mark.parametrize("tag", list(STEP_CASES))
def test_step_bf16_matches_bf16_emulating_oracle(tag):
    """The pin of the BENCHMARKED arithmetic: the throughput mode ("bf16", what bench.py times and train.py runs
    by default) against the oracle trainers running the same scenario with every conv operand rounded to bf16
    exactly where the kernels round it (oracle/pwg.py bf16_emulation; VQ, losses and Adam fp32 in both).
    Rounding is discontinuous, so even two CPU evaluations of that arithmetic differ (oracle/pwg.py): the
    emulation is run with float64 accumulation (the exact value), with fp32 accumulation and with permuted fp32
    accumulation, and every loss must lie within 3x the larger CPU deviation (+ 2e-3 relative at step 0, + 5e-2 after
    an optimizer update) of the float64-accumulated value.  Against the fp32 goldens the same losses sit at 1e-2 ... 1e-1 (bf16 arithmetic)."""
    from crank_amd import ops
    from oracle import pwg as opwg

    ops.set_precision("bf16")
    losses, models, trainer, fx, post = run_golden_case(tag, *_hip_factories(), device="cuda")
    torch.cuda.synchronize()
    runs = {}
    for acc in ("fp64", "fp32", "fp32-permuted"):
        with opwg.bf16_emulation(accumulate=acc):
            runs[acc] = run_golden_case(tag, *_oracle_factories(), device="cpu")
    ref = runs["fp64"][0]
    bad, report = [], []
    for s, got in enumerate(losses):
        for k, r in ref[s].items():
            r = float(r)
            if r == 0.0:
                continue
            noise = max(abs(float(runs[a][0][s][k]) - r) for a in ("fp32", "fp32-permuted"))
            err = abs(float(got.get(k, 0.0)) - r)
            report.append((err / abs(r), noise / abs(r), s, k))
            # step 0 is evaluated on identical parameters; later steps follow an Adam update of perturbed gradients and the
            # EMA re-initialisation of the codebooks (quirk Q2), after which single code flips move a commitment loss
            # by percents - rare events two CPU samples cannot bound, hence the wider floor
            if err > 3.0 * noise + (2e-3 if s == 0 else 5e-2) * abs(r) + 1e-6:
                bad.append((s, k, float(got.get(k, 0.0)), r, noise))
    report.sort(reverse=True)
    print(tag, "bf16 vs float64-accumulated emulation: largest relative loss deviations (kernel, cpu fp32, step, key)",
          [(f"{a:.1e}", f"{b:.1e}", s, k) for a, b, s, k in report[:4]])
    assert not bad, bad
    if tag not in CHAOTIC_POST:
        dref = runs["fp64"][4]["decoded"].detach().numpy()
        noise = max(_relmax(runs[a][4]["decoded"].detach().numpy(), dref) for a in ("fp32", "fp32-permuted"))
        err = _relmax(post["decoded"].cpu().numpy(), dref)
        print(tag, f"post-step decoded: kernel {err:.2e} / cpu fp32 {noise:.2e} of scale vs the float64-accumulated emulation")
        assert err < 3.0 * noise + 2e-3


@pytest.mark.parametrize("tag", ["vqvae", "lsgan", "cyclegan"])
def test_first_step_gradients_bf16_match_bf16_emulating_oracle(tag):
    """What the loss pin above cannot see after an update (Adam's first step is lr * sign(gradient): a wrong SCALE of a small
    term changes no loss of step 1 by more than the 5e-2 floor): the parameter gradients themselves, as every optimizer of
    the scenario receives them at its first step, in the benchmarked arithmetic (plain bf16) against the oracle trainers under
    bf16 emulation with float64 accumulation - per model, relative L2 over the whole gradient vector within 3x the spread of two
    fp32-accumulated CPU evaluations of the same arithmetic + 2e-3, and per tensor (against max(its own norm, 1 % of the
    model's)) within 3x + 1e-2.  A term whose gradient enters with a wrong factor moves whole tensors by that factor."""
    from crank_amd import ops
    from oracle import pwg as opwg
    from tests.helpers import run_golden_first_step_grads

    ops.set_precision("bf16")
    f = _hip_factories()
    got = run_golden_first_step_grads(tag, f[0], f[1], f[2], device="cuda")
    torch.cuda.synchronize()
    o = _oracle_factories()
    runs = {}
    for acc in ("fp64", "fp32", "fp32-permuted"):
        with opwg.bf16_emulation(accumulate=acc):
            runs[acc] = run_golden_first_step_grads(tag, o[0], o[1], o[2], device="cpu")
    assert set(got) == set(runs["fp64"]), (sorted(got), sorted(runs["fp64"]))
    bad, report = [], []
    for name, ref in runs["fp64"].items():
        keys = [k for k, v in ref.items() if v is not None and k in got[name]]
        assert keys, name

        def vec(d):
            return np.concatenate([np.asarray(d[k], dtype=np.float64).reshape(-1) for k in keys])

        r = vec(ref)
        rn = np.linalg.norm(r)
        noise = max(np.linalg.norm(vec(runs[a][name]) - r) for a in ("fp32", "fp32-permuted")) / rn
        err = np.linalg.norm(vec(got[name]) - r) / rn
        report.append((name, f"{err:.2e}", f"{noise:.2e}"))
        if not err <= 3.0 * noise + 2e-3:
            bad.append((name, err, noise))
        for k in keys:
            rk = np.asarray(ref[k], dtype=np.float64).reshape(-1)
            sc = max(np.linalg.norm(rk), 1e-2 * rn * np.sqrt(rk.size / r.size))
            nk = max(np.linalg.norm(np.asarray(runs[a][name][k], dtype=np.float64).reshape(-1) - rk) for a in ("fp32", "fp32-permuted")) / sc
            ek = np.linalg.norm(np.asarray(got[name][k], dtype=np.float64).reshape(-1) - rk) / sc
            if not ek <= 3.0 * nk + 1e-2:
                bad.append((name, k, ek, nk))
    print(tag, "first-step gradients, relative L2 (model, kernel, cpu fp32 spread):", report)
    assert not bad, bad[:8]


@pytest.mark.parametrize("mode", ["bf16x3", "bf16x3f"])
def test_vqvae2_forward_backward_vs_oracle(mode):
    """Forward values (decoded, encoded, indices) against the fp32 oracle in both parity modes; the parameter gradients in
    the mode whose backward is split-operand too (bf16x3f's backward is plain bf16: pinned by
    test_first_step_gradients_bf16_match_bf16_emulating_oracle)."""
    from crank_amd import ops
    from crank_amd.net.module.vqvae2 import VQVAE2
    from oracle.modules import OracleVQVAE2

    ops.set_precision(mode)
    try:
        conf = load_yaml(None)
        B, T, S = 2, 140, 3
        orac = OracleVQVAE2(conf, spkr_size=S).train()
        prod = VQVAE2(conf, spkr_size=S).train()
        fill_models({"G": orac})
        fill_models({"G": prod})
        batch = make_batch(B, T, S, seed=5)
        x = batch["in_feats"]
        dec_h = torch.cat([batch["lcf0"], batch["uv"]], -1)
        h = batch["org_h"].clone()
        h[:, :] = h[:, 0:1]
        oo = orac(x, None, dec_h, spkrvec=h, use_ema=False)
        po = prod(x.cuda(), None, dec_h.cuda(), spkrvec=h.cuda(), use_ema=False)
        for k in ["decoded"]:
            assert _relmax(po[k].detach().cpu().numpy(), oo[k].detach().numpy()) < 2e-4
        for n in range(2):
            assert _relmax(po["encoded"][n].detach().cpu().numpy(), oo["encoded"][n].detach().numpy()) < 2e-4
            assert (po["qidx"][n].cpu() == oo["qidx"][n]).float().mean() > 0.999
        w = torch.from_numpy(np.random.RandomState(0).standard_normal((B, T, 80)).astype(np.float32))

        def objective(o, wt):
            return (o["decoded"] * wt).sum() + sum(((o["encoded"][n] - o["emb_idx"][n].detach()) ** 2).mean() for n in range(2))

        objective(oo, w).backward()
        prod.zero_grad()
        objective(po, w.cuda()).backward()
        torch.cuda.synchronize()
        worst = ("", 0.0)
        for k, p in orac.named_parameters():
            if p.grad is None:
                continue
            e = _relmax(prod.grad_view(k).cpu().numpy(), p.grad.numpy())
            if e > worst[1]:
                worst = (k, e)
        print(mode, "worst G parameter-gradient error", worst)
        assert worst[1] < (1e-3 if mode == "bf16x3" else 1e-1), worst
    finally:
        ops.set_precision("bf16")


def _frame_mcd(a, b):
    """crank/bin/evaluate_mcd.py:76-77 applied frame-aligned: mean_t 10 / ln 10 * sqrt(2 sum_d (a - b)^2), dB."""
    d = (torch.as_tensor(a).double() - torch.as_tensor(b).double()) ** 2
    return float((10.0 / np.log(10.0) * torch.sqrt(2.0 * d.sum(-1))).mean())


@pytest.mark.parametrize("mode", ["bf16x3", "bf16x3f", "bf16"])
def test_conversion_matches_reference_golden(mode):
    """tests/golden/convert_vqvae.npz: the REFERENCE's VQVAE2 (imported in the authoring container) converting 4 x 200 frames
    on given parameters.  Both parity modes: decoded features within 1e-3, frame-aligned MCD < 1e-2 dB, >= 99.9 % identical
    code indices (north_star).  The throughput mode is held to what 8 mantissa bits allow and its figures are printed -
    the same numbers bench.py reports as `parity_gates`."""
    from crank_amd import ops
    from crank_amd.net.module.vqvae2 import VQVAE2
    from tests.helpers import golden

    fx = golden("convert_vqvae.npz")
    B, T, S, seed = [int(v) for v in fx["meta_B_T_nspk_seed"]]
    ops.set_precision(mode)
    try:
        conf = load_yaml(None)
        prod = VQVAE2(conf, spkr_size=S).eval()
        fill_models({"G": prod})
        batch = make_batch(B, T, S, in_dim=conf["input_size"], seed=seed, device="cuda")
        dec_h = torch.cat([batch["cv_lcf0"], batch["uv"]], -1)
        h = batch["cv_h"].clone()
        h[:, :] = h[:, 0:1]
        with torch.no_grad():
            po = prod(batch["in_feats"], None, dec_h, spkrvec=h, use_ema=False)
        dec = po["decoded"].cpu().numpy()
        err, mcd = _relmax(dec, fx["decoded"]), _frame_mcd(dec, fx["decoded"])
        same = [float((po["qidx"][i].cpu().numpy() == fx[f"qidx{i}"]).mean()) for i in range(2)]
        print(f"[{mode}] decoded rel err {err:.2e}, frame-aligned MCD {mcd:.2e} dB, identical indices {same}")
        if mode == "bf16":  # (a flipped index replaces a frame's code vector: the max-norm error is that of the worst frame)
            assert err < 0.6 and mcd < 3.0 and min(same) > 0.9
        else:
            assert err < 1e-3 and mcd < 1e-2 and min(same) >= 0.999
    finally:
        ops.set_precision("bf16")


def test_benchmarked_arithmetic_trains_like_fp32():
    """The arithmetic bench.py times ("bf16") and the cheaper parity mode ("bf16x3f") follow the fp32 reference arithmetic
    (the CPU oracle under the same trainer class, fp32) over a TRAINING RUN, not only over its first step: 120 steps of the
    vqvae trainer at the golden shape (B = 2, T = 96, 2 speakers), a fresh synthetic batch per step from a fixed seed,
    identical initial parameters.  Checked: the smoothed curves (mean over windows of 20 steps) of the generator loss, its
    reconstruction terms and the classifier / speaker-adversarial losses stay within 5 % (bf16x3f) / 8 % (bf16) of the
    oracle's (two fp32 oracle runs whose initial parameters differ by 3e-3 relative sit 0.6 % apart), the commitment terms
    within 30 %, the loss goes DOWN by about as much, and the codebook usage over the last 40 steps (normalised histogram of
    the chosen codes per quantizer; total-variation distance, printed) agrees where the oracle's codebook is alive."""
    from crank_amd import ops
    from crank_amd.net.trainer import TrainerWrapper

    steps, win, B, T, S = 120, 20, 2, 96, 2
    keys = ("G", "G_l1", "G_mse", "G_stft", "C", "SPKRADV", "G_commit0", "G_commit1")
    tight = 6  # the first six are held to the band; the commitment terms follow single code choices (quirk Q2 collapses the
    #            codebook of this scenario to one or two live codes) and already move by 5 - 8 % between two fp32 runs whose
    #            initial parameters differ by 3e-3: held to 0.3

    def run(factories, device, mode=None):
        if mode:
            ops.set_precision(mode)
        try:
            torch.manual_seed(1234)
            np.random.seed(1234)
            conf = load_yaml(None, trainer_type="vqvae", batch_size=B, batch_len=T)
            build_models, build_optim, build_criterion, build_sched = factories
            models = build_models(conf, S, None)
            fill_models(models)
            for m in models.values():
                m.train()
            optimizer = build_optim(conf, models)
            trainer = TrainerWrapper("vqvae", model=models, optimizer=optimizer, criterion=build_criterion(conf),
                                     dataloader={"spkrs": {f"spk{i}": i for i in range(S)}}, writer=None, expdir="/tmp/crank_amd_traj",
                                     conf=conf, feat_conf=conf["feature"], scheduler=build_sched(conf, optimizer), scaler=None, resume=0,
                                     device=device, n_jobs=1)
            curve, hist = [], [np.zeros(512), np.zeros(512)]
            for s in range(steps):
                batch = make_batch(B, T, S, seed=1000 + s, device=device)
                trainer.steps = 1
                trainer.check_custom_start()
                v = trainer.train(batch, phase="train")
                curve.append([float(v[k]) for k in keys])
                if s >= steps - 40:
                    with torch.no_grad():
                        enc_h = trainer._get_enc_h(batch)
                        dec_h, spkrvec = trainer._get_dec_h(batch)
                        o = models["G"].forward(batch["in_feats"], enc_h, dec_h, spkrvec=spkrvec, use_ema=False)
                    for i in range(2):
                        hist[i] += np.bincount(o["qidx"][i].reshape(-1).cpu().numpy(), minlength=512)[:512]
            return np.asarray(curve), [h / h.sum() for h in hist]
        finally:
            if mode:
                ops.set_precision("bf16")

    ref_curve, ref_hist = run(_oracle_factories(), "cpu")
    smooth = lambda c: c.reshape(steps // win, win, -1).mean(1)  # noqa: E731
    rs = smooth(ref_curve)
    print("fp32 oracle, smoothed", dict(zip(keys, np.round(rs[[0, -1]].T, 4).tolist())))
    assert rs[-1, 0] < rs[0, 0], "the oracle run itself does not train"
    for mode, band in (("bf16x3f", 0.05), ("bf16", 0.08)):
        curve, hist = run(_hip_factories(), "cuda", mode)
        assert np.isfinite(curve).all()
        gs = smooth(curve)
        dev = np.abs(gs - rs) / np.maximum(np.abs(rs), 1e-3)
        tv = [0.5 * float(np.abs(hist[i] - ref_hist[i]).sum()) for i in range(2)]
        used = [(int((hist[i] > 0).sum()), int((ref_hist[i] > 0).sum())) for i in range(2)]
        print(f"[{mode}] smoothed-curve relative deviation from the fp32 oracle, worst window per term:",
              dict(zip(keys, np.round(dev.max(0), 4).tolist())), "| first / last window of G:", np.round(gs[[0, -1], 0], 4).tolist(),
              "| codebook usage TV distance", np.round(tv, 3).tolist(), "codes used (here, oracle)", used)
        assert dev[:, :tight].max() < band, (mode, dev.max(0))
        assert dev[:, tight:].max() < 0.3, (mode, dev.max(0))
        # the run trains: the generator loss falls by at least 80 % of what the oracle's falls
        assert (gs[0, 0] - gs[-1, 0]) > 0.8 * (rs[0, 0] - rs[-1, 0]), (mode, gs[:, 0], rs[:, 0])
        # usage histograms are comparable only where the oracle's codebook is alive (with one live code the distance is 0 or 1)
        for i in range(2):
            if used[i][1] >= 8:
                assert tv[i] < 0.5, (mode, i, tv)


def test_full_size_step_runs_and_is_finite():
    """BASELINE configs[1] shape (B=64, T=500, 14 speakers) in the throughput mode:
    one full step, finite losses, codebook usage statistics consistent (sum of EMA
    cluster sizes equals its pre-smoothing total)."""
    from crank_amd import ops
    from crank_amd.bin.train import build_trainer

    ops.set_precision("bf16")
    torch.manual_seed(1234)
    conf = load_yaml(None, batch_size=64, batch_len=500)
    trainer = build_trainer(conf, 14, "/tmp/crank_amd_full")
    batch = make_batch(64, 500, 14, device="cuda")
    vals = trainer.train(batch)
    vals = trainer.train(batch)
    torch.cuda.synchronize()
    print({k: round(v, 5) for k, v in vals.items() if v})
    assert all(np.isfinite(v) for v in vals.values())
    for q in trainer.model["G"].quantizers:
        assert torch.isfinite(q.weight).all() and torch.isfinite(q.ema_w).all()


def test_full_size_lamb_step_moves_every_tensor_by_its_clamped_norm():
    """BASELINE configs[1] shape with optim.*.type = lamb (crank/net/trainer/utils.py:46-47): a size-independent property of
    the update w -= lr * r * u with r = clamp(||w||, 0, 10) / ||u|| per parameter tensor - after ONE step every tensor that
    has a gradient has moved by exactly lr * min(||w||, 10) in norm, whatever its gradient was (an all-zero tensor by
    lr * ||u||); a tensor no gradient reaches is where it was (the EMA codebooks are the quantizer's to move)."""
    from crank_amd import ops
    from crank_amd.bin.train import build_trainer

    ops.set_precision("bf16")
    torch.manual_seed(1234)
    conf = load_yaml(None, batch_size=64, batch_len=500)
    for m in conf["optim"]:
        conf["optim"][m]["type"] = "lamb"
    trainer = build_trainer(conf, 14, "/tmp/crank_amd_full_lamb")
    before = {k: m.flat.detach().clone() for k, m in trainer.model.items()}
    vals = trainer.train(make_batch(64, 500, 14, device="cuda"))
    torch.cuda.synchronize()
    assert all(np.isfinite(v) for v in vals.values())
    checked = 0
    for name, m in trainer.model.items():
        lr = conf["optim"][name]["lr"]
        skip = {off for off, _ in getattr(m, "ema_codebook_ranges", lambda: [])()}
        for key, off, shp in m._entries:
            cnt = int(np.prod(shp))
            if off in skip or cnt == 0:
                continue
            w0 = before[name][off: off + cnt].double()
            moved = (m.flat.detach()[off: off + cnt].double() - w0).norm().item()
            un = trainer.optimizer[name].upd[off: off + cnt].double().norm().item()  # ||u|| of the step (kept by FlatLamb)
            if un == 0.0:  # no gradient reached it (the last block's residual output conv feeds nothing): where it was
                assert moved == 0.0, (name, key, moved)
                continue
            # ratio 1 for an all-zero tensor (a bias at its initialisation): it moves by lr * ||u||
            want = lr * (min(w0.norm().item(), 10.0) if w0.norm().item() > 0.0 else un)
            assert abs(moved - want) <= 5e-3 * want + 1e-9, (name, key, moved, want)
            checked += 1
    print("tensors checked:", checked)
    assert checked > 100


def test_full_size_radam_first_step_is_a_plain_gradient_step():
    """BASELINE configs[1] shape with optim.*.type = radam (crank/net/trainer/utils.py:44-45): at step 1 the approximated SMA
    is shorter than 5, the update is the momentum-only one, and with m = (1 - beta1) g and the bias correction 1 - beta1 it
    is w -= lr * g exactly - checked on every parameter of G, SPKRADV and C against the gradients the step consumed (kept:
    clear_grads off), codebooks excluded (no gradient, moved by the EMA)."""
    from crank_amd import ops
    from crank_amd.bin.train import build_trainer

    ops.set_precision("bf16")
    torch.manual_seed(1234)
    conf = load_yaml(None, batch_size=64, batch_len=500)
    for m in conf["optim"]:
        conf["optim"][m]["type"] = "radam"
    trainer = build_trainer(conf, 14, "/tmp/crank_amd_full_radam")
    for o in trainer.optimizer.values():
        o.clear_grads = False
    before = {k: m.flat.detach().clone() for k, m in trainer.model.items()}
    vals = trainer.train(make_batch(64, 500, 14, device="cuda"))
    torch.cuda.synchronize()
    assert all(np.isfinite(v) for v in vals.values())
    for name, m in trainer.model.items():
        lr = conf["optim"][name]["lr"]
        keep = torch.ones(m.flat.numel(), dtype=torch.bool, device="cuda")
        for off, n in getattr(m, "ema_codebook_ranges", lambda: [])():
            keep[off: off + n] = False
        g = m.grad_flat.double()[keep]
        step = (before[name].double() - m.flat.detach().double())[keep]
        assert float(g.abs().max()) > 0.0
        # (the difference of two fp32 weights carries an ulp of the weight: 6e-8 * |w| against lr * |g|)
        err = (step - lr * g).abs().max().item()
        floor = 2.4e-7 * before[name].abs().max().item()
        assert err <= 1e-5 * lr * g.abs().max().item() + floor, (name, err, floor)
        assert trainer.optimizer[name].step_dev.item() == 1


@pytest.mark.parametrize("optim_type", ["adam", "radam", "lamb"])
def test_graph_replayed_steps_equal_eager_steps(optim_type):
    """conf["hip_graph"] (BaseTrainer.train_graphed / GraphedStep): three eager steps, a capture, replays - against the
    same steps run eagerly on an identically seeded trainer, to rounding level (the bound from the time the STFT-loss
    gradient used float atomics; test_replayed_vqvae_steps_equal_eager_steps_bit_for_bit holds the exact statement).
    With every optimizer of the factory (crank/net/trainer/utils.py:40-50): RAdam leaves its momentum-only regime behind
    step 5 - inside the replays, decided on the device from the step count the graph advances."""
    from crank_amd import ops
    from crank_amd.bin.train import build_trainer

    ops.set_precision("bf16")
    conf = load_yaml(None, batch_size=4, batch_len=160)
    for m in conf["optim"]:
        conf["optim"][m]["type"] = optim_type
    n_steps = 6 if optim_type == "adam" else 9
    runs = []
    for graphed in (False, True):
        torch.manual_seed(1234)
        trainer = build_trainer(conf, 5, "/tmp/crank_amd_graph")
        fill_models(trainer.model)
        vals = []
        for step in range(n_steps):
            batch = make_batch(4, 160, 5, seed=20 + step, device="cuda")
            v = trainer.train_graphed(batch) if graphed else trainer.train(batch)
            vals.append({k: float(x) for k, x in v.items()})
            trainer.steps += 1
        torch.cuda.synchronize()
        if graphed:
            assert any(slot[1] is not None for slot in trainer._graphs.values()), "no step was captured"
        runs.append((vals, {k: m.flat.detach().cpu().numpy().copy() for k, m in trainer.model.items()},
                     [q.weight.detach().cpu().numpy().copy() for q in trainer.model["G"].quantizers]))
    (ve, pe, ce), (vg, pg, cg) = runs
    for s in range(n_steps):
        for k, r in ve[s].items():
            assert abs(vg[s][k] - r) <= 1e-4 * abs(r) + 1e-6, (s, k, vg[s][k], r)
    for k in pe:
        assert np.abs(pg[k] - pe[k]).max() <= 1e-5, k
    for a, b in zip(cg, ce):
        assert np.abs(a - b).max() <= 1e-4 * np.abs(b).max()


def _run_steps(conf, n_spk, graphed, shapes, seed0=40):
    """`len(shapes)` optimisation steps of a freshly built, deterministically filled trainer; shapes[i] = (B, T) of step i."""
    import random

    from crank_amd.bin.train import build_trainer

    random.seed(1234)
    torch.manual_seed(1234)
    trainer = build_trainer(conf, n_spk, "/tmp/crank_amd_graph2")
    fill_models(trainer.model)
    trainer.steps = 1
    trainer.check_custom_start()
    vals = []
    for step, (B, T) in enumerate(shapes):
        batch = make_batch(B, T, n_spk, seed=seed0 + step, device="cuda")
        v = trainer.train_graphed(batch) if graphed else trainer.train(batch)
        vals.append({k: float(x) for k, x in v.items()})
        trainer.steps += 1
    torch.cuda.synchronize()
    state = {k: m.flat.detach().cpu().numpy().copy() for k, m in trainer.model.items()}
    books = [q.weight.detach().cpu().numpy().copy() for q in trainer.model["G"].quantizers]
    return vals, state, books, trainer


def _assert_same_run(eager, graphed, rtol=1e-4, exact=False):
    (ve, pe, ce, _), (vg, pg, cg, _) = eager, graphed
    if exact:
        assert ve == vg, [(s, k, ve[s][k], vg[s][k]) for s in range(len(ve)) for k in ve[s] if ve[s][k] != vg[s].get(k)][:5]
        for k in pe:
            assert np.array_equal(pg[k], pe[k]), (k, float(np.abs(pg[k] - pe[k]).max()))
        for a, b in zip(cg, ce):
            assert np.array_equal(a, b)
        return
    for s in range(len(ve)):
        assert set(ve[s]) == set(vg[s]), (s, sorted(set(ve[s]) ^ set(vg[s])))
        for k, r in ve[s].items():
            assert abs(vg[s][k] - r) <= rtol * abs(r) + 1e-6, (s, k, vg[s][k], r)
    for k in pe:
        assert np.abs(pg[k] - pe[k]).max() <= 1e-5, k
    for a, b in zip(cg, ce):
        assert np.abs(a - b).max() <= 1e-4 * np.abs(b).max()


@pytest.mark.parametrize("ttype,extra,steps", [
    # BASELINE configs[2]: the default discriminator WITH its dropout 0.25 (crank/bin/train.py:114): the masks come from
    # device-resident seeds (crk_seed_next), so the captured step draws a fresh mask per replay - and, both trainers
    # starting from the same seed state, exactly the masks of the eager run
    ("lsgan", {}, 7),
    # configs[3]: update_D shows D one of two fakes, drawn per step (trainer_cyclegan.py:166): one graph per outcome
    ("cyclegan", {"use_cyclic_training": True, "n_steps_cycle_start": 0}, 12),
    # configs[4]'s trainer with the per-step real / fake draw (trainer_stargan.py:90-93)
    ("stargan", {"use_cyclic_training": True, "n_steps_cycle_start": 0, "switch_update": True}, 12),
])
def test_every_trainer_replays_from_graphs_and_equals_eager(ttype, extra, steps):
    """conf["hip_graph"] for the GAN trainers (three of the five BASELINE configs): the same steps replayed from captured
    graphs and enqueued eagerly, identically seeded - loss values of every step, parameters and codebooks at the end."""
    from crank_amd import ops

    ops.set_precision("bf16")
    conf = load_yaml(None, batch_size=4, batch_len=160, trainer_type=ttype, n_steps_gan_start=0, **extra)
    assert conf["discriminator_dropout"] == 0.25
    shapes = [(4, 160)] * steps
    eager = _run_steps(conf, 5, False, shapes)
    graphed = _run_steps(conf, 5, True, shapes)
    tr = graphed[3]
    assert tr._graphs is not None, "a capture failed and the trainer fell back to eager steps"
    captured = [sig for sig, slot in tr._graphs.items() if slot[1] is not None]
    assert captured, "no step was captured"
    if ttype != "lsgan":
        assert len({sig[1] for sig in tr._graphs}) == 2, "both outcomes of the per-step draw should have occurred"
    _assert_same_run(eager, graphed)
    assert eager[0][-1]["D"] > 0


def test_replayed_vqvae_steps_equal_eager_steps_bit_for_bit():
    """Every kernel of the default step is deterministic now (the STFT-loss gradient was the exception: float atomics until
    the reconstruction losses became one launch without them), so the same steps replayed from a graph and enqueued
    eagerly agree to the bit: loss values of every step, parameters and codebooks after five updates."""
    from crank_amd import ops

    ops.set_precision("bf16")
    conf = load_yaml(None, batch_size=4, batch_len=120, trainer_type="vqvae")
    shapes = [(4, 120)] * 5
    eager = _run_steps(dict(conf), 14, False, shapes)
    graphed = _run_steps(dict(conf, hip_graph=True), 14, True, shapes)
    assert graphed[3]._graphs
    _assert_same_run(eager, graphed, exact=True)


def test_replayed_steps_see_a_codebook_written_between_two_replays():
    """A single process's EMA blend leaves the codebooks' search images current, so a captured step holds no launch that
    rebuilds them before its first search.  A codebook written BETWEEN two replays (load_state_dict, touch()) must still be
    searched as written: GraphedStep.step rebuilds the images when the generator's codebook count moved.  Same steps and the
    same write, eager and replayed: bit for bit."""
    import random

    from crank_amd import ops
    from crank_amd.bin.train import build_trainer

    ops.set_precision("bf16")
    runs = []
    for graphed in (False, True):
        conf = load_yaml(None, batch_size=4, batch_len=120, trainer_type="vqvae", hip_graph=graphed)
        random.seed(1234)
        torch.manual_seed(1234)
        trainer = build_trainer(conf, 14, "/tmp/crank_amd_graph3")
        fill_models(trainer.model)
        trainer.steps = 1
        trainer.check_custom_start()
        G = trainer.model["G"]
        vals = []
        for step in range(7):
            if step == 5:  # (steps 0-2 eager warm-ups, 3 the capture, 4 a replay)
                with torch.no_grad():
                    for q in G.quantizers:
                        q.weight.mul_(-1.0)  # every frame's nearest code changes; the next blend overwrites it from ema_w
                G.touch()
            batch = make_batch(4, 120, 14, seed=70 + step, device="cuda")
            v = trainer.train_graphed(batch) if graphed else trainer.train(batch)
            vals.append({k: float(x) for k, x in v.items()})
            trainer.steps += 1
        torch.cuda.synchronize()
        if graphed:
            assert any(slot[1] is not None for slot in trainer._graphs.values()), "no step was captured"
        runs.append((vals, {k: m.flat.detach().cpu().numpy().copy() for k, m in trainer.model.items()},
                     [q.weight.detach().cpu().numpy().copy() for q in G.quantizers], trainer))
    _assert_same_run(runs[0], runs[1], exact=True)


def test_graphs_of_two_batch_shapes_alternate():
    """A short last batch of an epoch (or a dev batch) between replays of the full-batch graph: every shape has device
    tables of its own (plane offsets are multiples of B*T, partial-sum offsets of the slot counts), a replay must see
    the tables of the shape it was captured with whatever ran in between."""
    from crank_amd import ops

    ops.set_precision("bf16")
    conf = load_yaml(None, batch_size=4, batch_len=160)
    A, Bs = (4, 160), (3, 96)
    shapes = [A, A, A, A, A, Bs, A, Bs, Bs, Bs, Bs, A, Bs, A]
    eager = _run_steps(conf, 5, False, shapes)
    graphed = _run_steps(conf, 5, True, shapes)
    tr = graphed[3]
    assert tr._graphs is not None and sum(slot[1] is not None for slot in tr._graphs.values()) == 2
    _assert_same_run(eager, graphed)


def test_capture_survives_garbage_that_owns_graphs():
    """A dropped trainer whose captured steps still wait for the garbage collector (trainer and GraphedStep reference each
    other) and a collection INSIDE the next capture: ~CUDAGraph synchronizes the device on ROCm, which is illegal while a
    stream captures and is raised inside a destructor - std::terminate, SIGABRT (met in round 4 when the collector
    happened to run inside the capture of the cyclegan step; ``tools/gc_capture_repro.py nofix lsgan vqvae`` reproduces it
    every time, profiles/round4_gc_capture_abort.txt).  GraphedStep collects BEFORE it captures and holds the collector
    off until the capture has ended (basetrainer.hold_collector_for_capture).  In a process of its own: a regression
    kills the process."""
    import subprocess
    import sys

    r = subprocess.run([sys.executable, "tools/gc_capture_repro.py", "fix", "lsgan", "vqvae"], cwd=REPO, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "CAPTURED-WITH-GARBAGE-OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "collected inside the capture: 0" in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("ttype", ["vqvae", "lsgan", "cyclegan", "stargan"])
def test_loss_values_of_a_step_leave_the_device_as_one_arena(ttype):
    """Every loss value a trainer reports is a result scalar of a loss op; the ops take those scalars out of the step's arena
    (ops._ScalarArena), so the values go to the host as ONE copy of the arena with no launch that stacks them: the step's
    LossValues carry the arena positions of their keys, and reading them gives what the tensors hold."""
    from crank_amd import ops
    from crank_amd.bin.train import build_trainer
    from crank_amd.utils import load_yaml

    ops.set_precision("bf16")
    over = dict(batch_size=4, batch_len=160, trainer_type=ttype)
    if ttype != "vqvae":
        over.update(n_steps_gan_start=0, n_steps_cycle_start=0, use_cyclic_training=ttype != "lsgan")
    conf = load_yaml(None, **over)
    torch.manual_seed(3)
    trainer = build_trainer(conf, 5, "/tmp/crank_amd_arena")
    fill_models(trainer.model)
    trainer.steps = 1
    trainer.check_custom_start()
    values = trainer.train(make_batch(4, 160, 5, seed=60, device="cuda"))
    assert values._pending is not None and values._pending[3] is not None, "a loss value lives outside the arena"
    keys, index = values._pending[0], values._pending[3]
    assert len(index) == len(keys) and len(set(index)) > 5  # (keys may share a scalar: a total with one term of weight 1)
    arena = trainer._arena.buf.clone()
    for k, i in zip(keys, index):
        assert values[k] == float(arena[i]), k
