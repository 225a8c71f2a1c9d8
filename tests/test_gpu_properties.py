"""GPU tests that do not need golden vectors: the "MCD vs ref" parity statement of
SURVEY.md section 8(d), the mcep configuration (BASELINE configs[4] shapes), and size-independent
properties at the full benchmark shape (B=64, T=500) where the CPU oracle would take minutes:
determinism, linearity of the backward pass in the output gradient, exact integer EMA
statistics, argmin optimality of the VQ indices, and bit-identity between the two window
shapes of the fused kernels (the bf16x3 parity tests only exercise the 4-wave shape)."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests.helpers import REPO, fill_models, make_batch
from crank_amd.utils import load_yaml

pytestmark = pytest.mark.gpu


def _mcd(a, b):
    """crank/bin/evaluate_mcd.py:76-77 applied frame-aligned (no DTW): mean_t 10/ln10 * sqrt(2 sum_d (a-b)^2)."""
    d = (a.double() - b.double()) ** 2
    return float((10.0 / math.log(10.0) * torch.sqrt(2.0 * d.sum(-1))).mean())


@pytest.mark.parametrize("mode", ["bf16x3", "bf16x3f"])
@pytest.mark.parametrize("feat", ["mlfb80", "mcep34"])
def test_mcd_between_gpu_and_oracle_conversion(feat, mode):
    """Convert the same utterances with identical weights on the GPU path and the CPU oracle
    (eval mode, converted speaker / F0 conditions): frame-aligned MCD must be ~0 dB - in both modes whose forward passes
    are split-operand ("bf16x3f" is the cheaper one: what a conversion run would use)."""
    from crank_amd import ops
    from crank_amd.net.module.vqvae2 import VQVAE2
    from oracle.modules import OracleVQVAE2

    ops.set_precision(mode)
    try:
        over = {} if feat == "mlfb80" else dict(input_feat_type="mcep", output_feat_type="mcep", input_size=34, output_size=34)
        conf = load_yaml(None, **over)
        D = conf["input_size"]
        B, T, S = 3, 200, 12 if feat == "mcep34" else 14
        orac = OracleVQVAE2(conf, spkr_size=S).eval()
        prod = VQVAE2(conf, spkr_size=S).eval()
        fill_models({"G": orac})
        fill_models({"G": prod})
        batch = make_batch(B, T, S, in_dim=D, seed=11)
        x = batch["in_feats"]
        dec_h = torch.cat([batch["cv_lcf0"], batch["uv"]], -1)  # converted F0, BaseTrainer._get_dec_h(use_cvfeats=True)
        h = batch["cv_h"].clone()
        h[:, :] = h[:, 0:1]
        with torch.no_grad():
            oo = orac(x, None, dec_h, spkrvec=h, use_ema=False)
            po = prod(x.cuda(), None, dec_h.cuda(), spkrvec=h.cuda(), use_ema=False)
        mcd = _mcd(po["decoded"].cpu(), oo["decoded"])
        scale = _mcd(oo["decoded"], torch.zeros_like(oo["decoded"]))
        print(f"[{feat}, {mode}] MCD(GPU conversion, oracle conversion) = {mcd:.2e} dB (features themselves: {scale:.1f} dB)")
        assert mcd < 1e-2
        for n in range(2):
            assert (po["qidx"][n].cpu() == oo["qidx"][n]).float().mean() > 0.999
    finally:
        ops.set_precision("bf16")


def test_mcep_configuration_step_runs():
    """BASELINE configs[4] layer shapes (34-dim mcep, 12 speakers, D input 34 + 1 + 32 = 67):
    one stargan step with the GAN and cycle terms enabled, finite losses."""
    from crank_amd import ops
    from crank_amd.bin.train import build_trainer

    ops.set_precision("bf16")
    torch.manual_seed(7)
    conf = load_yaml(None, trainer_type="stargan", batch_size=4, batch_len=300, input_feat_type="mcep",
                     output_feat_type="mcep", input_size=34, output_size=34, n_steps_gan_start=0,
                     use_cyclic_training=True, n_steps_cycle_start=0)
    trainer = build_trainer(conf, 12, "/tmp/crank_amd_mcep")
    batch = make_batch(4, 300, 12, in_dim=34, device="cuda")
    vals = trainer.train(batch)
    torch.cuda.synchronize()
    print({k: round(v, 5) for k, v in vals.items() if v})
    assert all(np.isfinite(v) for v in vals.values())


def _full_G(seed=3):
    from crank_amd.bin.train import get_model

    conf = load_yaml(None, batch_size=64, batch_len=500)
    torch.manual_seed(seed)
    model = get_model(conf, 14, "cuda")
    batch = make_batch(64, 500, 14, device="cuda", seed=seed)
    dec_h = torch.cat([batch["lcf0"], batch["uv"]], -1)
    h = batch["org_h"].clone()
    h[:, :] = h[:, 0:1]
    return model, batch, dec_h, h


def test_full_size_forward_backward_properties():
    """B=64, T=500: (1) two forwards are bit-identical, (2) the backward pass is linear in the
    output gradient (every kernel of the chain is), (3) parameter gradients are deterministic."""
    from crank_amd import ops

    ops.set_precision("bf16")
    model, batch, dec_h, h = _full_G()
    G = model["G"].train()
    x = batch["in_feats"]
    w = torch.randn(64, 500, 80, device="cuda", generator=torch.Generator("cuda").manual_seed(0))

    def run(scale):
        G.zero_grad()
        xi = x.clone().requires_grad_(True)
        o = G(xi, None, dec_h, spkrvec=h, use_ema=False)
        (o["decoded"] * (w * scale)).sum().backward()
        torch.cuda.synchronize()
        return o["decoded"].detach().clone(), xi.grad.clone(), G.grad_flat.clone()

    d1, gx1, gp1 = run(1.0)
    d2, gx2, gp2 = run(1.0)
    assert torch.equal(d1, d2), "forward is not deterministic"
    assert torch.equal(gx1, gx2) and torch.equal(gp1, gp2), "backward is not deterministic"
    _, gx4, gp4 = run(4.0)  # a power of two: fp32 / bf16 scaling is exact, so linearity holds to the bit
    assert torch.equal(gx4, gx1 * 4.0), float((gx4 - 4 * gx1).abs().max())
    rel = float((gp4 - 4 * gp1).abs().max() / gp4.abs().max())
    print("parameter-gradient linearity residual", rel)
    assert rel < 1e-6


@pytest.mark.parametrize("ttype,B", [("lsgan", 64), ("cyclegan", 32), ("stargan", 32), ("stargan_mcep", 32)])
def test_full_size_gan_steps_are_finite_repeatable_and_keep_the_ema_mass(ttype, B):
    """BASELINE configs[2] (lsgan, B = 64) and configs[3] / [4]'s per-GPU shape (cyclegan / stargan, B = 32) at T = 500
    with the default discriminator (dropout 0.25), throughput arithmetic - "stargan_mcep" = configs[4] as the recipe has it:
    34-dim mel-cepstra in and out, 12 speakers, discriminator 67 -> 1 (egs/vaevc/template/conf/mcep_vqvae_22050.yml:17-27) -:
    two GAN-phase steps are finite; an identically
    seeded second trainer reproduces the loss values (dropout masks included: the seeds live on the device); every EMA
    update adds exactly (1 - decay) * frames to a quantizer's cluster mass (Laplace smoothing redistributes, it does not
    create mass: crank/net/module/vqvae2.py:316-328), whatever the number of generator forwards of the trainer."""
    import random

    from crank_amd import ops
    from crank_amd.bin.train import build_trainer

    ops.set_precision("bf16")
    mcep = ttype == "stargan_mcep"
    dim, n_spk = (34, 12) if mcep else (80, 14)
    ttype = "stargan" if mcep else ttype
    over = dict(trainer_type=ttype, batch_size=B, batch_len=500, n_steps_gan_start=0)
    if ttype != "lsgan":
        over.update(use_cyclic_training=True, n_steps_cycle_start=0)
    if mcep:
        over.update(input_feat_type="mcep", output_feat_type="mcep", input_size=34, output_size=34, use_mcep_0th=False,
                    ignore_scaler=["mcep"])
    conf = load_yaml(None, **over)
    calls = {"n": 0}
    real, real_b = ops.vq_ema_apply_multi, ops.vq_ema_blend_multi  # (the blend of a forward: either entry point)

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)

    def counting_b(*a, **k):
        calls["n"] += 1
        return real_b(*a, **k)

    runs = []
    for rep in range(2):
        random.seed(7)
        torch.manual_seed(7)
        trainer = build_trainer(conf, n_spk, "/tmp/crank_amd_full_gan")
        trainer.steps = 1
        trainer.check_custom_start()
        assert trainer.gan_flag
        batch = make_batch(B, 500, n_spk, in_dim=dim, device="cuda", seed=9)
        if mcep:
            assert trainer.model["D"].stack.net.convs[0][1] == 34 + 1 + 32  # [mcep | uv | speaker embedding]: 67 -> 1
        calls["n"] = 0
        ops.vq_ema_apply_multi, ops.vq_ema_blend_multi = counting, counting_b
        try:
            vals = [{k: float(v) for k, v in trainer.train(batch).items()} for _ in range(2)]  # (.items(): the values arrive lazily)
        finally:
            ops.vq_ema_apply_multi, ops.vq_ema_blend_multi = real, real_b
        torch.cuda.synchronize()
        runs.append(vals)
        assert all(np.isfinite(v) for d in vals for v in d.values()), vals
        assert vals[1]["D"] > 0 and vals[1]["G"] > 0
        k = calls["n"]  # EMA updates so far (each covers both quantizers)
        assert k >= 4, k
        for q in trainer.model["G"].quantizers:
            mass = float(q.ema_size.double().sum())
            want = B * 500 * (1.0 - q.decay ** k)
            assert abs(mass - want) <= 1e-4 * want, (mass, want, k)
            assert torch.isfinite(q.weight).all() and torch.isfinite(q.ema_w).all()
    for s in range(2):
        for key, r in runs[0][s].items():
            assert abs(runs[1][s][key] - r) <= 1e-4 * abs(r) + 1e-6, (s, key, runs[1][s][key], r)


def test_full_size_vq_and_ema_properties():
    """N = 32 000 frames, K = 512: indices are optimal (no other code is closer in fp64 beyond
    fp32 resolution), the gathered vectors are the codebook rows, the EMA counts are the
    histogram and the fixed-point sums equal an fp64 scatter-add to 2^-28 per element."""
    from crank_amd import ops

    torch.manual_seed(5)
    N, D, K = 32000, 64, 512
    x = torch.randn(64, 500, D, device="cuda")
    cb = torch.randn(K, D, device="cuda") * 0.7
    e, qx, idx = ops.vq_apply(x, cb)
    torch.cuda.synchronize()
    assert torch.equal(e.reshape(N, D), cb[idx.reshape(-1)])
    xd, cd = x.reshape(N, D).double(), cb.double()
    dist = (cd * cd).sum(1)[None] - 2 * xd @ cd.T + (xd * xd).sum(1, keepdim=True)
    best = dist.min(1).values
    chosen = dist.gather(1, idx.reshape(-1, 1))[:, 0]
    slack = float((chosen - best).max())
    print("largest distance excess of a chosen code over the fp64 optimum", slack)
    assert slack < 1e-4 * float(dist.abs().max())
    same = float((dist.argmin(1) == idx.reshape(-1)).float().mean())
    print("agreement with the fp64 argmin", same)
    assert same > 0.9999
    # EMA statistics through the C ABI
    from crank_amd import _lib
    from crank_amd._lib import check, ptr, stream_ptr

    L = _lib.lib()
    counts = torch.empty(K, device="cuda", dtype=torch.int32)
    sums = torch.empty(D * K, device="cuda", dtype=torch.int64)
    scratch = torch.empty(L.crk_vq_ema_scratch_bytes(N, D, K), device="cuda", dtype=torch.uint8)
    xk = x.reshape(N, D).contiguous()
    check(L.crk_vq_ema_stats(ptr(xk), D, ptr(idx), N, D, K, ptr(counts), ptr(sums), ptr(scratch), stream_ptr()), "ema_stats")
    torch.cuda.synchronize()
    assert torch.equal(counts.long(), torch.bincount(idx.reshape(-1), minlength=K))
    ref = torch.zeros(K, D, device="cuda", dtype=torch.float64).index_add_(0, idx.reshape(-1), xd)
    got = sums.view(D, K).double().T * 2.0 ** -28
    err = float((got - ref).abs().max())
    nmax = int(counts.max())
    print("fixed-point sum error", err, "largest cluster", nmax)
    assert err <= nmax * 2.0 ** -29 * 1.01 + 1e-9  # <= half an ulp of 2^-28 per summand


_NW_SCRIPT = r"""
import sys, torch, numpy as np
sys.path.insert(0, %r)
from crank_amd import ops
from crank_amd.net.module.pwg import ResidualParallelWaveGANDiscriminator
ops.set_precision("bf16")
torch.manual_seed(0)
net = ResidualParallelWaveGANDiscriminator(in_channels=113, out_channels=1, kernel_size=5, layers=8, stacks=4, dropout=0.0)
g = torch.Generator().manual_seed(1)
x = torch.randn(3, 113, 300, generator=g).cuda().requires_grad_(True)
y = net(x)
(y * torch.randn(y.shape, generator=g).cuda()).sum().backward()
torch.cuda.synchronize()
np.savez(sys.argv[1], y=y.detach().cpu().numpy(), dx=x.grad.cpu().numpy(), gp=net.grad_flat.cpu().numpy())
"""


def test_window_shapes_of_the_fused_kernels_agree_bitwise(tmp_path):
    """The 8-wave / 256-frame, 6-wave / 192-frame and 4-wave / 128-frame variants of the fused stack
    kernels do the same arithmetic per frame, so outputs, input gradients and parameter gradients must
    be identical to the bit (the utterance-group weight-gradient partials are window independent)."""
    outs = {}
    for nw in ("4", "6", "8"):
        f = tmp_path / f"nw{nw}.npz"
        env = dict(os.environ, CRK_SK_NW=nw)
        r = subprocess.run([sys.executable, "-c", _NW_SCRIPT % REPO, str(f)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[nw] = np.load(f)
    for k in ("y", "dx", "gp"):
        a = outs["4"][k]
        assert np.isfinite(a).all()
        for other in ("6", "8"):
            assert np.array_equal(a, outs[other][k]), (k, other, float(np.abs(a - outs[other][k]).max()))


_V_SCRIPT = r"""
import sys, torch, numpy as np
sys.path.insert(0, %r)
from crank_amd import ops
from crank_amd.net.module.flat import FlatModel
from crank_amd.net.module.pwg import KIND_GENERATOR, KIND_PLAIN, KIND_RESIDUAL_D, HipStack
ops.set_precision("bf16")
cfgs = [dict(kind=KIND_GENERATOR, cin=128, cout=80, k=5, layers=8, stacks=4, aux=34, B=3, T=500, drop=0.0),
        dict(kind=KIND_GENERATOR, cin=80, cout=64, k=5, layers=8, stacks=4, aux=0, B=2, T=333, drop=0.0),
        dict(kind=KIND_GENERATOR, cin=64, cout=64, k=3, layers=6, stacks=3, aux=0, B=3, T=500, drop=0.0),
        dict(kind=KIND_GENERATOR, cin=80, cout=64, k=5, layers=8, stacks=4, aux=2, B=2, T=97, drop=0.0),
        dict(kind=KIND_GENERATOR, cin=80, cout=64, k=5, layers=4, stacks=2, aux=16, B=2, T=40, drop=0.0),
        dict(kind=KIND_RESIDUAL_D, cin=113, cout=1, k=5, layers=8, stacks=4, aux=0, B=3, T=300, drop=0.0),
        dict(kind=KIND_RESIDUAL_D, cin=113, cout=1, k=5, layers=8, stacks=4, aux=0, B=2, T=500, drop=0.25),
        # chains of plain convs: the speaker classifier (mlfb / mcep input), the speaker-adversarial net, with and
        # without an input gradient (the chain then stops in front of the transposed first conv)
        dict(kind=KIND_PLAIN, cin=80, cout=14, k=5, layers=8, stacks=1, aux=0, B=3, T=500, drop=0.0),
        dict(kind=KIND_PLAIN, cin=80, cout=14, k=5, layers=8, stacks=1, aux=0, B=2, T=333, drop=0.0, nodx=True),
        dict(kind=KIND_PLAIN, cin=34, cout=12, k=5, layers=8, stacks=1, aux=0, B=2, T=130, drop=0.0, nodx=True),
        dict(kind=KIND_PLAIN, cin=128, cout=14, k=3, layers=3, stacks=1, aux=0, B=3, T=500, drop=0.0),
        dict(kind=KIND_PLAIN, cin=128, cout=14, k=3, layers=3, stacks=1, aux=0, B=2, T=97, drop=0.0, nodx=True),
        dict(kind=KIND_PLAIN, cin=64, cout=64, k=3, layers=2, stacks=1, aux=0, B=2, T=40, drop=0.0),
        # B * T a multiple of 4 (above: 1500, 80, 900, 1000; not 666, 194): the channel-split data-gradient chain then keeps
        # the planes of the weight gradient (dG, dX, dS) as 4-frame records (StackBP::rec); three more such shapes
        dict(kind=KIND_GENERATOR, cin=128, cout=80, k=5, layers=8, stacks=4, aux=34, B=4, T=504, drop=0.0),
        dict(kind=KIND_GENERATOR, cin=64, cout=64, k=3, layers=6, stacks=3, aux=0, B=2, T=240, drop=0.0),
        dict(kind=KIND_RESIDUAL_D, cin=113, cout=1, k=5, layers=8, stacks=4, aux=0, B=2, T=496, drop=0.25)]
out = {}
for i, c in enumerate(cfgs):
    torch.manual_seed(10 + i)
    class M(FlatModel):
        def __init__(self):
            super().__init__()
            self.stack = HipStack(c["kind"], c["cin"], c["cout"], c["k"], c["layers"], stacks=c["stacks"], aux_channels=c["aux"], bias=True, dropout=c["drop"])
            self._alloc(self.stack.entries("", 0), self.stack.n_params, "cuda")
            self.stack.bind(self, 0)
            self.stack.init_parameters()
    m = M()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(c["B"], c["T"], c["cin"], generator=g).cuda().requires_grad_(not c.get("nodx", False))
    a = torch.randn(c["B"], c["T"], c["aux"], generator=g).cuda().requires_grad_(True) if c["aux"] else None
    torch.manual_seed(77)  # the dropout seed of the call comes from the torch RNG
    y = m.stack(x, c=a)
    (y * torch.randn(y.shape, generator=g).cuda()).sum().backward()
    torch.cuda.synchronize()
    out[f"y{i}"], out[f"gp{i}"] = y.detach().cpu().numpy(), m.grad_flat.cpu().numpy()
    if x.grad is not None:
        out[f"dx{i}"] = x.grad.cpu().numpy()
    if a is not None:
        out[f"dc{i}"] = a.grad.cpu().numpy()
np.savez(sys.argv[1], **out)
"""


def test_channel_split_stack_kernels_equal_the_frame_split_ones_bitwise(tmp_path):
    """stack2_kernels.hip (a wave owns 32 channels, weights straight from L2 in fragment order, 64*FT-frame windows)
    against stack_kernels.hip (a wave owns 32 frames, weights through LDS): the same products are accumulated in the
    same order per output element, so outputs, input / conditioning gradients and every parameter gradient must be
    identical to the bit - generator stacks with and without conditioning (34, 2, 16 channels), k = 3 and 5,
    several windows per utterance and utterances shorter than one, the discriminator with and without dropout -
    and for every window shape the planner may pick (CRK_S2_CFG: 128 / 192 / 160 = 96 + 64 rows)."""
    outs = {}
    for tag, env_over in (("v1", {"CRK_SK_V": "1"}), ("v2", {"CRK_SK_V": "2"}), ("v2s22", {"CRK_SK_V": "2", "CRK_S2_CFG": "22"}),
                          ("v2s32", {"CRK_SK_V": "2", "CRK_S2_CFG": "32"}),
                          # 160-row windows, frame half 0 three tiles / half 1 two (the planner's pick for the k = 3 stacks)
                          ("v2s322", {"CRK_SK_V": "2", "CRK_S2_CFG": "322"}),

                          # the data-gradient chain: channel-split (stack2b_kernels.hip, default) / frame-split with the folds
                          ("v2b1", {"CRK_SK_V": "2", "CRK_SKB_V": "1"}),
                          # the plain chains: channel-split (pstack2_kernels.hip, default) / frame-split (pstack_kernels.hip)
                          ("ps1", {"CRK_PS_V": "1"})):
        f = tmp_path / f"{tag}.npz"
        r = subprocess.run([sys.executable, "-c", _V_SCRIPT % REPO, str(f)], env=dict(os.environ, **env_over), capture_output=True, text=True,
                           timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[tag] = np.load(f)
    ref = outs["v1"]
    for tag in ("v2", "v2s22", "v2s32", "v2s322", "v2b1", "ps1"):
        for k in ref.files:
            assert np.isfinite(ref[k]).all(), k
            assert np.array_equal(ref[k], outs[tag][k]), (tag, k, float(np.abs(ref[k] - outs[tag][k]).max()), float(np.abs(ref[k]).max()))


@pytest.mark.parametrize("layers,cin,k", [(8, 80, 5), (3, 128, 3)])
def test_plain_stack_parameter_gradients_do_not_depend_on_whether_the_input_gradient_is_wanted(layers, cin, k):
    """The speaker classifier's input is data and the adversarial net's is detached in its own update: the data-gradient
    chain then stops in front of the transposed first conv.  Same parameter gradients, bit for bit, as with it."""
    from crank_amd import ops
    from crank_amd.net.module.pwg import ParallelWaveGANDiscriminator

    ops.set_precision("bf16")
    torch.manual_seed(5)
    net = ParallelWaveGANDiscriminator(in_channels=cin, out_channels=14, kernel_size=k, layers=layers, conv_channels=64,
                                       dilation_factor=1, nonlinear_activation="LeakyReLU",
                                       nonlinear_activation_params={"negative_slope": 0.2}, bias=True, use_weight_norm=True)
    x = torch.randn(4, cin, 500, device="cuda")
    w = torch.randn(4, 14, 500, device="cuda")
    grads = []
    for want_dx in (True, False):
        xi = x.clone().requires_grad_(want_dx)
        net.zero_grad()
        (net(xi) * w).sum().backward()
        torch.cuda.synchronize()
        grads.append(net.grad_flat.clone())
    assert grads[0].abs().max() > 0
    assert torch.equal(grads[0], grads[1])


def test_grouped_weight_norm_backward_and_preparation_equal_the_per_stack_ones_bitwise():
    """step_model defers the weight-norm backward (and the first-conv / head weight gradients) of the generator's four
    stacks to ONE launch each and prepares all stacks in one launch after the update.  Same gradients and same
    parameters after two steps, bit for bit, as with every stack doing its own (the path plain backward() takes)."""
    from crank_amd import ops
    from crank_amd.bin.train import build_trainer
    from crank_amd.utils import load_yaml
    from tests.helpers import fill_models, make_batch

    ops.set_precision("bf16")
    conf = load_yaml(None, batch_size=4, batch_len=160)
    results = []
    for grouped in (True, False):
        torch.manual_seed(7)
        trainer = build_trainer(conf, 5, "/tmp/crank_amd_grouped")
        fill_models(trainer.model)
        for opt in trainer.optimizer.values():
            opt.clear_grads = False
        trainer.group_stack_maintenance = grouped
        for step in range(2):
            trainer.train(make_batch(4, 160, 5, seed=30 + step, device="cuda"))
        torch.cuda.synchronize()
        results.append({k: (m.grad_flat.clone(), m.flat.detach().clone()) for k, m in trainer.model.items()})
    for k in results[0]:
        assert torch.equal(results[0][k][0], results[1][k][0]), f"gradients of {k} differ"
        assert torch.equal(results[0][k][1], results[1][k][1]), f"parameters of {k} differ"


@pytest.mark.parametrize("ttype", ["vqvae", "lsgan", "cyclegan"])
def test_gradients_joined_in_the_quantizer_backward_equal_autograd_accumulation_bitwise(ttype, monkeypatch):
    """With the commitment loss inside the quantizer op, the other consumers of the encoder outputs (the
    speaker-adversarial net) and of the top stack's qx (the last decoder's concatenation) read aliases handed out by the
    op, so their gradients are added inside its ONE backward launch (crk_vq_commit_bwd) instead of in accumulation
    launches of autograd's.  Same sums: gradients, parameters and losses of three steps identical to the bit with the
    aliases switched off (ops.VQ_JOIN)."""
    from crank_amd import ops
    from crank_amd.bin.train import build_trainer

    ops.set_precision("bf16")
    over = dict(batch_size=4, batch_len=160, trainer_type=ttype)
    if ttype != "vqvae":
        over.update(n_steps_gan_start=0, discriminator_dropout=0.0)
    if ttype == "cyclegan":
        over.update(n_steps_cycle_start=0, use_cyclic_training=True)
    conf = load_yaml(None, **over)
    assert conf["use_spkradv_training"]
    results = []
    for join in (True, False):
        monkeypatch.setattr(ops, "VQ_JOIN", join)
        torch.manual_seed(7)
        trainer = build_trainer(conf, 5, "/tmp/crank_amd_join")
        fill_models(trainer.model)
        trainer.steps = 1
        trainer.check_custom_start()
        for opt in trainer.optimizer.values():
            opt.clear_grads = False
        losses = []
        for step in range(3):
            v = trainer.train(make_batch(4, 160, 5, seed=40 + step, device="cuda"))
            losses.append({k: float(x) for k, x in v.items()})
        torch.cuda.synchronize()
        results.append(({k: (m.grad_flat.clone(), m.flat.detach().clone()) for k, m in trainer.model.items()}, losses))
    (a, la), (b, lb) = results
    assert la == lb, (la, lb)
    for k in a:
        assert a[k][0].abs().max() > 0
        for i, what in enumerate(("gradients", "parameters")):
            assert torch.equal(a[k][i], b[k][i]), f"{what} of {k} differ: {float((a[k][i] - b[k][i]).abs().max())}"


@pytest.mark.parametrize("graph", [False, True])
def test_classifier_update_on_a_second_stream_leaves_every_value_unchanged(graph, monkeypatch):
    """The speaker classifier's update reads nothing but the batch and writes nothing but its own state
    (trainer_vqvae.py:186-198 of the reference), so the VQ-VAE trainer enqueues it on a second stream next to the rest of the
    step (forked at the start, joined before the loss values are collected).  Parameters, gradients, Adam moments of every
    model, codebooks, cluster sizes, moving sums and every loss value are identical to the bit with the overlap switched off
    - stepping eagerly and replaying a captured step (the fork and the join are edges of the graph there)."""
    from crank_amd import config, ops
    from crank_amd.bin.train import build_trainer

    ops.set_precision("bf16")
    conf = load_yaml(None, batch_size=4, batch_len=160, trainer_type="vqvae", hip_graph=graph)
    assert conf["use_spkr_classifier"]
    results = []
    for overlap in ("1", "2", "0"):
        monkeypatch.setattr(config.cfg, "overlap_c", int(overlap))
        torch.manual_seed(7)
        trainer = build_trainer(conf, 5, "/tmp/crank_amd_overlap")
        fill_models(trainer.model)
        trainer.steps = 1
        trainer.check_custom_start()
        assert (trainer._classifier_stream(make_batch(4, 160, 5, seed=50, device="cuda"), "train") is not None) == (overlap != "0")
        for opt in trainer.optimizer.values():
            opt.clear_grads = False
        losses = []
        for step in range(6 if graph else 3):  # (graph: three eager steps, the capture, replays)
            batch = make_batch(4, 160, 5, seed=50 + step, device="cuda", full_length=True)
            v = trainer.train_graphed(batch) if graph else trainer.train(batch)
            losses.append({k: float(x) for k, x in v.items()})
        torch.cuda.synchronize()
        if graph:
            assert any(slot[1] is not None for slot in trainer._graphs.values()), "no step was captured"
        state = {k: (m.grad_flat.clone(), m.flat.detach().clone(), trainer.optimizer[k].exp_avg.clone(),
                     trainer.optimizer[k].exp_avg_sq.clone()) for k, m in trainer.model.items()}
        qs = trainer.model["G"].quantizers
        state["ema"] = (torch.cat([q.ema_size for q in qs]), torch.cat([q.weight.reshape(-1) for q in qs]),
                        torch.cat([q.ema_w.reshape(-1) for q in qs]), torch.cat([q.ema_size for q in qs]))
        results.append((state, losses))
    for (a, la) in results[:2]:
        b, lb = results[2]
        assert la == lb, (la, lb)
        for k in a:
            for i, what in enumerate(("gradients", "parameters", "exp_avg", "exp_avg_sq")):
                assert torch.equal(a[k][i], b[k][i]), f"{what} of {k} differ: {float((a[k][i] - b[k][i]).abs().max())}"
