"""GPU parity: HIP conv stacks (through the C ABI) vs the CPU oracle's torch modules on
identical weights and inputs: outputs, input gradients and every parameter gradient.

Tolerances: "bf16x3" (split-operand MFMA, ~fp32) must agree with the fp32 oracle to
2e-4 of the tensor's scale (the north-star bar is 1e-3 relative).  Plain "bf16" - the
throughput mode bench.py times - is pinned against the oracle's bf16-EMULATION mode
(oracle/pwg.py: operands rounded to bf16 exactly where the kernels round them): outputs,
input / conditioning gradients and every parameter gradient must lie as close to the
float64-accumulated emulation as two fp32 CPU evaluations of the same arithmetic do
(see _check_standalone); against the fp32 oracle the outputs additionally stay within
bf16-level agreement (3e-2 of scale)."""
import numpy as np
import pytest
import torch

from tests.helpers import deterministic_state

pytestmark = pytest.mark.gpu

TOL = {"bf16x3": 2e-4, "bf16": 3e-2}


def _rel(a, b):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    scale = b.abs().max().item() + 1e-12
    return ((a - b).abs().max().item()) / scale


def _load_same(prod, orac, seed=99):
    sd = orac.state_dict()
    vals = deterministic_state({k: tuple(v.shape) for k, v in sd.items()}, seed)
    orac.load_state_dict({k: torch.from_numpy(vals[k]) for k in sd})
    prod.load_state_dict({k: torch.from_numpy(vals[k]) for k in sd})


def _pick_input_away_from_kinks(orac, B, cin, T, n_seeds=24, extra=None):
    """LeakyReLU / ReLU gradients are discontinuous at 0: a pre-activation within the
    forward error (~1e-5) of zero can take a different branch on the GPU than in the
    oracle, which changes a handful of gradient entries by O(1) without being an error
    of either side.  Choose, among a few seeds, the input whose smallest hidden
    |pre-activation| is largest, so every unit is safely on one side."""
    best = (-1.0, None)
    mins = []

    def hook(_m, inp):
        mins.append(inp[0].detach().abs().min().item())

    hs = [m.register_forward_pre_hook(hook) for m in orac.modules()
          if isinstance(m, (torch.nn.LeakyReLU, torch.nn.ReLU))]
    for seed in range(n_seeds):
        rs = np.random.RandomState(100 + seed)
        x = torch.from_numpy(rs.standard_normal((B, cin, T)).astype(np.float32))
        mins.clear()
        with torch.no_grad():
            orac(x) if extra is None else orac(x, extra)
        m = min(mins) if mins else 1.0
        if m > best[0]:
            best = (m, x)
    for h in hs:
        h.remove()
    return best[1], best[0]


def _cos(a, b):
    a = a.detach().cpu().double().reshape(-1)
    b = b.detach().cpu().double().reshape(-1)
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def _rl2(a, b):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _oracle_run(orac, x, c, dy, aux_ch):
    xo = x.clone().requires_grad_(True)
    co = c.clone().requires_grad_(True) if aux_ch else None
    orac.zero_grad()
    yo = orac(xo, co) if aux_ch else orac(xo)
    if dy is None:
        dy = torch.from_numpy(np.random.RandomState(3).standard_normal(tuple(yo.shape)).astype(np.float32))
    (yo * dy).sum().backward()
    out = {"y": yo.detach().clone(), "dx": xo.grad.clone()}
    if aux_ch:
        out["dc"] = co.grad.clone()
    for k, p in orac.named_parameters():
        if p.grad is not None:  # e.g. the last block's conv1x1_out never gets a gradient
            out["d" + k] = p.grad.clone()
    return out, dy


def _check_standalone(prod, orac, cin, B, T, precision, lengths=None, aux_ch=0, prod_call=None):
    """precision "bf16x3": outputs, input gradients and every parameter gradient against the fp32 oracle, 2e-4 of
    scale.  precision "bf16" (the benchmarked arithmetic): against the oracle's bf16-emulation mode.  Rounding to
    bf16 is discontinuous, so two evaluations of the same arithmetic that only sum in a different order already
    disagree by percents in the gradients of an 8-layer net (oracle/pwg.py); the test therefore evaluates the
    emulation three ways on the CPU - float64 accumulation (the arithmetic's exact value), fp32, fp32 with permuted
    summation - and requires the kernel to lie as close to the float64 result as the fp32 evaluations do (relative
    L2 error within 3x the larger of theirs).  A wrong operand, tap, halo row or rounding mode is off by O(1)."""
    from crank_amd import ops
    from oracle import pwg as opwg

    ops.set_precision(precision)
    _load_same(prod, orac)
    rs = np.random.RandomState(3)
    c = torch.from_numpy(rs.standard_normal((B, aux_ch, T)).astype(np.float32)) if aux_ch else None
    if precision == "bf16":
        x = torch.from_numpy(np.random.RandomState(100).standard_normal((B, cin, T)).astype(np.float32))
        with opwg.bf16_emulation(accumulate="fp64"):
            ref, dy = _oracle_run(orac, x, c, None, aux_ch)
        with opwg.bf16_emulation(accumulate="fp32"):
            e32, _ = _oracle_run(orac, x, c, dy, aux_ch)
        with opwg.bf16_emulation(accumulate="fp32-permuted"):
            e32p, _ = _oracle_run(orac, x, c, dy, aux_ch)
        margin = float("nan")
    else:
        x, margin = _pick_input_away_from_kinks(orac, B, cin, T, extra=c)
        ref, dy = _oracle_run(orac, x, c, None, aux_ch)
    xp = x.cuda().requires_grad_(True)
    cp = c.cuda().requires_grad_(True) if aux_ch else None
    prod.zero_grad()
    yp = prod_call(xp, cp) if prod_call is not None else prod(xp)
    (yp * dy.cuda()).sum().backward()
    torch.cuda.synchronize()
    got = {"y": yp, "dx": xp.grad}
    if aux_ch:
        got["dc"] = cp.grad
    for k in ref:
        if k not in got:
            got[k] = prod.grad_view(k[1:])
    if precision == "bf16x3":
        errs = {k: _rel(got[k], ref[k]) for k in ref}
        worst = max(errs, key=errs.get)
        print(f"[{type(prod).__name__} bf16x3 B={B} T={T}] kink margin {margin:.1e} y {errs['y']:.2e} dx {errs['dx']:.2e} "
              f"worst {worst} {errs[worst]:.2e}")
        bad = {k: v for k, v in errs.items() if not (v < TOL["bf16x3"])}
        assert not bad, bad
    else:
        bad, worst = {}, ("", 0.0, 0.0)
        for k in ref:
            noise = max(_rl2(e32[k], ref[k]), _rl2(e32p[k], ref[k]))
            err = _rl2(got[k], ref[k])
            if err > 3.0 * noise + 2e-4:
                bad[k] = (err, noise)
            if err / (noise + 1e-6) > worst[1]:
                worst = (k, err / (noise + 1e-6), err)
        ny = max(_rl2(e32["y"], ref["y"]), _rl2(e32p["y"], ref["y"]))
        ndx = max(_rl2(e32["dx"], ref["dx"]), _rl2(e32p["dx"], ref["dx"]))
        print(f"[{type(prod).__name__} bf16 vs bf16-emulating oracle (float64-accumulated) B={B} T={T}] relative L2: y kernel "
              f"{_rl2(got['y'], ref['y']):.2e} / cpu fp32 {ny:.2e}; dx kernel {_rl2(got['dx'], ref['dx']):.2e} / cpu fp32 {ndx:.2e}; "
              f"largest kernel/cpu ratio {worst[1]:.2f} ({worst[0]}, {worst[2]:.2e})")
        assert not bad, bad
        # and against the fp32 reference arithmetic: bf16-level agreement of the outputs
        with torch.no_grad():
            yf = orac(x, c) if aux_ch else orac(x)
        e32o = _rel(yp, yf)
        print(f"   vs fp32 oracle: y {e32o:.2e} of scale")
        assert e32o < TOL["bf16"], e32o
    ops.set_precision("bf16")


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
@pytest.mark.parametrize("cfg", [
    dict(in_channels=80, out_channels=14, kernel_size=5, layers=1),    # a single conv
    dict(in_channels=80, out_channels=14, kernel_size=5, layers=8),    # speaker classifier C (train.py:78-89)
    dict(in_channels=128, out_channels=14, kernel_size=3, layers=3),   # SPKRADV classifier (spkradv.py:49-60)
    dict(in_channels=34, out_channels=2, kernel_size=3, layers=2),
])
def test_plain_stack(cfg, precision):
    from crank_amd.net.module.pwg import ParallelWaveGANDiscriminator
    from oracle import pwg

    kw = dict(conv_channels=64, dilation_factor=1, nonlinear_activation="LeakyReLU",
              nonlinear_activation_params={"negative_slope": 0.2}, bias=True, use_weight_norm=True)
    prod = ParallelWaveGANDiscriminator(**cfg, **kw)
    orac = pwg.ParallelWaveGANDiscriminator(**cfg, **kw)
    _check_standalone(prod, orac, cfg["in_channels"], B=3, T=150, precision=precision)


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
@pytest.mark.parametrize("cfg", [
    dict(in_channels=113, out_channels=1, kernel_size=5, layers=8, stacks=4),   # D (train.py:108-118)
    dict(in_channels=67, out_channels=15, kernel_size=3, layers=2, stacks=1),
])
def test_residual_discriminator(cfg, precision):
    from crank_amd.net.module.pwg import ResidualParallelWaveGANDiscriminator
    from oracle import pwg

    prod = ResidualParallelWaveGANDiscriminator(**cfg, dropout=0.0)
    orac = pwg.ResidualParallelWaveGANDiscriminator(**cfg, dropout=0.0)
    _check_standalone(prod, orac, cfg["in_channels"], B=2, T=200, precision=precision)


def test_tile_boundaries_and_ragged_T():
    """T not a multiple of the 128-frame tile, T smaller than a tile, T == 1 tile."""
    from crank_amd.net.module.pwg import ParallelWaveGANDiscriminator
    from oracle import pwg

    kw = dict(in_channels=80, out_channels=14, kernel_size=5, layers=4, conv_channels=64)
    prod = ParallelWaveGANDiscriminator(**kw)
    orac = pwg.ParallelWaveGANDiscriminator(**kw)
    for T in [17, 128, 129, 500]:
        _check_standalone(prod, orac, 80, B=2, T=T, precision="bf16x3")


def test_dropout_mask_is_consistent_between_forward_and_backward():
    """Residual D with dropout: the backward must regenerate the forward's keep mask.
    Checked through a finite-difference-free identity: with dropout the network is still
    piecewise linear in its FIRST-conv bias direction only through kept units, so we
    compare dx against a second backward of the same graph (determinism) and check the
    keep rate through the first block's effect on the output."""
    from crank_amd import ops
    from crank_amd.net.module.pwg import ResidualParallelWaveGANDiscriminator

    ops.set_precision("bf16x3")
    torch.manual_seed(0)
    prod = ResidualParallelWaveGANDiscriminator(in_channels=20, out_channels=1, kernel_size=3, layers=2, stacks=1,
                                                dropout=0.25)
    x = torch.randn(2, 20, 100, device="cuda", requires_grad=True)
    y = prod(x)
    g1, = torch.autograd.grad(y.sum(), x, retain_graph=True)
    g2, = torch.autograd.grad(y.sum(), x)
    assert torch.isfinite(y).all() and torch.isfinite(g1).all()
    assert torch.equal(g1, g2)
    y2 = prod(x)  # new seed -> different mask
    assert (y - y2).abs().max().item() > 0
    # directional derivative with the mask pinned by restarting the net's device-side seed sequence
    v = torch.randn_like(x)
    eps = 1e-2
    prod.stack.net.reseed(7)
    ya = prod(x.detach().requires_grad_(True))
    prod.stack.net.reseed(7)
    xb = x.detach().clone().requires_grad_(True)
    yb = prod(xb)
    gb, = torch.autograd.grad(yb.sum(), xb)
    prod.stack.net.reseed(7)
    yc = prod(x.detach() + eps * v)
    prod.stack.net.reseed(7)
    yd = prod(x.detach() - eps * v)
    assert torch.equal(ya, yb)
    fd = ((yc.double().sum() - yd.double().sum()) / (2 * eps)).item()
    an = (gb * v).sum().item()
    print("dropout directional derivative fd", fd, "analytic", an)
    assert abs(fd - an) <= 2e-2 * max(1.0, abs(an))
    ops.set_precision("bf16")


@pytest.mark.parametrize("T", [1, 3, 40, 257])
def test_gated_stack_edge_lengths(T):
    """Utterances shorter than the receptive field / than one window, and one frame past a window:
    the fused kernels' halo, guard-row and bounds-check logic against the oracle (bf16x3)."""
    from crank_amd.net.module.pwg import ResidualParallelWaveGANDiscriminator
    from oracle import pwg

    cfg = dict(in_channels=113, out_channels=1, kernel_size=5, layers=8, stacks=4)
    prod = ResidualParallelWaveGANDiscriminator(**cfg, dropout=0.0)
    orac = pwg.ResidualParallelWaveGANDiscriminator(**cfg, dropout=0.0)
    _check_standalone(prod, orac, 113, B=2, T=T, precision="bf16x3")


class _GenStack:
    """A generator stack (ParallelWaveGANGenerator call sites crank/net/module/vqvae2.py:237-273) as a model of its own."""

    def __new__(cls, **kw):
        from crank_amd.net.module.flat import FlatModel
        from crank_amd.net.module.pwg import KIND_GENERATOR, HipStack

        class M(FlatModel):
            def __init__(self):
                super().__init__()
                self.stack = HipStack(KIND_GENERATOR, kw["in_channels"], kw["out_channels"], kw["kernel_size"], kw["layers"],
                                      stacks=kw["stacks"], aux_channels=kw["aux_channels"], bias=True)
                self._alloc(self.stack.entries("", 0), self.stack.n_params, "cuda")
                self.stack.bind(self, 0)
                self.stack.init_parameters()

            def forward(self, x, c=None):
                return self.stack(x.transpose(1, 2), c=None if c is None else c.transpose(1, 2)).transpose(1, 2)

        return M()


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
@pytest.mark.parametrize("cfg,T", [
    (dict(in_channels=128, out_channels=80, kernel_size=5, layers=8, stacks=4, aux_channels=34), 500),  # dec0: aux, 3 windows of the 8-wave shape
    (dict(in_channels=80, out_channels=64, kernel_size=5, layers=8, stacks=4, aux_channels=0), 500),    # enc0
    (dict(in_channels=64, out_channels=64, kernel_size=3, layers=6, stacks=3, aux_channels=0), 333),    # enc1 / dec1
    (dict(in_channels=80, out_channels=64, kernel_size=5, layers=8, stacks=4, aux_channels=2), 130),    # enc0 with encoder_f0
])
def test_generator_stack(cfg, T, precision):
    """The four encoder / decoder stacks of G at the benchmark's utterance length (T = 500 spans several windows
    of the fused kernels, with the conditioning chunk on dec0): forward, dx, dc and every parameter gradient."""
    from oracle import pwg

    prod = _GenStack(**cfg)
    orac = pwg.ParallelWaveGANGenerator(**cfg, upsample_conditional_features=False)
    aux = cfg["aux_channels"]
    if aux:
        _check_standalone(prod, orac, cfg["in_channels"], B=2, T=T, precision=precision, aux_ch=aux, prod_call=lambda x, c: prod(x, c))
    else:
        orac_call = orac

        class O(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.m = orac_call

            def forward(self, x):
                return self.m(x, None)

            def state_dict(self, *a, **k):
                return self.m.state_dict(*a, **k)

            def load_state_dict(self, sd, strict=True):
                return self.m.load_state_dict(sd, strict)

            def named_parameters(self, *a, **k):
                return self.m.named_parameters(*a, **k)

        _check_standalone(prod, O(), cfg["in_channels"], B=2, T=T, precision=precision)
