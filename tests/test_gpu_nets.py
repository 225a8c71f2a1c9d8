"""GPU parity: HIP conv stacks (through the C ABI) vs the CPU oracle's torch modules on
identical weights and inputs: outputs, input gradients and every parameter gradient.

Tolerances: "bf16x3" (split-operand MFMA, ~fp32) must agree with the fp32 oracle to
2e-4 of the tensor's scale (the north-star bar is 1e-3 relative).  Plain "bf16" - the
throughput mode bench.py times - is pinned against the oracle's bf16-EMULATION mode
(oracle/pwg.py: operands rounded to bf16 exactly where the kernels round them, fp32
accumulation) to the same 1e-3 of scale, outputs, input gradients and every parameter
gradient; against the fp32 oracle it additionally stays within bf16-level agreement
(3e-2 of scale on outputs)."""
import numpy as np
import pytest
import torch

from tests.helpers import deterministic_state

pytestmark = pytest.mark.gpu

TOL = {"bf16x3": 2e-4, "bf16": 3e-2, "bf16_vs_emulation": 1e-3}


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


def _check_standalone(prod, orac, cin, B, T, precision, lengths=None, aux_ch=0, prod_call=None):
    """precision "bf16x3": vs the fp32 oracle; "bf16": vs the oracle in bf16-emulation mode (the pin of the
    benchmarked arithmetic) and, loosely, vs the fp32 oracle.  `aux_ch` > 0: the stack takes a conditioning
    input c (B, aux_ch, T) (generator)."""
    import contextlib

    from crank_amd import ops
    from oracle import pwg as opwg

    ops.set_precision(precision)
    _load_same(prod, orac)
    emu = precision == "bf16"
    ctx = opwg.bf16_emulation if emu else contextlib.nullcontext
    rs = np.random.RandomState(3)
    c = torch.from_numpy(rs.standard_normal((B, aux_ch, T)).astype(np.float32)) if aux_ch else None
    with ctx():
        x, margin = _pick_input_away_from_kinks(orac, B, cin, T, extra=c)
        xo = x.clone().requires_grad_(True)
        co = c.clone().requires_grad_(True) if aux_ch else None
        orac.zero_grad()
        yo = orac(xo, co) if aux_ch else orac(xo)
        dy = torch.from_numpy(rs.standard_normal(tuple(yo.shape)).astype(np.float32))
        (yo * dy).sum().backward()
    xp = x.cuda().requires_grad_(True)
    cp = c.cuda().requires_grad_(True) if aux_ch else None
    prod.zero_grad()
    yp = prod_call(xp, cp) if prod_call is not None else prod(xp)
    (yp * dy.cuda()).sum().backward()
    torch.cuda.synchronize()
    pairs = {"dx": (xp.grad, xo.grad)}
    if aux_ch:
        pairs["dc"] = (cp.grad, co.grad)
    for k, p in orac.named_parameters():
        if p.grad is not None:  # e.g. the last block's conv1x1_out never gets a gradient
            pairs["d" + k] = (prod.grad_view(k), p.grad)
    errs = {"y": _rel(yp, yo)}
    errs.update({k: _rel(a, b) for k, (a, b) in pairs.items()})
    worst = max(errs, key=errs.get)
    print(f"[{type(prod).__name__} {precision}{' vs bf16-emulating oracle' if emu else ''} B={B} T={T}] kink margin "
          f"{margin:.1e} y {errs['y']:.2e} dx {errs['dx']:.2e} worst {worst} {errs[worst]:.2e}")
    tol = TOL["bf16_vs_emulation"] if emu else TOL[precision]
    bad = {k: v for k, v in errs.items() if not (v < tol)}
    assert not bad, bad
    if emu:
        # and against the fp32 reference arithmetic: bf16-level agreement of the outputs
        with torch.no_grad():
            yf = orac(x, c) if aux_ch else orac(x)
        e32 = _rel(yp, yf)
        print(f"   vs fp32 oracle: y {e32:.2e}")
        assert e32 < TOL["bf16"], e32
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
    # directional derivative with the mask pinned through the torch RNG seed
    v = torch.randn_like(x)
    eps = 1e-2
    torch.manual_seed(7)
    ya = prod(x.detach().requires_grad_(True))
    torch.manual_seed(7)
    xb = x.detach().clone().requires_grad_(True)
    yb = prod(xb)
    gb, = torch.autograd.grad(yb.sum(), xb)
    torch.manual_seed(7)
    yc = prod(x.detach() + eps * v)
    torch.manual_seed(7)
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
