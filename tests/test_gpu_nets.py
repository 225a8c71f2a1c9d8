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
        # A weight_g gradient is the inner product <v, dW> / ||v|| of a row: it can cancel to a value far below the terms it
        # sums (one output channel, three frames: ResidualParallelWaveGANDiscriminator T = 3 sits at 2e-4 of ITS OWN value
        # with every other tensor at 2e-5), so its error is measured against what it is a sum of - ||dW|| of the row, which
        # under weight normalisation (g ~ ||v||) is the row norm of the weight_v gradient - when that is the larger scale.
        for k in ref:
            kv = k[:-1] + "v"
            if k.endswith("weight_g") and kv in ref:
                rown = ref[kv].detach().double().flatten(1).norm(dim=1).max().item()
                scale = max(ref[k].detach().abs().max().item(), rown) + 1e-12
                errs[k] = (got[k].detach().cpu().double() - ref[k].detach().cpu().double()).abs().max().item() / scale
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
    eps = 2e-3
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
    # (central differences across the LeakyReLU kinks of D: the error shrinks with eps, 5.8 % at 1e-2, 2.2 % at 2e-3; a mask
    # that differed between forward and backward would be off by tens of percent)
    assert abs(fd - an) <= 4e-2 * max(1.0, abs(an))
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
    _generator_stack_case(cfg, T, precision, B=2)


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_generator_stack_weight_gradient_groups_that_cross_utterances(precision):
    """A 6-block stack's weight gradient takes groups of 64-frame chunks (net.hip::stack_cpg: as many groups as fill the
    compute units), not of whole utterances.  B = 9, T = 300: 5 chunks per utterance (the last one 44 frames: with the
    32-frame chunks of bf16x3 its second half is partly behind the utterance's end), groups of 2 - every other group starts
    inside an utterance or ends in the next one.  Every parameter gradient against the oracle, as test_generator_stack."""
    _generator_stack_case(dict(in_channels=64, out_channels=64, kernel_size=3, layers=6, stacks=3, aux_channels=0), 300, precision, B=9)


def _generator_stack_case(cfg, T, precision, B):
    from oracle import pwg

    prod = _GenStack(**cfg)
    orac = pwg.ParallelWaveGANGenerator(**cfg, upsample_conditional_features=False)
    aux = cfg["aux_channels"]
    if aux:
        _check_standalone(prod, orac, cfg["in_channels"], B=B, T=T, precision=precision, aux_ch=aux, prod_call=lambda x, c: prod(x, c))
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

        _check_standalone(prod, O(), cfg["in_channels"], B=B, T=T, precision=precision)


@pytest.mark.parametrize("cfg,T", [
    (dict(in_channels=128, out_channels=80, kernel_size=5, layers=8, stacks=4, aux_channels=34), 500),  # dec0 (packed conditioning tile)
    (dict(in_channels=80, out_channels=64, kernel_size=5, layers=8, stacks=4, aux_channels=0), 500),    # enc0
    (dict(in_channels=64, out_channels=64, kernel_size=3, layers=6, stacks=3, aux_channels=0), 500),    # enc1 / dec1: 160-row windows
    (dict(in_channels=64, out_channels=64, kernel_size=3, layers=6, stacks=3, aux_channels=0), 333),
    (dict(in_channels=80, out_channels=64, kernel_size=5, layers=8, stacks=4, aux_channels=2), 500),    # enc0 with encoder_f0
    (dict(in_channels=80, out_channels=64, kernel_size=5, layers=8, stacks=4, aux_channels=0), 70),     # short: the frame-split pairing
])
def test_generator_stack_split_forward_plain_backward(cfg, T):
    """`bf16x3f` on a generator stack: the channel-split split-operand forward (stack2x_kernels.hip) against the fp32 oracle at
    the bf16x3 tolerance (2e-4 of scale) and against the frame-split split-operand forward (precision "bf16x3": the same sums
    in the same order) to 2e-5; the plain-bf16 backward behind it (it reads the hi planes this forward wrote, lane records
    included) against the oracle's gradients at bf16 accuracy."""
    from oracle import pwg

    prod = _GenStack(**cfg)
    orac = pwg.ParallelWaveGANGenerator(**cfg, upsample_conditional_features=False)
    aux = cfg["aux_channels"]
    _check_split_forward(prod, orac, cfg["in_channels"], aux, 2, T, f"gated {cfg['kernel_size']}x{cfg['layers']} aux {aux}",
                         orac_call=(None if aux else (lambda x: orac(x, None))))


@pytest.mark.parametrize("cfg,T", [
    (dict(in_channels=80, out_channels=14, kernel_size=5, layers=8), 500),    # speaker classifier C: 192-row windows
    (dict(in_channels=128, out_channels=14, kernel_size=3, layers=3), 500),   # SPKRADV classifier: 128-row windows
    (dict(in_channels=80, out_channels=14, kernel_size=5, layers=1), 150),    # a single conv
    (dict(in_channels=34, out_channels=2, kernel_size=3, layers=2), 77),
])
def test_plain_chain_split_forward_plain_backward(cfg, T):
    """The same for the plain conv chains (pstack2x_kernels.hip in front of the plain-bf16 pstack2 / weight-gradient kernels)."""
    from crank_amd.net.module.pwg import ParallelWaveGANDiscriminator
    from oracle import pwg

    kw = dict(conv_channels=64, dilation_factor=1, nonlinear_activation="LeakyReLU",
              nonlinear_activation_params={"negative_slope": 0.2}, bias=True, use_weight_norm=True)
    prod = ParallelWaveGANDiscriminator(**cfg, **kw)
    orac = pwg.ParallelWaveGANDiscriminator(**cfg, **kw)
    _check_split_forward(prod, orac, cfg["in_channels"], 0, 3, T, f"plain {cfg['kernel_size']}x{cfg['layers']}")


def _check_split_forward(prod, orac, cin, aux, B, T, tag, orac_call=None):
    from crank_amd import ops

    _load_same(prod, orac)
    rs = np.random.RandomState(3)
    c = torch.from_numpy(rs.standard_normal((B, aux, T)).astype(np.float32)) if aux else None
    x = torch.from_numpy(np.random.RandomState(100).standard_normal((B, cin, T)).astype(np.float32))

    class O(torch.nn.Module):  # (a generator stack without conditioning still takes the argument)
        def __init__(self):
            super().__init__()
            self.m = orac

        def forward(self, x):
            return orac_call(x)

        def named_parameters(self, *a, **k):
            return self.m.named_parameters(*a, **k)

    ref, dy = _oracle_run(O() if orac_call is not None else orac, x, c, None, aux)

    def run(mode):
        ops.set_precision(mode)
        try:
            xp = x.cuda().requires_grad_(True)
            cp = c.cuda().requires_grad_(True) if aux else None
            prod.zero_grad()
            yp = prod(xp, cp) if aux else prod(xp)
            (yp * dy.cuda()).sum().backward()
            torch.cuda.synchronize()
            got = {"y": yp.detach().clone(), "dx": xp.grad.clone()}
            if aux:
                got["dc"] = cp.grad.clone()
            for k in ref:
                if k not in got:
                    got[k] = prod.grad_view(k[1:]).detach().clone()
            return got
        finally:
            ops.set_precision("bf16")

    got = run("bf16x3f")
    full = run("bf16x3")
    ey, e3 = _rel(got["y"], ref["y"]), _rel(got["y"], full["y"])
    worst = max((k for k in ref if k != "y"), key=lambda k: _rl2(got[k], ref[k]))
    print(f"[bf16x3f {tag} T={T}] y vs fp32 oracle {ey:.2e}, vs frame-split bf16x3 {e3:.2e}; "
          f"gradients (plain bf16 backward): worst relative L2 {worst} {_rl2(got[worst], ref[worst]):.2e}")
    assert ey < TOL["bf16x3"], ey
    assert e3 < 2e-5, e3
    bad = {k: _rl2(got[k], ref[k]) for k in ref if k != "y" and not (_rl2(got[k], ref[k]) < 5e-2 and _cos(got[k], ref[k]) > 0.998)}
    assert not bad, bad


# ---------------------------------------------------------------------------------------------------------------------
# Deterministic pin of the benchmarked arithmetic (plain bf16): WHERE the kernels round, tap / channel wiring, rounding mode.
# The whole-network comparisons above are statistical because fp32 summation-order noise (1e-6) flips bf16 roundings and
# the flips compound.  Here the operands are chosen so that every sum is EXACT in fp32 and in float64 alike:
#   * effective weights are +-1/4: weight_v rows hold sixteen +-1 entries (||v|| = 4 exactly), weight_g = 1
#     -> w = g v / ||v|| without rounding on either side;
#   * inputs, conditioning, output gradients and biases are multiples of 1/8.
# Every conv output (16 products of at most 8 significant bits and a power of two) is then exactly representable, the
# kernel and the float64-accumulated emulation see IDENTICAL values in front of every bf16 rounding, and summation order
# cannot matter.  What is left: the gate's transcendentals (hardware exp / rcp against libm, ~1e-7) - a tanh / sigmoid / z
# value lands on the other side of a rounding boundary about once per 3e4 elements and moves one frame's outputs by
# <= 1 bf16 ulp x 1/4 - and the fp32 sums over ~450 frames in the weight gradients (~1e-6 of scale).  So the bounds are
# fixed numbers: a plain conv must agree to 2e-5 of scale in the max norm, every tensor; a gated stack's outputs and
# input / conditioning gradients must agree to 2e-5 of scale on >= 98 % of their entries, to 5e-5 in relative L2 and to
# 2e-3 of scale everywhere (measured on MI355X: bit-identical), its parameter gradients to 3e-4 of scale (measured: 8e-5).
# A rounding at another site, truncation instead of round-to-nearest-even, or a miswired tap is off by >= 1e-3 on most
# entries.
def _exact_state(shapes, seed):
    rs = np.random.RandomState(seed)
    out = {}
    for name in sorted(shapes):
        shp = tuple(shapes[name])
        if name.endswith("weight_v"):
            cout, n = shp[0], int(np.prod(shp[1:]))
            nz = 16 if n >= 16 else 4
            v = np.zeros((cout, n), dtype=np.float32)
            for r in range(cout):
                v[r, rs.choice(n, nz, replace=False)] = rs.choice([-1.0, 1.0], nz)
            out[name] = v.reshape(shp)
            out[name[: -len("weight_v")] + "weight_g"] = np.full((cout, 1, 1), np.sqrt(nz) * 0.25, dtype=np.float32)
        elif name.endswith("bias"):
            out[name] = (rs.randint(-4, 5, size=shp) / 8.0).astype(np.float32)
    for name in shapes:
        assert name in out, name
    return out


def _eighths(rs, shape, lim=16):
    return torch.from_numpy((rs.randint(-lim, lim + 1, size=shape) / 8.0).astype(np.float32))


def _pin_metrics(got, ref):
    g = got.detach().cpu().double()
    r = ref.detach().cpu().double()
    scale = r.abs().max().item() + 1e-30
    err = (g - r).abs()
    return {"max": err.max().item() / scale, "rl2": float(err.norm() / (r.norm() + 1e-30)),
            "frac>2e-5": float((err > 2e-5 * scale).double().mean())}


def _pin_case(case):
    from crank_amd.net.module.pwg import ParallelWaveGANDiscriminator
    from oracle import pwg as opwg

    if case == "plain_conv_k5":
        cfg = dict(in_channels=80, out_channels=14, kernel_size=5, layers=1)
        kw = dict(conv_channels=64, dilation_factor=1, nonlinear_activation="LeakyReLU",
                  nonlinear_activation_params={"negative_slope": 0.2}, bias=True, use_weight_norm=True)
        prod = ParallelWaveGANDiscriminator(**cfg, **kw) if torch.cuda.is_available() else None
        orac = opwg.ParallelWaveGANDiscriminator(**cfg, **kw)
        return cfg, 0, prod, orac, (lambda m, x, c: m(x))
    if case == "gated_k3_one_block":
        cfg = dict(in_channels=64, out_channels=64, kernel_size=3, layers=1, stacks=1, aux_channels=0)
    else:  # dilation is 2**(l % layers_per_stack): dilation 2 needs a second block
        cfg = dict(in_channels=128, out_channels=80, kernel_size=5, layers=2, stacks=1, aux_channels=34)
    prod = _GenStack(**cfg) if torch.cuda.is_available() else None
    orac = opwg.ParallelWaveGANGenerator(**cfg, upsample_conditional_features=False)
    return cfg, cfg["aux_channels"], prod, orac, (lambda m, x, c: m(x, c))


def _pin_oracle(orac, call, x, c, dy, accumulate):
    from oracle import pwg as opwg

    with opwg.bf16_emulation(accumulate=accumulate):
        xo = x.clone().requires_grad_(True)
        co = c.clone().requires_grad_(True) if c is not None else None
        orac.zero_grad()
        yo = call(orac, xo, co)
        (yo * dy).sum().backward()
    ref = {"y": yo.detach().clone(), "dx": xo.grad.clone()}
    if co is not None:
        ref["dc"] = co.grad.clone()
    for k, p in orac.named_parameters():
        if p.grad is not None:
            ref["d" + k] = p.grad.clone()
    return ref


PIN_CASES = ["plain_conv_k5", "gated_k3_one_block", "gated_k5_dil12_aux34"]


def _pin_inputs(cfg, aux, B=3, T=150):
    rs = np.random.RandomState(17)
    x = _eighths(rs, (B, cfg["in_channels"], T))
    c = _eighths(rs, (B, aux, T)) if aux else None
    dy = _eighths(rs, (B, cfg["out_channels"], T))
    return x, c, dy


@pytest.mark.parametrize("case", PIN_CASES)
def test_bf16_single_layer_deterministic_pin(case):
    """Plain bf16 (what bench.py times): forward, dx, dc and every weight / bias gradient of one plain conv, of a
    one-block gated stack (k3, dilation 1) and of a two-block gated stack with conditioning (k5, dilations 1 and 2,
    aux 34) against the float64-accumulated bf16 emulation, on exactly representable operands, with fixed bounds."""
    from crank_amd import ops

    ops.set_precision("bf16")
    cfg, aux, prod, orac, call = _pin_case(case)
    sd = orac.state_dict()
    vals = _exact_state({k: tuple(v.shape) for k, v in sd.items()}, 321)
    orac.load_state_dict({k: torch.from_numpy(vals[k]) for k in sd})
    prod.load_state_dict({k: torch.from_numpy(vals[k]) for k in sd})
    x, c, dy = _pin_inputs(cfg, aux)
    ref = _pin_oracle(orac, call, x, c, dy, "fp64")

    xp = x.cuda().requires_grad_(True)
    cp = c.cuda().requires_grad_(True) if aux else None
    prod.zero_grad()
    yp = call(prod, xp, cp)
    (yp * dy.cuda()).sum().backward()
    torch.cuda.synchronize()
    got = {"y": yp, "dx": xp.grad}
    if aux:
        got["dc"] = cp.grad
    for k in ref:
        if k not in got:
            got[k] = prod.grad_view(k[1:])

    strict = case == "plain_conv_k5"
    bad, worst = {}, ("", 0.0)
    for k in ref:
        m = _pin_metrics(got[k], ref[k])
        if m["max"] > worst[1]:
            worst = (k, m["max"])
        if strict:
            ok = m["max"] <= 2e-5
        elif k in ("y", "dx", "dc"):  # activations and their gradients: exact up to the rare gate flips
            ok = m["frac>2e-5"] <= 2e-2 and m["rl2"] <= 5e-5 and m["max"] <= 2e-3
        else:  # parameter gradients: fp32 sums over B * T frames (and the weight-norm backward's dot products) on top
            ok = m["max"] <= 3e-4 and m["rl2"] <= 1e-4
        if not ok:
            bad[k] = m
    my, mx = _pin_metrics(got["y"], ref["y"]), _pin_metrics(got["dx"], ref["dx"])
    print(f"[pin {case}] y max {my['max']:.1e} rl2 {my['rl2']:.1e} frac>2e-5 {my['frac>2e-5']:.1e}; dx max {mx['max']:.1e} "
          f"rl2 {mx['rl2']:.1e} frac>2e-5 {mx['frac>2e-5']:.1e}; worst max-norm {worst[0]} {worst[1]:.1e}")
    assert not bad, bad


@pytest.mark.parametrize("layers,cin,k,T", [(8, 80, 5, 500), (3, 128, 3, 500), (8, 34, 5, 130), (3, 128, 3, 97)])
@pytest.mark.parametrize("want_dx", [True, False])
def test_classifier_cross_entropy_as_one_op_equals_the_composed_ops(layers, cin, k, T, want_dx):
    """ops.net_ce - classifier and cross entropy as one autograd node whose backward hands the unnormalised gradient to the
    chain and lets its first kernel take the loss scale on the device - against net_apply + cross_entropy + the scaling
    launch: the same loss, input and parameter gradients bit for bit (the same unnormalised gradient, the same product with
    the same factor).  Targets carry ignore_index rows."""
    from crank_amd import ops
    from crank_amd.net.module.pwg import ParallelWaveGANDiscriminator

    ops.set_precision("bf16")
    torch.manual_seed(11)
    net = ParallelWaveGANDiscriminator(in_channels=cin, out_channels=14, kernel_size=k, layers=layers, conv_channels=64,
                                       dilation_factor=1, nonlinear_activation="LeakyReLU",
                                       nonlinear_activation_params={"negative_slope": 0.2}, bias=True, use_weight_norm=True)
    B = 4
    x = torch.randn(B, cin, T, device="cuda")
    tgt = torch.randint(0, 14, (B, T), device="cuda")
    tgt[1, T // 3: T // 2] = -100
    res = []
    for fused in (True, False):
        xi = x.clone().requires_grad_(want_dx)
        net.zero_grad()
        if fused:
            loss = net.forward_ce(xi, tgt)
        else:
            logits = net(xi).transpose(1, 2)
            loss = ops.cross_entropy(logits.reshape(-1, 14), tgt.reshape(-1))
        (1.7 * loss).backward()
        torch.cuda.synchronize()
        res.append((loss.item(), net.grad_flat.clone(), xi.grad.clone() if want_dx else None))
    ref = torch.nn.functional.cross_entropy(net(x).transpose(1, 2).reshape(-1, 14).double(), tgt.reshape(-1), ignore_index=-100)
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=2e-6)
    np.testing.assert_allclose(res[0][0], ref.item(), rtol=1e-5)
    assert res[0][1].abs().max() > 0
    assert torch.equal(res[0][1], res[1][1])
    if want_dx:
        assert torch.equal(res[0][2], res[1][2])


def test_compute_entry_points_never_allocate_and_check_how_they_are_paired(capfd):
    """The boundary's contract (include/crank_hip.h): crk_net_reserve(net, B, T) makes everything a batch
    shape needs, crk_net_forward / crk_net_backward allocate nothing (crk_debug_alloc_count does not move over two
    forward + backward passes), refuse a shape that was not reserved with CRK_ERR_ARG - nothing launched - and a backward
    whose flags do not describe the forward that filled its workspace (the plane layouts differ) is refused the same way
    instead of reading planes that were never written."""
    import ctypes

    from crank_amd import _lib, ops
    from crank_amd._lib import ptr, stream_ptr

    L = _lib.lib()
    ops.set_precision("bf16")
    net = ops.HipNet(kind=0, in_ch=80, out_ch=64, kernel_size=5, layers=4, stacks=2, res_ch=64, gate_ch=128, skip_ch=64, aux_ch=0,
                     conv_ch=64, causal=0, use_bias=1, slope=0.2, dropout=0.0)
    B, T = 3, 200
    g = torch.Generator().manual_seed(0)
    params = (0.1 * torch.randn(net.n_params, generator=g)).abs().add_(0.05).cuda()  # (weight_g must not be ~0)
    grads = torch.zeros_like(params)
    x = torch.randn(B, T, 80, generator=g).cuda()
    dy = torch.randn(B, T, 64, generator=g).cuda()
    y, dx = torch.empty(B, T, 64, device="cuda"), torch.empty(B, T, 80, device="cuda")

    def fwd(Bc, flags, saved):
        return L.crk_net_forward(net.handle, ptr(params), 1, ptr(x), 80, None, 0, ptr(y), 64, ptr(saved), Bc, T, flags, 0, stream_ptr())

    def bwd(Bc, flags, saved):
        return L.crk_net_backward(net.handle, ptr(params), 1, ptr(grads), ptr(x), 80, None, 0, ptr(dy), 64, ptr(dx), 80, 1.0, None, 0,
                                  ptr(saved), Bc, T, flags, 0, stream_ptr())

    saved = torch.empty(L.crk_net_saved_bytes(net.handle, B, T) // 4 + 1, device="cuda")
    # not reserved: refused, nothing allocated
    a0 = L.crk_debug_alloc_count()
    assert fwd(B, 0, saved) == 1 and bwd(B, 0, saved) == 1
    assert L.crk_debug_alloc_count() == a0
    assert "crk_net_reserve" in capfd.readouterr().err
    assert L.crk_net_scratch_bytes(net.handle, B, T) > 0
    assert L.crk_net_reserve(net.handle, B, T) == 0
    a1 = L.crk_debug_alloc_count()
    assert a1 > a0
    for _ in range(2):
        assert fwd(B, 0, saved) == 0 and bwd(B, 0, saved) == 0
    torch.cuda.synchronize()
    assert L.crk_debug_alloc_count() == a1, "a compute entry point allocated"
    assert torch.isfinite(dx).all() and float(grads.abs().max()) > 0
    assert L.crk_net_reserve(net.handle, B, T) == 0 and L.crk_debug_alloc_count() == a1  # idempotent
    # a smaller batch with other slot counts was not reserved either
    assert fwd(2, 0, saved) == 1
    # pairing: plain forward, then a backward that claims the split-operand forward wrote the planes (and the other way round)
    PRECISE, FWD_PRECISE, BWD_PLAIN = 1, 32, 64
    assert fwd(B, 0, saved) == 0
    assert bwd(B, FWD_PRECISE, saved) == 1 and bwd(B, PRECISE, saved) == 1
    assert fwd(B, PRECISE, saved) == 0  # the documented bf16x3 forward ...
    assert bwd(B, FWD_PRECISE, saved) == 1  # ... does not pair with the bf16x3f backward (the advisor's case)
    assert bwd(B, PRECISE, saved) == 0
    assert fwd(B, PRECISE | BWD_PLAIN, saved) == 0
    assert bwd(B, 0, saved) == 1 and bwd(B, FWD_PRECISE, saved) == 0
    torch.cuda.synchronize()
    assert "do not match the forward" in capfd.readouterr().err


def test_the_fast_kernel_generations_are_what_runs_at_the_benchmark_shape():
    """Every older kernel generation stays in the library as a fallback that computes the same values (that is what the
    bitwise tests compare), so a plan that starts refusing a shape does not fail a test - it only costs time: the bf16x3f
    forward ran on the frame-split kernels for part of round 6 (2.94 ms per step against 1.99) because the predicate
    handed the planner a half-filled shape.  crk_debug_net_paths names what the compute entry points pick; pinned here
    for the nets of the benchmark (configs[1] / configs[2]: 64 and 128 x 500 frames; the trainer's own models)."""
    from crank_amd import _lib, ops
    from crank_amd.bin.train import get_model
    from crank_amd.utils import load_yaml

    L = _lib.lib()
    ops.set_precision("bf16")
    conf = load_yaml(None, batch_size=64, batch_len=500, trainer_type="lsgan")
    m = get_model(conf, 14, "cuda")
    def nets_of(obj, depth=0, out=None, visited=None):
        out = [] if out is None else out
        visited = set() if visited is None else visited
        if id(obj) in visited or depth > 4:
            return out
        visited.add(id(obj))
        if isinstance(obj, ops.HipNet):
            out.append(obj)
        elif isinstance(obj, (list, tuple)):
            for o in obj:
                nets_of(o, depth + 1, out, visited)
        elif hasattr(obj, "__dict__") and not isinstance(obj, torch.Tensor):
            for o in list(vars(obj).values()) + list(getattr(obj, "_modules", {}).values()):
                nets_of(o, depth + 1, out, visited)
        return out

    seen = {0: 0, 1: 0, 2: 0}
    for name, model in m.items():
        for net in nets_of(model):
            kind = None
            for B, T in ((64, 500), (128, 500)):
                paths = L.crk_debug_net_paths(net.handle, B, T)
                assert paths > 0, (name, B, T, paths, "a net of the step on a fallback generation")
                kind = 2 if paths & 8 else (1 if paths & 4 else 0)
                if kind == 0:
                    assert paths & 1 and paths & 2, (name, B, T, paths, "generator stack: plain / bf16x3f forward not channel-split")
            seen[kind] += 1
    assert seen[0] == 4 and seen[1] >= 1 and seen[2] >= 2, seen  # four generator stacks, D, C and SPKRADV
