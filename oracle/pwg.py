"""Oracle (TEST INFRASTRUCTURE): PyTorch-CPU restatement of the ``parallel_wavegan``
networks that crank instantiates.

The package itself is a third-party dependency that is NOT under /root/reference
(``tools/requirements.txt:9`` un-pinned; submodule ``.gitmodules:1-3`` not vendored)
and is not installed in this image.  The definitions below restate its published
0.4.x/0.5.x behaviour as relied upon by crank's call sites
(``crank/net/module/vqvae2.py:237-273``, ``crank/net/module/spkradv.py:49-60``,
``crank/bin/train.py:78-128``); see SURVEY.md Appendix A.0-A.5.  Parity of these
stacks against the third-party code is therefore unpinned; constructor-level tests
check state-dict key names and parameter counts.

Upstream notice: the classes restated here (argument lists, attribute and state-dict key
names, layer arithmetic) are those of kan-bayashi/ParallelWaveGAN, published under the MIT
License, Copyright (c) 2019 Tomoki Hayashi.  No upstream source text is in this file; the
permission notice of that licence applies to the design it follows:
"Permission is hereby granted, free of charge, to any person obtaining a copy of this
software and associated documentation files (the "Software"), to deal in the Software
without restriction ... THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY KIND".
"""
import contextlib
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

# ---------------------------------------------------------------------------------------
# bf16-emulation mode (TEST INFRASTRUCTURE, like everything else in oracle/).
#
# The reference computes in fp32.  The product's throughput mode ("bf16": what bench.py
# times) feeds bf16 operands to the matrix cores and accumulates in fp32.  To pin THAT
# arithmetic - not just the fp32 one - the oracle can round exactly where the HIP kernels
# round (crank_amd/csrc/stack_kernels.hip, pstack_kernels.hip):
#   * every conv consumes bf16(input) and bf16(w) with w = g*v/||v|| formed in fp32;
#     bias add, accumulation, gate, residual stream and skip sum stay fp32;
#   * every data gradient consumes bf16(output gradient) and bf16(w); every weight gradient
#     consumes bf16(output gradient) and the forward's bf16 input; bias gradients are sums of
#     the bf16 output gradients;
#   * the gate backward reads tanh / sigmoid from bf16 planes;
#   * the out-conv of a gated block receives d(out) = sqrt(.5)*dX: its data gradient rounds
#     bf16(sqrt(.5)*dX), its weight/bias gradients round bf16(dX) and scale afterwards.
# Rounding is round-to-nearest-even (torch's fp32 -> bf16 cast = v_cvt_pk_bf16_f32).
#
# What such an emulation can and cannot pin.  Rounding an activation to bf16 is discontinuous: two evaluations of
# the SAME arithmetic that differ only in the order of the fp32 accumulation (MFMA vs MKL, or MKL with the input
# channels permuted) disagree by ~1e-6 before a rounding, by a whole bf16 ulp (4e-3) on the few elements that sit
# on a rounding boundary after it, and the disagreement compounds through the layers and flips ReLU / LeakyReLU
# derivative masks.  Measured on the 8-layer speaker classifier (CPU against CPU, this file): outputs 4e-3, input
# gradient 5e-2 of scale in the max norm.  So the whole-network comparison is statistical: `accumulate` selects
#   "fp32"           torch's fp32 convolution of the rounded operands,
#   "fp32-permuted"  the same with the input channels summed in another order,
#   "fp64"           products and sums in float64 - the arithmetic's exact value before the next rounding,
# and a kernel is accepted when it lies as close to the "fp64" result as the two fp32 evaluations do
# (tests/test_gpu_nets.py).  Layer by layer, on identical operands, the agreement is fp32-accumulation tight.
_EMULATE_BF16 = False
_ACCUMULATE = "fp32"


@contextlib.contextmanager
def bf16_emulation(on=True, accumulate="fp32"):
    global _EMULATE_BF16, _ACCUMULATE
    assert accumulate in ("fp32", "fp32-permuted", "fp64")
    old = (_EMULATE_BF16, _ACCUMULATE)
    _EMULATE_BF16, _ACCUMULATE = bool(on), accumulate
    try:
        yield
    finally:
        _EMULATE_BF16, _ACCUMULATE = old


def _conv_acc(x, w, b, padding, dilation):
    if _ACCUMULATE == "fp64":
        return F.conv1d(x.double(), w.double(), None if b is None else b.double(), padding=padding, dilation=dilation).float()
    if _ACCUMULATE == "fp32-permuted":
        perm = torch.randperm(x.shape[1], generator=torch.Generator().manual_seed(x.shape[1]))
        return F.conv1d(x[:, perm].contiguous(), w[:, perm].contiguous(), b, padding=padding, dilation=dilation)
    return F.conv1d(x, w, b, padding=padding, dilation=dilation)


def _conv_dx(xshape, w, g, padding, dilation):
    if _ACCUMULATE == "fp64":
        return torch.nn.grad.conv1d_input(xshape, w.double(), g.double(), padding=padding, dilation=dilation).float()
    if _ACCUMULATE == "fp32-permuted":
        perm = torch.randperm(g.shape[1], generator=torch.Generator().manual_seed(g.shape[1] + 1))
        return torch.nn.grad.conv1d_input(xshape, w[perm].contiguous(), g[:, perm].contiguous(), padding=padding, dilation=dilation)
    return torch.nn.grad.conv1d_input(xshape, w, g, padding=padding, dilation=dilation)


def _conv_dw(x, wshape, g, padding, dilation):
    if _ACCUMULATE == "fp64":
        return torch.nn.grad.conv1d_weight(x.double(), wshape, g.double(), padding=padding, dilation=dilation).float()
    if _ACCUMULATE == "fp32-permuted":  # another order of the sum over frames: batch entries reversed
        return torch.nn.grad.conv1d_weight(x.flip(0).contiguous(), wshape, g.flip(0).contiguous(), padding=padding, dilation=dilation)
    return torch.nn.grad.conv1d_weight(x, wshape, g, padding=padding, dilation=dilation)


def bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float32)


class _EmuConv1d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, padding, dilation, gunscale):
        xb, wb = bf16_round(x), bf16_round(w)
        ctx.save_for_backward(xb, wb)
        ctx.geom = (padding, dilation, float(gunscale), b is not None, tuple(x.shape), tuple(w.shape))
        return _conv_acc(xb, wb, b, padding, dilation)

    @staticmethod
    def backward(ctx, dy):
        xb, wb = ctx.saved_tensors
        padding, dilation, gs, has_b, xshape, wshape = ctx.geom
        gb = bf16_round(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _conv_dx(xshape, wb, gb, padding, dilation)
        gw = gb if gs == 1.0 else bf16_round(dy / gs)
        if ctx.needs_input_grad[1]:
            dw = _conv_dw(xb, wshape, gw, padding, dilation) * gs
        if has_b and ctx.needs_input_grad[2]:
            db = (gw.double().sum(dim=(0, 2)).float() if _ACCUMULATE == "fp64" else gw.sum(dim=(0, 2))) * gs
        return dx, dw, db, None, None, None


class _EmuGate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xa, xb):
        t, s = torch.tanh(xa), torch.sigmoid(xb)
        ctx.save_for_backward(bf16_round(t), bf16_round(s))
        return t * s

    @staticmethod
    def backward(ctx, dz):
        t, s = ctx.saved_tensors
        return dz * s * (1.0 - t * t), dz * t * s * (1.0 - s)


class Conv1d(nn.Conv1d):
    """A.0: kaiming-normal(relu) weight, zero bias."""

    grad_unscale = 1.0  # see bf16_emulation: out-conv of a gated block

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)

    def reset_parameters(self):
        nn.init.kaiming_normal_(self.weight, nonlinearity="relu")
        if self.bias is not None:
            nn.init.constant_(self.bias, 0.0)

    def forward(self, x):
        if _EMULATE_BF16:
            return _EmuConv1d.apply(x, self.weight, self.bias, self.padding[0], self.dilation[0], self.grad_unscale)
        return super().forward(x)


class Conv1d1x1(Conv1d):
    def __init__(self, in_channels, out_channels, bias):
        super().__init__(
            in_channels, out_channels, kernel_size=1, padding=0, dilation=1, bias=bias
        )


class ResidualBlock(nn.Module):
    """A.2 gated residual block."""

    def __init__(
        self,
        kernel_size=3,
        residual_channels=64,
        gate_channels=128,
        skip_channels=64,
        aux_channels=80,
        dropout=0.0,
        dilation=1,
        bias=True,
        use_causal_conv=False,
    ):
        super().__init__()
        self.dropout = dropout
        if use_causal_conv:
            padding = (kernel_size - 1) * dilation
        else:
            assert (kernel_size - 1) % 2 == 0, "Not support even number kernel size."
            padding = (kernel_size - 1) // 2 * dilation
        self.use_causal_conv = use_causal_conv
        self.conv = Conv1d(
            residual_channels,
            gate_channels,
            kernel_size,
            padding=padding,
            dilation=dilation,
            bias=bias,
        )
        if aux_channels > 0:
            self.conv1x1_aux = Conv1d1x1(aux_channels, gate_channels, bias=False)
        else:
            self.conv1x1_aux = None
        gate_out_channels = gate_channels // 2
        self.conv1x1_out = Conv1d1x1(gate_out_channels, residual_channels, bias=bias)
        self.conv1x1_out.grad_unscale = math.sqrt(0.5)
        self.conv1x1_skip = Conv1d1x1(gate_out_channels, skip_channels, bias=bias)

    def forward(self, x, c):
        residual = x
        x = F.dropout(x, p=self.dropout, training=self.training)
        x = self.conv(x)
        x = x[:, :, : residual.size(-1)] if self.use_causal_conv else x
        splitdim = 1
        xa, xb = x.split(x.size(splitdim) // 2, dim=splitdim)
        if c is not None:
            assert self.conv1x1_aux is not None
            c = self.conv1x1_aux(c)
            ca, cb = c.split(c.size(splitdim) // 2, dim=splitdim)
            xa, xb = xa + ca, xb + cb
        x = _EmuGate.apply(xa, xb) if _EMULATE_BF16 else torch.tanh(xa) * torch.sigmoid(xb)
        s = self.conv1x1_skip(x)
        x = (self.conv1x1_out(x) + residual) * math.sqrt(0.5)
        return x, s


def _apply_weight_norm(module):
    def _f(m):
        if isinstance(m, (nn.Conv1d, nn.Conv2d)):
            nn.utils.weight_norm(m)

    module.apply(_f)


def _remove_weight_norm(module):
    def _f(m):
        try:
            nn.utils.remove_weight_norm(m)
        except ValueError:
            return

    module.apply(_f)


class ParallelWaveGANGenerator(nn.Module):
    """A.1 (only the configuration crank uses: no upsampling network)."""

    def __init__(
        self,
        in_channels=1,
        out_channels=1,
        kernel_size=3,
        layers=30,
        stacks=3,
        residual_channels=64,
        gate_channels=128,
        skip_channels=64,
        aux_channels=80,
        aux_context_window=2,
        dropout=0.0,
        bias=True,
        use_weight_norm=True,
        use_causal_conv=False,
        upsample_conditional_features=True,
        upsample_net="ConvInUpsampleNetwork",
        upsample_params=None,
    ):
        super().__init__()
        if upsample_conditional_features:
            raise NotImplementedError(
                "oracle restates only upsample_conditional_features=False "
                "(crank/net/module/vqvae2.py:252,271)"
            )
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.aux_channels = aux_channels
        self.layers = layers
        self.stacks = stacks
        self.kernel_size = kernel_size
        assert layers % stacks == 0
        layers_per_stack = layers // stacks
        self.first_conv = Conv1d1x1(in_channels, residual_channels, bias=True)
        self.upsample_net = None
        self.conv_layers = nn.ModuleList()
        for layer in range(layers):
            dilation = 2 ** (layer % layers_per_stack)
            self.conv_layers.append(
                ResidualBlock(
                    kernel_size=kernel_size,
                    residual_channels=residual_channels,
                    gate_channels=gate_channels,
                    skip_channels=skip_channels,
                    aux_channels=aux_channels,
                    dilation=dilation,
                    dropout=dropout,
                    bias=bias,
                    use_causal_conv=use_causal_conv,
                )
            )
        self.last_conv_layers = nn.ModuleList(
            [
                nn.ReLU(inplace=True),
                Conv1d1x1(skip_channels, skip_channels, bias=True),
                nn.ReLU(inplace=True),
                Conv1d1x1(skip_channels, out_channels, bias=True),
            ]
        )
        if use_weight_norm:
            _apply_weight_norm(self)

    def forward(self, x, c):
        x = self.first_conv(x)
        skips = 0
        for f in self.conv_layers:
            x, h = f(x, c)
            skips = skips + h
        skips = skips * math.sqrt(1.0 / len(self.conv_layers))
        x = skips
        for f in self.last_conv_layers:
            x = f(x)
        return x

    def remove_weight_norm(self):
        _remove_weight_norm(self)

    @property
    def receptive_field_size(self):
        layers_per_cycle = self.layers // self.stacks
        dilations = [2 ** (i % layers_per_cycle) for i in range(self.layers)]
        return (self.kernel_size - 1) * sum(dilations) + 1


class ParallelWaveGANDiscriminator(nn.Module):
    """A.4 plain dilated conv stack + LeakyReLU."""

    def __init__(
        self,
        in_channels=1,
        out_channels=1,
        kernel_size=3,
        layers=10,
        conv_channels=64,
        dilation_factor=1,
        nonlinear_activation="LeakyReLU",
        nonlinear_activation_params={"negative_slope": 0.2},
        bias=True,
        use_weight_norm=True,
    ):
        super().__init__()
        assert (kernel_size - 1) % 2 == 0, "Not support even number kernel size."
        assert dilation_factor > 0, "Dilation factor must be > 0."
        self.conv_layers = nn.ModuleList()
        conv_in_channels = in_channels
        for i in range(layers - 1):
            if i == 0:
                dilation = 1
            else:
                dilation = i if dilation_factor == 1 else dilation_factor ** i
                conv_in_channels = conv_channels
            padding = (kernel_size - 1) // 2 * dilation
            self.conv_layers += [
                Conv1d(
                    conv_in_channels,
                    conv_channels,
                    kernel_size=kernel_size,
                    padding=padding,
                    dilation=dilation,
                    bias=bias,
                ),
                getattr(nn, nonlinear_activation)(
                    inplace=True, **nonlinear_activation_params
                ),
            ]
        padding = (kernel_size - 1) // 2
        self.conv_layers += [
            Conv1d(
                conv_in_channels if layers == 1 else conv_channels,
                out_channels,
                kernel_size=kernel_size,
                padding=padding,
                bias=bias,
            )
        ]
        if use_weight_norm:
            _apply_weight_norm(self)

    def forward(self, x):
        for f in self.conv_layers:
            x = f(x)
        return x

    def remove_weight_norm(self):
        _remove_weight_norm(self)


class ResidualParallelWaveGANDiscriminator(nn.Module):
    """A.3 gated-residual discriminator (no aux input)."""

    def __init__(
        self,
        in_channels=1,
        out_channels=1,
        kernel_size=3,
        layers=30,
        stacks=3,
        residual_channels=64,
        gate_channels=128,
        skip_channels=64,
        dropout=0.0,
        bias=True,
        use_weight_norm=True,
        use_causal_conv=False,
        nonlinear_activation="LeakyReLU",
        nonlinear_activation_params={"negative_slope": 0.2},
    ):
        super().__init__()
        assert (kernel_size - 1) % 2 == 0, "Not support even number kernel size."
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.layers = layers
        self.stacks = stacks
        self.kernel_size = kernel_size
        assert layers % stacks == 0
        layers_per_stack = layers // stacks
        self.first_conv = nn.Sequential(
            Conv1d1x1(in_channels, residual_channels, bias=True),
            getattr(nn, nonlinear_activation)(
                inplace=True, **nonlinear_activation_params
            ),
        )
        self.conv_layers = nn.ModuleList()
        for layer in range(layers):
            dilation = 2 ** (layer % layers_per_stack)
            self.conv_layers.append(
                ResidualBlock(
                    kernel_size=kernel_size,
                    residual_channels=residual_channels,
                    gate_channels=gate_channels,
                    skip_channels=skip_channels,
                    aux_channels=-1,
                    dilation=dilation,
                    dropout=dropout,
                    bias=bias,
                    use_causal_conv=use_causal_conv,
                )
            )
        self.last_conv_layers = nn.ModuleList(
            [
                getattr(nn, nonlinear_activation)(
                    inplace=True, **nonlinear_activation_params
                ),
                Conv1d1x1(skip_channels, skip_channels, bias=True),
                getattr(nn, nonlinear_activation)(
                    inplace=True, **nonlinear_activation_params
                ),
                Conv1d1x1(skip_channels, out_channels, bias=True),
            ]
        )
        if use_weight_norm:
            _apply_weight_norm(self)

    def forward(self, x):
        x = self.first_conv(x)
        skips = 0
        for f in self.conv_layers:
            x, h = f(x, None)
            skips = skips + h
        skips = skips * math.sqrt(1.0 / len(self.conv_layers))
        x = skips
        for f in self.last_conv_layers:
            x = f(x)
        return x

    def remove_weight_norm(self):
        _remove_weight_norm(self)
