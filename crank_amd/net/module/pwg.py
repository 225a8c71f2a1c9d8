"""HIP-backed convolutional stacks with the constructor surface of the third-party
`parallel_wavegan` classes crank instantiates (call sites
crank/net/module/vqvae2.py:237-273, crank/net/module/spkradv.py:49-60,
crank/bin/train.py:78-128).  The arithmetic is in libcrank_hip.so (net.hip).
"""
import torch

from ... import ops
from .flat import FlatModel, net_keys

KIND_GENERATOR, KIND_RESIDUAL_D, KIND_PLAIN = 0, 1, 2


class HipStack:
    """One stack living at `base` inside its owner's flat parameter block."""

    def __init__(self, kind, in_channels, out_channels, kernel_size, layers, stacks=1, aux_channels=0,
                 conv_channels=64, use_causal_conv=False, bias=True, negative_slope=0.2, dropout=0.0):
        self.kind = kind
        self.layers, self.stacks, self.kernel_size = layers, stacks, kernel_size
        self.net = ops.HipNet(
            kind=kind, in_ch=in_channels, out_ch=out_channels, kernel_size=kernel_size, layers=layers,
            stacks=max(stacks, 1), res_ch=64, gate_ch=128, skip_ch=64, aux_ch=max(aux_channels, 0),
            conv_ch=conv_channels, causal=int(bool(use_causal_conv)), use_bias=int(bool(bias)),
            slope=float(negative_slope), dropout=float(dropout),
        )
        self.owner, self.base = None, 0

    @property
    def n_params(self):
        return self.net.n_params

    def entries(self, prefix, base):
        return [(prefix + k, base + off, shp) for (k, off, shp) in net_keys(self.kind, self.net.convs)]

    def bind(self, owner, base):
        self.owner, self.base = owner, base
        owner.register_net(self.net, base)

    @torch.no_grad()
    def init_parameters(self):
        """PWG init: kaiming-normal(relu) v, g = ||v||, bias 0."""
        flat = self.owner.flat.data
        for (cout, cin, k, off_b, off_g, off_v, dil, role, layer) in self.net.convs:
            v = torch.randn(cout, cin * k, device=flat.device) * float((2.0 / (cin * k)) ** 0.5)
            flat[self.base + off_v: self.base + off_v + cout * cin * k] = v.reshape(-1)
            flat[self.base + off_g: self.base + off_g + cout] = v.norm(dim=1)
            if off_b >= 0:
                flat[self.base + off_b: self.base + off_b + cout] = 0.0

    @property
    def receptive_field_size(self):
        lpc = self.layers // self.stacks
        return (self.kernel_size - 1) * sum(2 ** (i % lpc) for i in range(self.layers)) + 1

    def __call__(self, x, c=None, dx_scale=1.0, out=None):
        """x: (B,T,in) channel-last; c: (B,T,aux) or None -> (B,T,out); out = (buffer, column): see ops.net_apply."""
        return ops.net_apply(self.net, self.owner, self.base, x, c, dx_scale, out=out)

    def ce(self, x, target, dx_scale=1.0, ignore_index=-100):
        """Mean cross entropy of this stack's output against `target` (B,T) as one op (ops.net_ce): what
        nn.CrossEntropyLoss(ignore_index)(stack(x).reshape(-1, classes), target.reshape(-1)) returns."""
        return ops.net_ce(self.net, self.owner, self.base, x, target, dx_scale, ignore_index)


class _StandaloneStack(FlatModel):
    """A model that is exactly one stack (speaker classifier C, discriminator D).
    Called like the reference calls them: (B,C,T) in, (B,C_out,T) out."""

    def __init__(self, stack, device):
        super().__init__()
        self.stack = stack
        self._alloc(stack.entries("", 0), stack.n_params, device)
        stack.bind(self, 0)
        stack.init_parameters()

    def forward(self, x):
        y = self.stack(x.transpose(1, 2))
        return y.transpose(1, 2)

    def forward_ce(self, x, target, ignore_index=-100):
        """cross entropy of forward(x) (B,C_out,T) against target (B,T) as one op (not in the reference: its trainers
        compose the two, trainer_vqvae.py:186-198); stacks without dropout only."""
        return self.stack.ce(x.transpose(1, 2), target, ignore_index=ignore_index)


class ParallelWaveGANDiscriminator(_StandaloneStack):
    def __init__(self, in_channels=1, out_channels=1, kernel_size=3, layers=10, conv_channels=64, dilation_factor=1,
                 nonlinear_activation="LeakyReLU", nonlinear_activation_params={"negative_slope": 0.2}, bias=True,
                 use_weight_norm=True, device="cuda"):
        if dilation_factor != 1 or nonlinear_activation != "LeakyReLU" or not use_weight_norm:
            raise NotImplementedError("only the configuration crank uses is implemented "
                                      "(dilation_factor=1, LeakyReLU, weight norm)")
        stack = HipStack(KIND_PLAIN, in_channels, out_channels, kernel_size, layers, conv_channels=conv_channels,
                         bias=bias, negative_slope=nonlinear_activation_params.get("negative_slope", 0.2))
        super().__init__(stack, device)


class ResidualParallelWaveGANDiscriminator(_StandaloneStack):
    def __init__(self, in_channels=1, out_channels=1, kernel_size=3, layers=30, stacks=3, residual_channels=64,
                 gate_channels=128, skip_channels=64, dropout=0.0, bias=True, use_weight_norm=True,
                 use_causal_conv=False, nonlinear_activation="LeakyReLU",
                 nonlinear_activation_params={"negative_slope": 0.2}, device="cuda"):
        if (residual_channels, gate_channels, skip_channels) != (64, 128, 64) or not use_weight_norm:
            raise NotImplementedError("channel widths are fixed at 64/128/64 as in crank/bin/train.py:108-118")
        stack = HipStack(KIND_RESIDUAL_D, in_channels, out_channels, kernel_size, layers, stacks=stacks,
                         use_causal_conv=use_causal_conv, bias=bias, dropout=dropout,
                         negative_slope=nonlinear_activation_params.get("negative_slope", 0.2))
        super().__init__(stack, device)
