"""Flat-parameter model base.

Each of the step's models (G, D, C, SPKRADV) keeps ALL its trainable tensors in one
flat fp32 block (`flat`) with a matching flat gradient block (`grad_flat`):

* the HIP stacks read weight_g / weight_v / bias straight from the block by offset,
* one Adam launch updates a whole model, one RCCL all-reduce moves a whole model's
  gradients,
* `state_dict()` / `load_state_dict()` expose the reference's key names and shapes
   as views, so reference checkpoints
  (crank/net/trainer/basetrainer.py:131-140, crank/bin/train.py:134-142) load as is.

Parameter gradients are written into `grad_flat` by the backward kernels themselves
(autograd is used for activations only), so `zero_grad()` here is a memset and
`optimizer.step()` never walks a parameter list.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

PWG_ROLE_NAMES = {1: "conv", 2: "conv1x1_aux", 3: "conv1x1_out", 4: "conv1x1_skip"}


def net_keys(kind, convs):
    """Reference state-dict names for the convs of one stack, in table order.
    convs: list of (cout, cin, k, off_b, off_g, off_v, dilation, role, layer)."""
    names = []
    for (cout, cin, k, off_b, off_g, off_v, dil, role, layer) in convs:
        if role == 0:
            base = "first_conv" if kind == 0 else "first_conv.0"
        elif role in PWG_ROLE_NAMES:
            base = f"conv_layers.{layer}.{PWG_ROLE_NAMES[role]}"
        elif role == 5:
            base = "last_conv_layers.1"
        elif role == 6:
            base = "last_conv_layers.3"
        elif role == 7:
            base = f"conv_layers.{2 * layer}"
        else:
            raise ValueError(role)
        if off_b >= 0:
            names.append((base + ".bias", off_b, (cout,)))
        names.append((base + ".weight_g", off_g, (cout, 1, 1)))
        names.append((base + ".weight_v", off_v, (cout, cin, k)))
    return names


def python_conv_table(kind, in_ch, out_ch, kernel_size, layers, stacks=1, aux_ch=0, conv_ch=64, use_bias=True):
    """Pure-Python mirror of the library's parameter layout (crk_net_conv_info), used to
    check key names / shapes on machines without a GPU."""
    convs, off = [], 0

    def add(role, layer, cout, cin, k, dil, bias):
        nonlocal off
        ob = -1
        if bias:
            ob = off
            off += cout
        og = off
        off += cout
        ov = off
        off += cout * cin * k
        convs.append((cout, cin, k, ob, og, ov, dil, role, layer))

    if kind in (0, 1):
        lps = layers // stacks
        add(0, -1, 64, in_ch, 1, 1, True)
        for l in range(layers):
            add(1, l, 128, 64, kernel_size, 2 ** (l % lps), use_bias)
            if aux_ch > 0:
                add(2, l, 128, aux_ch, 1, 1, False)
            add(3, l, 64, 64, 1, 1, use_bias)
            add(4, l, 64, 64, 1, 1, use_bias)
        add(5, -1, 64, 64, 1, 1, True)
        add(6, -1, out_ch, 64, 1, 1, True)
    else:
        cin = in_ch
        for i in range(layers - 1):
            add(7, i, conv_ch, cin, kernel_size, 1 if i == 0 else i, use_bias)
            cin = conv_ch
        add(7, layers - 1, out_ch, cin, kernel_size, 1, use_bias)
    return convs, off


class FlatModel(nn.Module):
    """Base: subclasses call `_alloc(entries, buffers, device)` once."""

    def __init__(self):
        super().__init__()
        self.version = 1
        self.skip_param_grads = False
        self._entries = []  # (key, offset, shape)
        self._bufs = OrderedDict()

    def _alloc(self, entries, n_total, device):
        self._entries = list(entries)
        self.flat = nn.Parameter(torch.zeros(n_total, device=device, dtype=torch.float32))
        self.grad_flat = torch.zeros(n_total, device=device, dtype=torch.float32)
        self.flat.grad = self.grad_flat
        # True while grad_flat is known to be all zero: every backward of this package that writes parameter gradients
        # clears the flag (code that writes grad_flat by other means must do the same)
        self.grads_clean = True
        self._keepalive = []  # workspaces the deferred weight-gradient launch of finish_grads() still reads

    def view(self, key):
        for k, off, shp in self._entries:
            if k == key:
                return self.flat.data[off: off + int(np.prod(shp))].view(shp)
        raise KeyError(key)

    def grad_view(self, key):
        from ... import ops

        for k, off, shp in self._entries:
            if k == key:
                return self.grad_flat[off: off + int(np.prod(shp))].view(shp)
        raise KeyError(key)

    def offset_of(self, key):
        for k, off, _ in self._entries:
            if k == key:
                return off
        raise KeyError(key)

    # per-step bookkeeping flags: plain Python values that nn.Module.__setattr__ would run its parameter / buffer / module
    # registry checks for (~4 us each, a few dozen times per step)
    _PLAIN = frozenset(("version", "grads_clean", "defer_wnorm", "skip_param_grads", "_wnorm_pending", "_keepalive",
                        "_commits", "training", "codebook_epoch", "_trained_codebooks"))

    def __setattr__(self, name, value):
        if name in self._PLAIN:
            object.__setattr__(self, name, value)
        else:
            super().__setattr__(name, value)

    def touch(self, by_optimizer=False):
        """Call after any in-place parameter change (optimizer step, checkpoint load).  by_optimizer: the change is an
        optimizer step (the generator keeps state that such a step cannot invalidate: VQVAE2.touch)."""
        self.version += 1

    # ---- the model's conv stacks as a group (one launch for all of them instead of one each) ----
    def register_net(self, net, base):
        if not hasattr(self, "_nets"):
            self._nets = []
        self._nets.append((net, base))

    def finish_grads(self):
        """The weight-norm backward the stacks' backward passes left pending (``defer_wnorm``), one launch for all of
        them.  Anything that reads ``grad_flat`` comes after this."""
        if getattr(self, "_wnorm_pending", False):
            from ... import ops

            ops.nets_wnorm_bwd([n for n, _ in self._nets])
            self._wnorm_pending = False
        self._keepalive = []

    def prepare_nets(self, bump_step=None):
        """Weight preparation of every stack for the current parameters in one launch (each stack would otherwise
        prepare itself, a launch each, on its next forward).  bump_step: the step count of an optimizer whose
        ``step(defer_bump=True)`` left it to this launch."""
        from ... import ops

        nets = getattr(self, "_nets", None) or []
        ops.nets_prepare([n for n, _ in nets], [self.flat.data_ptr() + 4 * b for _, b in nets], self.version, bump_step)

    def zero_grad(self, set_to_none=False):
        from ... import ops

        self.finish_grads()
        if getattr(self, "grads_clean", False):
            return  # zeroed by the optimizer step that consumed it (crk_adam_step clear_grads) and not written since
        self.grad_flat.zero_()
        self.flat.grad = self.grad_flat
        self.grads_clean = True

    # ---- reference-compatible checkpoints ----
    def state_dict(self, *args, **kwargs):
        sd = OrderedDict()
        for k, off, shp in self._entries:
            sd[k] = self.flat.data[off: off + int(np.prod(shp))].view(shp).clone()
        for k, b in self._bufs.items():
            sd[k] = b.clone()
        return sd

    def load_state_dict(self, sd, strict=True):
        mine = {k for k, _, _ in self._entries} | set(self._bufs)
        # constants that are rebuilt from the configuration at construction (the on-the-fly mel layer's basis and scaler
        # statistics): the reference's checkpoints carry them, older crank_amd checkpoints do not - either loads
        optional = {k for k in self._bufs if k.startswith("preprocess_layer.")}
        missing, unexpected = mine - set(sd) - optional, set(sd) - mine
        if strict and (missing or unexpected):
            raise RuntimeError(f"state_dict mismatch: missing {sorted(missing)[:5]} unexpected {sorted(unexpected)[:5]}")
        with torch.no_grad():
            for k, off, shp in self._entries:
                if k in sd:
                    self.flat.data[off: off + int(np.prod(shp))].copy_(
                        torch.as_tensor(sd[k]).to(self.flat.device, torch.float32).reshape(-1))
            for k, b in self._bufs.items():
                if k in sd:
                    b.copy_(torch.as_tensor(sd[k]).to(b.device, b.dtype).reshape(b.shape))
        self.touch()

    def to(self, *args, **kwargs):
        dev = kwargs.get("device", args[0] if args else None)
        if dev is not None and torch.device(dev).type != self.flat.device.type:
            raise RuntimeError("crank_amd models live on the GPU they were built on (no CPU path)")
        return self
