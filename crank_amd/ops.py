"""torch.autograd bindings over the C ABI (plumbing only: device memory, streams,
autograd graph).  Every op here launches HIP kernels from libcrank_hip.so; there is no
alternative implementation.
"""
import ctypes

import torch

from . import _lib
from .config import cfg
from ._lib import check, ptr, stream_ptr

# "bf16": single bf16 MFMA per product (fast path, what bench.py measures)
# "bf16x3": hi/lo split operands, three MFMAs per product (~fp32 accuracy; parity tests)
# "bf16x3f": the forward passes in bf16x3, the backward passes in plain bf16 (an experiment: loss VALUES at fp32
#            accuracy on given parameters, gradients at bf16 accuracy)
_PRECISION = cfg.precision


def set_precision(name):
    global _PRECISION
    if name not in ("bf16", "bf16x3", "bf16x3f"):
        raise ValueError(f"unknown precision {name!r} (bf16 | bf16x3 | bf16x3f)")
    _PRECISION = name


def get_precision():
    return _PRECISION


# flags of crk_net_forward / crk_net_backward (include/crank_hip.h)
CRK_FLAG_PRECISE = 1
CRK_FLAG_NO_PARAM_GRAD = 2
CRK_FLAG_NO_SAVE = 4
CRK_FLAG_DEFER_WNORM = 8
CRK_FLAG_SEED_ON_DEVICE = 16
CRK_FLAG_FWD_PRECISE = 32
CRK_FLAG_BWD_PLAIN = 64


def _flags(skip_param_grads=False, no_save=False, precision=None, defer_wnorm=False, backward=False):
    prec = precision or _PRECISION
    if prec == "bf16x3f":  # forward: precise; backward: plain arithmetic on the planes a precise forward wrote
        pbits = CRK_FLAG_FWD_PRECISE if backward else (CRK_FLAG_PRECISE | CRK_FLAG_BWD_PLAIN)
    else:
        pbits = CRK_FLAG_PRECISE if prec == "bf16x3" else 0
    return (pbits | (CRK_FLAG_NO_PARAM_GRAD if skip_param_grads else 0) | (CRK_FLAG_NO_SAVE if no_save else 0)
            | (CRK_FLAG_DEFER_WNORM if defer_wnorm else 0))


class _ScalarArena:
    """One small device buffer per training step out of which the loss ops take their result scalars (means, counts):
    the step's loss values then sit side by side and go to the host as ONE copy of the buffer, without the launch that
    stacked them.  Nothing here is ever written by a torch in-place op (the slices share a version counter)."""

    def __init__(self, device, n=256):
        self.buf = torch.empty(n, device=device, dtype=torch.float32)
        self.pos = 0


_ARENA = None


def begin_scalar_arena(device):
    """Called by a trainer at the start of a step (CUDA only); None switches the arena off."""
    global _ARENA
    dev = torch.device(device) if device is not None else None
    _ARENA = _ScalarArena(dev) if (dev is not None and dev.type == "cuda") else None
    return _ARENA


def scalar_arena():
    return _ARENA


def _scalars(n, device):
    """n fp32 result scalars on `device`: a 16-byte aligned slice of the step's arena when one is open, else a new tensor."""
    a = _ARENA
    if a is not None and a.buf.device == device and a.pos + n <= a.buf.numel():
        t = a.buf[a.pos: a.pos + n]
        a.pos += (n + 3) & ~3
        return t
    return torch.empty(n, device=device, dtype=torch.float32)


def _rows(t):
    """View a (..., C) fp32 tensor as frames x channels with a row stride; returns
    (tensor_to_keep_alive, ld).  Copies only if the layout cannot be expressed."""
    assert t.dtype == torch.float32, t.dtype
    if t.dim() == 3:
        B, T, C = t.shape
        if t.stride(2) == 1 and (B == 1 or t.stride(0) == T * t.stride(1)) and t.stride(1) >= C:
            return t, t.stride(1)
    elif t.dim() == 2:
        if t.stride(1) == 1 and t.stride(0) >= t.size(1):
            return t, t.stride(0)
    t = t.contiguous()
    return t, t.size(-1)


class HipNet:
    """One convolutional stack handle (kinds: see include/crank_hip.h)."""

    def __init__(self, **desc):
        L = _lib.lib()
        self.desc = _lib.NetDesc(**desc)
        self.handle = L.crk_net_create(ctypes.byref(self.desc))
        if not self.handle:
            raise RuntimeError(f"crk_net_create failed for {desc}")
        self.n_params = L.crk_net_param_count(self.handle)
        self.convs = []
        buf = (ctypes.c_longlong * 9)()
        for i in range(L.crk_net_conv_count(self.handle)):
            check(L.crk_net_conv_info(self.handle, i, buf), "crk_net_conv_info")
            self.convs.append(tuple(int(v) for v in buf))
        self.in_ch, self.out_ch = desc["in_ch"], desc["out_ch"]
        self.aux_ch = desc.get("aux_ch", 0)
        self.dropout = float(desc.get("dropout", 0.0))
        self._saved_bytes = {}
        # dropout seeds live on the device: `seed_state` is advanced by a one-thread launch per forward call
        # (crk_seed_next), so nothing about a call is a host value - no .item() per call, and a step with dropout can be
        # captured in a HIP graph and still draw fresh masks on every replay.  The start value comes from torch's CPU
        # generator (torch.manual_seed before building the model makes runs repeatable); reseed() pins it later.
        self.seed_state = None
        if self.dropout > 0:
            self.reseed(int(torch.randint(0, 2 ** 62, (1,)).item()))

    def reseed(self, value):
        """Restart this net's dropout-seed sequence (tests pin masks with it, like torch.manual_seed would for torch's
        dropout); under data parallelism every rank offsets its sequence by its rank."""
        from . import parallel

        v = (int(value) + 0x632BE59BD9B4E019 * parallel.rank()) & 0x3FFFFFFFFFFFFFFF
        if self.seed_state is None:
            self.seed_state = torch.empty(1, dtype=torch.int64, device="cuda")
        self.seed_state.fill_(v)

    def next_seed(self):
        """A fresh device-resident seed for one forward call (and its backward)."""
        out = torch.empty(1, dtype=torch.int64, device=self.seed_state.device)
        check(_lib.lib().crk_seed_next(ptr(self.seed_state), ptr(out), stream_ptr()), "crk_seed_next")
        return out

    def saved_bytes(self, B, T):
        """crk_net_saved_bytes, remembered per batch shape.  A shape seen for the first time is reserved here
        (crk_net_reserve: the handle's gradient planes, partial sums and tables - the compute entry points never allocate);
        that happens in the eager steps in front of a graph capture, never inside one."""
        key = (B, T, _PRECISION)
        v = self._saved_bytes.get(key)
        if v is None:
            check(_lib.lib().crk_net_reserve(self.handle, B, T), "crk_net_reserve")
            v = self._saved_bytes[key] = _lib.lib().crk_net_saved_bytes(self.handle, B, T)
        return v

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.lib().crk_net_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class _NetFn(torch.autograd.Function):
    """y = net(x, c).  `owner` supplies the flat parameter / gradient blocks."""

    @staticmethod
    def forward(ctx, x, c, flat, net, owner, offset, dx_scale, no_save=False, ybuf=None, ycol=0):
        L = _lib.lib()
        B, T = x.shape[0], x.shape[1]
        xk, ldx = _rows(x)
        ck, ldc = (None, 0) if c is None else _rows(c)
        if ybuf is None:
            y = torch.empty(B, T, net.out_ch, device=x.device, dtype=torch.float32)
        else:  # the output lands in a column slice of a wider buffer (what a later concatenation would build)
            y = ybuf[..., ycol: ycol + net.out_ch]
        ldy = y.stride(1)
        nbytes = net.saved_bytes(B, T)
        saved = torch.empty(max(nbytes // 4, 1), device=x.device, dtype=torch.float32)
        seed_t = net.next_seed() if net.dropout > 0 else None  # device-resident (CRK_FLAG_SEED_ON_DEVICE)
        params = flat.data_ptr() + 4 * offset
        check(
            L.crk_net_forward(net.handle, params, owner.version, ptr(xk), ldx, ptr(ck), ldc, ptr(y), ldy,
                              ptr(saved), B, T, _flags(no_save=no_save) | (CRK_FLAG_SEED_ON_DEVICE if seed_t is not None else 0), ptr(seed_t),
                              stream_ptr()),
            "crk_net_forward",
        )
        ctx.net, ctx.owner, ctx.offset, ctx.dx_scale, ctx.seed = net, owner, offset, dx_scale, seed_t
        ctx.precision = _PRECISION  # the backward must read the saved planes the way this forward wrote them
        ctx.ld = (ldx, ldc)
        ctx.version = owner.version
        ctx.save_for_backward(xk, ck if ck is not None else torch.empty(0, device=x.device), flat)
        ctx.saved_ws = saved
        ctx.has_c = c is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        net, owner = ctx.net, ctx.owner
        xk, ck, flat = ctx.saved_tensors
        if not ctx.has_c:
            ck = None
        ldx, ldc = ctx.ld
        B, T = xk.shape[0], xk.shape[1]
        dyk, lddy = _rows(dy)
        dx = torch.empty(B, T, net.in_ch, device=dy.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        dc = None
        if ctx.has_c and ctx.needs_input_grad[1]:
            dc = torch.empty(B, T, net.aux_ch, device=dy.device, dtype=torch.float32)
        if ctx.version != owner.version:
            raise RuntimeError("parameters were modified between forward and backward of a crank_amd net")
        skip = owner.skip_param_grads
        defer = (not skip) and getattr(owner, "defer_wnorm", False)
        if not skip:
            owner.grads_clean = False
            if defer:
                owner._wnorm_pending = True
                # the deferred weight gradients of the plain convs read planes of the forward's workspace: it has to
                # outlive this backward call, until owner.finish_grads()
                owner._keepalive.append(ctx.saved_ws)
        params = flat.data_ptr() + 4 * ctx.offset
        grads = owner.grad_flat.data_ptr() + 4 * ctx.offset
        check(
            L.crk_net_backward(net.handle, params, owner.version, grads, ptr(xk), ldx, ptr(ck), ldc, ptr(dyk), lddy,
                               ptr(dx), net.in_ch, float(ctx.dx_scale), ptr(dc), net.aux_ch, ptr(ctx.saved_ws), B, T,
                               _flags(skip, precision=ctx.precision, defer_wnorm=defer, backward=True) | (CRK_FLAG_SEED_ON_DEVICE if ctx.seed is not None else 0),
                               ptr(ctx.seed), stream_ptr()),
            "crk_net_backward",
        )
        return dx, dc, None, None, None, None, None, None, None, None


class _NetCEFn(torch.autograd.Function):
    """loss = cross_entropy(net(x), target) as ONE autograd node: the net's backward reads the unnormalised gradient
    softmax - onehot the loss kernel left and takes the factor upstream gradient / count itself
    (crk_net_backward_scaled) - no scaling launch, no (N, classes) gradient tensor in between.  The trainers' classifier
    and speaker-adversarial losses (trainer_vqvae.py:177-198, :294-315)."""

    @staticmethod
    def forward(ctx, x, flat, net, owner, offset, dx_scale, target, ignore_index, no_save=False):
        L = _lib.lib()
        B, T = x.shape[0], x.shape[1]
        xk, ldx = _rows(x)
        C = net.out_ch
        y = torch.empty(B, T, C, device=x.device, dtype=torch.float32)
        saved = torch.empty(max(net.saved_bytes(B, T) // 4, 1), device=x.device, dtype=torch.float32)
        params = flat.data_ptr() + 4 * offset
        check(L.crk_net_forward(net.handle, params, owner.version, ptr(xk), ldx, None, 0, ptr(y), C, ptr(saved), B, T,
                                _flags(no_save=no_save), 0, stream_ptr()), "crk_net_forward")
        tk = target.reshape(-1).contiguous()
        assert tk.numel() == B * T, (tk.shape, B, T)
        out = _scalars(2, x.device)
        dl = torch.empty(B * T, C, device=x.device, dtype=torch.float32)
        check(L.crk_ce_fwd(ptr(y), C, ptr(tk), B * T, C, int(ignore_index), ptr(out), ptr(dl), ptr(_loss_scratch(x.device)),
                           stream_ptr()), "crk_ce_fwd")
        ctx.net, ctx.owner, ctx.offset, ctx.dx_scale = net, owner, offset, dx_scale
        ctx.precision = _PRECISION
        ctx.ldx, ctx.version = ldx, owner.version
        ctx.save_for_backward(xk, flat, dl, out)
        ctx.saved_ws = saved
        return out[0]

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        net, owner = ctx.net, ctx.owner
        xk, flat, dl, out = ctx.saved_tensors
        B, T = xk.shape[0], xk.shape[1]
        C = net.out_ch
        dx = torch.empty(B, T, net.in_ch, device=g.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        if ctx.version != owner.version:
            raise RuntimeError("parameters were modified between forward and backward of a crank_amd net")
        skip = owner.skip_param_grads
        defer = (not skip) and getattr(owner, "defer_wnorm", False)
        if not skip:
            owner.grads_clean = False
            if defer:
                owner._wnorm_pending = True
                owner._keepalive.append(ctx.saved_ws)
        params = flat.data_ptr() + 4 * ctx.offset
        grads = owner.grad_flat.data_ptr() + 4 * ctx.offset
        gk = g.contiguous().reshape(1)
        flags = _flags(skip, precision=ctx.precision, defer_wnorm=defer, backward=True)
        rc = L.crk_net_backward_scaled(net.handle, params, owner.version, grads, ptr(xk), ctx.ldx, None, 0, ptr(dl), C, ptr(dx),
                                       net.in_ch, float(ctx.dx_scale), None, 0, ptr(ctx.saved_ws), B, T, flags, 0, ptr(gk),
                                       ptr(out), stream_ptr())
        if rc == 3:  # not a fused chain of plain convs: scale, then the plain backward
            res = torch.empty_like(dl)
            check(L.crk_ce_bwd(ptr(dl), B * T, C, ptr(out), ptr(gk), ptr(res), stream_ptr()), "crk_ce_bwd")
            rc = L.crk_net_backward(net.handle, params, owner.version, grads, ptr(xk), ctx.ldx, None, 0, ptr(res), C, ptr(dx),
                                    net.in_ch, float(ctx.dx_scale), None, 0, ptr(ctx.saved_ws), B, T, flags, 0, stream_ptr())
        check(rc, "crk_net_backward_scaled")
        return dx, None, None, None, None, None, None, None, None


def net_ce(net, owner, offset, x, target, dx_scale=1.0, ignore_index=-100):
    """cross_entropy(net(x).reshape(-1, classes), target.reshape(-1)) for a net without conditioning input and dropout."""
    if net.dropout > 0:
        raise ValueError("net_ce: nets with dropout go through net_apply + cross_entropy")
    return _NetCEFn.apply(x, owner.flat, net, owner, offset, dx_scale, target, ignore_index, not torch.is_grad_enabled())


def nets_wnorm_bwd(nets):
    """Pending weight-norm backward of several stacks (backward with ``owner.defer_wnorm``) in one launch."""
    arr = (ctypes.c_void_p * len(nets))(*[n.handle for n in nets])
    check(_lib.lib().crk_nets_wnorm_bwd(len(nets), arr, stream_ptr()), "crk_nets_wnorm_bwd")


def nets_prepare(nets, param_ptrs, version, bump_step=None):
    """Weight preparation of several stacks (parameter blocks at ``param_ptrs``) in one launch; bump_step: an Adam step
    count (adam_step(..., defer_bump=True)) advanced in the same launch."""
    arr = (ctypes.c_void_p * len(nets))(*[n.handle for n in nets])
    par = (ctypes.c_void_p * len(nets))(*param_ptrs)
    check(_lib.lib().crk_nets_prepare(len(nets), arr, par, version, ptr(bump_step), stream_ptr()), "crk_nets_prepare")


def net_apply(net, owner, offset, x, c=None, dx_scale=1.0, out=None):
    """out = (buffer (B,T,W), first column): write the result into that column slice and return the slice."""
    # without autograd nothing will ever read the per-layer activations: tell the library
    ybuf, ycol = out if out is not None else (None, 0)
    return _NetFn.apply(x, c, owner.flat, net, owner, offset, dx_scale, not torch.is_grad_enabled(), ybuf, ycol)


class _AliasCatFn(torch.autograd.Function):
    """torch.cat(xs, -1) for tensors that ARE the adjacent column slices of one buffer (their producers wrote them
    there): returns the buffer, launches nothing; the backward hands out column slices of the incoming gradient."""

    @staticmethod
    def forward(ctx, *xs):
        B, T = xs[0].shape[0], xs[0].shape[1]
        W = sum(x.shape[2] for x in xs)
        ctx.widths = [x.shape[2] for x in xs]
        return xs[0].as_strided((B, T, W), (T * W, W, 1))

    @staticmethod
    def backward(ctx, d):
        out, off = [], 0
        for w in ctx.widths:
            out.append(d[..., off: off + w])
            off += w
        return tuple(out)


def cat_channels(xs):
    """torch.cat(xs, dim=-1); free when the pieces already sit side by side in one buffer."""
    xs = list(xs)
    if len(xs) > 1 and all(x.dim() == 3 and x.dtype == torch.float32 for x in xs):
        B, T = xs[0].shape[0], xs[0].shape[1]
        W = sum(x.shape[2] for x in xs)
        base, off, ok = xs[0].data_ptr(), 0, xs[0].storage_offset() * 4 + B * T * W * 4 <= xs[0].untyped_storage().nbytes()
        for x in xs:
            ok = ok and x.shape[:2] == (B, T) and x.stride() == (T * W, W, 1) and x.data_ptr() == base + 4 * off
            off += x.shape[2]
        if ok:
            return _AliasCatFn.apply(*xs)
    return torch.cat(xs, dim=-1)


# ------------------------------------------------------------------------------------
def vq_image_bytes(K, D):
    """Size of a codebook image (crk_vq_image_bytes); 0: no image for this shape."""
    return int(_lib.lib().crk_vq_image_bytes(int(K), int(D)))


def vq_image_build(codebooks, images):
    """What the split-f16 search derives from a codebook alone, written once (crk_vq_image_build_multi): <= 4 (K, 64)
    codebooks and their image buffers per launch."""
    for i in range(0, len(codebooks), 4):
        cb, im = codebooks[i: i + 4], images[i: i + 4]
        check(_lib.lib().crk_vq_image_build_multi(len(cb), _parr(cb), _iarr([c.shape[0] for c in cb]), int(cb[0].shape[1]),
                                                  _parr(im), stream_ptr()), "crk_vq_image_build_multi")


def _vq_call(xk, ldx, addk, ldadd, codebook, idx, e, qx, mk=None, commit_out=None, image=None):
    """The quantizer launch: fused entry point (input sum and commitment partials inside the search kernel) where the
    shape allows it, the separate kernels otherwise.  Returns the effective input (x, or x + add).  image: the codebook's
    prepared image (vq_image_build), which the caller keeps valid, or None."""
    L = _lib.lib()
    B, T, D = xk.shape
    K = codebook.shape[0]
    ldq = qx.stride(1) if qx is not None else D
    xsum = None
    if addk is not None or commit_out is not None or image is not None:
        xsum = torch.empty(B, T, D, device=xk.device, dtype=torch.float32) if addk is not None else None
        rc = L.crk_vq_forward_fused(ptr(xk), ldx, ptr(addk), ldadd, ptr(xsum), D, ptr(codebook), B * T, D, K, ptr(idx), ptr(e), D,
                                    ptr(qx), ldq, ptr(mk), ptr(commit_out),
                                    ptr(_loss_scratch(xk.device)) if commit_out is not None else None, ptr(image), stream_ptr())
        if rc == 0:
            return (xsum, D) if xsum is not None else (xk, ldx)
        if rc != 3:
            check(rc, "crk_vq_forward_fused")
        if addk is not None:  # unsupported shape: compose
            xsum = xk + addk
            xk, ldx = xsum, D
    check(L.crk_vq_forward(ptr(xk), ldx, ptr(codebook), B * T, D, K, ptr(idx), ptr(e), D, ptr(qx), ldq, stream_ptr()),
          "crk_vq_forward")
    if commit_out is not None:
        check(L.crk_masked_loss_fwd(ptr(xk), ldx, ptr(e), D, 0.0, ptr(mk), B * T, D, 1, ptr(commit_out),
                                    ptr(_loss_scratch(xk.device)), stream_ptr()), "crk_masked_loss_fwd")
    return xk, ldx


class _VQFn(torch.autograd.Function):
    """(e, qx, idx, xin) = quantize(x (+ add) (B,T,D), codebook (K,D)); straight-through backward.  xin: the quantizer's
    input when it is a sum formed inside the op (add given), else None."""

    @staticmethod
    def forward(ctx, x, codebook, owner, cb_offset, qbuf=None, qcol=0, want=3, add=None, image=None):
        xk, ldx = _rows(x)
        addk, ldadd = (None, 0) if add is None else _rows(add)
        B, T, D = xk.shape
        K = codebook.shape[0]
        # want: bit 0 the gathered code vectors e, bit 1 the straight-through value qx (a caller that only needs the
        # indices - the EMA side effect of a forward whose outputs are discarded - saves both writes)
        e = torch.empty(B, T, D, device=x.device, dtype=torch.float32) if want & 1 else None
        qx = None
        if want & 2:
            qx = torch.empty(B, T, D, device=x.device, dtype=torch.float32) if qbuf is None else qbuf[..., qcol: qcol + D]
        idx = torch.empty(B, T, device=x.device, dtype=torch.int64)
        xin, _ = _vq_call(xk, ldx, addk, ldadd, codebook, idx, e, qx, image=image)
        ctx.owner, ctx.cb_offset, ctx.K, ctx.D = owner, cb_offset, K, D
        ctx.has_add = add is not None
        ctx.save_for_backward(idx)
        ctx.mark_non_differentiable(idx)
        ctx.set_materialize_grads(False)
        return e, qx, idx, (xin if add is not None else None)

    @staticmethod
    def backward(ctx, de, dqx, _didx, dxin):
        (idx,) = ctx.saved_tensors
        if de is not None and ctx.owner is not None and not ctx.owner.skip_param_grads:
            # dictionary loss path (ema_flag false): d codebook[k] += sum_{idx==k} de
            ctx.owner.grads_clean = False
            g = ctx.owner.grad_flat[ctx.cb_offset: ctx.cb_offset + ctx.K * ctx.D].view(ctx.K, ctx.D)
            g.index_add_(0, idx.reshape(-1), de.reshape(-1, ctx.D))
        dx = dqx if dxin is None else (dxin if dqx is None else dqx + dxin)
        return dx, None, None, None, None, None, None, (dx if ctx.has_add else None), None


def vq_apply(x, codebook, owner=None, cb_offset=0, qx_out=None, want_e=True, want_qx=True, add=None, image=None):
    """qx_out = (buffer (B,T,W), first column): the straight-through value is written into that column slice.
    want_e / want_qx False: that output is not produced (None).  add: the quantizer's input is x + add; the fourth result
    is that sum (None without add)."""
    qbuf, qcol = qx_out if qx_out is not None else (None, 0)
    r = _VQFn.apply(x, codebook, owner, cb_offset, qbuf, qcol, (1 if want_e else 0) | (2 if want_qx else 0), add, image)
    return r if add is not None else r[:3]


# False: the second consumers of a quantizer's input and output read the tensors themselves and autograd accumulates their
# gradients in launches of its own (the equality test sets it; the values are the same bit for bit)
VQ_JOIN = True
# False: the search kernel derives the codebook's operand planes and norms in every workgroup of every call (rounds 1 - 4)
# instead of copying the image prepared once per codebook update (the equality test sets it)
VQ_IMAGE = True


class _VQCommitFn(torch.autograd.Function):
    """(e, qx, idx, commit, xin, x_alias, qx_alias) with commit = masked mean of (x - e)^2, the commitment loss the trainers
    form from x and e.detach() (trainer_vqvae.py:227-237), x = x (+ add).  Forward: the search kernel forms the sum and the
    loss partials itself (ONE launch + the finishing one; an addition, the search, a loss pass and its finish otherwise).
    x_alias / qx_alias (alias=True, else None) are x and qx again, as outputs of their own: a second consumer of either
    tensor (the speaker-adversarial net takes the encoder outputs, spkradv.py:74-76; the last decoder's concatenation
    takes every stack's qx, vqvae2.py:186-188) reads the alias, and its gradient then arrives HERE instead of in an
    accumulation launch of autograd's - the backward joins the straight-through gradient, the loss gradient and those two
    in ONE launch, each sum the one autograd would have formed.  EMA codebooks only (e carries no gradient)."""

    @staticmethod
    def forward(ctx, x, codebook, mask, qbuf=None, qcol=0, add=None, alias=False, image=None):
        xk, ldx = _rows(x)
        addk, ldadd = (None, 0) if add is None else _rows(add)
        B, T, D = xk.shape
        e = torch.empty(B, T, D, device=x.device, dtype=torch.float32)
        qx = torch.empty(B, T, D, device=x.device, dtype=torch.float32) if qbuf is None else qbuf[..., qcol: qcol + D]
        idx = torch.empty(B, T, device=x.device, dtype=torch.int64)
        mk = None
        if mask is not None:
            mk = mask.reshape(-1).contiguous()
            mk = mk.view(torch.uint8) if mk.dtype == torch.bool else mk.to(torch.uint8)
            assert mk.numel() == B * T, (mk.numel(), B * T)
        out = _scalars(2, x.device)
        xin, ldin = _vq_call(xk, ldx, addk, ldadd, codebook, idx, e, qx, mk, out, image=image)
        ctx.geom = (B, T, D, ldin)
        ctx.has_m = mk is not None
        ctx.has_add = add is not None
        ctx.save_for_backward(xin, e, mk if mk is not None else out, out)
        ctx.mark_non_differentiable(idx, e)
        ctx.set_materialize_grads(False)
        return (e, qx, idx, out[0], (xin if add is not None else None),
                (x.view_as(x) if alias else None), (qx.view_as(qx) if alias else None))

    @staticmethod
    def backward(ctx, _de, dqx, _didx, dcommit, dxin, dxal=None, dqx2=None):
        if dxin is not None:  # the sum is used outside the op too (the cyclic forward hands it to a second decode)
            dqx = dxin if dqx is None else dqx + dxin

        def plus(a, b):
            return b if a is None else (a if b is None else a + b)

        if dcommit is None:
            dsum = plus(dqx, dqx2)
            return plus(dsum, dxal), None, None, None, None, (dsum if ctx.has_add else None), None, None
        L = _lib.lib()
        xk, e, mk, out = ctx.saved_tensors
        B, T, D, ldx = ctx.geom
        mk = mk if ctx.has_m else None
        dx = torch.empty(B, T, D, device=xk.device, dtype=torch.float32)
        g = dcommit.contiguous().reshape(1)
        if dxal is not None or dqx2 is not None:
            # the tensor added to x inside the op takes everything but the second consumer of x itself
            dsum = torch.empty_like(dx) if (ctx.has_add and dxal is not None) else None
            a = [(None, 0) if t is None else _rows(t) for t in (dqx, dqx2, dxal)]
            rc = L.crk_vq_commit_bwd(ptr(xk), ldx, ptr(e), D, ptr(mk), B * T, D, ptr(out), ptr(g), ptr(dx), D, ptr(dsum), D,
                                     ptr(a[0][0]), a[0][1], ptr(a[1][0]), a[1][1], ptr(a[2][0]), a[2][1], stream_ptr())
            if rc == 0:
                return dx, None, None, None, None, ((dx if dsum is None else dsum) if ctx.has_add else None), None, None
            if rc != 3:
                check(rc, "crk_vq_commit_bwd")
            dqx = plus(dqx, dqx2)  # unaligned geometry: separate additions
        addk, ldadd = (None, 0) if dqx is None else _rows(dqx)
        check(L.crk_masked_loss_bwd_acc(ptr(xk), ldx, ptr(e), D, 0.0, ptr(mk), B * T, D, 1, ptr(out), ptr(g), ptr(dx), D,
                                        None, 0, ptr(addk), ldadd, None, stream_ptr()), "crk_masked_loss_bwd_acc")
        return plus(dx, dxal), None, None, None, None, (dx if ctx.has_add else None), None, None


def vq_commit_apply(x, codebook, mask, qx_out=None, add=None, alias=False, image=None):
    """(e, qx, idx, commit[, x + add]); alias=True appends (x_alias, qx_alias) - see _VQCommitFn."""
    qbuf, qcol = qx_out if qx_out is not None else (None, 0)
    r = _VQCommitFn.apply(x, codebook, mask, qbuf, qcol, add, alias, image)
    head = r[:5] if add is not None else r[:4]
    return head + r[5:] if alias else head  # (e, qx, idx, commit[, x + add][, x_alias, qx_alias])


def vq_ema_stats(x, idx, counts, sums):
    """Integer EMA statistics of one quantizer call into caller-owned buffers: counts (K) int32, sums (D*K)
    int64 2^-28 fixed point (crk_vq_ema_stats overwrites both)."""
    L = _lib.lib()
    xk, ldx = _rows(x)
    K, D = counts.numel(), sums.numel() // counts.numel()
    N = idx.numel()
    nbytes = L.crk_vq_ema_scratch_bytes(N, D, K)
    if nbytes < 0:
        raise ValueError(f"vq_ema_stats: unsupported codebook size K={K}")
    scratch = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    check(L.crk_vq_ema_stats(ptr(xk), ldx, ptr(idx), N, D, K, ptr(counts), ptr(sums), ptr(scratch), stream_ptr()),
          "crk_vq_ema_stats")


def vq_ema_partial(x, idx, D, K):
    """First half of vq_ema_stats: the per-chunk tables of one quantizer call; returns (scratch, N) for
    vq_ema_reduce_multi."""
    L = _lib.lib()
    xk, ldx = _rows(x)
    N = idx.numel()
    nbytes = L.crk_vq_ema_scratch_bytes(N, D, K)
    if nbytes < 0:
        raise ValueError(f"vq_ema_partial: unsupported codebook size K={K}")
    scratch = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    check(L.crk_vq_ema_partial(ptr(xk), ldx, ptr(idx), N, D, K, ptr(scratch), stream_ptr()), "crk_vq_ema_partial")
    return scratch, N


def _parr(tensors):
    return (ctypes.c_void_p * len(tensors))(*[ptr(t) for t in tensors])


def _iarr(vals):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


def vq_ema_partial_multi(xs, idxs, Ds, Ks):
    """vq_ema_partial for up to 4 quantizer calls in one launch; returns [(scratch, N), ...]."""
    L = _lib.lib()
    rows = [_rows(x) for x in xs]
    Ns = [i.numel() for i in idxs]
    scr = []
    for x, N, D, K in zip(xs, Ns, Ds, Ks):
        nbytes = L.crk_vq_ema_scratch_bytes(N, D, K)
        if nbytes < 0:
            raise ValueError(f"vq_ema_partial: unsupported codebook size K={K}")
        scr.append(torch.empty(nbytes, device=x.device, dtype=torch.uint8))
    check(L.crk_vq_ema_partial_multi(len(xs), _parr([r[0] for r in rows]), _iarr([r[1] for r in rows]), _parr(idxs), _iarr(Ns),
                                     _iarr(Ds), _iarr(Ks), _parr(scr), stream_ptr()), "crk_vq_ema_partial_multi")
    return list(zip(scr, Ns))


def vq_ema_reduce_size_multi(scratches, Ns, Ds, Ks, counts, sums, ema_sizes, decay, eps):
    """vq_ema_reduce_multi + the cluster-size half of vq_ema_apply_multi in one launch (single process)."""
    check(_lib.lib().crk_vq_ema_reduce_size_multi(len(Ns), _parr(scratches), _iarr(Ns), _iarr(Ds), _iarr(Ks), _parr(counts),
                                                  _parr(sums), _parr(ema_sizes), float(decay), float(eps), stream_ptr()),
          "crk_vq_ema_reduce_size_multi")


def vq_ema_blend_multi(sums, ema_sizes, ema_ws, codebooks, Ds, Ks, decay, images=None):
    """The blend half of vq_ema_apply_multi (cluster sizes already updated).  images: one prepared-image buffer per codebook
    (vq_image_bytes) - the blend leaves them current in the same launch where the shapes allow (returns True), else they are
    untouched (returns False: the caller rebuilds them)."""
    L = _lib.lib()
    if images is not None and len(Ds) <= 4:
        rc = L.crk_vq_ema_blend_image_multi(len(Ds), _parr(sums), _parr(ema_sizes), _parr(ema_ws), _parr(codebooks), _iarr(Ds),
                                            _iarr(Ks), float(decay), _parr(images), stream_ptr())
        if rc == 0:
            return True
        if rc != 3:
            check(rc, "crk_vq_ema_blend_image_multi")
    check(L.crk_vq_ema_blend_multi(len(Ds), _parr(sums), _parr(ema_sizes), _parr(ema_ws), _parr(codebooks), _iarr(Ds),
                                   _iarr(Ks), float(decay), stream_ptr()), "crk_vq_ema_blend_multi")
    return False


def vq_ema_reduce_multi(scratches, Ns, Ds, Ks, counts, sums):
    """Per-chunk tables of several quantizer calls -> their integer statistics, one launch."""
    check(_lib.lib().crk_vq_ema_reduce_multi(len(Ns), _parr(scratches), _iarr(Ns), _iarr(Ds), _iarr(Ks), _parr(counts),
                                             _parr(sums), stream_ptr()), "crk_vq_ema_reduce_multi")


def vq_ema_apply_multi(counts, sums, ema_sizes, ema_ws, codebooks, Ds, Ks, decay, eps):
    """vq_ema_apply for several quantizers: one size launch and one blend launch for all of them."""
    check(_lib.lib().crk_vq_ema_apply_multi(len(Ds), _parr(counts), _parr(sums), _parr(ema_sizes), _parr(ema_ws),
                                            _parr(codebooks), _iarr(Ds), _iarr(Ks), float(decay), float(eps), stream_ptr()), "crk_vq_ema_apply_multi")


def vq_ema_apply(counts, sums, ema_size, ema_w, codebook, decay, eps):
    """EMA blend + Laplace smoothing + codebook write-back from (all-reduced) integer statistics, in place."""
    D, K = ema_w.shape
    check(_lib.lib().crk_vq_ema_apply(ptr(counts), ptr(sums), ptr(ema_size), ptr(ema_w), ptr(codebook), D, K,
                                      float(decay), float(eps), stream_ptr()), "crk_vq_ema_apply")


def vq_ema_update(x, idx, ema_size, ema_w, codebook, decay, eps, reduce_fn=None):
    """EMA codebook update (in place).  `reduce_fn(counts, sums)` all-reduces the integer
    statistics under data parallelism."""
    D, K = ema_w.shape
    counts = torch.empty(K, device=x.device, dtype=torch.int32)
    sums = torch.empty(D * K, device=x.device, dtype=torch.int64)
    vq_ema_stats(x, idx, counts, sums)
    if reduce_fn is not None:
        reduce_fn(counts, sums)
    vq_ema_apply(counts, sums, ema_size, ema_w, codebook, decay, eps)


# ------------------------------------------------------------------------------------
_scratch = {}


def _loss_scratch(device):
    key = (device.type, device.index, stream_ptr())  # per stream: losses on different streams must not share partials
    if key not in _scratch:
        _scratch[key] = torch.empty(_lib.lib().crk_loss_scratch_floats(), device=device, dtype=torch.float32)
    return _scratch[key]


def _as2d(t):
    """(tensor, ld, N, D) view of a tensor as frames x channels."""
    if t.dim() == 3:
        k, ld = _rows(t)
        return k, ld, k.shape[0] * k.shape[1], k.shape[2]
    if t.dim() == 2:
        k, ld = _rows(t)
        return k, ld, k.shape[0], k.shape[1]
    k = t.contiguous().reshape(-1, 1)
    return k, 1, k.numel(), 1


class _MaskedLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, mask, mode, yconst):
        L = _lib.lib()
        xk, ldx, N, Dm = _as2d(x)
        if y is not None:
            yk, ldy, Ny, Dy = _as2d(y)
            assert (Ny, Dy) == (N, Dm), (x.shape, y.shape)
        else:
            yk, ldy = None, 0
        mk = None
        if mask is not None:
            mk = mask.reshape(-1).contiguous()
            mk = mk.view(torch.uint8) if mk.dtype == torch.bool else mk.to(torch.uint8)
            assert mk.numel() == N, (mk.numel(), N)
        out = _scalars(2, x.device)
        check(L.crk_masked_loss_fwd(ptr(xk), ldx, ptr(yk), ldy, float(yconst), ptr(mk), N, Dm, mode, ptr(out),
                                    ptr(_loss_scratch(x.device)), stream_ptr()), "crk_masked_loss_fwd")
        ctx.mode, ctx.yconst, ctx.geom = mode, float(yconst), (N, Dm, ldx, ldy)
        ctx.has_y, ctx.has_m = y is not None, mk is not None
        ctx.save_for_backward(xk, yk if yk is not None else out, mk if mk is not None else out, out)
        ctx.xshape = x.shape
        ctx.yshape = y.shape if y is not None else None
        return out[0]

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        xk, yk, mk, out = ctx.saved_tensors
        N, Dm, ldx, ldy = ctx.geom
        yk = yk if ctx.has_y else None
        mk = mk if ctx.has_m else None
        dx = torch.empty(N, Dm, device=g.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        dy = torch.empty(N, Dm, device=g.device, dtype=torch.float32) if (ctx.has_y and ctx.needs_input_grad[1]) else None
        g = g.contiguous().reshape(1)
        check(L.crk_masked_loss_bwd(ptr(xk), ldx, ptr(yk), ldy, ctx.yconst, ptr(mk), N, Dm, ctx.mode, ptr(out), ptr(g),
                                    ptr(dx), Dm, ptr(dy), Dm, stream_ptr()), "crk_masked_loss_bwd")
        return (dx.view(ctx.xshape) if dx is not None else None, dy.view(ctx.yshape) if dy is not None else None,
                None, None, None)


class _MaskedBothFn(torch.autograd.Function):
    """(L1 mean, MSE mean) of the same masked pair in one pass; gradients only for the means that are differentiated."""

    @staticmethod
    def forward(ctx, x, y, mask):
        L = _lib.lib()
        xk, ldx, N, Dm = _as2d(x)
        yk, ldy, Ny, Dy = _as2d(y)
        assert (Ny, Dy) == (N, Dm), (x.shape, y.shape)
        mk = None
        if mask is not None:
            mk = mask.reshape(-1).contiguous()
            mk = mk.view(torch.uint8) if mk.dtype == torch.bool else mk.to(torch.uint8)
            assert mk.numel() == N, (mk.numel(), N)
        out = _scalars(4, x.device)
        check(L.crk_masked_loss_both_fwd(ptr(xk), ldx, ptr(yk), ldy, ptr(mk), N, Dm, ptr(out), ptr(_loss_scratch(x.device)),
                                         stream_ptr()), "crk_masked_loss_both_fwd")
        ctx.geom = (N, Dm, ldx, ldy)
        ctx.has_m = mk is not None
        ctx.save_for_backward(xk, yk, mk if mk is not None else out, out)
        ctx.xshape = x.shape
        ctx.set_materialize_grads(False)
        return out[0], out[2]

    @staticmethod
    def backward(ctx, g1, g2):
        L = _lib.lib()
        xk, yk, mk, out = ctx.saved_tensors
        N, Dm, ldx, ldy = ctx.geom
        mk = mk if ctx.has_m else None
        if not ctx.needs_input_grad[0] or (g1 is None and g2 is None):
            return None, None, None
        dx = None
        for mode, g in ((0, g1), (1, g2)):
            if g is None:
                continue
            nxt = torch.empty(N, Dm, device=xk.device, dtype=torch.float32)
            gg = g.contiguous().reshape(1)
            check(L.crk_masked_loss_bwd_acc(ptr(xk), ldx, ptr(yk), ldy, 0.0, ptr(mk), N, Dm, mode, ptr(out[2 * mode: 2 * mode + 2]),
                                            ptr(gg), ptr(nxt), Dm, None, 0, ptr(dx), Dm, None, stream_ptr()),
                  "crk_masked_loss_bwd_acc")
            dx = nxt
        return dx.view(ctx.xshape), None, None


def masked_both_loss(x, y, mask=None):
    """(mean |x-y|, mean (x-y)^2) over the masked frames, one pass; y carries no gradient."""
    return _MaskedBothFn.apply(x, y, mask)


def masked_mean_loss(x, y, mask=None, mode="l1", yconst=0.0):
    """mean over masked frames of |x-y| or (x-y)^2; y=None uses the constant yconst."""
    return _MaskedLossFn.apply(x, y, mask, 0 if mode == "l1" else 1, yconst)


class _CEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        L = _lib.lib()
        lk, ldl = _rows(logits)
        N, C = lk.shape
        tk = target.contiguous()
        out = _scalars(2, logits.device)
        dl = torch.empty(N, C, device=logits.device, dtype=torch.float32)
        check(L.crk_ce_fwd(ptr(lk), ldl, ptr(tk), N, C, int(ignore_index), ptr(out), ptr(dl),
                           ptr(_loss_scratch(logits.device)), stream_ptr()), "crk_ce_fwd")
        ctx.save_for_backward(dl, out)
        ctx.geom = (N, C)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        dl, out = ctx.saved_tensors
        N, C = ctx.geom
        res = torch.empty(N, C, device=g.device, dtype=torch.float32)
        g = g.contiguous().reshape(1)
        check(L.crk_ce_bwd(ptr(dl), N, C, ptr(out), ptr(g), ptr(res), stream_ptr()), "crk_ce_bwd")
        return res, None, None


def cross_entropy(logits, target, ignore_index=-100):
    return _CEFn.apply(logits, target, ignore_index)


class _STFTLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, resolutions, windows, logratio):
        L = _lib.lib()
        xk, ldx = _rows(x)
        yk, ldy = _rows(y)
        B, T, Dm = xk.shape
        out = _scalars(1, x.device)
        w = 1.0 / len(resolutions)
        ctx.multi = len(resolutions) <= 4 and all(win <= 64 for _, _, win in resolutions)
        ctx.unit = None
        if ctx.multi and ctx.needs_input_grad[0] and not cfg.stft_two_pass:
            # the loss will be differentiated: loss and gradient (for an upstream gradient of 1) in ONE pass over the DFTs
            ctx.unit = torch.zeros(B, T, Dm, device=x.device, dtype=torch.float32)
            check(L.crk_stft_loss_multi_fwd_grad(ptr(xk), ldx, ptr(yk), ldy, B, T, Dm, len(resolutions),
                                                 _iarr([r[0] for r in resolutions]), _iarr([r[1] for r in resolutions]),
                                                 _iarr([r[2] for r in resolutions]), _parr(windows), float(logratio), ptr(out),
                                                 ptr(ctx.unit), Dm, ptr(_loss_scratch(x.device)), stream_ptr()),
                  "crk_stft_loss_multi_fwd_grad")
        elif ctx.multi:  # every resolution in one launch
            check(L.crk_stft_loss_multi_fwd(ptr(xk), ldx, ptr(yk), ldy, B, T, Dm, len(resolutions),
                                            _iarr([r[0] for r in resolutions]), _iarr([r[1] for r in resolutions]),
                                            _iarr([r[2] for r in resolutions]), _parr(windows), float(logratio), ptr(out),
                                            ptr(_loss_scratch(x.device)), stream_ptr()), "crk_stft_loss_multi_fwd")
        else:
            for i, ((n_fft, hop, win), wt) in enumerate(zip(resolutions, windows)):
                check(L.crk_stft_loss_fwd(ptr(xk), ldx, ptr(yk), ldy, B, T, Dm, n_fft, hop, win, ptr(wt), float(logratio), w,
                                          1 if i > 0 else 0, ptr(out), ptr(_loss_scratch(x.device)), stream_ptr()),
                      "crk_stft_loss_fwd")
        ctx.save_for_backward(xk, yk, *windows)
        ctx.res, ctx.logratio, ctx.ld = resolutions, float(logratio), (ldx, ldy)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        if ctx.unit is not None:
            unit, ctx.unit = ctx.unit, None
            return unit.mul_(g), None, None, None, None
        xk, yk, *windows = ctx.saved_tensors
        B, T, Dm = xk.shape
        ldx, ldy = ctx.ld
        dx = torch.zeros(B, T, Dm, device=g.device, dtype=torch.float32)
        g = g.contiguous().reshape(1)
        w = 1.0 / len(ctx.res)
        if ctx.multi:
            check(L.crk_stft_loss_multi_bwd(ptr(xk), ldx, ptr(yk), ldy, B, T, Dm, len(ctx.res), _iarr([r[0] for r in ctx.res]),
                                            _iarr([r[1] for r in ctx.res]), _iarr([r[2] for r in ctx.res]), _parr(windows),
                                            ctx.logratio, ptr(g), ptr(dx), Dm, stream_ptr()), "crk_stft_loss_multi_bwd")
            return dx, None, None, None, None
        for (n_fft, hop, win), wt in zip(ctx.res, windows):
            check(L.crk_stft_loss_bwd(ptr(xk), ldx, ptr(yk), ldy, B, T, Dm, n_fft, hop, win, ptr(wt), ctx.logratio, w,
                                      ptr(g), ptr(dx), Dm, stream_ptr()), "crk_stft_loss_bwd")
        return dx, None, None, None, None


_tw_tables = {}


def _stft_tables(resolutions, windows):
    """Twiddle tables (cos * window, -sin * window per bin) of the fused reconstruction-loss kernels, built once per
    (n_fft, win_length, window tensor)."""
    L = _lib.lib()
    out = []
    for (n_fft, _, win), w in zip(resolutions, windows):
        key = (int(n_fft), int(win), w.data_ptr(), w.device.index)
        hit = _tw_tables.get(key)
        if hit is None:
            tab = torch.empty(L.crk_stft_twiddle_floats(int(n_fft), int(win)), device=w.device, dtype=torch.float32)
            check(L.crk_stft_twiddles(int(n_fft), int(win), ptr(w), ptr(tab), stream_ptr()), "crk_stft_twiddles")
            hit = (w, tab)  # (the window is kept alive: its address is the key)
            if not (w.is_cuda and torch.cuda.is_current_stream_capturing()):
                _tw_tables[key] = hit  # a table first built inside a capture lives in the graph's pool: rebuilt by every replay, never cached
        out.append(hit[1])
    return out


def recon_supported(T, resolutions):
    ia = [_iarr([r[i] for r in resolutions]) for i in range(3)]
    return bool(_lib.lib().crk_recon_supported(int(T), len(resolutions), ia[0], ia[1], ia[2]))


class _ReconFn(torch.autograd.Function):
    """(L1 mean, MSE mean, multi-resolution STFT loss) of decoded features against their target - the three terms the
    trainers form on the same pair (trainer_vqvae.py:215-225).  Frames that do not overlap (crk_recon_supported; quirk
    Q1 makes the default stft_params such): ONE launch for the two sums, every DFT and the STFT loss's unit gradient
    (compact, no atomics) + a finishing launch; backward ONE launch.  Other geometries: one pass for both means, one
    over the DFTs with atomics into a dense unit gradient; backward one launch."""

    @staticmethod
    def forward(ctx, x, y, mask, resolutions, windows, logratio):
        L = _lib.lib()
        xk, ldx = _rows(x)
        yk, ldy = _rows(y)
        B, T, Dm = xk.shape
        N = B * T
        mk = None
        if mask is not None:
            mk = mask.reshape(-1).contiguous()
            mk = mk.view(torch.uint8) if mk.dtype == torch.bool else mk.to(torch.uint8)
            assert mk.numel() == N, (mk.numel(), N)
        out = _scalars(5, x.device)
        scr = _loss_scratch(x.device)
        nres = len(resolutions)
        ia = [_iarr([r[i] for r in resolutions]) for i in range(3)]
        ctx.fused = recon_supported(T, resolutions) and not cfg.recon_dense
        ctx.geom = (N, Dm, ldx, ldy)
        ctx.has_m = mk is not None
        ctx.xshape = x.shape
        ctx.set_materialize_grads(False)
        if ctx.fused:
            tabs = _stft_tables(resolutions, windows)
            gc = (torch.empty(L.crk_recon_grad_floats(B, T, Dm, nres, ia[1], ia[2]), device=x.device, dtype=torch.float32)
                  if ctx.needs_input_grad[0] else None)
            check(L.crk_recon_loss_fwd(ptr(xk), ldx, ptr(yk), ldy, ptr(mk), B, T, Dm, nres, ia[0], ia[1], ia[2], _parr(tabs),
                                       float(logratio), ptr(out), ptr(gc), ptr(scr), stream_ptr()), "crk_recon_loss_fwd")
            ctx.unit, ctx.res = gc, resolutions
            ctx.save_for_backward(xk, yk, mk if mk is not None else out, out)
            return out[0], out[2], out[4]
        check(L.crk_masked_loss_both_fwd(ptr(xk), ldx, ptr(yk), ldy, ptr(mk), N, Dm, ptr(out), ptr(scr), stream_ptr()),
              "crk_masked_loss_both_fwd")
        unit = torch.zeros(B, T, Dm, device=x.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        if unit is not None:
            check(L.crk_stft_loss_multi_fwd_grad(ptr(xk), ldx, ptr(yk), ldy, B, T, Dm, nres, ia[0], ia[1], ia[2], _parr(windows),
                                                 float(logratio), ptr(out[4:]), ptr(unit), Dm, ptr(scr), stream_ptr()),
                  "crk_stft_loss_multi_fwd_grad")
        else:
            check(L.crk_stft_loss_multi_fwd(ptr(xk), ldx, ptr(yk), ldy, B, T, Dm, nres, ia[0], ia[1], ia[2], _parr(windows),
                                            float(logratio), ptr(out[4:]), ptr(scr), stream_ptr()), "crk_stft_loss_multi_fwd")
        ctx.unit = unit
        ctx.save_for_backward(xk, yk, mk if mk is not None else out, out)
        return out[0], out[2], out[4]

    @staticmethod
    def backward(ctx, g1, g2, g3):
        L = _lib.lib()
        xk, yk, mk, out = ctx.saved_tensors
        N, Dm, ldx, ldy = ctx.geom
        mk = mk if ctx.has_m else None
        unit, ctx.unit = ctx.unit, None
        if g3 is not None and unit is None:
            raise RuntimeError("the STFT term of a recon loss can be differentiated once")
        if g1 is None and g2 is None and g3 is None:
            return None, None, None, None, None, None
        if ctx.fused:
            B, T = xk.shape[0], xk.shape[1]
            ia = [_iarr([r[i] for r in ctx.res]) for i in range(3)]
            gs = [None if g is None else g.contiguous().reshape(1) for g in (g1, g2, g3)]
            dx = torch.empty(N, Dm, device=xk.device, dtype=torch.float32)
            check(L.crk_recon_loss_bwd(ptr(xk), ldx, ptr(yk), ldy, ptr(mk), B, T, Dm, len(ctx.res), ia[0], ia[1], ia[2],
                                       ptr(out), ptr(unit), ptr(gs[0]), ptr(gs[1]), ptr(gs[2]), ptr(dx), Dm, stream_ptr()),
                  "crk_recon_loss_bwd")
            return dx.view(ctx.xshape), None, None, None, None, None
        dx, scale = (unit, g3.contiguous().reshape(1)) if g3 is not None else (None, None)
        for mode, g in ((0, g1), (1, g2)):
            if g is None:
                continue
            nxt = torch.empty(N, Dm, device=xk.device, dtype=torch.float32)
            gg = g.contiguous().reshape(1)
            check(L.crk_masked_loss_bwd_acc(ptr(xk), ldx, ptr(yk), ldy, 0.0, ptr(mk), N, Dm, mode, ptr(out[2 * mode: 2 * mode + 2]),
                                            ptr(gg), ptr(nxt), Dm, None, 0, ptr(dx), Dm, ptr(scale), stream_ptr()),
                  "crk_masked_loss_bwd_acc")
            dx, scale = nxt, None
        if scale is not None:  # only the STFT term is differentiated
            dx = dx.mul_(scale)
        return dx.view(ctx.xshape), None, None, None, None, None


def recon_loss(x, y, mask, resolutions, windows, logratio=0.0):
    """(L1, MSE, STFT loss) of x against y (mask: frames of the two means; the STFT loss ignores it); window lengths <= 64."""
    return _ReconFn.apply(x, y, mask, tuple(resolutions), tuple(windows), logratio)


def stft_loss(x, y, resolutions, windows, logratio=0.0):
    """resolutions: list of (n_fft, hop_length, win_length) as torch.stft receives them."""
    return _STFTLossFn.apply(x, y, tuple(resolutions), tuple(windows), logratio)


# ------------------------------------------------------------------------------------
_ONES = {}
_WCONST = {}


def one_like(t):
    """A cached tensor of ones shaped like t: the root gradient of a backward pass (``backward()`` would fill a new one every
    call).  _WeightedSumFn recognises it and hands out cached constants instead of launching."""
    key = (t.device, t.dtype, tuple(t.shape))
    if key not in _ONES:
        _ONES[key] = torch.ones_like(t)
    return _ONES[key]


def _farr(vals):
    return (ctypes.c_float * len(vals))(*[float(v) for v in vals])


class _WeightedSumFn(torch.autograd.Function):
    """total = sum_i w_i * term_i (+ constant) over 0-dim device scalars: one launch (stack + multiply + sum were three);
    backward one launch - none when the upstream gradient is the cached 1 of one_like()."""

    @staticmethod
    def forward(ctx, weights, constant, *terms):
        ts = [t.reshape(1) if t.is_contiguous() else t.contiguous().reshape(1) for t in terms]
        out = _scalars(1, terms[0].device)
        check(_lib.lib().crk_weighted_sum(len(ts), _parr(ts), _farr(weights), float(constant), ptr(out), stream_ptr()),
              "crk_weighted_sum")
        ctx.weights = tuple(float(w) for w in weights)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        ws = ctx.weights
        one = _ONES.get((g.device, g.dtype, tuple(g.shape)))
        key = (ws, g.device)
        cached = one is not None and g.data_ptr() == one.data_ptr()
        if cached and key not in _WCONST:
            if g.is_cuda and torch.cuda.is_current_stream_capturing():
                cached = False  # a host-to-device copy cannot be captured: this pass launches, the next eager one fills the cache
            else:
                _WCONST[key] = torch.tensor(ws, device=g.device, dtype=torch.float32)
        if cached:
            grads = _WCONST[key]
        else:
            grads = torch.empty(len(ws), device=g.device, dtype=torch.float32)
            check(_lib.lib().crk_weighted_sum_bwd(len(ws), _farr(ws), ptr(g.contiguous().reshape(1)), ptr(grads), stream_ptr()),
                  "crk_weighted_sum_bwd")
        return (None, None) + tuple(grads[i] for i in range(len(ws)))


def weighted_sum(terms, weights, constant=0.0):
    """sum_i weights[i] * terms[i] + constant for 0-dim fp32 device tensors (<= 16 of them)."""
    return _WeightedSumFn.apply(tuple(weights), float(constant), *terms)


# ------------------------------------------------------------------------------------
def _label_runs(idx):
    """(tensor to hand to the kernels, run): a (B,T) label tensor in which every frame reads its utterance's first label -
    the stride-0 view ``h[:, 0:1].expand(-1, T)`` of a contiguous h - is passed as h itself with run = T (the lookup
    kernels then read idx[n - n % run]: no filled copy); anything else contiguous with run = 1."""
    B, T = idx.shape
    if T > 1 and idx.stride(1) == 0 and (B == 1 or idx.stride(0) == T):
        return idx, T
    return idx.contiguous(), 1


class _ConcatEmbedFn(torch.autograd.Function):
    """out = cat([a, b, table[idx]], -1); backward routes the embedding slice into the
    owner's flat gradient (table lives there) and returns da / db slices."""

    @staticmethod
    def forward(ctx, a, b, table, idx, owner, tab_offset, flat):
        L = _lib.lib()
        B, T = idx.shape
        ak, lda = (None, 0) if a is None else _rows(a)
        bk, ldb = (None, 0) if b is None else _rows(b)
        ca = 0 if a is None else a.shape[-1]
        cb = 0 if b is None else b.shape[-1]
        E = table.shape[1]
        out = torch.empty(B, T, ca + cb + E, device=idx.device, dtype=torch.float32)
        ik, run = _label_runs(idx)
        check(L.crk_concat_embed_run(ptr(ak), lda, ca, ptr(bk), ldb, cb, ptr(table), E, ptr(ik), run, B * T, ptr(out),
                                     ca + cb + E, stream_ptr()), "crk_concat_embed_run")
        ctx.geom = (ca, cb, E, table.shape[0], run, B * T)
        ctx.owner, ctx.tab_offset = owner, tab_offset
        ctx.save_for_backward(ik)
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        (ik,) = ctx.saved_tensors
        ca, cb, E, rows, run, N = ctx.geom
        dk, ld = _rows(dout)
        if ctx.owner is not None and not ctx.owner.skip_param_grads:
            ctx.owner.grads_clean = False
            g = ctx.owner.grad_flat[ctx.tab_offset: ctx.tab_offset + rows * E]
            scratch = torch.empty(L.crk_embed_bwd_scratch_floats(N, E, rows), device=dk.device, dtype=torch.float32)
            check(L.crk_embed_bwd_run(ptr(dk), ld, ca + cb, E, ptr(ik), run, N, rows, ptr(g), ptr(scratch), stream_ptr()),
                  "crk_embed_bwd_run")
        da = dk[..., :ca] if (ca and ctx.needs_input_grad[0]) else None
        db = dk[..., ca:ca + cb] if (cb and ctx.needs_input_grad[1]) else None
        return da, db, None, None, None, None, None


def concat_embed(a, b, table, idx, owner=None, tab_offset=0, flat=None):
    return _ConcatEmbedFn.apply(a, b, table, idx, owner, tab_offset, flat)


def adam_step(flat, grad, exp_avg, exp_avg_sq, lr_dev, step_dev, beta1=0.9, beta2=0.999, eps=1e-8, clear_grads=False,
              defer_bump=False):
    check(_lib.lib().crk_adam_step(ptr(flat), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), flat.numel(), ptr(lr_dev),
                                   ptr(step_dev), beta1, beta2, eps, (1 if clear_grads else 0) | (2 if defer_bump else 0),
                                   stream_ptr()), "crk_adam_step")


def radam_step(flat, grad, exp_avg, exp_avg_sq, lr_dev, step_dev, beta1=0.9, beta2=0.999, eps=1e-8, clear_grads=False,
               defer_bump=False):
    check(_lib.lib().crk_radam_step(ptr(flat), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), flat.numel(), ptr(lr_dev),
                                    ptr(step_dev), beta1, beta2, eps, (1 if clear_grads else 0) | (2 if defer_bump else 0),
                                    stream_ptr()), "crk_radam_step")


def lamb_tile():
    return int(_lib.lib().crk_lamb_tile())


def lamb_step(flat, grad, exp_avg, exp_avg_sq, upd, tiles, tensors, part, ratio_out, lr_dev, step_dev, beta1=0.9, beta2=0.999,
              eps=1e-6, clear_grads=False, defer_bump=False):
    """tiles: int32 [n_tiles, 4] = (offset, length, tensor, 0); tensors: int32 [n_tensors, 2] = (first tile, tiles)."""
    assert tiles.dtype == torch.int32 and tensors.dtype == torch.int32 and tiles.is_contiguous() and tensors.is_contiguous()
    assert upd.numel() >= flat.numel() and part.numel() >= 2 * tiles.shape[0]
    assert ratio_out is None or ratio_out.numel() >= tensors.shape[0]
    check(_lib.lib().crk_lamb_step(ptr(flat), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), ptr(upd), ptr(tiles), tiles.shape[0],
                                   ptr(tensors), tensors.shape[0], ptr(part), ptr(ratio_out), ptr(lr_dev), ptr(step_dev),
                                   beta1, beta2, eps, (1 if clear_grads else 0) | (2 if defer_bump else 0), stream_ptr()),
          "crk_lamb_step")


def logmel(raw, T, n_fft, hop, win_length, window, mel_basis, eps=1e-10, mean=None, std=None, center=False):
    L = _lib.lib()
    raw = raw.contiguous()
    B, n = raw.shape
    n_mels = mel_basis.shape[1]
    out = torch.empty(B, T, n_mels, device=raw.device, dtype=torch.float32)
    check(L.crk_logmel_fwd(ptr(raw), n, B, n, T, n_fft, hop, win_length, ptr(window), ptr(mel_basis), n_mels, float(eps),
                           ptr(mean), ptr(std), ptr(out), n_mels, 1 if center else 0, stream_ptr()), "crk_logmel_fwd")
    return out
