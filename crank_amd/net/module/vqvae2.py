"""Hierarchical VQ-VAE generator G on the HIP stacks.

Same public surface as the reference class (crank/net/module/vqvae2.py:38-283):
``forward(x, enc_h, dec_h, spkrvec, use_ema, encoder_detach) -> dict`` and
``cycle_forward(...) -> list[dict]`` with the keys of ``make_dict`` (:197-209),
attributes ``spkr_embedding``, ``encoder_receptive_size``, ``decoder_receptive_size``,
``conf`` and a reference-compatible ``state_dict``.  Internally everything stays
channel-last (B,T,C): the reference's (B,C,T) transposes (:88-91) have no counterpart.
"""
import torch

from ... import ops, parallel
from .flat import FlatModel
from .mlfb import LogMelFilterBankLayer
from .pwg import KIND_GENERATOR, HipStack


class Quantizer:
    """VQ codebook with EMA update (crank/net/module/vqvae2.py:286-347).  Lives inside
    its owner's flat block (embedding.weight) plus two buffers (ema_size, ema_w)."""

    def __init__(self, owner, prefix, emb_dim, emb_size, decay=0.99, eps=1e-5, ema_flag=False, bdt_flag=False):
        self.owner, self.prefix = owner, prefix
        self.emb_dim, self.emb_size = emb_dim, emb_size
        self.decay, self.eps, self.ema_flag, self.bdt_flag = decay, eps, ema_flag, bdt_flag
        self.cb_offset = None
        self.training = True
        self.bucket, self.slot = None, 0  # set by the owning generator: the shared EMA-statistics message (C2)
        self._img, self._img_epoch = None, -1  # the codebook image of the search kernel and the owner's epoch it was built at

    def entries(self, base):
        self.cb_offset = base
        return [(self.prefix + "embedding.weight", base, (self.emb_size, self.emb_dim))]

    @property
    def n_params(self):
        return self.emb_size * self.emb_dim

    def make_buffers(self, device):
        if not self.ema_flag:
            return {}
        self.ema_size = torch.zeros(self.emb_size, device=device)
        self.ema_w = torch.randn(self.emb_dim, self.emb_size, device=device)
        return {self.prefix + "ema_size": self.ema_size, self.prefix + "ema_w": self.ema_w}

    @property
    def weight(self):
        o = self.cb_offset
        return self.owner.flat.data[o: o + self.n_params].view(self.emb_size, self.emb_dim)

    @torch.no_grad()
    def init_parameters(self):
        self.weight.uniform_(-1.0 / self.emb_size, 1.0 / self.emb_size)

    def image_stale(self):
        """The prepared form of the codebook the search kernel reads (split-f16 operand planes, squared norms, scales:
        ops.vq_image_build) is valid for ONE state of the codebook.  The owner counts the codebooks' states
        (``codebook_epoch``: advanced by the EMA blend, load_state_dict, touch(), and by optimizer steps when a codebook
        is trained); an image built at an older epoch is rebuilt before the next search - here, or for all quantizers of a
        forward in one launch (VQVAE2.refresh_images).  None: this shape has no image (the kernel derives everything per call)."""
        if self._img is None:
            # (an owner that does not count codebook states cannot keep an image valid: none)
            nbytes = ops.vq_image_bytes(self.emb_size, self.emb_dim) if (ops.VQ_IMAGE and hasattr(self.owner, "codebook_epoch")) else 0
            self._img = torch.empty(nbytes, device=self.owner.flat.device, dtype=torch.uint8) if nbytes else False
        if self._img is False:
            return None
        return self._img_epoch != getattr(self.owner, "codebook_epoch", 0)

    def image(self):
        stale = self.image_stale()
        if stale is None:
            return None
        if stale:
            ops.vq_image_build([self.weight], [self._img])
            self._img_epoch = getattr(self.owner, "codebook_epoch", 0)
        return self._img

    def quantize(self, x, use_ema=True, pending=None, commit_mask=None, want_commit=False, qx_out=None, want_e=True,
                 want_qx=True, add=None, alias=False):
        """x: (B,T,D) channel-last -> (embed_idx (B,T,D), embed_idx_qx (B,T,D), idx (B,T)).
        EMA (training, ema_flag, use_ema): the integer statistics of this call are written into the owner's
        message bucket; with `pending` (a list, the generator's decode) the exchange and the blend are left to
        the caller's ``flush_ema`` - one message for all quantizers of the forward -
        otherwise they happen here."""
        # add (not in the reference's signature): the quantizer's input is x + add - the decoder's "enc[n] + dec"
        # (vqvae2.py:177) formed inside the search kernel; self.xin is that input afterwards (x itself without add)
        # alias (with the commitment loss only): x_alias / qx_alias are x and qx again for their second consumers, whose
        # gradients then join this op's backward launch (ops._VQCommitFn); None otherwise
        self.commit = self.x_alias = self.qx_alias = None
        if want_commit and self.ema_flag:  # commitment loss inside the op (its backward joins the straight-through one)
            alias = alias and torch.is_grad_enabled() and x.requires_grad
            r = ops.vq_commit_apply(x, self.weight, commit_mask, qx_out=qx_out, add=add, alias=alias, image=self.image())
            e, qx, idx, self.commit = r[:4]
            if alias:
                self.x_alias, self.qx_alias = r[-2:]
            xsum = r[4] if add is not None else None
        else:
            r = ops.vq_apply(x, self.weight, None if self.ema_flag else self.owner, self.cb_offset, qx_out=qx_out,
                             want_e=want_e, want_qx=want_qx, add=add, image=self.image())
            e, qx, idx = r[:3]
            xsum = r[-1] if add is not None else None
        self.xin = x = x if add is None else xsum
        if self.training and self.ema_flag and use_ema:
            # lookup used the OLD codebook; statistics use every frame
            if self.bucket is None:
                self.bucket = parallel.EmaBucket([(self.emb_dim, self.emb_size)], x.device)
            if pending is not None:  # a generator forward: tables, reduce and blend for all its quantizers in flush_ema
                self._partial = None
                self._ema_in = (x.detach(), idx)
                pending.append(self)
            else:
                counts, sums = self.bucket.views(self.slot)
                ops.vq_ema_stats(x.detach(), idx, counts, sums)
                self._partial = None
                flush_ema([self])
        return e, qx, idx

    def apply_ema(self):
        counts, sums = self.bucket.views(self.slot)
        ops.vq_ema_apply(counts, sums, self.ema_size, self.ema_w, self.weight, self.decay, self.eps)
        self.owner.touch_codebook()

    def __call__(self, x, use_ema=True):
        if self.bdt_flag:
            x = x.transpose(1, 2)
        e, qx, idx = self.quantize(x, use_ema=use_ema)
        if self.bdt_flag:
            qx = qx.transpose(1, 2)
        return e, qx, idx


def flush_ema(pending):
    """One all-reduce for the statistics of every quantizer in `pending` (they share a bucket), then the blends."""
    if not pending:
        return
    buckets = []
    for q in pending:
        if not any(q.bucket is b for b in buckets):
            buckets.append(q.bucket)
    views = [q.bucket.views(q.slot) for q in pending]
    chunks = [list(range(i, min(i + 4, len(pending)))) for i in range(0, len(pending), 4)]  # the multi entry points take <= 4
    # the per-chunk tables of the calls that left their inputs (a generator forward): one launch per <= 4 calls
    for ch in chunks:
        todo = [i for i in ch if getattr(pending[i], "_ema_in", None) is not None]
        if todo:
            parts = ops.vq_ema_partial_multi([pending[i]._ema_in[0] for i in todo], [pending[i]._ema_in[1] for i in todo],
                                             [pending[i].emb_dim for i in todo], [pending[i].emb_size for i in todo])
            for i, pt in zip(todo, parts):
                pending[i]._partial, pending[i]._ema_in = pt, None
    fused = all(getattr(q, "_partial", None) is not None for q in pending)
    same = len({(q.decay, q.eps) for q in pending}) == 1
    if fused and same and not parallel.is_dist():
        # single process: nothing to exchange between the statistics and the update - tables -> statistics + cluster
        # sizes in one launch, the blend in another
        current = []
        for ch in chunks:
            ops.vq_ema_reduce_size_multi([pending[i]._partial[0] for i in ch], [pending[i]._partial[1] for i in ch],
                                         [pending[i].emb_dim for i in ch], [pending[i].emb_size for i in ch],
                                         [views[i][0] for i in ch], [views[i][1] for i in ch],
                                         [pending[i].ema_size for i in ch], pending[0].decay, pending[0].eps)
            # ... which also leaves the codebooks' search images current where every quantizer of the launch keeps one
            imgs = [pending[i]._img if pending[i].image_stale() is not None else None for i in ch]
            if ops.vq_ema_blend_multi([views[i][1] for i in ch], [pending[i].ema_size for i in ch], [pending[i].ema_w for i in ch],
                                      [pending[i].weight for i in ch], [pending[i].emb_dim for i in ch],
                                      [pending[i].emb_size for i in ch], pending[0].decay,
                                      images=imgs if all(im is not None for im in imgs) else None):
                current += [pending[i] for i in ch]
        for q in pending:
            q.owner.touch_codebook()
            q._partial = None
        for q in current:  # (after EVERY blend of the call has advanced its owner's count)
            q._img_epoch = q.owner.codebook_epoch
        pending.clear()
        return
    if fused:
        for ch in chunks:
            ops.vq_ema_reduce_multi([pending[i]._partial[0] for i in ch], [pending[i]._partial[1] for i in ch],
                                    [pending[i].emb_dim for i in ch], [pending[i].emb_size for i in ch],
                                    [views[i][0] for i in ch], [views[i][1] for i in ch])
    for b in buckets:
        b.reduce()
    if fused and same:
        for ch in chunks:
            ops.vq_ema_apply_multi([views[i][0] for i in ch], [views[i][1] for i in ch], [pending[i].ema_size for i in ch],
                                   [pending[i].ema_w for i in ch], [pending[i].weight for i in ch],
                                   [pending[i].emb_dim for i in ch], [pending[i].emb_size for i in ch], pending[0].decay,
                                   pending[0].eps)
        for q in pending:
            q.owner.touch_codebook()
    else:
        for q in pending:
            q.apply_ema()
    for q in pending:
        q._partial = None
    pending.clear()


class VQVAE2(FlatModel):
    can_skip_decoder = True  # forward(need_decoded=False)
    can_commit = True        # forward(want_commit=True, commit_mask=...)
    can_pair_f0 = True       # dec_h may be the pair (lcf0, uv)

    def __init__(self, conf, spkr_size=0, scaler=None, device="cuda"):
        super().__init__()
        self.codebook_epoch, self._trained_codebooks = 0, not conf["ema_flag"]
        self.conf = conf
        self.spkr_size = spkr_size
        self.encoder_receptive_size = 0
        self.decoder_receptive_size = 0
        if conf["use_sinc_conv"]:
            raise NotImplementedError("use_sinc_conv is a dead branch in the reference "
                                      "(crank/net/module/vqvae2.py:76-82 cannot construct its layer)")
        nst = conf["n_vq_stacks"]
        self.encoders, self.decoders, self.quantizers = [], [], []
        entries, off = [], 0

        def place(stack, prefix):
            nonlocal off
            entries.extend(stack.entries(prefix, off))
            base = off
            off += stack.n_params
            return base

        bases = []
        for n in range(nst):  # crank/net/module/vqvae2.py:211-283
            if n == 0:
                e_in, e_out = conf["input_size"], conf["emb_dim"][0]
                e_aux = 2 if conf["encoder_f0"] else 0
                d_in = sum(conf["emb_dim"][i] for i in range(nst))
                d_out = conf["output_size"]
                d_aux = 2 if conf["decoder_f0"] else 0
                d_aux += conf["spkr_embedding_size"] if conf["use_spkr_embedding"] else spkr_size
            else:
                e_in, e_out, e_aux = conf["emb_dim"][n - 1], conf["emb_dim"][n], 0
                d_in, d_out, d_aux = conf["emb_dim"][n], conf["emb_dim"][n - 1], 0
            common = dict(kernel_size=conf["kernel_size"][n], layers=conf["n_layers"][n] * conf["n_layers_stacks"][n],
                          stacks=conf["n_layers_stacks"][n], use_causal_conv=conf["causal"], bias=True)
            enc = HipStack(KIND_GENERATOR, e_in, e_out, aux_channels=e_aux, **common)
            dec = HipStack(KIND_GENERATOR, d_in, d_out, aux_channels=d_aux, **common)
            self.encoders.append(enc)
            self.decoders.append(dec)
            self.encoder_receptive_size += enc.receptive_field_size
            self.decoder_receptive_size += dec.receptive_field_size
            self.quantizers.append(Quantizer(self, f"quantizers.{n}.", conf["emb_dim"][n], conf["emb_size"][n],
                                             ema_flag=conf["ema_flag"], bdt_flag=True))
        # flat layout in the reference's registration order: encoders, decoders, quantizers, spkr_embedding
        for n in range(nst):
            bases.append(("enc", n, place(self.encoders[n], f"encoders.{n}.")))
        for n in range(nst):
            bases.append(("dec", n, place(self.decoders[n], f"decoders.{n}.")))
        for n in range(nst):
            q = self.quantizers[n]
            entries.extend(q.entries(off))
            off += q.n_params
        self.emb_offset = None
        if conf["use_spkr_embedding"]:
            self.emb_offset = off
            self.emb_size = conf["spkr_embedding_size"]
            entries.append(("spkr_embedding.weight", off, (spkr_size, self.emb_size)))
            off += spkr_size * self.emb_size
        self._alloc(entries, off, device)
        for kind, n, base in bases:
            (self.encoders if kind == "enc" else self.decoders)[n].bind(self, base)
        for s in self.encoders + self.decoders:
            s.init_parameters()
        for q in self.quantizers:
            q.init_parameters()
            self._bufs.update(q.make_buffers(device))
        if conf["ema_flag"]:
            bucket = parallel.EmaBucket([(q.emb_dim, q.emb_size) for q in self.quantizers], device)
            for i, q in enumerate(self.quantizers):
                q.bucket, q.slot = bucket, i
        if self.emb_offset is not None:
            self.spkr_table.normal_()  # nn.Embedding default init
        # keys in reference order for state_dict: buffers follow each quantizer's weight
        if conf["use_raw"]:
            ms = scaler["mlfb"] if conf["use_preprocessed_scaler"] else None
            f = conf["feature"]
            self.preprocess_layer = LogMelFilterBankLayer(
                fs=f["fs"], hop_size=f["hop_size"], fft_size=f["fftl"], win_length=f["win_length"],
                window=conf["raw_window_type"], center=False, n_mels=f["mlfb_dim"], fmin=f["fmin"], fmax=f["fmax"],
                scaler=ms, device=device)
            # the reference's G state_dict carries the layer's constants (crank/net/module/mlfb.py:34, 119-128):
            # same key names here, so its checkpoints load strictly and ours load there
            pl = self.preprocess_layer
            self._bufs["preprocess_layer.mlfb_layer.mel_basis"] = pl.mel_basis
            if pl.mean is not None:
                self._bufs["preprocess_layer.scaler_layer.mean"] = pl.mean
                self._bufs["preprocess_layer.scaler_layer.std"] = pl.std
        self.touch()

    # ---- plumbing ----
    def touch_codebook(self):
        """A codebook was written (the EMA blends call this): the quantizers' prepared images are stale."""
        self.codebook_epoch += 1

    def ema_codebook_ranges(self):
        """(offset, length) of every EMA-maintained codebook in the flat parameter block: the ranges an optimizer must leave
        alone for `touch(by_optimizer=True)` not to age the search images (FlatAdam checks its moments over them)."""
        return [(q.cb_offset, q.n_params) for q in self.quantizers if q.ema_flag]

    def touch(self, by_optimizer=False):
        """Any in-place parameter change.  An optimizer step leaves an EMA codebook as it is (its gradient is zero and Adam's
        update of a zero-gradient element with zero moments is exactly zero), so it only ages the images when a codebook
        is trained by gradient - or when an optimizer state with non-zero moments over a codebook was loaded
        (FlatAdam.load_state_dict sets `_trained_codebooks` then: the invariant is checked, not assumed)."""
        super().touch()
        if not by_optimizer or self._trained_codebooks:
            self.codebook_epoch += 1

    def refresh_images(self, force=False):
        """Rebuild the stale codebook images of all quantizers in ONE launch (start of a forward).  force: every image,
        whatever its epoch says (GraphedStep.step after a codebook was written between two replays: the captured step's
        first search relies on the images its own last EMA blend leaves and holds no launch that would rebuild them)."""
        todo = [q for q in self.quantizers if (q.image_stale() is not None if force else q.image_stale())]
        if todo:
            ops.vq_image_build([q.weight for q in todo], [q._img for q in todo])
            for q in todo:
                q._img_epoch = self.codebook_epoch

    def train(self, mode=True):
        super().train(mode)
        for q in self.quantizers:
            q.training = mode
        return self

    @property
    def spkr_table(self):
        o = self.emb_offset
        return self.flat.data[o: o + self.spkr_size * self.emb_size].view(self.spkr_size, self.emb_size)

    def spkr_embedding(self, h):
        """nn.Embedding-like lookup used by the trainers for D's conditioning
        (crank/net/trainer/trainer_lsgan.py:205); tracked for autograd."""
        return ops.concat_embed(None, None, self.spkr_table, h, self, self.emb_offset, self.flat)

    def _get_dec_h(self, dec_h, spkrvec):  # vqvae2.py:154-158
        """dec_h: the conditioning tensor, or a pair of tensors that are to be concatenated (can_pair_f0)."""
        a, b = dec_h if isinstance(dec_h, (tuple, list)) else (dec_h, None)
        if spkrvec is not None:
            return ops.concat_embed(a, b, self.spkr_table, spkrvec, self, self.emb_offset, self.flat)
        return dec_h if b is None else torch.cat([a, b], dim=-1)

    def _pre(self, x):
        return self.preprocess_layer(x) if self.conf["use_raw"] else x

    # ---- reference surface (all tensors channel-last) ----
    def encode(self, x, enc_h=None):  # vqvae2.py:160-169
        # the encoders write side by side into one buffer: the concatenation the speaker-adversarial net takes
        # (spkradv.py:74-76) then exists already (ops.cat_channels)
        out = []
        cur = x
        dims = self.conf["emb_dim"][: self.conf["n_vq_stacks"]]
        ebuf = torch.empty(x.shape[0], x.shape[1], sum(dims), device=x.device, dtype=torch.float32)
        col = 0
        for n in range(self.conf["n_vq_stacks"]):
            cur = self.encoders[n](cur, c=enc_h if n == 0 else None, out=(ebuf, col))
            col += dims[n]
            out.append(cur)
        return out

    def decode(self, enc, dec_h, use_ema=True, detach=False, need_decoded=True, commit_mask=None, want_commit=False):
        # vqvae2.py:171-190
        dec = None
        emb_idxs, qxs, qidxs = [], [], []
        self._commits = []
        self.refresh_images()
        # the quantized values land side by side (top stack first, the order of the concatenation the last decoder takes)
        nst = self.conf["n_vq_stacks"]
        qdims = [self.conf["emb_dim"][n] for n in reversed(range(nst))]
        qbuf = torch.empty(enc[0].shape[0], enc[0].shape[1], sum(qdims), device=enc[0].device, dtype=torch.float32) \
            if (need_decoded and not detach) else None
        qcol = 0
        pending = []  # EMA statistics of this forward: exchanged as one message after the last quantizer
        cat_in = []
        self._enc_alias = [None] * nst
        for n in reversed(range(self.conf["n_vq_stacks"])):
            # enc[n] + dec is formed inside the quantizer op; the sum replaces the caller's list entry (quirk Q6).
            # top stack: the reference adds the integer 0 (vqvae2.py:172,177), an identity
            e, qx, qi = self.quantizers[n].quantize(enc[n], add=dec, use_ema=use_ema, pending=pending, commit_mask=commit_mask,
                                                    want_commit=want_commit, alias=want_commit and not detach and ops.VQ_JOIN,
                                                    qx_out=(qbuf, qcol) if qbuf is not None else None,
                                                    # a forward whose decoded output nobody reads (need_decoded=False, no
                                                    # autograd): the code vectors are never looked at, the bottom stack's
                                                    # straight-through value neither
                                                    want_e=need_decoded or torch.is_grad_enabled(),
                                                    want_qx=need_decoded or torch.is_grad_enabled() or n != 0)
            enc[n] = self.quantizers[n].xin  # mutates the caller's list (quirk Q6)
            qcol += self.conf["emb_dim"][n]
            self._commits.append(self.quantizers[n].commit)
            self._enc_alias[n] = self.quantizers[n].x_alias
            if n == 0:
                flush_ema(pending)
            if detach and qx is not None:
                qx = qx.detach()
            emb_idxs.append(e)
            qxs.append(qx)
            # a stack's qx feeds its own decoder and the last decoder's concatenation: the concatenation reads the alias
            cat_in.append(qx if (n == 0 or self.quantizers[n].qx_alias is None) else self.quantizers[n].qx_alias)
            qidxs.append(qi)
            if n != 0:
                dec = self.decoders[n](qx, c=None)
            elif need_decoded:
                dec = self.decoders[n](ops.cat_channels(cat_in), c=dec_h)
            else:
                dec = None  # nothing downstream of the last decoder has a side effect (no quantizer, no EMA) - see forward()
        return enc, dec, emb_idxs, qxs, qidxs

    @staticmethod
    def make_dict(enc, dec, emb_idxs, qidxs, enc_unmod):  # vqvae2.py:197-209 (already (B,T,D))
        return {
            "encoded": list(enc),
            "encoded_unmod": list(enc_unmod) if enc_unmod is not None else None,
            "decoded": dec,
            "emb_idx": emb_idxs[::-1],
            "qidx": qidxs[::-1],
        }

    can_reuse_encoded = True  # forward(encoded=previous_result["encoder_out"] or encode_out(x, enc_h))

    def encode_out(self, x, enc_h):
        """The encoders alone, in the form ``forward(encoded=...)`` takes: (parameter version, autograd on, outputs, identity
        of x, identity of enc_h)."""
        return (self.version, torch.is_grad_enabled(), tuple(self.encode(self._pre(x), enc_h=enc_h)), id(x), id(enc_h))

    def forward(self, x, enc_h, dec_h, spkrvec=None, use_ema=True, encoder_detach=False, need_decoded=True,
                commit_mask=None, want_commit=False, encoded=None):
        """need_decoded=False (not in the reference): the caller only reads the encoder side of the result and wants
        the EMA side effect - the speaker-adversarial update (trainer_vqvae.py:163-184) runs a full forward and uses
        ``encoded`` alone.  The last decoder then is dead code: its output is discarded and it updates nothing, so it
        is not launched; every quantizer (and the decoders in front of one) still runs.
        want_commit=True (not in the reference): the result carries ``commit[n]``, the masked (commit_mask, frames)
        mean of (encoded[n] - emb_idx[n].detach())^2 the trainers otherwise form themselves.
        encoded (not in the reference): ``result["encoder_out"]`` of an earlier forward on the SAME x, enc_h and parameters
        (the caller's promise; a parameter update in between is noticed and the encoders run again).  The encoders are
        deterministic functions of those three, so a second forward that differs only in the decoder's conditioning - the
        adversarial pass of the GAN trainers, trainer_lsgan.py:122-131 - recomputes identical tensors; reusing them also
        sends both passes' gradients through ONE encoder backward instead of two (the same sums)."""
        # (outputs recorded without autograd cannot serve a forward that will be differentiated; the reverse is fine)
        # ... and outputs of other features / conditioning are not this call's: the tuple names the objects it was computed from
        if encoded is not None and (encoded[0] != self.version or (torch.is_grad_enabled() and not encoded[1])
                                    or encoded[3] != id(x) or encoded[4] != id(enc_h)):
            encoded = None
        if encoded is not None:
            enc = list(encoded[2])
        else:
            enc = self.encode(self._pre(x), enc_h=enc_h)
        encoder_out = encoded if encoded is not None else (self.version, torch.is_grad_enabled(), tuple(enc), id(x), id(enc_h))
        dec_h = self._get_dec_h(dec_h, spkrvec) if need_decoded else None
        enc_unmod = list(enc)  # the encoder outputs themselves: decode() rebinds, never writes in place
        enc, dec, emb_idxs, _, qidxs = self.decode(enc, dec_h, use_ema=use_ema, detach=encoder_detach,
                                                   need_decoded=need_decoded, commit_mask=commit_mask,
                                                   want_commit=want_commit)
        # (the encoder outputs as the quantizer ops handed them on, where they did: same values, same memory - whoever
        # reads them next to the quantizers sends its gradient into the quantizers' backward launch, see decode())
        enc_unmod = [a if a is not None else t for a, t in zip(self._enc_alias, enc_unmod)]
        out = self.make_dict(enc, dec, emb_idxs, qidxs, enc_unmod)
        out["encoder_out"] = encoder_out
        if want_commit and all(c is not None for c in self._commits):
            out["commit"] = self._commits[::-1]
        return out

    def cycle_forward(self, x, org_enc_h, org_dec_h, cv_enc_h, cv_dec_h, org_spkrvec, cv_spkrvec, encoded=None):
        # vqvae2.py:101-152; encoded: encode_out(x, org_enc_h) of the same parameters - the first cycle's first encode (see forward)
        # ... and outputs of other features / conditioning are not this call's: the tuple names the objects it was computed from
        if encoded is not None and (encoded[0] != self.version or (torch.is_grad_enabled() and not encoded[1])
                                    or encoded[3] != id(x) or encoded[4] != id(org_enc_h)):
            encoded = None
        x = self._pre(x) if encoded is None else x
        org_dec_h = self._get_dec_h(org_dec_h, org_spkrvec)
        cv_dec_h = self._get_dec_h(cv_dec_h, cv_spkrvec)
        outputs = []
        for cyc in range(self.conf["n_cycles"]):
            enc = list(encoded[2]) if (cyc == 0 and encoded is not None) else self.encode(x, enc_h=org_enc_h)
            org_unmod, cv_unmod = list(enc), list(enc)
            org_enc, org_dec, org_emb, _, org_q = self.decode(enc, org_dec_h)
            # the reference hands the SAME (already offset) list to the second decode and
            # both dicts alias it (quirk Q6): snapshot semantics are reproduced by sharing
            cv_enc, cv_dec, cv_emb, _, cv_q = self.decode(enc, cv_dec_h)
            enc2 = self.encode(cv_dec, enc_h=cv_enc_h)
            recon_unmod = list(enc2)
            recon_enc, recon_dec, recon_emb, _, recon_q = self.decode(enc2, org_dec_h)
            outputs.append({
                "org": self.make_dict(org_enc, org_dec, org_emb, org_q, org_unmod),
                "cv": self.make_dict(cv_enc, cv_dec, cv_emb, cv_q, cv_unmod),
                "recon": self.make_dict(recon_enc, recon_dec, recon_emb, recon_q, recon_unmod),
            })
            x = recon_dec.clone().detach()
        return outputs
