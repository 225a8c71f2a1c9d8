"""LSGAN trainer: after ``n_steps_gan_start`` every step updates the discriminator D
(real -> 1, fake -> 0) and the generator (VQ-VAE losses + adversarial term, fake -> 1),
in the order ``train_first`` says; before that it is the plain VQ-VAE step.
Follows crank/net/trainer/trainer_lsgan.py (train :59-72, forward_lsgan :74-82,
update_G :84-113, update_D :115-144, loss terms :146-181, _check_gan_start :183-192,
get_D_inputs :194-206).
"""
import torch

from .trainer_vqvae import VQVAETrainer


class LSGANTrainer(VQVAETrainer):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.gan_flag = False
        self._check_gan_start()
        self.stop_generator = False

    def check_custom_start(self):
        self._check_cycle_start()
        self._check_gan_start()

    def _G_step_is_last_of_main_update(self):
        # GAN phase with train_first G: update_D follows update_G and runs the generator with its new parameters;
        # a stopped generator takes no step at all
        return not (self.gan_flag and (self.conf["train_first"] == "G" or self.stop_generator))

    def _classifier_is_independent(self):
        # the GAN phase keeps the classifier's update in line: next to the discriminator's kernels the second stream
        # measured slower (configs[2]: 3.87 -> 3.91 ms), next to the plain VQ-VAE step faster (1.57 -> 1.53 ms)
        return super()._classifier_is_independent() and not self.gan_flag

    def _main_update(self, batch, loss, phase):  # trainer_lsgan.py:59-72: once the GAN phase has begun it replaces the VQ-VAE update
        if self.gan_flag:
            return self.forward_lsgan(batch, loss, phase=phase)
        return super()._main_update(batch, loss, phase)

    def _shared_encoded(self, batch, enc_h, need_grad=True):
        """{"encoded": ...} for G.forward: the encoder outputs of this step's batch, computed once (with autograd where the
        step differentiates) and handed to every generator forward of the step that runs on the same parameters - the
        discriminator update's detached forward, the generator update's reconstruction and adversarial forwards.  The
        encoders are deterministic in (features, enc_h, parameters); VQVAE2.forward checks the parameter version itself.
        need_grad False: the caller only uses the result detached and nobody after it in this step can reuse it (the
        discriminator update behind the generator's: G's parameters have moved, the cache misses) - the encoders then run
        without autograd instead of saving planes for a backward that never comes."""
        G = self.model["G"]
        if not getattr(G, "can_reuse_encoded", False):
            return {}
        want = need_grad and torch.is_grad_enabled()
        c = getattr(self, "_enc_shared", None)
        if c is None or c[0] is not batch or c[2] is not enc_h or c[1][0] != G.version or (want and not c[1][1]):
            with torch.set_grad_enabled(want):
                c = self._enc_shared = (batch, G.encode_out(batch["in_feats"], enc_h), enc_h)
        return {"encoded": c[1]}

    def _D_update_shares_with_G(self, phase):
        """The discriminator update's encode is worth differentiating only if the generator update of this step comes after it."""
        return phase == "train" and self.conf["train_first"] != "G"

    def forward_lsgan(self, batch, loss, phase="train"):
        self._enc_shared = None
        order = [self.update_G, self.update_D] if self.conf["train_first"] == "G" else [self.update_D, self.update_G]
        for fn in order:
            loss = fn(batch, loss, phase=phase)
        self._enc_shared = None  # (the step's autograd graph and its batch are not kept alive past the step)
        loss["objective"] = 0.0
        loss.add("objective", 1.0, loss["G"])
        loss.add("objective", 1.0, loss["D"])
        return loss

    def _discriminate(self, x):
        return self.model["D"](x.transpose(1, 2)).transpose(1, 2)

    def _discriminate_many(self, xs):
        """[self._discriminate(x) for x in xs] - as ONE call over the concatenated batch where the discriminator is the HIP
        stack: its launches (first conv, gated stack, head, their data and weight gradients, the weight-norm backward) then
        exist once instead of len(xs) times.  The frames of an utterance never see another utterance (windows are cut per
        utterance), so every output equals the separate call's; the parameter gradients are the same sums in another order.
        Dropout: one seed for the joint call - independent masks per utterance all the same."""
        D = self.model["D"]
        if len(xs) < 2 or not hasattr(D, "flat") or not xs[0].is_cuda or any(x.shape != xs[0].shape for x in xs):
            return [self._discriminate(x) for x in xs]
        out = self._discriminate(torch.cat(xs, dim=0))
        return list(torch.split(out, xs[0].shape[0], dim=0))

    def _masked_const_mse(self, sample, mask, value):
        """criterion["mse"](sample.masked_select(mask), const) of the reference as a masked mean."""
        target = torch.ones_like(sample) if value == 1 else torch.zeros_like(sample)
        return self.criterion["fmse"](sample, target, mask=mask)

    def _adv_side(self, batch):
        """Decoder conditioning and speaker labels of the adversarial branch: the conversion target's when
        ``cvadv_flag`` is set, the source speaker's otherwise (trainer_lsgan.py:96-101, 118-124)."""
        cv = bool(self.conf["cvadv_flag"])
        _, dec_h, spkrvec = self._cond(batch, cv=cv)
        return dec_h, spkrvec, batch["cv_h" if cv else "org_h"]

    def update_G(self, batch, loss, phase="train"):
        enc_h, dec_h, spkrvec = self._cond(batch)
        feats, G = batch["in_feats"], self.model["G"]
        for name in ("SPKRADV", "D"):  # their weight gradients from the G step are thrown away (Q7)
            self._discard_grads(name, True)
        outputs = G.forward(feats, enc_h, dec_h, spkrvec, **self._shared_encoded(batch, enc_h))
        loss = self.calculate_vqvae_loss(batch, outputs, loss)
        if self.conf["use_spkradv_training"]:
            loss = self.calculate_spkradv_loss(batch, outputs, loss, phase=phase)
        adv_dec_h, adv_spkrvec, h = self._adv_side(batch)
        detach = self.conf["encoder_detach"]
        # (the adversarial pass differs from the first in the decoder's conditioning only: the encoders would recompute what
        # they just produced - and be differentiated twice)
        reuse = {"encoded": outputs.get("encoder_out")} if getattr(G, "can_reuse_encoded", False) else {}
        adv = G.forward(feats, enc_h, adv_dec_h, spkrvec=adv_spkrvec, use_ema=not detach, encoder_detach=detach, **reuse)
        loss = self.calculate_adv_loss(batch, adv["decoded"], h, batch["decoder_mask"], loss)
        if phase == "train" and not self.stop_generator:
            self.step_model(loss, model="G")
        for name in ("SPKRADV", "D"):
            self._discard_grads(name, False)
        return loss

    def update_D(self, batch, loss, phase="train"):
        enc_h, mask = self._cond(batch)[0], batch["decoder_mask"]  # (the step's one enc_h object: the shared encoders are keyed on it)
        dec_h, spkrvec, h = self._adv_side(batch)
        grad_on = torch.is_grad_enabled()
        shared = self._shared_encoded(batch, enc_h, need_grad=self._D_update_shares_with_G(phase))  # (with autograd only if the generator update reads them too)
        with torch.no_grad():  # only the detached decoding is used
            outputs = self.model["G"].forward(batch["in_feats"], enc_h, dec_h, spkrvec, **shared)
        with torch.set_grad_enabled(grad_on):
            real, fake = self._discriminate_many([self.get_D_inputs(batch, batch["in_feats"], label="org"),
                                                  self.get_D_inputs(batch, outputs["decoded"].detach(), label="cv")])
            loss = self.calculate_discriminator_loss(real, batch["org_h"], mask, loss, label="real")
            loss = self.calculate_discriminator_loss(fake, h, mask, loss, label="fake")
            if phase == "train":
                self.step_model(loss, model="D")
        return loss

    def calculate_adv_loss(self, batch, decoded, h, mask, loss):
        fake = self._discriminate(self.get_D_inputs(batch, decoded, label="cv"))
        if self.conf["acgan_flag"]:
            fake, spkr_cls = torch.split(fake, [1, self.n_spkrs], dim=2)
            loss = self.calculate_acgan_loss(spkr_cls, h, loss)
        loss["D_adv"] = self._masked_const_mse(fake, mask, 1)
        loss.add("G", self.conf["alpha"]["adv"], loss["D_adv"])
        return loss

    def calculate_discriminator_loss(self, sample, h, mask, loss, label="real", updates=None):
        if self.conf["acgan_flag"]:
            sample, spkr_cls = torch.split(sample, [1, self.n_spkrs], dim=2)
            loss = self.calculate_acgan_loss(spkr_cls, h, loss, label=label, model="D")
        loss[f"D_{label}"] = self._masked_const_mse(sample, mask, 1 if label == "real" else 0)
        if updates is None or label in updates:
            loss.add("D", self.conf["alpha"][label], loss[f"D_{label}"])
        return loss

    def calculate_acgan_loss(self, spkr_cls, h, loss, label="adv", model="G"):
        loss[f"D_acgan_{label}"] = self._ce(spkr_cls, h)
        if not (self.conf["use_real_only_acgan"] and label == "fake"):
            loss.add(model, self.conf["alpha"]["acgan"], loss[f"D_acgan_{label}"])
        return loss

    def _check_gan_start(self):
        if self.steps > self.conf["n_steps_gan_start"]:
            self.gan_flag = True
            if self.conf["n_steps_stop_generator"] > 0:
                self.stop_generator = True
        if self.steps > self.conf["n_steps_gan_start"] + self.conf["n_steps_stop_generator"]:
            self.stop_generator = False

    def get_D_inputs(self, batch, feats, label="org"):
        if (self.conf["use_D_spkrcode"] and self.conf["use_spkr_embedding"] and feats.is_cuda and feats.dtype == torch.float32
                and hasattr(self.model["G"], "spkr_table")):
            # [feats | uv | spkr_embedding(h).detach()] (trainer_lsgan.py:194-206) as ONE launch: the gather-and-concatenate
            # kernel of the generator's own conditioning, the table read as a constant (no gradient: the reference detaches)
            from ... import ops

            h = self._filled_labels(batch, batch[f"{label}_h"], label)
            uv = batch["uv"] if self.conf["use_D_uv"] else None
            return ops.concat_embed(feats, uv, self.model["G"].spkr_table.detach(), h, None, 0, None)
        parts = [feats]
        if self.conf["use_D_uv"]:
            parts.append(batch["uv"])
        if self.conf["use_D_spkrcode"]:
            if not self.conf["use_spkr_embedding"]:
                parts.append(batch[f"{label}_h_onehot"])
            else:
                h = batch[f"{label}_h"].clone()
                h[:, :] = h[:, 0:1]  # drop the -100 pads
                parts.append(self.model["G"].spkr_embedding(h).detach())
        return torch.cat(parts, dim=-1).float()
