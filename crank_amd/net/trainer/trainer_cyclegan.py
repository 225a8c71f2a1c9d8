"""CycleGAN trainer: cyclic generator update with adversarial terms on the ``org`` and
``cv`` decodings, discriminator trained on real vs. a randomly chosen fake.
Follows crank/net/trainer/trainer_cyclegan.py (update_G :52-76, update_D :78-93,
calculate_cycleadv_loss :95-123, calculate_cycle_discriminator_loss :125-179),
including its quirks: D is conditioned on the ``cv`` speaker code for both
adversarial terms, and the discriminator loss always reads ``outputs[0]``.
The Python RNG decides which fake D sees; under data parallelism every rank seeds
``random`` identically so the choice is shared.
"""
import torch

from .trainer_lsgan import LSGANTrainer


class CycleGANTrainer(LSGANTrainer):
    def _classifier_is_independent(self):
        return False  # update_G classifies converted features with C in every step of the GAN phase

    def _draw_step_choices(self):
        # update_D shows the discriminator one of the two fakes per cycle (trainer_cyclegan.py:166)
        if not self.gan_flag:
            return ()
        return tuple(self.rng.choice(["org_fake", "cv_fake"]) for _ in range(self.conf["n_cycles"]))

    def update_G(self, batch, loss, phase="train"):
        enc_h, dec_h, spkrvec = self._cond(batch)
        enc_h_cv, dec_h_cv, spkrvec_cv = self._cond(batch, cv=True)
        self._discard_grads("SPKRADV", True)
        self._discard_grads("D", True)
        self._discard_grads("C", True)
        outs = self.model["G"].cycle_forward(batch["in_feats"], enc_h, dec_h, enc_h_cv, dec_h_cv, spkrvec, spkrvec_cv,
                                             **self._shared_encoded(batch, enc_h))
        loss = self.calculate_vqvae_loss(batch, outs[0]["org"], loss)
        loss = self.calculate_cyclevqvae_loss(batch, outs, loss)
        if self.conf["use_spkradv_training"]:
            loss = self.calculate_spkradv_loss(batch, outs[0]["org"], loss, phase=phase)
        loss = self.calculate_cycleadv_loss(batch, outs, loss)
        if phase == "train" and not self.stop_generator:
            self.step_model(loss, model="G")
        for m in ["SPKRADV", "D", "C"]:
            self._discard_grads(m, False)
        return loss

    def update_D(self, batch, loss, phase="train"):
        enc_h, dec_h, spkrvec = self._cond(batch)
        enc_h_cv, dec_h_cv, spkrvec_cv = self._cond(batch, cv=True)
        grad_on = torch.is_grad_enabled()
        shared = self._shared_encoded(batch, enc_h, need_grad=self._D_update_shares_with_G(phase))  # (the first encode of the cycle: the same tensors in both updates of the step)
        with torch.no_grad():  # decodings are only used detached
            outs = self.model["G"].cycle_forward(batch["in_feats"], enc_h, dec_h, enc_h_cv, dec_h_cv, spkrvec,
                                                 spkrvec_cv, **shared)
        with torch.set_grad_enabled(grad_on):
            loss = self.calculate_cycle_discriminator_loss(batch, outs, loss)
            if phase == "train":
                self.step_model(loss, model="D")
        return loss

    def calculate_cycleadv_loss(self, batch, outputs, loss):
        mask = batch["decoder_mask"]
        for c in range(self.conf["n_cycles"]):
            for io in ["org", "cv"]:
                lbl = f"{c}cyc_{io}"
                d_out = self._discriminate(self.get_D_inputs(batch, outputs[c][io]["decoded"], label="cv"))
                if self.conf["acgan_flag"]:
                    d_out, spkr_cls = torch.split(d_out, [1, self.n_spkrs], dim=2)
                    loss[f"D_acgan_adv_{lbl}"] = self._ce(spkr_cls, batch[f"{io}_h"])
                    loss.add("G", self.conf["alpha"]["acgan"], loss[f"D_acgan_adv_{lbl}"])
                    # the reference mask-selects only on this branch (trainer_cyclegan.py:110):
                    loss[f"D_adv_{lbl}"] = self._masked_const_mse(d_out, mask, 1)
                else:
                    # ... and takes an UNMASKED mean otherwise (:117-119)
                    loss[f"D_adv_{lbl}"] = self.criterion["mse"](d_out, torch.ones_like(d_out))
                loss.add("G", self.conf["alpha"]["adv"], loss[f"D_adv_{lbl}"])
        return loss

    def calculate_cycle_discriminator_loss(self, batch, outputs, loss):
        a = self.conf["alpha"]
        for c in range(self.conf["n_cycles"]):
            lbl = f"{c}cyc"
            sample = dict(zip(("real", "org_fake", "cv_fake"), self._discriminate_many([
                self.get_D_inputs(batch, batch["in_feats"], label="org"),
                self.get_D_inputs(batch, outputs[0]["org"]["decoded"].detach(), label="org"),
                self.get_D_inputs(batch, outputs[0]["cv"]["decoded"].detach(), label="cv")])))
            if self.conf["acgan_flag"]:
                for k in list(sample.keys()):
                    h = batch["org_h"] if k in ["real", "org_fake"] else batch["cv_h"]
                    sample[k], spkr_cls = torch.split(sample[k], [1, self.n_spkrs], dim=2)
                    loss[f"D_ce_{k}_{lbl}"] = self._ce(spkr_cls, h)
                    if not (self.conf["use_real_only_acgan"] and k == "org_fake"):
                        loss.add("D", a["acgan"], loss[f"D_ce_{k}_{lbl}"])
            loss[f"D_real_{lbl}"] = self._masked_const_mse(sample["real"], batch["decoder_mask"], 1)
            fake_key = self._choose(["org_fake", "cv_fake"])
            mask = batch["cycle_decoder_mask"] if fake_key == "org_fake" else batch["decoder_mask"]
            loss[f"D_fake_{lbl}"] = self._masked_const_mse(sample[fake_key], mask, 0)
            loss.add("D", a["fake"], loss[f"D_fake_{lbl}"])
            loss.add("D", a["real"], loss[f"D_real_{lbl}"])
        return loss
