"""StarGAN trainer: cyclic generator update with the adversarial term on the converted
(``cv``) decoding; D is trained on real features vs. a plain forward pass conditioned
on the conversion target, optionally updating on only one of the two per step.
Follows crank/net/trainer/trainer_stargan.py (update_G :51-80, update_D :82-118).
"""
import torch

from .trainer_lsgan import LSGANTrainer


class StarGANTrainer(LSGANTrainer):
    def _classifier_is_independent(self):
        return False  # update_G classifies converted features with C in every step of the GAN phase

    def _draw_step_choices(self):
        # with switch_update the discriminator is updated on real OR fake samples, drawn per step (trainer_stargan.py:90-93)
        if not (self.gan_flag and self.conf["switch_update"]):
            return ()
        return (self.rng.choice(["real", "fake"]),)

    def update_G(self, batch, loss, phase="train"):
        enc_h, dec_h, spkrvec = self._cond(batch)
        enc_h_cv, dec_h_cv, spkrvec_cv = self._cond(batch, cv=True)
        for m in ["SPKRADV", "D", "C"]:
            self._discard_grads(m, True)
        outs = self.model["G"].cycle_forward(batch["in_feats"], enc_h, dec_h, enc_h_cv, dec_h_cv, spkrvec, spkrvec_cv,
                                             **self._shared_encoded(batch, enc_h))
        if self.conf["use_vqvae_loss"]:
            loss = self.calculate_vqvae_loss(batch, outs[0]["org"], loss)
        loss = self.calculate_cyclevqvae_loss(batch, outs, loss)
        if self.conf["use_spkradv_training"]:
            for label in ["cv", "recon"]:
                loss = self.calculate_spkradv_loss(batch, outs[0][label], loss, label=label, phase=phase)
        loss = self.calculate_adv_loss(batch, outs[0]["cv"]["decoded"], batch["cv_h"], batch["decoder_mask"], loss)
        if phase == "train" and not self.stop_generator:
            self.step_model(loss, model="G")
        for m in ["SPKRADV", "D", "C"]:
            self._discard_grads(m, False)
        return loss

    def update_D(self, batch, loss, phase="train"):
        enc_h_cv, dec_h_cv, spkrvec_cv = self._cond(batch, cv=True)
        updates = self._choose(["real", "fake"]) if self.conf["switch_update"] else ["real", "fake"]
        grad_on = torch.is_grad_enabled()
        shared = self._shared_encoded(batch, enc_h_cv, need_grad=self._D_update_shares_with_G(phase))  # (the generator update's encoders where enc_h_cv is its enc_h: no F0 on the encoder)
        with torch.no_grad():  # only the detached decoding is used
            outputs = self.model["G"].forward(batch["in_feats"], enc_h_cv, dec_h_cv, spkrvec_cv, **shared)
        with torch.set_grad_enabled(grad_on):
            # (the real pass does not depend on the generator's forward: both samples go through D together)
            real, fake = self._discriminate_many([self.get_D_inputs(batch, batch["in_feats"], label="org"),
                                                  self.get_D_inputs(batch, outputs["decoded"].detach(), label="cv")])
            loss = self.calculate_discriminator_loss(real, batch["org_h"], batch["decoder_mask"], loss, label="real",
                                                     updates=updates)
            loss = self.calculate_discriminator_loss(fake, batch["cv_h"], batch["decoder_mask"], loss, label="fake",
                                                     updates=updates)
            if phase == "train":
                self.step_model(loss, model="D")
        return loss
