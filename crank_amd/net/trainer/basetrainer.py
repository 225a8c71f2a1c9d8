"""Step loop and conditioning helpers shared by the four trainers.

Mirrors the parts of crank/net/trainer/basetrainer.py that drive the optimisation
step (:26-46 TrainerWrapper, :117-129 run, :131-140 save_model, :153-167 _tr_step,
:200-309 loss bookkeeping and conditioning).  Waveform generation / HDF5 dumping of
dev and eval batches (:322-435) is outside the hot path and
is replaced by returning the converted features.

The trainers only rely on the object contract of SURVEY.md section 8b (model /
criterion / optimizer / scheduler dicts), so the same classes also drive the CPU
oracle modules in the parity tests and in bench.py's cpu_baseline leg.
"""
import logging
from pathlib import Path

import torch

from ... import parallel


def TrainerWrapper(trainer_type, **ka):
    from . import CycleGANTrainer, LSGANTrainer, StarGANTrainer, VQVAETrainer

    table = {"vqvae": VQVAETrainer, "lsgan": LSGANTrainer, "cyclegan": CycleGANTrainer, "stargan": StarGANTrainer}
    if trainer_type not in table:
        raise NotImplementedError("conf['trainer_type']: {} is not supported.".format(trainer_type))
    return table[trainer_type](**ka)


def to_device(batch, device):
    return {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}


class LossBook(dict):
    """The reference's loss dict (basetrainer.py:200-206): totals per model start at 0.0
    and become tensors as terms are added."""

    def __init__(self):
        super().__init__(objective=0.0, G=0.0, D=0.0, C=0.0, SPKRADV=0.0)


class BaseTrainer(object):
    def __init__(self, model, optimizer, criterion, dataloader, writer, expdir, conf, feat_conf, scheduler=None,
                 scaler=None, resume=0, device="cuda", n_jobs=-1):
        self.model, self.optimizer, self.criterion = model, optimizer, criterion
        if parallel.is_dist():  # loss means become shares of the global mean (parallel.py, C3)
            self.criterion = parallel.wrap_criterion(criterion)
        self.dataloader, self.writer = dataloader, writer
        self.expdir = Path(expdir)
        self.conf, self.feat_conf = conf, feat_conf
        self.scheduler, self.scaler = scheduler, scaler
        self.device, self.n_jobs = device, n_jobs
        self.spkrs = dataloader["spkrs"]
        self.n_spkrs = len(self.spkrs)
        self.resume_steps = self.steps = resume
        self._step_schedulers(explicit=True)
        self.finish_train = False

    # ------------------------------------------------------------------ loop
    def run(self, flag="train", tdir=None):
        self.flag = flag
        if flag != "train":
            return self._run_eval(flag)
        while not self.finish_train:
            self._tr_step()
        logging.info("Finish training")

    def _tr_step(self):
        for batch in self.dataloader["train"]:
            batch = to_device(batch, self.device)
            values = self.train(batch, phase="train")
            if self.steps % self.conf["n_steps_print_loss"] == 0:
                self._print_loss_values(values, phase="train")
            self._dev_step()
            if self.resume_steps != self.steps and self.steps % self.conf["n_steps_save_model"] == 0:
                self.save_model()
            self.steps += 1
            self._step_schedulers(explicit=True)
            if self.steps > self.conf["n_steps"]:
                self.finish_train = True
            self.check_custom_start()
            if self.finish_train:
                break

    def _dev_step(self):
        ds = self.conf["dev_steps"]
        if self.steps % ds == 0 and self.steps > ds - 1 and self.steps != self.resume_steps and "dev" in self.dataloader:
            values = None
            for i, batch in enumerate(self.dataloader["dev"]):
                values = self.dev(to_device(batch, self.device))
                if i > 0:
                    break
            if values is not None:
                self._print_loss_values(values, phase="dev")

    def _run_eval(self, flag):
        out = []
        if flag == "eval":
            for batch in self.dataloader["eval"]:
                out.append(self.eval(to_device(batch, self.device)))
        elif flag == "reconstruction":
            for key in ["train", "dev"]:
                for batch in self.dataloader[key]:
                    out.append(self.reconstruction(to_device(batch, self.device)))
        return out

    def _step_schedulers(self, explicit=False):
        if self.scheduler is None:
            return
        scheds = self.scheduler.values() if isinstance(self.scheduler, dict) else [self.scheduler]
        for s in scheds:
            s.step(self.steps)

    def save_model(self):
        """checkpoint_{steps}steps.pkl = {"steps", "model": {name: state_dict}}
        (basetrainer.py:131-140); optimizer state is not saved, like the reference."""
        self.expdir.mkdir(parents=True, exist_ok=True)
        state = {"steps": self.steps, "model": {"G": self.model["G"].state_dict()}}
        for m in ["SPKRADV", "D", "C"]:
            if m in self.model:
                state["model"][m] = self.model[m].state_dict()
        torch.save(state, self.expdir / "checkpoint_{}steps.pkl".format(self.steps))

    def check_custom_start(self):
        pass

    # ------------------------------------------------------------------ losses
    def _get_loss_dict(self):
        if "_reset" in self.criterion:
            self.criterion["_reset"]()
        return LossBook()

    def _parse_loss(self, loss):
        """floats for logging (basetrainer.py:208-215) with ONE device->host copy instead
        of an .item() per key."""
        values = dict(LossBook())
        keys = [k for k, v in loss.items() if isinstance(v, torch.Tensor)]
        if keys:
            vec = torch.stack([loss[k].detach().reshape(()).float() for k in keys])
            if parallel.is_dist():  # per-rank shares -> global values
                torch.distributed.all_reduce(vec)
            flat = vec.tolist()
            for k, v in zip(keys, flat):
                values[k] = values.get(k, 0.0) + v
        for k in loss:
            values.setdefault(k, 0.0)
        return values

    def _print_loss_values(self, values, phase="train"):
        logging.info("{} iterations: {}".format(phase, self.steps))
        for k, v in sorted(values.items()):
            if v != 0.0:
                logging.info("{}: {}".format(k, v))

    def _flush_writer(self, loss, phase):
        if self.writer is None or self.steps % self.conf["n_steps_print_loss"] != 0:
            return
        w = self.writer.get(phase) if isinstance(self.writer, dict) else None
        if w is None:
            return
        for k, v in loss.items():
            if isinstance(v, torch.Tensor):
                w.add_scalar("loss/{}".format(k), v.item(), self.steps)
        w.flush()

    # ------------------------------------------------------------------ conditioning
    def _get_enc_h(self, batch, use_cvfeats=False, cv_spkr_name=None):  # basetrainer.py:253-258
        if self.conf["encoder_f0"]:
            return self._get_f0_condition(batch, cv_spkr_name, use_cvfeats)
        return None

    def _get_dec_h(self, batch, use_cvfeats=False, cv_spkr_name=None):  # basetrainer.py:260-275
        h, h_onehot = self._get_spkr_conditions(batch, cv_spkr_name, use_cvfeats)
        f0 = self._get_f0_condition(batch, cv_spkr_name, use_cvfeats) if self.conf["decoder_f0"] else None
        if not self.conf["use_spkr_embedding"]:
            return (torch.cat([f0, h_onehot], dim=-1) if f0 is not None else h_onehot), None
        return f0, h

    def _get_f0_condition(self, batch, cv_spkr_name, use_cvfeats=False):  # basetrainer.py:277-289
        if cv_spkr_name is not None:
            lcf0 = self._get_cvf0(batch, cv_spkr_name)
        else:
            lcf0 = batch["cv_lcf0"] if use_cvfeats else batch["lcf0"]
        return torch.cat([lcf0, batch["uv"]], dim=-1)

    def _get_spkr_conditions(self, batch, cv_spkr_name, use_cvfeats=False):  # basetrainer.py:291-309
        if cv_spkr_name is not None:
            B, T, _ = batch["in_feats"].shape
            num = self.spkrs[cv_spkr_name]
            h = torch.full((B, T), num, dtype=torch.long, device=batch["in_feats"].device)
            h_onehot = torch.zeros(B, T, self.n_spkrs, device=h.device)
            h_onehot[..., num] = 1.0
        else:
            key = "cv" if use_cvfeats else "org"
            h = batch[f"{key}_h"].clone()
            h_onehot = batch[f"{key}_h_onehot"]
        h[:, :] = h[:, 0:1]  # overwrite the -100 pads with the utterance's label
        return h, h_onehot

    def _get_cvf0(self, batch, spkr_name):
        """F0 linear transform org -> spkr in the log domain on the scaler statistics
        (basetrainer.py:311-320 + dataset.convert_f0)."""
        if self.scaler is None:
            raise ValueError("converting F0 to a named speaker needs the feature scaler")
        import numpy as np

        out = []
        for n in range(batch["in_feats"].size(0)):
            lcf0 = self.scaler["lcf0"].inverse_transform(batch["lcf0"][n].detach().cpu().numpy())
            org, cv = self.scaler[batch["org_spkr_name"][n]]["lcf0"], self.scaler[spkr_name]["lcf0"]
            conv = (lcf0 - org.mean_) / np.sqrt(org.var_) * np.sqrt(cv.var_) + cv.mean_
            out.append(torch.tensor(self.scaler["lcf0"].transform(conv)))
        return torch.stack(out, dim=0).float().to(batch["in_feats"].device)
