"""Model construction, checkpoint I/O and the training entry point
(crank/bin/train.py: get_model :56-131, load_checkpoint :134-142, main :145-231).

Two data sides: ``--scpdir/--featdir`` is the reference's (list directories, feats.scp,
scaler.pkl; crank/bin/train.py:166-204) with the corpus packed into HBM and every batch assembled on
the device (crank_amd/net/trainer/dataset.py; reading HDF5 needs h5py, which the benchmark image does
not carry); without them the loader yields synthetic batches of the same layout (what bench.py
measures).  Everything from ``get_model`` on is the same product path.
"""
import argparse
import logging
import random
import re
import sys
from pathlib import Path

import numpy as np
import torch

from ..net.module.pwg import ParallelWaveGANDiscriminator, ResidualParallelWaveGANDiscriminator
from ..net.module.spkradv import SpeakerAdversarialNetwork
from ..net.module.vqvae2 import VQVAE2
from ..net.trainer import TrainerWrapper
from ..net.trainer.utils import get_criterion, get_optimizer, get_scheduler
from ..synthetic import make_batch
from ..utils import load_yaml, open_featsscp, open_scpdir


def get_model(conf, spkr_size=0, device="cuda", scaler=None):
    if torch.device(device).type != "cuda":
        raise RuntimeError("crank_amd models run on an MI355X (device='cuda' under ROCm); there is no CPU path")
    models = {"G": VQVAE2(conf, spkr_size=spkr_size, scaler=scaler, device=device)}
    if conf["use_spkradv_training"]:
        models["SPKRADV"] = SpeakerAdversarialNetwork(conf, spkr_size, device=device)
    if conf["use_spkr_classifier"]:
        models["C"] = ParallelWaveGANDiscriminator(
            in_channels=conf["input_size"], out_channels=spkr_size, kernel_size=conf["spkr_classifier_kernel_size"],
            layers=conf["n_spkr_classifier_layers"], conv_channels=64, dilation_factor=1,
            nonlinear_activation="LeakyReLU", nonlinear_activation_params={"negative_slope": 0.2}, bias=True,
            use_weight_norm=True, device=device)
    if conf["trainer_type"] in ["lsgan", "cyclegan", "stargan"]:
        cin = conf["input_size"] + (1 if conf["use_D_uv"] else 0)
        if conf["use_D_spkrcode"]:
            cin += conf["spkr_embedding_size"] if conf["use_spkr_embedding"] else spkr_size
        if conf["gan_type"] != "lsgan":
            raise ValueError("gan_type must be lsgan (the reference leaves other types undefined, train.py:103-104)")
        cout = 1 + (spkr_size if conf["acgan_flag"] else 0)
        if not conf["use_residual_network"]:
            raise NotImplementedError("use_residual_network: false is broken in the reference "
                                      "(train.py:121 multiplies an int by a list) and is not offered")
        models["D"] = ResidualParallelWaveGANDiscriminator(
            in_channels=cin, out_channels=cout, kernel_size=conf["discriminator_kernel_size"],
            layers=conf["n_discriminator_layers"] * conf["n_discriminator_stacks"],
            stacks=conf["n_discriminator_stacks"], dropout=conf["discriminator_dropout"], device=device)
    return models


def load_checkpoint(model, checkpoint):
    state = torch.load(checkpoint, map_location="cpu")
    model["G"].load_state_dict(state["model"]["G"])
    for m in ["D", "C", "SPKRADV"]:
        if m in state["model"] and m in model:
            model[m].load_state_dict(state["model"][m])
    return model, state["steps"]


class SyntheticLoader:
    """Endless stream of synthetic batches with the dataset's dict layout."""

    def __init__(self, conf, n_spkrs, device, n_batches=None, seed=1234):
        self.conf, self.n_spkrs, self.device, self.n_batches, self.seed = conf, n_spkrs, device, n_batches, seed

    def __iter__(self):
        i = 0
        dim = self.conf["input_size"]
        while self.n_batches is None or i < self.n_batches:
            yield make_batch(self.conf["batch_size"], self.conf["batch_len"], self.n_spkrs, in_dim=dim,
                             out_dim=self.conf["output_size"], seed=self.seed + i, use_raw=self.conf["use_raw"],
                             fftl=self.conf["feature"]["fftl"], hop_size=self.conf["feature"]["hop_size"],
                             device=self.device)
            i += 1


def load_recipe_data(conf, scpdir, featdir, flag="train", featsscp=None, reader=None, device="cuda"):
    """crank/bin/train.py:166-181,204: scp dict, scaler.pkl and the dataloader dict of a prepared recipe."""
    import joblib

    from ..net.trainer.utils import get_dataloader

    featdir = Path(featdir) / conf["feature"]["label"]
    scp = {}
    for phase in ["train", "dev", "eval"]:
        scp[phase] = open_scpdir(Path(scpdir) / phase)
        scp[phase]["feats"] = open_featsscp(featdir / phase / "feats.scp")
    if flag == "eval" and featsscp not in (None, "None"):
        scp["eval"]["feats"] = open_featsscp(featsscp)
    scaler = joblib.load(featdir / "scaler.pkl")
    return scp, scaler, get_dataloader(conf, scp, scaler, flag=flag, reader=reader, device=device)


def build_trainer(conf, n_spkrs, expdir, device="cuda", resume=0, checkpoint=None, grad_reduce_fn=None, scaler=None,
                  dataloader=None):
    model = get_model(conf, n_spkrs, device, scaler=scaler)
    if checkpoint is not None:
        model, resume = load_checkpoint(model, checkpoint)
    optimizer = get_optimizer(conf, model, grad_reduce_fn=grad_reduce_fn)
    criterion = get_criterion(conf, device=device)
    scheduler = get_scheduler(conf, optimizer)
    if dataloader is None:
        spkrs = {f"spk{i}": i for i in range(n_spkrs)}
        dataloader = {"spkrs": spkrs, "train": SyntheticLoader(conf, n_spkrs, device)}
    return TrainerWrapper(conf["trainer_type"], model=model, optimizer=optimizer, criterion=criterion,
                          dataloader=dataloader, writer=None, expdir=expdir, conf=conf, feat_conf=conf["feature"],
                          scheduler=scheduler, scaler=scaler, resume=resume, device=device, n_jobs=1)


def main():
    ap = argparse.ArgumentParser(description="crank_amd training driver (a prepared recipe, or synthetic data)")
    ap.add_argument("--scpdir", type=str, default=None, help="scp directory of the recipe (with --featdir)")
    ap.add_argument("--featdir", type=str, default=None, help="feature directory of the recipe")
    ap.add_argument("--flag", default="train", choices=["train"])
    ap.add_argument("--conf", type=str, default=None, help="recipe YAML merged over the defaults")
    ap.add_argument("--expdir", type=str, default="exp")
    ap.add_argument("--n_spkrs", type=int, default=14)
    ap.add_argument("--n_steps", type=int, default=None)
    ap.add_argument("--checkpoint", type=str, default=None)
    args = ap.parse_args()
    logging.basicConfig(level=logging.INFO, stream=sys.stdout, format="%(asctime)s %(levelname)s: %(message)s")
    random.seed(1234)
    np.random.seed(1234)
    torch.manual_seed(1234)  # crank/bin/train.py:49-51
    conf = load_yaml(args.conf)
    if args.n_steps is not None:
        conf["n_steps"] = args.n_steps
    ckpt = args.checkpoint
    if ckpt is None:
        found = sorted(Path(args.expdir).glob("checkpoint_*steps.pkl"),
                       key=lambda p: int(re.findall(r"checkpoint_(\d+)steps", p.name)[0]))
        ckpt = str(found[-1]) if found else None
    if (args.scpdir is None) != (args.featdir is None):
        ap.error("--scpdir and --featdir go together")
    if args.scpdir is not None:
        scp, scaler, dataloader = load_recipe_data(conf, args.scpdir, args.featdir, flag=args.flag)
        trainer = build_trainer(conf, len(scp["train"]["spkrs"]), args.expdir, checkpoint=ckpt, scaler=scaler,
                                dataloader=dataloader)
    else:
        trainer = build_trainer(conf, args.n_spkrs, args.expdir, checkpoint=ckpt)
    trainer.run("train")


if __name__ == "__main__":
    main()
