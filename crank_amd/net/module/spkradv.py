"""Speaker-adversarial network behind a gradient-reversal layer
(crank/net/module/spkradv.py:20-81).  The reversal (-lambda on the way back,
spkradv.py:63-72) is folded into the stack's input-gradient scale, so it costs nothing.
"""
import torch

from ... import ops

from .flat import FlatModel
from .pwg import KIND_PLAIN, HipStack


class SpeakerAdversarialNetwork(FlatModel):
    def __init__(self, conf, spkr_size=0, device="cuda"):
        super().__init__()
        self.conf = conf
        self.spkr_size = spkr_size
        self.scale = float(conf["spkradv_lambda"])
        self.classifier = HipStack(  # spkradv.py:49-60
            KIND_PLAIN,
            in_channels=sum(conf["emb_dim"][: conf["n_vq_stacks"]]),
            out_channels=spkr_size,
            kernel_size=conf["spkradv_kernel_size"],
            layers=conf["n_spkradv_layers"],
            conv_channels=64,
            bias=True,
            negative_slope=0.2,
        )
        self._alloc(self.classifier.entries("classifier.", 0), self.classifier.n_params, device)
        self.classifier.bind(self, 0)
        self.classifier.init_parameters()

    def forward(self, x, detach=False):
        """x: list of (B,T,emb_dim) encodings -> (B,T,n_spkrs) logits."""
        x = ops.cat_channels(x)
        if detach:
            x = x.detach()
        return self.classifier(x, dx_scale=-self.scale)

    def forward_ce(self, x, target, detach=False, ignore_index=-100):
        """cross entropy of forward(x, detach) against target (B,T) as one op (the reference's trainers compose the two,
        trainer_vqvae.py:177-184, :294-315)."""
        x = ops.cat_channels(x)
        if detach:
            x = x.detach()
        return self.classifier.ce(x, target, dx_scale=-self.scale, ignore_index=ignore_index)
