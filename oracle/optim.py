"""TEST INFRASTRUCTURE - CPU restatement of the two non-Adam optimizers the reference's factory can return
(crank/net/trainer/utils.py:40-50: ``toptim.RAdam(model.parameters(), lr=lr)`` and ``Lamb(model.parameters(), lr=lr)``).

Both come from third-party packages that are absent from /root/reference and from this image (tools/requirements.txt
names ``torch-optimizer`` and ``pytorch_lamb`` without a version): **parity unpinned** against those packages.  What is
restated is their published update rule with the defaults the reference's call leaves in place:

* ``RAdam``  - Liu et al., "On the Variance of the Adaptive Learning Rate and Beyond" (ICLR 2020), Alg. 2, as
  torch_optimizer states it: betas (0.9, 0.999), eps 1e-8, weight_decay 0; the second moment is updated before the first,
  ``N_sma >= 5`` selects the rectified update, the bias correction of the second moment sits inside the step size and
  ``eps`` is added to the UNcorrected ``sqrt(v)``.  Cross-check available here: ``torch.optim.RAdam`` (same algorithm, the
  bias correction on the other side of ``eps``, threshold ``> 5``) - tests/test_oracle_cpu.py holds the two together.
* ``Lamb``   - You et al., "Large Batch Optimization for Deep Learning" (ICLR 2020), Alg. 2, as pytorch_lamb states it:
  betas (0.9, 0.999), eps 1e-6, weight_decay 0, no bias correction ("paper v3 does not use debiasing"), per parameter
  tensor ``trust_ratio = clamp(||w||, 0, 10) / ||adam_step||`` and 1 where either norm is 0.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import torch


class RAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self):
        for group in self.param_groups:
            lr, (beta1, beta2), eps, wd = group["lr"], group["betas"], group["eps"], group["weight_decay"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                grad = p.grad
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                exp_avg, exp_avg_sq = st["exp_avg"], st["exp_avg_sq"]
                exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
                exp_avg.mul_(beta1).add_(grad, alpha=1 - beta1)
                st["step"] += 1
                t = st["step"]
                beta2_t = beta2 ** t
                n_sma_max = 2.0 / (1.0 - beta2) - 1.0
                n_sma = n_sma_max - 2.0 * t * beta2_t / (1.0 - beta2_t)
                if n_sma >= 5:
                    step_size = lr * math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_sma_max - 4) * (n_sma - 2) / n_sma
                                               * n_sma_max / (n_sma_max - 2)) / (1 - beta1 ** t)
                else:
                    step_size = lr / (1 - beta1 ** t)
                if wd != 0:
                    p.add_(p, alpha=-wd * lr)
                if n_sma >= 5:
                    p.addcdiv_(exp_avg, exp_avg_sq.sqrt().add_(eps), value=-step_size)
                else:
                    p.add_(exp_avg, alpha=-step_size)


class Lamb(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self):
        for group in self.param_groups:
            lr, (beta1, beta2), eps, wd = group["lr"], group["betas"], group["eps"], group["weight_decay"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                grad = p.grad
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                exp_avg, exp_avg_sq = st["exp_avg"], st["exp_avg_sq"]
                st["step"] += 1
                exp_avg.mul_(beta1).add_(grad, alpha=1 - beta1)
                exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
                weight_norm = p.pow(2).sum().sqrt().clamp(0, 10)
                adam_step = exp_avg / exp_avg_sq.sqrt().add(eps)
                if wd != 0:
                    adam_step.add_(p, alpha=wd)
                adam_norm = adam_step.pow(2).sum().sqrt()
                trust_ratio = 1.0 if (weight_norm == 0 or adam_norm == 0) else float(weight_norm / adam_norm)
                st["weight_norm"], st["adam_norm"], st["trust_ratio"] = weight_norm, adam_norm, trust_ratio
                p.add_(adam_step, alpha=-lr * trust_ratio)


def make_optimizer(optim_type, params, lr):
    """crank/net/trainer/utils.py:40-50 ``return_optim``."""
    if optim_type == "adam":
        return torch.optim.Adam(params, lr=lr)
    if optim_type == "radam":
        return RAdam(params, lr=lr)
    if optim_type == "lamb":
        return Lamb(params, lr=lr)
    raise ValueError("Invalid optimizer type")
