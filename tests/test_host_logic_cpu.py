"""CPU: the product trainers (host logic: loss algebra, update order, conditioning)
driven with the oracle modules must reproduce the golden step vectors that were
produced by the REFERENCE's own trainer / VQVAE2 / SpeakerAdversarialNetwork classes
(tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch

from tests.helpers import STEP_CASES, compare_losses, run_golden_case, state_summary


def _oracle_factories():
    from oracle import modules as om

    def sched(conf, optimizer):
        return {m: torch.optim.lr_scheduler.StepLR(o, conf["optim"][m]["decay_step_size"], conf["optim"][m]["decay_size"])
                for m, o in optimizer.items()}

    return (lambda conf, n: om.get_model(conf, n), om.get_optimizer, om.get_criterion, sched)


@pytest.mark.parametrize("tag", list(STEP_CASES))
def test_trainer_matches_reference_step(tag):
    torch.set_num_threads(4)
    bm, bo, bc, bs = _oracle_factories()
    losses, models, trainer, fx, post = run_golden_case(tag, bm, bo, bc, bs)
    bad = compare_losses(losses, fx, rtol=2e-4)
    assert not bad, bad
    np.testing.assert_allclose(post["decoded"].numpy(), fx["post_decoded"], rtol=1e-3, atol=1e-4)
    assert (post["qidx"][0].numpy() == fx["post_qidx0"]).mean() > 0.999
    assert (post["qidx"][1].numpy() == fx["post_qidx1"]).mean() > 0.999
    summ = state_summary(models)
    for k, v in summ.items():
        np.testing.assert_allclose(v, fx[k], rtol=2e-3, atol=2e-4, err_msg=k)
