"""Worker of tests/test_gpu_dp.py (not a test module): runs the PRODUCT trainer - FlatAdam with the C1
all-reduce, the generator's EmaBucket (C2), the wrapped criteria fed by prepare_step (C3) - on this rank's
shard of a fixed batch (or on the whole batch when launched without torchrun) and writes what the test
compares: reduced gradients and loss values of step 1, codebooks / EMA state / parameter checksums after 3 steps.

    python tests/dp_gpu_worker.py OUT.npz TRAINER GLOBAL_B T [PRECISION [MODE [CLIP [STEPS]]]]

MODE "graph": the steps go through ``trainer.train_graphed`` (three eager steps, then the step replayed as a chain of HIP
graphs with the host-issued collectives between them).  CLIP: ``clip_grad_norm`` of every optimizer (0 = off).
"""
import os
import random
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def _log_path(out):
    multi = "RANK" in os.environ
    res = f"{out}.rank{os.environ['RANK']}.npz" if multi else out
    return res.replace(".npz", "") + ".log"


def main():
    out, ttype, B, T = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    precision = sys.argv[5] if len(sys.argv) > 5 else "bf16"
    mode = sys.argv[6] if len(sys.argv) > 6 else "eager"
    clip = float(sys.argv[7]) if len(sys.argv) > 7 else 0.0
    n_steps = int(sys.argv[8]) if len(sys.argv) > 8 else 3
    # everything this process writes to stderr - native abort messages of the HIP runtime / RCCL / the watchdog thread
    # included - goes to the rank's log file; the test prints it when a worker fails
    log = open(_log_path(out), "w", buffering=1)
    os.dup2(log.fileno(), 2)

    def note(msg):
        log.write(msg + "\n")
        log.flush()

    from crank_amd import ops, parallel
    from crank_amd.bin.train import build_trainer
    from crank_amd.utils import load_yaml
    from tests.helpers import fill_models, make_batch

    rank, world, _ = parallel.init_from_env()
    note(f"rank {rank} of {world}: {ttype} B={B} T={T} {precision} {mode} clip={clip} steps={n_steps} "
         f"backend={torch.distributed.get_backend() if torch.distributed.is_initialized() else None}")
    torch.cuda.set_device(0)  # every rank shares the one GPU of the test box
    ops.set_precision(precision)
    S = 3
    over = dict(trainer_type=ttype, batch_size=B // world, batch_len=T)
    if ttype != "vqvae":
        over.update(discriminator_dropout=0.0, n_steps_gan_start=0)
    if ttype in ("cyclegan", "stargan"):
        over.update(use_cyclic_training=True, n_steps_cycle_start=0)
    conf = load_yaml(None, **over)
    if clip:
        for m in conf["optim"]:
            conf["optim"][m]["clip_grad_norm"] = clip
    random.seed(1234)
    np.random.seed(1234)
    torch.manual_seed(1234)
    parallel.seed_shared_python_rng(1234)
    trainer = build_trainer(conf, S, "/tmp/crank_amd_dp", grad_reduce_fn=parallel.install())
    fill_models(trainer.model)
    for opt in trainer.optimizer.values():
        opt.clear_grads = False  # the gradients of step 0 are compared below
    trainer.steps = 1
    trainer.check_custom_start()
    res = {"world": np.array(world), "rank": np.array(rank)}
    for step in range(n_steps):
        full = make_batch(B, T, S, seed=11 + step, device="cuda")
        batch = parallel.shard_batch(full, rank, world) if world > 1 else full
        random.seed(99 + rank)  # rank-specific global draws (what a dataset does) must not matter
        random.random()
        vals = trainer.train_graphed(batch) if mode == "graph" else trainer.train(batch)
        torch.cuda.synchronize()
        note(f"step {step} done")
        if step == n_steps - 1:
            for k, v in vals.items():
                res[f"last_loss/{k}"] = np.array(float(v))
        if step == 0:
            for k, v in vals.items():
                res[f"loss/{k}"] = np.array(float(v))
            for name, m in trainer.model.items():
                res[f"grad/{name}"] = m.grad_flat.detach().cpu().numpy()
                if name == "SPKRADV":  # (by reference key too: what an oracle's named parameters are compared with)
                    for key, _, _ in m._entries:
                        res[f"gradkey/{name}/{key}"] = m.grad_view(key).detach().cpu().numpy()
    G = trainer.model["G"]
    for i, q in enumerate(G.quantizers):
        res[f"codebook{i}"] = q.weight.detach().cpu().numpy()
        res[f"ema_size{i}"] = q.ema_size.detach().cpu().numpy()
        res[f"ema_w{i}"] = q.ema_w.detach().cpu().numpy()
    for name, m in trainer.model.items():
        res[f"flat/{name}"] = m.flat.detach().cpu().numpy()
    if mode == "graph":
        res["n_graphs"] = np.array(0 if trainer._graphs is None else sum(s[1] is not None for s in trainer._graphs.values()))
        res["n_segments"] = np.array(0 if trainer._graphs is None else
                                     max([len(s[1].segments) for s in trainer._graphs.values() if s[1] is not None] + [0]))
    multi = torch.distributed.is_available() and torch.distributed.is_initialized()
    np.savez(f"{out}.rank{rank}" if multi else out, **res)
    note("results written")
    if multi:
        torch.distributed.barrier()
        # graphs (they hold the pool the step's tensors live in) before the communicator
        import gc

        trainer._graphs = None
        del trainer
        gc.collect()
        torch.cuda.synchronize()
        torch.distributed.destroy_process_group()
        note("process group destroyed")
    note("DONE")


if __name__ == "__main__":
    main()
