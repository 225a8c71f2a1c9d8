#!/usr/bin/env python
"""Benchmark of the hot path: one optimisation step of the VQ-VAE trainer
(BASELINE.json configs[1]: vqvae trainer, VCC2020-shaped 80-dim mlfb, batch 64 x 500
frames per GPU, 14 speakers, bf16 MFMA compute) on synthetic inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL over xGMI)

Rank 0 prints ONE JSON line.  `value` = frames processed by all ranks / max-over-ranks
wall time of exactly K steps (barrier + synchronize on both sides).  Weak scaling: every
rank trains on its own 64 utterances; gradients, VQ-EMA statistics and masked-mean
normalisers are all-reduced (crank_amd/parallel.py).

`roofline` is measured in a second pass of K identical steps with HIP events recorded
around every conv-class kernel on its launch stream (the first pass, which defines
`value`, runs without events so they cannot perturb it); it reports the kernel class
with the largest summed time.  `cpu_baseline` (rank 0, N=1 only) times the CPU oracle
driven by the same trainer class on a bounded sample.
"""
import argparse
import ctypes
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense, MI355X_MICROARCH.md (AMD's 5 PF figure is 2:1 sparse)
HBM_PEAK_GBS = 8000.0           # HBM3E spec peak, same guide (6 290 GB/s measured with a float4 copy)
KERNEL_CLASSES = {0: "conv_tile_kernel (generic per-layer conv; fallback path)",
                  1: "stack_fwd_kernel (all gated residual blocks of a stack, forward)",
                  2: "stack_bwd_kernel (data-gradient chain of a gated stack)",
                  3: "wgrad_kernel (table weight gradient; fallback path)",
                  4: "pstack_kernel (fused plain-conv chains: C, SPKRADV, first conv, heads; both directions)",
                  5: "stack_wgrad_kernel (weight gradients of the gated blocks)",
                  6: "pstack_wgrad_kernel (weight gradients of the plain convs)"}


def pmc_traffic(kernel_name):
    """HBM bytes per launch of `kernel_name` from the committed counter pass (tools/pmc_traffic.sh ->
    profiles/pmc_traffic.csv: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of this same
    benchmark).  FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md section HBM); KB -> bytes."""
    path = os.path.join(REPO, "profiles", "pmc_traffic.csv")
    if not os.path.exists(path):
        return None
    import csv

    key = kernel_name.split(" ")[0]
    rd = wr = n = 0.0
    for r in csv.DictReader(open(path)):
        if key in r["kernel"]:
            k = float(r["launches"])
            rd += float(r["FETCH_SIZE_avg_raw"]) * k
            wr += float(r["WRITE_SIZE_avg_raw"]) * k
            n += k
    return None if n == 0 else (2.0 * rd + wr) / n * 1024.0


def stacks_alone(model_G, B, T, iters=10):
    """the four gated-residual stacks of G (enc0, enc1, dec1, dec0) forward + backward in
    isolation on synthetic activations: 3 x 2 x 1 269 760 FLOP per frame (forward, data gradient,
    weight gradient).  Returns TFLOP/s and the fraction of the dense bf16 MFMA peak."""
    stacks = list(model_G.encoders) + list(model_G.decoders)
    dev = model_G.flat.device
    ins = []
    for st in stacks:
        net = st.net
        x = torch.randn(B, T, net.in_ch, device=dev, requires_grad=True)
        c = torch.randn(B, T, net.aux_ch, device=dev) if net.aux_ch > 0 else None
        ins.append((st, x, c))

    def once():
        for st, x, c in ins:
            y = st(x, c=c) if c is not None else st(x)
            y.backward(torch.ones_like(y))

    once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        once()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    flop = 3 * 2 * 1269760.0 * B * T
    return {"ms": dt * 1e3, "tflops": flop / dt / 1e12, "frac_of_mfma_peak": flop / dt / 1e12 / MFMA_BF16_PEAK_TFLOPS,
            "what": "enc0+enc1+dec1+dec0 forward+backward in isolation, 7.62 MFLOP/frame (SURVEY 8d)"}


def cpu_baseline(conf_over, n_spkrs, budget_s=20.0):
    """The oracle (PyTorch fp32 ops on the host cores) under the same trainer class."""
    import copy

    from crank_amd.net.trainer import TrainerWrapper
    from crank_amd.synthetic import make_batch
    from crank_amd.utils import load_yaml
    from oracle import modules as om

    # the step is hundreds of small convolutions over 64..128 channels: intra-op threading
    # beyond a few cores only adds synchronisation (256 threads measured 50x SLOWER than
    # 16 on the GPU box's host), so the baseline uses 16 cores and says so
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    Bc, T = 4, 500
    conf = load_yaml(None, **copy.deepcopy(conf_over))
    conf["batch_size"] = Bc
    torch.manual_seed(1234)
    models = om.get_model(conf, n_spkrs)
    for m in models.values():
        m.train()
    optimizer = om.get_optimizer(conf, models)
    trainer = TrainerWrapper(conf["trainer_type"], model=models, optimizer=optimizer, criterion=om.get_criterion(conf),
                             dataloader={"spkrs": {f"spk{i}": i for i in range(n_spkrs)}}, writer=None,
                             expdir="/tmp/crank_amd_cpu", conf=conf, feat_conf=conf["feature"], scheduler=None,
                             scaler=None, resume=0, device="cpu", n_jobs=1)
    batch = make_batch(Bc, T, n_spkrs, seed=1234)
    t0 = time.perf_counter()
    trainer.train(batch)  # warm-up, also sizes the sample
    one = time.perf_counter() - t0
    if one > budget_s:  # already over budget: the warm-up step is the sample
        steps, dt = 1, one
    else:
        steps = int(max(1, min(10, budget_s / max(one, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(steps):
            trainer.train(batch)
        dt = time.perf_counter() - t0
    return {"value": Bc * T * steps / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"CPU oracle (PyTorch fp32 ops) vqvae step, B={Bc} x T={T}, {steps} steps after 1 warm-up, "
                      f"{dt / steps * 1e3:.0f} ms/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--trainer", default="vqvae", choices=["vqvae", "lsgan", "cyclegan", "stargan"])
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    from crank_amd import _lib, ops, parallel
    from crank_amd.bin.train import build_trainer
    from crank_amd.synthetic import make_batch
    from crank_amd.utils import load_yaml

    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ops.set_precision("bf16")
    n_spkrs, T, B = 14, 500, args.batch
    conf_over = dict(trainer_type=args.trainer, batch_size=B, batch_len=T)
    if args.trainer != "vqvae":
        conf_over.update(n_steps_gan_start=0)
        if args.trainer in ("cyclegan", "stargan"):
            conf_over.update(use_cyclic_training=True, n_steps_cycle_start=0)
    conf = load_yaml(None, **conf_over)
    torch.manual_seed(1234)
    np.random.seed(1234)
    parallel.seed_shared_python_rng(1234)
    grad_reduce = parallel.install()
    trainer = build_trainer(conf, n_spkrs, "/tmp/crank_amd_bench", device=dev, grad_reduce_fn=grad_reduce)
    trainer.steps = 1
    trainer.check_custom_start()
    batch = make_batch(B, T, n_spkrs, seed=1234 + rank, device=dev)  # resident in HBM before timing

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def run(k):
        barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            trainer.train(batch)
        barrier()
        return time.perf_counter() - t0

    for _ in range(args.warmup):
        vals = trainer.train(batch)
    dt = run(args.steps)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    frames = world * B * T * args.steps
    out = {
        "metric": "train frames/sec",
        "value": frames / dt,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic",
        "config": {"workload": f"{args.trainer} trainer step (all sub-updates), VCC2020-shaped synthetic 80-dim mlfb, "
                               f"{B} utterances x {T} frames per GPU, {n_spkrs} speakers",
                   "trainer": args.trainer, "global_batch": B * world, "batch_len": T, "parallelism": f"dp{world}"},
        "loss_G": vals.get("G"),
    }

    if not args.no_roofline:
        L = _lib.lib()
        L.crk_prof_enable(1)
        dt2 = run(args.steps)
        L.crk_prof_enable(0)
        best = None
        per_class = {}
        for cls, name in KERNEL_CLASSES.items():
            cnt, ms, fl = ctypes.c_longlong(), ctypes.c_double(), ctypes.c_double()
            L.crk_prof_report(cls, ctypes.byref(cnt), ctypes.byref(ms), ctypes.byref(fl))
            if cnt.value:
                per_class[name] = {"launches": cnt.value, "avg_us": ms.value / cnt.value * 1e3,
                                   "total_ms_per_step": ms.value / args.steps,
                                   "tflops": fl.value / (ms.value * 1e-3) / 1e12}
                # rank by kernel time: an event-bracketed empty kernel reads ~6.3 us, which would let a
                # class of many short launches outrank the kernel that really dominates
                net = ms.value - 0.0063 * cnt.value
                if best is None or net > best[4]:
                    best = (name, ms.value, fl.value, cnt.value, net)
        if best is not None:
            avg_s = best[1] * 1e-3 / best[3]
            flops_launch = best[2] / best[3]
            tfl = flops_launch / avg_s / 1e12
            traffic = pmc_traffic(best[0])  # HBM bytes per launch (committed PMC pass; equals the algorithmic bytes)
            # the roofline that bounds this kernel: whichever of (FLOP / MFMA peak, bytes / HBM peak) is the longer time
            t_mfma = flops_launch / (MFMA_BF16_PEAK_TFLOPS * 1e12)
            t_hbm = (traffic or 0.0) / (HBM_PEAK_GBS * 1e9)
            if traffic and t_hbm > t_mfma:
                ach = traffic / avg_s / 1e9
                roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}
            else:
                roof = {"bound": "mfma", "achieved": tfl, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": tfl / MFMA_BF16_PEAK_TFLOPS}
            roof.update({"kernel": best[0], "traffic": traffic, "avg_launch_us": avg_s * 1e6,
                         "flops_per_launch": flops_launch, "mfma_frac": tfl / MFMA_BF16_PEAK_TFLOPS,
                         "hbm_frac": (traffic / avg_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "ms_per_step_with_events": dt2 / args.steps * 1e3, "classes": per_class})
            out["roofline"] = roof
        try:
            out["stacks_alone"] = stacks_alone(trainer.model["G"], B, T)
        except Exception as e:  # never let the side measurement take the bench line down
            out["stacks_alone"] = {"error": repr(e)[:200]}
        # whole-step figure next to it: conv-GEMM FLOP of the step / step time
        flop_per_frame = {"vqvae": 11.47e6, "lsgan": 28.10e6}.get(args.trainer)
        if flop_per_frame:
            out["step_mfma_frac"] = (frames / dt) * flop_per_frame / (world * MFMA_BF16_PEAK_TFLOPS * 1e12)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(conf_over, n_spkrs)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
