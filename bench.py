#!/usr/bin/env python
"""Benchmark of the hot path: one optimisation step of the VQ-VAE trainer
(BASELINE.json configs[1]: vqvae trainer, VCC2020-shaped 80-dim mlfb, batch 64 x 500
frames per GPU, 14 speakers, bf16 MFMA compute) on synthetic inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL over xGMI)

Rank 0 prints ONE JSON line.  `value` = frames processed by all ranks / max-over-ranks
wall time of exactly K steps (barrier + synchronize on both sides).  At N = 1 a step is
`trainer.train_graphed`'s replay of the captured HIP graph of `trainer.train(batch)` (the
product's `hip_graph` mode: the same kernels in the same order, one launch per step; field
`launch`); the eagerly enqueued step is timed next to it (`eager_ms_per_step`; `--no-graph`
makes it the headline).  N > 1 under RCCL replays the step as ONE HIP graph with its collectives captured
(parallel.graph_collectives; CRANK_AMD_DP_GRAPH_COLLECTIVES=0 or the gloo backend: a chain of HIP graphs cut at every
collective, the collectives issued from the host between the replays, GraphedStep.segments); every rank replays or, if
one of them could not capture, all step eagerly.  Weak scaling: every rank trains on its own 64 utterances;
gradients, VQ-EMA statistics and masked-mean normalisers are all-reduced (crank_amd/parallel.py).
`--force-dist` runs that data-parallel code path in a world of ONE rank over RCCL (every collective issued, every
graph segment replayed): the N = 1 line carries its time as `dp_path_world_of_one` - the per-step price of the
collectives' launches and the segment boundaries before any xGMI traffic.

`roofline` is measured in a second pass of K identical steps with HIP events recorded
around every conv-class kernel on its launch stream (the first pass, which defines
`value`, runs without events so they cannot perturb it); it reports the kernel class
with the largest summed time, every class against the bound that is its own (algorithmic
FLOP / dense bf16 MFMA peak vs algorithmic bytes / HBM peak, whichever takes longer).
N=1 only, after the timed region: `parity_mode` (the same step in the bf16x3 arithmetic
that meets the 1e-3 bar against the fp32 reference), `other_configs` (BASELINE configs[2],
the lsgan step) and `cpu_baseline` (the CPU oracle under the same trainer class on bounded
samples of the configs[0] and configs[1] shapes).

`python bench.py --gpus N` without a torchrun environment starts its own N ranks
(torch.distributed.run on 127.0.0.1) and relays rank 0's JSON line.
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
DEC0_FWD_MAC = 64 * 128 + 8 * (128 * (64 * 5 + 34) + 128 * 64) + 64 * 64 + 64 * 80  # last decoder, forward, per frame
STEP_MFLOP = {"vqvae": (11.47e6 - 2 * DEC0_FWD_MAC) / 1e6, "lsgan": (28.10e6 - 2 * DEC0_FWD_MAC) / 1e6}
HBM_PEAK_GBS = 8000.0           # HBM3E spec peak, same guide (6 290 GB/s measured with a float4 copy)
KERNEL_CLASSES = {0: "conv_tile_kernel (generic per-layer conv; fallback path)",
                  1: "stack_fwd_kernel (stack2_fwd_kernel / stack_fwd_kernel: all gated residual blocks of a stack, forward)",
                  2: "stack_bwd_kernel (stack2_bwd_kernel / stack_bwd_kernel: data-gradient chain of a gated stack)",
                  3: "wgrad_kernel (table weight gradient; fallback path)",
                  4: "pstack_kernel (fused plain-conv chains: C, SPKRADV, first conv, heads; both directions)",
                  5: "stack_wgrad_kernel (weight gradients of the gated blocks)",
                  6: "pstack_wgrad_kernel (weight gradients of the plain convs)",
                  7: "vq_forward_f16_kernel (codebook search + gather + straight-through; 520 B per frame, SURVEY 8d)",
                  8: "logmel_kernel (on-the-fly log-mel front end; use_raw only)"}
HBM_CLASSES = (7, 8)  # priced against the HBM peak whatever their FLOP count (a4 / a5 / a18 are bandwidth rows)


def pmc_traffic(kernel_name):
    """HBM bytes per launch of `kernel_name` from the committed counter pass (tools/pmc_traffic.sh ->
    profiles/pmc_traffic.csv: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of this same
    benchmark).  FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md section HBM); KB -> bytes."""
    path = os.path.join(REPO, "profiles", "pmc_traffic.csv")
    if not os.path.exists(path):
        return None, None
    import csv

    tag_path = os.path.join(REPO, "profiles", "pmc_traffic.commit")  # written next to the CSV by the session that measured it
    tag = open(tag_path).read().strip() if os.path.exists(tag_path) else "an earlier tree (no tag file)"

    key = kernel_name.split(" ")[0]
    # both generations of the forward kernel / of the data-gradient chain
    keys = {"stack_fwd_kernel": (key, "stack2_fwd_kernel"), "stack_bwd_kernel": (key, "stack2_bwd_kernel")}.get(key, (key,))
    rd = wr = n = 0.0
    for r in csv.DictReader(open(path)):
        if any(k in r["kernel"] for k in keys):
            k = float(r["launches"])
            rd += float(r["FETCH_SIZE_avg_raw"]) * k
            wr += float(r["WRITE_SIZE_avg_raw"]) * k
            n += k
    return (None if n == 0 else (2.0 * rd + wr) / n * 1024.0), tag


def class_report(L, steps):
    """Every conv-class kernel recorded since crk_prof_enable(1) against the bound that is its own; returns the `roofline`
    object of the class with the largest summed kernel time (None: nothing was recorded)."""
    best = None
    per_class = {}
    for cls, name in KERNEL_CLASSES.items():
        cnt, ms, fl, by = ctypes.c_longlong(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        L.crk_prof_report(cls, ctypes.byref(cnt), ctypes.byref(ms), ctypes.byref(fl))
        L.crk_prof_report_bytes(cls, ctypes.byref(by))
        if cnt.value:
            sec = ms.value * 1e-3
            tfl, gbs = fl.value / sec / 1e12, by.value / sec / 1e9
            # the conv-GEMM kernels of the gated stacks (forward, data gradient, weight gradient) are priced against the
            # bf16 MFMA roofline (north_star's 30 % target): the bf16 planes they exchange through HBM are an
            # implementation choice, not algorithmic bytes.  The byte figure (planes included) stays next to it.  The
            # other classes (chains of 1x1 / narrow convs) get whichever of the two bounds is the longer time.
            gemm_class = cls in (1, 2, 5)
            bound = "mfma" if (gemm_class or by.value / (HBM_PEAK_GBS * 1e9) <= fl.value / (MFMA_BF16_PEAK_TFLOPS * 1e12)) and cls not in HBM_CLASSES else "hbm"
            per_class[name] = {"launches": cnt.value, "launches_per_step": cnt.value / steps, "avg_us": ms.value / cnt.value * 1e3,
                               "total_ms_per_step": ms.value / steps, "tflops": tfl,
                               "mfma_frac": tfl / MFMA_BF16_PEAK_TFLOPS, "GBps_incl_saved_planes": gbs,
                               "hbm_frac_incl_saved_planes": gbs / HBM_PEAK_GBS, "bound": bound,
                               "frac": gbs / HBM_PEAK_GBS if bound == "hbm" else tfl / MFMA_BF16_PEAK_TFLOPS}
            # rank by kernel time: an event-bracketed empty kernel reads ~6.3 us, which would let a
            # class of many short launches outrank the kernel that really dominates
            net = ms.value - 0.0063 * cnt.value
            if best is None or net > best[1]:
                best = (name, net)
    if best is None:
        return None
    c = per_class[best[0]]
    traffic, traffic_of = pmc_traffic(best[0])
    if c["bound"] == "hbm":
        roof = {"bound": "hbm", "achieved": c["GBps_incl_saved_planes"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": c["hbm_frac_incl_saved_planes"]}
    else:
        roof = {"bound": "mfma", "achieved": c["tflops"], "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": c["mfma_frac"]}
    roof.update({"kernel": best[0], "traffic": traffic,
                 "traffic_source": "static: profiles/pmc_traffic.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, "
                                   f"tools/pmc_traffic.sh; not re-measured in this run; measured on {traffic_of})",
                 "avg_launch_us": c["avg_us"], "mfma_frac": c["mfma_frac"],
                 "measured": "HIP events around every launch of eager steps on ONE stream (CRANK_AMD_OVERLAP_C=0 for "
                             "this pass; the timed step runs the classifier's update on a second stream, where a "
                             "kernel's begin-to-end time includes the compute units it shares)",
                 "hbm_frac_incl_saved_planes": c["hbm_frac_incl_saved_planes"], "classes": per_class})
    return roof


def stacks_alone(model_G, B, T, iters=20):
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
        # as in the step: the stacks' weight-norm backward waits for ONE launch over all of them (FlatModel.finish_grads)
        model_G.defer_wnorm = True
        try:
            for st, x, c in ins:
                y = st(x, c=c) if c is not None else st(x)
                if tuple(y.shape) not in ones:
                    ones[tuple(y.shape)] = torch.ones_like(y)
                torch.autograd.grad(y, x, ones[tuple(y.shape)])  # (dx returned, weight gradients into the flat block)
        finally:
            model_G.defer_wnorm = False
            model_G.finish_grads()

    ones = {}
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    # replayed from a captured graph like the step itself: eight autograd calls and ~20 launches per pass cost the host
    # about as long to issue as the GPU needs to run them
    graph, how = None, "eager"
    import gc

    from crank_amd.net.trainer.basetrainer import hold_collector_for_capture

    gc_was_on = hold_collector_for_capture()  # (an object owning a HIP graph must not be finalized inside a capture)
    try:
        from crank_amd import parallel

        parallel.drain_backend_watchdog()  # (N > 1: the RCCL watchdog must have nothing to poll during a capture)
        side = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
            once()
        how = "hip graph replay"
    except Exception as e:  # measure eagerly rather than not at all
        print(f"[bench] stacks_alone not capturable ({e!r}); timing it eagerly", file=sys.stderr)
        graph = None
        torch.cuda.synchronize()
    finally:
        if gc_was_on:
            gc.enable()
    run = graph.replay if graph is not None else once
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    flop = 3 * 2 * 1269760.0 * B * T
    return {"ms": dt * 1e3, "tflops": flop / dt / 1e12, "frac_of_mfma_peak": flop / dt / 1e12 / MFMA_BF16_PEAK_TFLOPS, "launch": how,
            "what": "enc0+enc1+dec1+dec0 forward+backward (data + weight gradients, grouped weight-norm backward) in isolation, "
                    "7.62 MFLOP/frame (SURVEY 8d)"}


def logmel_alone(B, T, dev, iters=30):
    """The on-the-fly log-mel front end (crank/net/module/mlfb.py:134-171; `use_raw` recipes only) at the benchmark's
    utterance shape: B waveforms of fftl + hop * T - 1 samples -> (B, T + 1, 80).  Algorithmic bytes: every sample once + the
    features written (SURVEY 8d puts this row on the HBM roofline; its actual bound is the vector ALU)."""
    from crank_amd.net.module.mlfb import LogMelFilterBankLayer

    hop, nfft = 128, 1024
    ns = nfft + hop * T - 1
    layer = LogMelFilterBankLayer(fs=22050, hop_size=hop, fft_size=nfft, win_length=nfft, window="hann", center=False, n_mels=80,
                                  fmin=80, fmax=7600, device=dev)
    x = 0.1 * torch.randn(B, ns, device=dev)
    for _ in range(3):
        y = layer(x)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(iters):
        layer(x)
    ev[1].record()
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) / iters * 1e3
    by = 4.0 * B * ns + 4.0 * y.numel()
    return {"us_per_call": us, "frames": int(y.shape[0] * y.shape[1]), "GBps": by / us / 1e3, "hbm_frac": by / us / 1e3 / HBM_PEAK_GBS,
            "kernel": "logmel_wave_kernel + lm_prep_kernel (HIP events around back-to-back calls)",
            "what": f"{B} waveforms x {ns} samples (fftl 1024, hop 128) -> {tuple(y.shape)} log-mel, fp32"}


def parity_gates(dev):
    """BASELINE.md section 3 ("parity gates reported with every timing") and the metric's "MCD vs ref", evaluated after the
    timed region against fixtures produced by the REFERENCE's own classes (tests/golden/make_golden.py; no oracle and no
    reference on the GPU box), for the arithmetic that was just timed ("bf16") and for the two parity modes:
      given parameters (tests/golden/convert_vqvae.npz: the reference VQVAE2 converting 4 x 200 frames to the target speaker):
        decoded-feature max relative error, frame-aligned MCD in dB (crank/bin/evaluate_mcd.py:76-77 without the DTW: both
        sides convert the same frames), fraction of identical code indices per quantizer;
      one training scenario (tests/golden/step_vqvae.npz: two steps of the reference's VQVAETrainer): max relative error of
        the losses of the first step (a pure forward comparison) and of both steps (the second follows an update)."""
    import math

    from crank_amd import ops
    from crank_amd.bin.train import get_model
    from crank_amd.net.trainer.utils import get_criterion, get_optimizer, get_scheduler
    from crank_amd.synthetic import make_batch
    from crank_amd.utils import load_yaml
    from tests.helpers import fill_models, golden, run_golden_case

    fx = golden("convert_vqvae.npz")
    B, T, S, seed = [int(v) for v in fx["meta_B_T_nspk_seed"]]
    factories = (lambda conf, n, scaler=None: get_model(conf, n, dev, scaler=scaler), get_optimizer,
                 lambda conf: get_criterion(conf, dev), get_scheduler)
    ref = torch.from_numpy(fx["decoded"]).double()
    feat_db = float((10.0 / math.log(10.0) * torch.sqrt(2.0 * (ref ** 2).sum(-1))).mean())
    out = {"fixtures": "tests/golden/convert_vqvae.npz, step_vqvae.npz (generated by importing the reference; fp32 on CPU)",
           "mcd_of_the_features_against_zero_dB": feat_db, "tolerance": "north_star: 1e-3 relative, VQ indices bit-exact"}
    for mode in ("bf16", "bf16x3f", "bf16x3"):
        ops.set_precision(mode)
        try:
            torch.manual_seed(1234)
            conf = load_yaml(None)
            G = get_model(conf, S, dev)["G"].eval()
            fill_models({"G": G})
            batch = make_batch(B, T, S, in_dim=conf["input_size"], seed=seed, device=dev)
            dec_h = torch.cat([batch["cv_lcf0"], batch["uv"]], -1)
            h = batch["cv_h"].clone()
            h[:, :] = h[:, 0:1]
            with torch.no_grad():
                o = G(batch["in_feats"], None, dec_h, spkrvec=h, use_ema=False)
            dec = o["decoded"].double().cpu()
            mcd = float((10.0 / math.log(10.0) * torch.sqrt(2.0 * ((dec - ref) ** 2).sum(-1))).mean())
            rec = {"decoded_rel_err": float((dec - ref).abs().max() / ref.abs().max()), "mcd_vs_ref_dB": mcd,
                   "qidx_identical": [float((o["qidx"][i].cpu().numpy() == fx[f"qidx{i}"]).mean()) for i in range(2)]}
            losses, _, _, sfx, _ = run_golden_case("vqvae", *factories, device=dev)
            torch.cuda.synchronize()

            def worst(steps_):
                w = 0.0
                for s_ in steps_:
                    for k in [f for f in sfx.files if f.startswith(f"loss{s_}/")]:
                        r = float(sfx[k])
                        if abs(r) > 1e-5:
                            w = max(w, abs(float(losses[s_].get(k.split("/", 1)[1], 0.0)) - r) / abs(r))
                return w

            rec["loss_rel_err_first_step"] = worst([0])
            rec["loss_rel_err_two_steps"] = worst(range(len(losses)))
            rec["within_1e-3"] = bool(rec["decoded_rel_err"] < 1e-3 and rec["loss_rel_err_first_step"] < 1e-3)
            out[mode] = rec
        except Exception as e:  # a gate that cannot run is reported, it does not take the line down
            out[mode] = {"error": repr(e)[:200]}
        finally:
            ops.set_precision("bf16")
    return out


def respawn_under_torchrun(n):
    """`python bench.py --gpus N` from a plain shell: start N ranks of this script on this node."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def _cpu_step_rate(conf_over, n_spkrs, Bc, T, budget_s, max_steps):
    """The oracle (PyTorch fp32 ops on the host cores) under the same trainer class."""
    import copy

    from crank_amd.net.trainer import TrainerWrapper
    from crank_amd.synthetic import make_batch
    from crank_amd.utils import load_yaml
    from oracle import modules as om

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
    trainer.steps = 1
    trainer.check_custom_start()  # (GAN / cycle phases start from trainer.steps: the lsgan sample must be in its GAN phase)
    batch = make_batch(Bc, T, n_spkrs, seed=1234)
    # median of >= 5 steps after 2 warm-ups - inside a time budget (the default bench run must stay short):
    # the first warm-up sizes the sample
    t0 = time.perf_counter()
    trainer.train(batch)
    one = time.perf_counter() - t0
    trainer.train(batch)
    steps = int(max(1, min(max_steps, budget_s / max(one, 1e-3))))
    times = []
    for _ in range(steps):
        t0 = time.perf_counter()
        trainer.train(batch)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return Bc * T / med, f"B={Bc} x T={T}, {n_spkrs} speakers, median of {steps} steps after 2 warm-ups, {med * 1e3:.0f} ms/step"


def cpu_baseline(conf_over):
    """the CPU oracle ("port": the reference's own trainer cannot run without the absent
    parallel_wavegan package) at the configs[1] shape (the benchmarked workload: `value`) and at the configs[0]
    shape (the reference's own CPU-runnable toy case)."""
    # the step is hundreds of small convolutions over 64..128 channels: intra-op threading
    # beyond a few cores only adds synchronisation (256 threads measured 50x SLOWER than
    # 16 on the GPU box's host), so the baseline uses 16 cores and says so
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    v2, s2 = _cpu_step_rate(conf_over, 14, 64, 500, budget_s=14.0, max_steps=7)
    v1, s1 = _cpu_step_rate(conf_over, 2, 2, 500, budget_s=5.0, max_steps=9)
    return {"value": v2, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "CPU oracle (PyTorch fp32 ops) vqvae step at the benchmarked shape (configs[1]): " + s2,
            "configs0": {"value": v1, "unit": "frames/s", "sample": "configs[0] shape (2-speaker toy, batch 2): " + s1}}


def dp_path_world_of_one(args, headline_ms):
    """The data-parallel code path (5 collectives per vqvae step - C2 + C3, G's gradients started / finished around the
    classifier's update, C2, SPKRADV + C gradients, loss values - captured with the step under RCCL) timed in a process group of one rank
    over RCCL, in a process of its own (`bench.py --force-dist`): what the segment boundaries and the collectives'
    launches cost per step before any xGMI traffic.  A failure of that process is reported, it cannot take this line down."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--force-dist", "--steps", str(min(args.steps, 50)), "--warmup", "10",
           "--batch", str(args.batch), "--no-roofline", "--no-extras", "--no-cpu-baseline"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=dict(os.environ))
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            what = [ln.strip()[:300] for ln in r.stderr.splitlines() if "what()" in ln or "HIP error" in ln or "Error" in ln][:4]
            return {"error": f"rc {r.returncode}", "printed_its_line": bool(line), "what": what, "stderr_tail": r.stderr[-300:]}
        d = json.loads(line[-1])
        return {"ms_per_step": d["ms_per_step"], "eager_ms_per_step": d["eager_ms_per_step"], "launch": d["launch"],
                "dist_backend": d["dist_backend"], "over_headline": d["ms_per_step"] / headline_ms,
                "what": "same step with the data-parallel path forced on in a world of one rank (RCCL): every collective issued - "
                        "captured with the step (one graph) unless CRANK_AMD_DP_GRAPH_COLLECTIVES=0 (a chain of graphs)"}
    except Exception as e:
        return {"error": repr(e)[:200]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--trainer", default="vqvae", choices=["vqvae", "lsgan", "cyclegan", "stargan"])
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the parity-mode and configs[2] timings")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "bf16x3f", "bf16x3"],
                    help="arithmetic of the timed steps (default bf16, the headline; the others are what `parity_mode` reports)")
    ap.add_argument("--no-graph", action="store_true",
                    help="enqueue every step eagerly (default at N=1: the step is replayed from a captured HIP graph, "
                         "trainer.train_graphed / conf hip_graph; N>1: a chain of graphs with the collectives between them)")
    ap.add_argument("--force-dist", action="store_true",
                    help="N=1 only: a process group of one rank (backend nccl = RCCL) with the data-parallel code path on")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(respawn_under_torchrun(args.gpus))

    from crank_amd import _lib, config, ops, parallel
    from crank_amd.bin.train import build_trainer
    from crank_amd.synthetic import make_batch
    from crank_amd.utils import load_yaml

    if args.force_dist:
        if args.gpus != 1:
            raise SystemExit("--force-dist is the world-of-one measurement: use it with --gpus 1")
        import socket

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.update(CRANK_AMD_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=os.environ.get("MASTER_PORT", str(port)))
        config.reload()  # (the package read the environment when it was imported)
    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist_on = parallel.is_dist()
    local = local % max(1, torch.cuda.device_count())  # (ranks sharing a GPU: the gloo test on the one-GPU box)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ops.set_precision(args.precision)
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
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # The product's graph-replay mode (the same kernels in the same order, launched from captured HIP graphs).  N>1: the
    # step is a chain of graphs cut at every collective (GraphedStep.segments), the collectives run between the replays;
    # every rank replays or none does.
    graphed = None
    if not args.no_graph:
        from crank_amd.net.trainer.basetrainer import GraphedStep

        try:
            graphed = GraphedStep(trainer, batch, warmup=3)
        except (RuntimeError, ValueError) as e:
            print(f"[bench] rank {rank}: step not capturable ({e}); running eagerly", file=sys.stderr)
            torch.cuda.synchronize()
            graphed = None
        # every rank replays or none does (a capture exchanges nothing, and an eager rank issues the collectives of a
        # replaying one: the ranks are still paired up here whatever happened)
        if not parallel.agree_on_capture(graphed is not None):
            graphed = None

    def run(k, replay=True):
        barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            graphed.step() if (graphed is not None and replay) else trainer.train(batch)
        barrier()
        return time.perf_counter() - t0

    for _ in range(args.warmup):
        vals = graphed.step() if graphed is not None else trainer.train(batch)
    dt = run(args.steps)
    dt_eager = run(args.steps, replay=False) if graphed is not None else dt
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if torch.distributed.get_backend() == "nccl" else "cpu")
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
        "dtype": args.precision,
        "data": "synthetic",
        "config": {"workload": f"{args.trainer} trainer step (all sub-updates), VCC2020-shaped synthetic 80-dim mlfb, "
                               f"{B} utterances x {T} frames per GPU, {n_spkrs} speakers",
                   "trainer": args.trainer, "global_batch": B * world, "batch_len": T, "parallelism": f"dp{world}"},
        "loss_G": vals.get("G"),
        "launch": ("eager" if graphed is None else "hip graph replay (trainer.train_graphed)" if not dist_on else
                   "one hip graph, the RCCL collectives captured with the step" if len(graphed.segments) == 1 else
                   f"chain of {len(graphed.segments)} hip graphs with the host-issued collectives between them"),
        "eager_ms_per_step": dt_eager / args.steps * 1e3,
        "world_size_seen": world,
        "dist_backend": torch.distributed.get_backend() if dist_on else None,
    }
    if args.force_dist:
        out["forced_dist_world_of_one"] = True

    if not args.no_roofline:
        L = _lib.lib()
        # one stream for this pass: next to the classifier's update on its second stream (the timed step) a kernel's
        # begin-to-end time includes the time it shares compute units with that stream's kernels - the roofline prices
        # each kernel's OWN duration
        with config.override(overlap_c=0):
            L.crk_prof_enable(1)
            dt2 = run(args.steps, replay=False)  # the events are recorded from the host around each launch
            L.crk_prof_enable(0)
        roof = class_report(L, args.steps)
        if roof is not None:
            roof["ms_per_step_with_events"] = dt2 / args.steps * 1e3
            out["roofline"] = roof
        try:
            out["stacks_alone"] = stacks_alone(trainer.model["G"], B, T)
        except Exception as e:  # never let the side measurement take the bench line down
            out["stacks_alone"] = {"error": repr(e)[:200]}
        # whole-step figure next to it: conv-GEMM FLOP the step EXECUTES and needs / step time (SURVEY 8d's counting rule).
        # SURVEY's 11.47 / 28.10 MFLOP per frame count the last decoder of the speaker-adversarial update's generator forward
        # (445 440 MAC per frame); that launch is dead work and is not executed (VQVAE2.forward(need_decoded=False)).
        flop_per_frame = {"vqvae": STEP_MFLOP["vqvae"] * 1e6, "lsgan": STEP_MFLOP["lsgan"] * 1e6}.get(args.trainer)
        if flop_per_frame:
            out["step_mfma_frac"] = (frames / dt) * flop_per_frame / (world * MFMA_BF16_PEAK_TFLOPS * 1e12)

    if world == 1 and not args.no_extras and args.trainer == "vqvae":
        def timed(tr, k, w=3):
            for _ in range(w):
                tr.train(batch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k):
                tr.train(batch)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / k

        # the arithmetic that meets 1e-3 against the fp32 reference's goldens (tests/test_gpu_step.py), same workload, replayed
        # from a HIP graph like the headline: "bf16x3f" = forward passes as 3 bf16 MFMAs on split operands (~fp32 loss
        # values), backward passes in plain bf16 - every loss of every golden step within 1e-3
        # (test_step_bf16x3_forward_only_mode); "bf16x3" = both directions split (parameters after the step within 5e-3 too)
        from crank_amd.net.trainer.basetrainer import GraphedStep

        def replayed(mode, k=20):
            ops.set_precision(mode)
            try:
                g = GraphedStep(trainer, batch, warmup=2)
                for _ in range(3):
                    g.step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(k):
                    g.step()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / k
            finally:
                ops.set_precision("bf16")

        try:
            tf = replayed("bf16x3f")
            out["parity_mode"] = {"dtype": "bf16x3f", "ms_per_step": tf * 1e3, "frames_per_s": B * T / tf, "launch": "hip graph replay",
                                  "what": "same step, forward passes as 3 bf16 MFMAs on split operands (~fp32), backward passes in "
                                          "plain bf16: the cheapest mode whose losses meet 1e-3 against the fp32 reference's goldens on "
                                          "every golden step (tests/test_gpu_step.py); the timed `value` runs plain bf16, pinned against "
                                          "the bf16-emulating oracle (test_gpu_step.py, test_gpu_nets.py)"}
        except Exception as e:
            out["parity_mode"] = {"error": repr(e)[:200]}
        try:
            t3 = replayed("bf16x3", k=10)
            out["parity_mode_both_directions"] = {"dtype": "bf16x3", "ms_per_step": t3 * 1e3, "frames_per_s": B * T / t3,
                                                  "launch": "hip graph replay"}
        except Exception as e:
            out["parity_mode_both_directions"] = {"error": repr(e)[:200]}
        try:  # BASELINE configs[2]: lsgan step in the GAN phase, same shapes
            over3 = dict(trainer_type="lsgan", batch_size=B, batch_len=T, n_steps_gan_start=0)
            tr3 = build_trainer(load_yaml(None, **over3), n_spkrs, "/tmp/crank_amd_bench3", device=dev)
            tr3.steps = 1
            tr3.check_custom_start()
            tl_eager = timed(tr3, 10)
            g3 = GraphedStep(tr3, batch, warmup=0)  # default D, dropout 0.25: the masks' seeds live on the device
            for _ in range(3):
                g3.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                g3.step()
            torch.cuda.synchronize()
            tl = (time.perf_counter() - t0) / 20
            rec3 = {"ms_per_step": tl * 1e3, "frames_per_s": B * T / tl, "value": B * T / tl, "unit": "frames/s", "dtype": "bf16",
                    "launch": "hip graph replay", "eager_ms_per_step": tl_eager * 1e3,
                    "step_mfma_frac": B * T / tl * STEP_MFLOP["lsgan"] * 1e6 / (MFMA_BF16_PEAK_TFLOPS * 1e12),
                    "what": "configs[2]: VQ-VAE + residual D (dropout 0.25) + spkradv, GAN phase, "
                            f"{B} x {T} frames, 20 replayed steps"}
            if not args.no_roofline:  # the same per-class event pass as the headline's, over 10 eager lsgan steps
                Lr = _lib.lib()
                with config.override(overlap_c=0):
                    Lr.crk_prof_enable(1)
                    t_ev = timed(tr3, 10, w=0)
                    Lr.crk_prof_enable(0)
                roof3 = class_report(Lr, 10)
                if roof3 is not None:
                    roof3["ms_per_step_with_events"] = t_ev * 1e3
                    roof3["traffic"] = None  # (the committed counter pass is of the vqvae step)
                    roof3["traffic_source"] = None
                    rec3["roofline"] = roof3
                    rec3["conv_class_launches_per_step"] = sum(c["launches_per_step"] for c in roof3["classes"].values())
            out["other_configs"] = {"lsgan": rec3}
            del tr3, g3
        except Exception as e:
            out["other_configs"] = {"error": repr(e)[:200]}

    if rank == 0 and world == 1 and not args.no_extras and args.trainer == "vqvae":
        try:  # north_star's "STFT / mel filterbank" kernel: off the benchmarked path (use_raw is false), timed by itself
            out["logmel_use_raw"] = logmel_alone(B, T, dev)
        except Exception as e:
            out["logmel_use_raw"] = {"error": repr(e)[:200]}
    if rank == 0 and world == 1 and not args.no_extras and not args.force_dist and args.trainer == "vqvae":
        out["dp_path_world_of_one"] = dp_path_world_of_one(args, out["ms_per_step"])
    if rank == 0 and world == 1 and not args.no_extras:
        try:
            out["parity_gates"] = parity_gates(dev)
            g = out["parity_gates"].get(args.precision, {})
            out["mcd_vs_ref_dB"] = g.get("mcd_vs_ref_dB")  # the metric's "MCD vs ref" for the arithmetic that was timed
        except Exception as e:
            out["parity_gates"] = {"error": repr(e)[:200]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(dict(trainer_type=args.trainer))
        if args.trainer == "vqvae" and not args.no_extras and isinstance(out.get("other_configs", {}).get("lsgan"), dict):
            try:  # configs[2] next to its own CPU number (bounded: ~8 s of host time)
                torch.set_num_threads(min(16, os.cpu_count() or 1))
                v3, s3 = _cpu_step_rate(dict(trainer_type="lsgan", n_steps_gan_start=0), 14, 64, 500, budget_s=6.0, max_steps=3)
                out["other_configs"]["lsgan"]["cpu_baseline"] = {"value": v3, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                                                                 "sample": "CPU oracle lsgan step (GAN phase) at the configs[2] shape: " + s3}
            except Exception as e:
                out["other_configs"]["lsgan"]["cpu_baseline"] = {"error": repr(e)[:200]}
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if dist_on:
        torch.distributed.barrier()
        graphed = None  # the graphs (and the pool the step's tensors live in) go before the communicator
        del trainer
        import gc

        gc.collect()
        torch.cuda.synchronize()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
