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
import gc
import logging
import os
import random
from pathlib import Path

import torch

from ... import parallel
from ...config import cfg


def TrainerWrapper(trainer_type, **ka):
    from . import CycleGANTrainer, LSGANTrainer, StarGANTrainer, VQVAETrainer

    table = {"vqvae": VQVAETrainer, "lsgan": LSGANTrainer, "cyclegan": CycleGANTrainer, "stargan": StarGANTrainer}
    if trainer_type not in table:
        raise NotImplementedError("conf['trainer_type']: {} is not supported.".format(trainer_type))
    return table[trainer_type](**ka)


def to_device(batch, device):
    return {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}


_WEIGHTS = {}


def _weight_vector(ws, like):
    key = (ws, like.device, like.dtype)
    if key not in _WEIGHTS:
        _WEIGHTS[key] = torch.tensor(ws, device=like.device, dtype=like.dtype)
    return _WEIGHTS[key]


class LossBook(dict):
    """The reference's loss dict (basetrainer.py:200-206): totals per model start at 0.0
    and become tensors as terms are added.

    ``add(key, w, term)`` is the reference's ``loss[key] += w * term``.  The terms are only
    recorded; the total is formed when it is read (``loss[key]``, ``items()``), as ONE stack +
    multiply + sum, and its backward is ONE multiply: the reference's chain of 0-dim torch ops
    (a multiply and an add per term forward, a multiply per term backward) was ~50 launches
    of ~3.5 us per step next to kernels that take 10 - 90 us."""

    def __init__(self):
        super().__init__(objective=0.0, G=0.0, D=0.0, C=0.0, SPKRADV=0.0)
        self._pending = {}

    def add(self, key, w, term):
        self._pending.setdefault(key, []).append((float(w), term))

    def _settle(self, key):
        pend = self._pending.pop(key, None)
        if not pend:
            return
        base = super().__getitem__(key) if key in self else 0.0
        if isinstance(base, torch.Tensor):
            pend = [(1.0, base)] + pend
        # a zero weight (default alpha["mse"]) contributes exactly 0 to value and gradient: leave the term out of
        # the graph (its backward kernels would only produce zeros); its own value stays in the book for logging
        tens = [(w, t) for w, t in pend if isinstance(t, torch.Tensor) and w != 0.0]
        const = sum(w * t for w, t in pend if not isinstance(t, torch.Tensor)) + (0.0 if isinstance(base, torch.Tensor) else base)
        if not tens:
            val = const
        elif len(tens) == 1 and tens[0][0] == 1.0 and const == 0.0:
            val = tens[0][1]
        elif len(tens) <= 16 and all(t.is_cuda and t.dtype == torch.float32 and t.numel() == 1 for _, t in tens):
            from ... import ops  # (GPU scalars: the total and its backward as one launch each)

            val = ops.weighted_sum([t for _, t in tens], [w for w, _ in tens], const)
        else:
            vec = torch.stack([t.reshape(()) for _, t in tens])
            val = (vec * _weight_vector(tuple(w for w, _ in tens), vec)).sum()
            if const != 0.0:
                val = val + const
        super().__setitem__(key, val)

    def __getitem__(self, key):
        self._settle(key)
        return super().__getitem__(key)

    def __setitem__(self, key, value):
        self._pending.pop(key, None)
        super().__setitem__(key, value)

    def items(self):
        for k in list(self._pending):
            self._settle(k)
        return super().items()

    def values(self):
        for k in list(self._pending):
            self._settle(k)
        return super().values()


class LossValues(dict):
    """``dict[str, float]`` of a step's loss values whose numbers arrive asynchronously: the keys are there at
    once, the floats are filled in from the host buffer the first time anything reads a value."""

    def __init__(self, keys, host, event, zero_keys, index=None):
        """index (optional): key i is element index[i] of the host vector (the step's scalar arena), else element i."""
        super().__init__(LossBook())
        for k in list(keys) + list(zero_keys):
            super().setdefault(k, 0.0)
        self._pending = (list(keys), host, event, index) if host is not None else None

    def _resolve(self):
        if self._pending is not None:
            keys, host, event, index = self._pending
            self._pending = None
            if event is not None:
                event.synchronize()
            vals = host.tolist()
            if index is not None:
                vals = [vals[i] for i in index]
            for k, v in zip(keys, vals):
                super().__setitem__(k, super().__getitem__(k) + v)

    def __getitem__(self, k):
        self._resolve()
        return super().__getitem__(k)

    def get(self, k, default=None):
        self._resolve()
        return super().get(k, default)

    def items(self):
        self._resolve()
        return super().items()

    def values(self):
        self._resolve()
        return super().values()

    def copy(self):
        self._resolve()
        return dict(super().items())

    def __repr__(self):
        self._resolve()
        return super().__repr__()

    def __eq__(self, other):
        self._resolve()
        return super().__eq__(other)

    __hash__ = None


def hold_collector_for_capture():
    """No garbage collection while this thread captures: a collected object that owns a HIP graph (an earlier GraphedStep
    of a dropped trainer - trainer and step reference each other, so they wait for the collector) makes a HIP call in
    ~CUDAGraph that is illegal while this thread captures, raised inside the destructor: std::terminate, SIGABRT ("Fatal
    Python error: Aborted ... Garbage-collecting", met in round 4 when the collector happened to run inside a capture;
    profiles/round4_gc_capture_abort.txt, tools/gc_capture_repro.py).  torch's own graph context collects before a capture
    only under torch.compiler.config.force_cudagraph_gc.  Collects now and switches the collector off; returns whether it
    was on (the caller switches it back on after the capture)."""
    was_on = gc.isenabled()
    # until nothing is left: one pass can free objects whose finalizers make more garbage (an autograd graph kept alive by a
    # dropped trainer goes in stages - round 5: 84 objects survived a single pass and were collected inside the capture)
    for _ in range(8):
        if gc.collect() == 0:
            break
    gc.disable()
    return was_on


class GraphedStep:
    """One optimisation step captured as HIP graphs (torch.cuda.CUDAGraph) and replayed: the ~100 kernel launches of
    a step cost the host about as long to enqueue as the GPU needs to run them, so a replay takes the host out of the
    loop.  The step reads its batch from the tensors it was captured with: ``step(batch)`` copies a new batch into them
    (same shapes) and replays; with ``step()`` the captured tensors are used as they are (synthetic, resident batches).
    Host-side decisions of the captured step are frozen: a graph belongs to one phase of a trainer
    (``BaseTrainer._mode_signature``), one batch shape and one outcome of the trainer's per-step random choices
    (``BaseTrainer._draw_step_choices``); learning rates, Adam step counts and dropout seeds live in device memory and
    keep advancing.

    Data parallelism: the collectives of a step (parallel.py: C3 counts, C2 EMA statistics, C1 gradients, loss values)
    are issued from the host.  Each one ends the running capture and starts the next (``collective``), so a step is a
    chain of graphs that share one memory pool, replayed in order with the collectives in between - whatever the
    backend (RCCL, or gloo in the one-GPU tests).  A single process has no collectives and is one graph.

    ``warmup`` eager steps on the batch precede the capture (they are real optimisation steps): the library's lazy
    allocations must have happened before a capture.  Pass 0 when the trainer has already run steps of this shape."""

    def __init__(self, trainer, batch, warmup=3, choices=()):
        refuse = cfg.test_refuse_capture_rank  # test hook: this rank cannot capture its step
        if refuse is not None and parallel.is_dist() and parallel.rank() == int(refuse):
            raise RuntimeError("capture refused on this rank (CRANK_AMD_TEST_REFUSE_CAPTURE_RANK)")
        self.trainer = trainer
        self.batch = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
        self.choices = tuple(choices)
        self.stream = torch.cuda.Stream()
        if warmup:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                for _ in range(warmup):
                    trainer._pending_choices = list(self.choices)
                    trainer.train(self.batch)
            torch.cuda.current_stream().wait_stream(self.stream)
        torch.cuda.synchronize()
        writer, trainer.writer = trainer.writer, None  # the writer reads values on the host: not inside a capture
        parallel.drain_backend_watchdog()  # (RCCL: nothing left for the watchdog thread to poll while capturing)
        self.pool = torch.cuda.graph_pool_handle()
        self.segments = []  # (graph, what the host does after its replay: None | ("reduce", tensors) | ("start", key, tensor) | ("finish", key))
        self._ctx = None
        gc_was_on = hold_collector_for_capture()
        try:
            self._open()
            parallel._segmenter = self
            trainer._pending_choices = list(self.choices)
            try:
                self.values = trainer.train(self.batch)
            finally:
                parallel._segmenter = None
            if self.segments:  # after the last collective nothing may follow: a graph without nodes cannot be replayed
                self._tick = torch.zeros(1, device="cuda")
                self._tick.add_(1.0)
            self._close(None)
        except BaseException:
            self._abort()
            raise
        finally:
            trainer.writer = writer
            if gc_was_on:
                gc.enable()
        self._cb_state = self._codebook_state()
        self._keys = list(self.values._pending[0]) if self.values._pending else []
        self._vec = self.values._pending[1] if self.values._pending else None
        self._index = self.values._pending[3] if self.values._pending else None
        self._zero = [k for k in self.values.keys() if k not in self._keys]

    # ---- capture of one segment
    def _open(self):
        # capture_error_mode "thread_local": under hipStreamCaptureModeGlobal (torch's default) a HIP call that is illegal
        # during capture fails on EVERY thread of the process - ProcessGroupNCCL's watchdog thread polls the events of
        # collectives that were issued before the capture with hipEventQuery every 100 ms, gets
        # hipErrorStreamCaptureUnsupported, terminates with the exception and the process dies with SIGABRT (round 3:
        # sporadic abort of the RCCL test, never with gloo - gloo has no watchdog).  Everything a captured step launches
        # is enqueued by this thread, so thread-local checking loses nothing.
        g = torch.cuda.CUDAGraph()
        mode = cfg.capture_mode  # ("global" reproduces the round-3 abort)
        self._ctx = (g, torch.cuda.graph(g, pool=self.pool, stream=self.stream, capture_error_mode=mode))
        self._ctx[1].__enter__()

    def _close(self, tensor):
        g, ctx = self._ctx
        self._ctx = None
        ctx.__exit__(None, None, None)
        self.segments.append((g, tensor))

    def _abort(self):
        if self._ctx is not None:
            try:
                self._ctx[1].__exit__(None, None, None)
            except Exception:
                pass
            self._ctx = None

    def collective(self, t):
        """parallel.all_reduce_sum inside the captured step: a segment boundary.  Nothing is exchanged while capturing (a
        capture executes nothing: `t` holds stale values, and a live collective would only leave work for the backend's
        watchdog to poll during the next capture).  The ranks stay aligned without it: a replayed step issues exactly the
        collectives of an eager step, in the same order on the same tensors - so even a rank whose capture fails, and
        which therefore steps eagerly, pairs up with ranks that replay (``agree_on_capture`` then takes all of them to the
        eager path together)."""
        self._close(("reduce", list(t) if isinstance(t, (list, tuple)) else [t]))
        self._open()

    def collective_start(self, t):
        """parallel.all_reduce_start inside the captured step: a replay issues the all-reduce of `t` here WITHOUT waiting for
        it (RCCL: on the collective's own stream) and goes on replaying; the segment up to ``collective_finish`` - work that
        does not touch `t` - runs beside the collective."""
        key = len(self.segments)
        self._close(("start", key, t))
        self._open()
        return key

    def collective_finish(self, key):
        self._close(("finish", key))
        self._open()

    def _codebook_state(self):
        return tuple(getattr(m, "codebook_epoch", -1) for m in self.trainer.model.values())

    def step(self, batch=None):
        if self._codebook_state() != self._cb_state:
            # a codebook was written outside the captured step (load_state_dict, touch(), another graph's or an eager step's
            # blend under data parallelism): an EMA blend of a single process leaves the search images current, so the step
            # may have been captured WITHOUT the launch that rebuilds them before its first search - do it here
            for m in self.trainer.model.values():
                if hasattr(m, "refresh_images"):
                    m.refresh_images(force=True)
            self._cb_state = self._codebook_state()
        if batch is not None:
            for k, v in batch.items():
                if isinstance(v, torch.Tensor):
                    self.batch[k].copy_(v, non_blocking=True)
                else:
                    self.batch[k] = v
        inflight = {}
        for g, act in self.segments:
            g.replay()
            if act is None:
                continue
            if act[0] == "reduce":
                for t in act[1]:
                    parallel.all_reduce_now(t)
            elif act[0] == "start":
                inflight[act[1]] = parallel.all_reduce_async(act[2])
            else:  # "finish": the stream waits for the collective (the host does not)
                work = inflight.pop(act[1], None)
                if work is not None:
                    work.wait()
        # the values of THIS replay: the device vector is rewritten by the next one
        if self._vec is None:
            return LossValues(self._keys, None, None, self._zero)
        host = torch.empty(self._vec.shape, dtype=self._vec.dtype, pin_memory=True)
        host.copy_(self._vec, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        return LossValues(self._keys, host, done, self._zero, self._index)


class BaseTrainer(object):
    def __init__(self, model, optimizer, criterion, dataloader, writer, expdir, conf, feat_conf, scheduler=None,
                 scaler=None, resume=0, device="cuda", n_jobs=-1):
        self.model, self.optimizer, self.criterion = model, optimizer, criterion
        self._graphs = {}  # captured steps by (phase, batch shape, random choices); None: capture failed, eager for good
        self._pending_choices = None  # per-step random choices drawn ahead of train() (train_graphed)
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
        # The random choices inside the cyclegan / stargan losses (crank/net/trainer/trainer_cyclegan.py:166,
        # trainer_stargan.py:91) come from a generator of the trainer's own, started from the global state at
        # construction: a run seeded like the reference's (random.seed before building the trainer) sees the
        # reference's sequence, and later draws of the dataset from the global stream - rank-specific under data
        # parallelism - cannot make ranks choose differently (their collectives would no longer pair up).
        self.rng = random.Random()
        self.rng.setstate(random.getstate())

    # ------------------------------------------------------------------ loop
    def run(self, flag="train", tdir=None):
        self.flag = flag
        if flag != "train":
            return self._run_eval(flag)
        while not self.finish_train:
            self._tr_step()
        logging.info("Finish training")

    # ------------------------------------------------------------------ HIP-graph replay of the training step
    def _mode_signature(self):
        """Everything host-side that selects which kernels a train() call launches (a graph is valid for one value)."""
        return tuple(sorted((k, v) for k, v in vars(self).items()
                            if (k.endswith("_flag") or k == "stop_generator") and isinstance(v, bool)))

    def _draw_step_choices(self):
        """The random choices train() will take this step (cyclegan: which fake the discriminator sees, stargan with
        ``switch_update``: which side it updates on), drawn from the trainer's generator in the order train() asks for
        them.  A captured graph freezes host decisions, so there is one graph per outcome and the host picks."""
        return ()

    def _choose(self, options):
        """``self.rng.choice(options)``, or the value ``train_graphed`` drew ahead for this step."""
        if self._pending_choices:
            return self._pending_choices.pop(0)
        return self.rng.choice(options)

    def train_graphed(self, batch):
        """``train(batch)`` replayed from captured graphs (conf["hip_graph"]).  The first three steps of every
        (phase, batch shape, random choice) run eagerly - lazy allocations, and they are the warm-up of the capture -
        then the step is captured once and replayed.  Falls back to ``train`` for good if a capture fails."""
        choices = tuple(self._draw_step_choices())
        if self._graphs is None:
            self._pending_choices = list(choices)
            return self.train(batch)
        sig = (self._mode_signature(), choices,
               tuple((k, tuple(v.shape), str(v.dtype)) for k, v in sorted(batch.items()) if isinstance(v, torch.Tensor)))
        slot = self._graphs.setdefault(sig, [0, None])
        if slot[1] is None:
            slot[0] += 1
            if slot[0] <= 3:
                self._pending_choices = list(choices)
                return self.train(batch)
            try:
                slot[1] = GraphedStep(self, batch, warmup=0, choices=choices)
            except (RuntimeError, ValueError) as e:
                logging.warning("hip_graph: the step is not capturable (%s); running eagerly", e)
                slot[1] = None
                torch.cuda.synchronize()
            # data parallel: every rank replays or none does (all ranks reach this point in the same step - the
            # signature is built from the shared mode flags, the shared random choices and the per-rank batch shape)
            if not parallel.agree_on_capture(slot[1] is not None):
                if slot[1] is not None:
                    logging.warning("hip_graph: another rank could not capture the step; running eagerly")
                self._graphs = None
                self._pending_choices = list(choices)
                return self.train(batch)
            # (the capture itself executes nothing: the step of this call is the first replay)
        values = slot[1].step(batch)
        if self.writer is not None and self.steps % self.conf["n_steps_print_loss"] == 0:
            w = self.writer.get("train") if isinstance(self.writer, dict) else None
            if w is not None:
                for k, v in values.items():
                    w.add_scalar("loss/{}".format(k), v, self.steps)
                w.flush()
        return values

    def _tr_step(self):
        for batch in self.dataloader["train"]:
            batch = to_device(batch, self.device)
            values = self.train_graphed(batch) if self.conf.get("hip_graph") else self.train(batch, phase="train")
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
    def _get_loss_dict(self, batch=None):
        if batch is not None:
            parallel.prepare_step(batch, self.conf)  # C3: the step's mask / target counts, one message
        return LossBook()

    def _parse_loss(self, loss):
        """floats for logging (basetrainer.py:208-215) through ONE device->host copy instead of an .item() per
        key - and without making the host wait for it: on the device the copy is enqueued into pinned memory
        behind the step, the returned dict reads it (waiting for the copy's event) when a value is first
        looked at.  A loop that logs every n-th step keeps the GPU fed across steps."""
        keys = [k for k, v in loss.items() if isinstance(v, torch.Tensor)]
        others = [k for k in loss if k not in keys]
        if not keys:
            return LossValues(keys, None, None, others)
        index = None if parallel.is_dist() else self._arena_index([loss[k] for k in keys])
        if index is not None:
            # every value is a result scalar of a loss op in the step's arena (ops._ScalarArena): the arena IS the vector
            vec = self._arena.buf
        else:
            vec = torch.stack([loss[k].detach().reshape(()).float() for k in keys])
        if parallel.is_dist():  # per-rank shares -> global values
            parallel.all_reduce_sum(vec)
        if not vec.is_cuda:
            return LossValues(keys, vec, None, others)
        if torch.cuda.is_current_stream_capturing():
            # inside a captured step (GraphedStep): the values stay in a device vector the graph rewrites on every
            # replay; they are fetched when somebody reads them
            return LossValues(keys, vec, None, others, index)
        host = torch.empty(vec.shape, dtype=vec.dtype, pin_memory=True)
        host.copy_(vec, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        return LossValues(keys, host, done, others, index)

    _arena = None

    def _open_arena(self):
        """The step's scalar arena (GPU only): opened by train() before the first loss op of a step."""
        from ... import ops

        self._arena = ops.begin_scalar_arena(self.device if torch.device(self.device).type == "cuda" else None)

    def _close_arena(self):
        """End of train(): loss ops called outside a step (a criterion used directly, bench.py's stacks_alone, a second
        trainer of the process) allocate their own result scalars again instead of slicing this step's buffer - which, after
        a capture, lives in the graph's private pool and is rewritten by every replay.  The step's own values keep the buffer
        alive through their reference.  The per-step caches that pin the batch go with it."""
        from ... import ops

        ops.begin_scalar_arena(None)
        self._cond_cache = self._label_cache = None

    def _arena_index(self, tensors):
        """Positions of the 0-dim fp32 tensors in the open arena, or None when any of them lives elsewhere."""
        a = self._arena
        if a is None:
            return None
        base = a.buf.untyped_storage().data_ptr()
        index = []
        for t in tensors:
            if not (t.is_cuda and t.dtype == torch.float32 and t.numel() == 1 and t.untyped_storage().data_ptr() == base):
                return None
            index.append(t.storage_offset() - a.buf.storage_offset())
        return index

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
        if not self.conf["use_spkr_embedding"]:
            f0 = self._get_f0_condition(batch, cv_spkr_name, use_cvfeats) if self.conf["decoder_f0"] else None
            return (torch.cat([f0, h_onehot], dim=-1) if f0 is not None else h_onehot), None
        # with a speaker embedding the generator concatenates [lcf0 | uv | embedding] itself: a model that takes the two
        # F0 streams as a pair (VQVAE2.can_pair_f0) saves the intermediate concatenation
        pair = getattr(self.model.get("G"), "can_pair_f0", False)
        f0 = self._get_f0_condition(batch, cv_spkr_name, use_cvfeats, pair=pair) if self.conf["decoder_f0"] else None
        return f0, h

    def _get_f0_condition(self, batch, cv_spkr_name, use_cvfeats=False, pair=False):  # basetrainer.py:277-289
        if cv_spkr_name is not None:
            lcf0 = self._get_cvf0(batch, cv_spkr_name)
        else:
            lcf0 = batch["cv_lcf0"] if use_cvfeats else batch["lcf0"]
        return (lcf0, batch["uv"]) if pair else torch.cat([lcf0, batch["uv"]], dim=-1)

    def _get_spkr_conditions(self, batch, cv_spkr_name, use_cvfeats=False):  # basetrainer.py:291-309
        if cv_spkr_name is not None:
            B, T, _ = batch["in_feats"].shape
            num = self.spkrs[cv_spkr_name]
            h = torch.full((B, T), num, dtype=torch.long, device=batch["in_feats"].device)
            h_onehot = torch.zeros(B, T, self.n_spkrs, device=h.device)
            h_onehot[..., num] = 1.0
        else:
            key = "cv" if use_cvfeats else "org"
            h = batch[f"{key}_h"]
            h_onehot = batch[f"{key}_h_onehot"]
        # the utterance's label on every frame, -100 pads included (basetrainer.py:303-308 clones and overwrites)
        return self._filled_labels(batch, h, cv_spkr_name if cv_spkr_name is not None else ("cv" if use_cvfeats else "org")), h_onehot

    def _filled_labels(self, batch, h, key):
        """``h`` with every frame carrying its utterance's first label (the reference's ``h[:, :] = h[:, 0:1]`` on a clone,
        basetrainer.py:303-308, trainer_lsgan.py:202-204).  On the GPU a stride-0 VIEW of the batch's label tensor: the
        lookup kernels read the first label of each utterance themselves (ops._label_runs), nothing is copied.  On the
        CPU (and for label tensors that are not contiguous) the filled copy, made once per (batch, key) and step."""
        if h.is_cuda and h.dim() == 2 and h.is_contiguous():
            return h[:, 0:1].expand(-1, h.shape[1])
        cache = getattr(self, "_label_cache", None)
        if cache is None or cache[0] is not batch:
            cache = self._label_cache = (batch, {})
        key = (key, h.data_ptr(), h._version)  # (a label tensor rewritten in place between calls is a new entry)
        if key not in cache[1]:
            cache[1][key] = h[:, 0:1].expand(-1, h.shape[1]).contiguous()
        return cache[1][key]

    # ------------------------------------------------------------------ decode side
    def _scaler_stats(self):
        """Device copies (float64) of the scaler statistics the decode side needs, built once."""
        if self.scaler is None:
            raise ValueError("converting F0 to a named speaker needs the feature scaler")
        if getattr(self, "_stats", None) is None:
            from .dataset import ScalerStats

            names = sorted(self.spkrs, key=self.spkrs.get)
            self._stats = ScalerStats(self.scaler, names, [self.conf["output_feat_type"], "lcf0"],
                                      list(self.conf.get("ignore_scaler", [])), self.device)
        return self._stats

    def _decode_f0(self, batch, cv_names, cv_lcf0=False, f0=False, normed=False):
        """crk_decode_f0 on a padded batch: float64 (B,T,1) tensors for the requested outputs."""
        from ... import _lib
        from ..._lib import check, ptr, stream_ptr

        st = self._scaler_stats()
        lcf0 = batch["lcf0"].contiguous()
        if not lcf0.is_cuda:
            raise RuntimeError("F0 conversion runs on the device; the batch is on " + str(lcf0.device))
        B, T = lcf0.shape[:2]
        dev = lcf0.device
        org = torch.tensor([self.spkrs[n] for n in batch["org_spkr_name"]], dtype=torch.int32, device=dev)
        cv = torch.tensor([self.spkrs[n] for n in cv_names], dtype=torch.int32, device=dev)
        uv = batch["uv"].contiguous()
        mk = lambda on: torch.empty(B, T, 1, dtype=torch.float64, device=dev) if on else None  # noqa: E731
        o_cv, o_f0, o_n = mk(cv_lcf0), mk(f0), mk(normed)
        g = st.lcf0_host or (0.0, 1.0)
        check(_lib.lib().crk_decode_f0(ptr(lcf0), ptr(uv), B, T, ptr(org), ptr(cv), g[0], g[1], 1 if st.lcf0_host else 0,
                                       ptr(st.spk_mean), ptr(st.spk_std), ptr(o_cv), ptr(o_f0), ptr(o_n), stream_ptr()), "decode_f0")
        return o_cv, o_f0, o_n

    def _get_cvf0(self, batch, spkr_name):
        """F0 linear transform org -> spkr in the log domain on the scaler statistics, renormalised
        (basetrainer.py:311-320 + dataset.py:288-293): float64 on the device, returned as float32."""
        _, _, normed = self._decode_f0(batch, [spkr_name] * batch["lcf0"].shape[0], normed=True)
        return normed.float()

    def _store_features(self, batch, outputs, cv_spkr_name=None):
        """basetrainer.py:346-386 without the file names: per utterance the de-normalised converted
        features, converted F0 and their normalised versions, cut to the utterance length.  The
        arithmetic runs on the padded batch on the device; slicing is the only per-utterance work."""
        from .dataset import scaler_apply

        st = self._scaler_stats()
        ftype = self.conf["output_feat_type"]
        B = outputs["decoded"].shape[0]
        names = [cv_spkr_name or n for n in batch["org_spkr_name"]]
        cv_cf0, f0, normed = self._decode_f0(batch, names, cv_lcf0=True, f0=True, normed=True)
        feat = outputs["decoded"].detach().float().contiguous()
        rm = None
        if ftype == "mcep" and not self.conf.get("use_mcep_0th", False):  # basetrainer.py:360-368
            feat = torch.cat([batch["mcep_0th"], feat], dim=-1)
            rm = torch.cat([batch["mcep_0th"], batch["in_feats"]], dim=-1)
        inv = (lambda x: scaler_apply(x, *st.feat[ftype], inverse=True)) if ftype in st.feat else (lambda x: x)
        den, rden = inv(feat), (inv(rm) if rm is not None else None)
        flens = [int(v) for v in batch["flen"].tolist()]
        out = []
        for n in range(B):
            L = flens[n]
            d = {"feats": den[n, :L], "normed_feat": feat[n, :L], "lcf0": cv_cf0[n, :L], "uv": batch["uv"][n, :L],
                 "f0": f0[n, :L], "normed_lcf0": normed[n, :L], "org_spkr_name": batch["org_spkr_name"][n],
                 "cv_spkr_name": names[n], "flbl": batch["flbl"][n]}
            if ftype == "mcep":
                d["cap"] = batch["cap"][n, :L] if "cap" in batch else None
                d["rmcep"] = rden[n, :L] if rden is not None else None
            out.append(d)
        return out
