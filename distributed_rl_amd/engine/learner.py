"""GPU learner process: the runtime around ApexEngine / R2D2Engine (replaces Learner.optimize_loop +
the Replay thread of the reference, ape_x/learner.py:57-99, ape_x/replay.py:24-35, on an MI355X).

Loop: ``AsyncIngest.service()`` (engine/ingest.py: a host thread stages experience blocks into pinned memory, a side
HIP stream DMAs them into the HBM replay while the learner keeps stepping; only two tiny tree launches per block touch
the learner stream - no sleep-based back-pressure) → one graph replay per learner step → every
``publish_period`` steps / ``publish_interval`` seconds publish the flat fp32 parameters (a D2D snapshot on the learner
stream, D2H + register write on a side stream / thread: the loop never waits for PCIe) → periodic checkpoint in the
reference's ``model_<t>.pth`` format (+ sidecar with optimizer moments, target net and step so ``--resume`` works).
Wall-clock phase timers (ingest / step / publish / save) and the ingest counters go to the metrics sinks every
``fit_timing`` steps.
"""
from __future__ import annotations

import os
import threading
import time
from collections import deque
from typing import Optional

import torch

from ..config import ApexConfig, R2D2Config
from ..models import DuelingDQN, DuelingLSTMDQN
from ..transport import ExperienceChannel, ParamChannel


# exit status of a learner process whose step guard fired (engine/guard.py: a bounded in-kernel wait timed out, the step was
# discarded, parameters / optimizer moments / priorities are intact): EX_TEMPFAIL. The launcher restarts the learner from
# the checkpoint written on the way out, with the path that timed out switched off (runtime/launcher.py heal_learner).
EXIT_GUARD = 75
FAULT_NOTE = "learner_fault.json"


class _AsyncPublisher:
    """Parameter publication off the learner's critical path: ``publish()`` snapshots the flat fp32 parameters with one
    D2D copy on the learner stream (a consistent version, ~10 us); a side stream takes it from there and a thread hands
    it to the parameter channel once the stream work has completed. What the side stream does depends on what the
    channel asks for (transport.ParamChannel.wants_host / wants_device): a register in shared memory needs a host copy
    (D2H into pinned memory), a broadcast lane over RCCL takes the DEVICE snapshot as it is - no PCIe at all
    (reference: ``SET params`` of a state_dict moved to the actor device, ape_x/learner.py:83-84). A publication that
    is still in flight is not queued behind: the channel is a latest-value register, the next period
    publishes."""

    def __init__(self, flat: torch.Tensor, params: ParamChannel):
        self.flat, self.params = flat, params
        self.snap = torch.empty_like(flat)
        self.want_host = bool(getattr(params, "wants_host", True))
        self.want_dev = bool(getattr(params, "wants_device", False))
        self.host = torch.empty(flat.numel(), dtype=torch.float32).pin_memory() if self.want_host else None
        self.stream = torch.cuda.Stream(flat.device)
        self._busy = threading.Event()
        self._go = threading.Event()
        self._stop = False
        self._done_ev = None
        self.version = 0
        self.skipped = 0
        self.error = None
        self.last_us = 0.0                 # wall clock of the channel's part of the last publication
        self._th = threading.Thread(target=self._run, name="drl-publish", daemon=True)
        self._th.start()

    def _run(self) -> None:
        if self.flat.is_cuda:
            torch.cuda.set_device(self.flat.device)
        while True:
            self._go.wait()
            self._go.clear()
            if self._stop:
                return
            try:
                self._done_ev.synchronize()
                t0 = time.perf_counter()
                with torch.cuda.stream(self.stream):         # collectives of a lane are ordered on OUR side stream
                    self.version = self.params.publish2(self.host.numpy() if self.want_host else None,
                                                        self.snap if self.want_dev else None)
                self.stream.synchronize()                    # the snapshot is free again only when the lanes have read it
                self.last_us = (time.perf_counter() - t0) * 1e6
            except BaseException as e:  # noqa: BLE001
                self.error = e
            finally:
                self._busy.clear()

    def publish(self, wait: bool = False) -> int:
        if self.error is not None:
            raise RuntimeError("parameter publisher died") from self.error
        if self._busy.is_set():
            if not wait:
                self.skipped += 1
                return self.version
            while self._busy.is_set():
                time.sleep(0.0005)
        self._busy.set()
        cur = torch.cuda.current_stream(self.flat.device)
        self.snap.copy_(self.flat, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(cur)
        self.stream.wait_event(ev)
        with torch.cuda.stream(self.stream):
            if self.want_host:
                self.host.copy_(self.snap, non_blocking=True)
            self._done_ev = torch.cuda.Event()
            self._done_ev.record(self.stream)
        self._go.set()
        if wait:
            while self._busy.is_set():
                time.sleep(0.0005)
        return self.version

    def close(self) -> None:
        self._stop = True
        self._go.set()
        self._th.join(2.0)


class HipLearner:
    sync_every = 4096            # learner steps between two device-wide synchronizes of optimize_loop (see there)

    def __init__(self, algorithm: str, n_action: int, cfg, exp: ExperienceChannel, params: ParamChannel, metrics=None,
                 replay_size: int = 100_000, device="cuda", heartbeat=None, dist=None, use_graph: bool = True,
                 publish_interval: float = 0.05, max_blocks_per_step: int = 8, seed: int = 0, save_replay: bool = False,
                 reset_channels: bool = True):
        self.algorithm, self.cfg = algorithm, cfg
        self.exp, self.params, self.metrics, self.heartbeat, self.dist = exp, params, metrics, heartbeat, dist
        self.device = torch.device(device)
        self.is_r2d2 = algorithm == "r2d2"
        if reset_channels:                   # DEL params / DEL experience (ape_x/learner.py:37, replay.py:21)
            self.params.reset()
            self.exp.reset()                 # (a learner that takes over a running system keeps both: parameter versions go on)
        replay_size = self._fit_capacity(replay_size, cfg)
        if self.is_r2d2:
            from ..replay.device_seq import DeviceSequenceReplay
            from .r2d2 import R2D2Engine
            self.replay = DeviceSequenceReplay(replay_size, self.device, cfg, seed=seed)
            self.engine = R2D2Engine(n_action, cfg, self.replay, self.device, dist=dist, use_graph=use_graph)
        else:
            from ..replay.device_per import DevicePER
            from .apex import ApexEngine
            # planes per transition of capacity: ~1.1-1.3 with the actors' plane-pool blocks; 8 if the producers send full
            # stacks (codec kind 1: DRL_PLANE_BLOCKS=0 actors, external tools)
            ppi = float(os.environ.get("DRL_PLANES_PER_ITEM", "1.3"))
            self.replay = DevicePER(replay_size, self.device, batch_size=cfg.batch_size, seed=seed, planes_per_item=ppi)
            self.engine = ApexEngine(n_action, cfg, self.replay, self.device, dist=dist, use_graph=use_graph)
        self.publish_interval = publish_interval
        self.save_replay = save_replay          # also write model_<t>.replay.pth (the whole HBM replay: opt-in, large)
        self.max_blocks = max_blocks_per_step
        self.step = 0
        self._last_pub = 0.0
        self.ingested = 0
        self.max_ahead = 3                   # loop turns queued ahead of the GPU
        self.pipe = None
        self._pub = None
        if self.device.type == "cuda":
            from .ingest import AsyncIngest
            # staging buffers in flight: a block lives ~7 learner steps between its retire and its commit (one more step
            # launch, the host's run-ahead, the DMA), so 4 buffers capped the ingest at ~0.5 blocks per learner step - at
            # 100 K transitions/s the actors of the system bench were waiting for ring slots
            self.pipe = AsyncIngest(exp, self.replay, self.device, depth=48)
            # shm register: also keep the two newest versions in a device double buffer that actors on THIS GPU read
            # device to device (transport.ShmParams); rank 0 publishes, so only it exports one
            if hasattr(params, "enable_device_mirror") and (dist is None or dist.rank == 0):
                params.enable_device_mirror(self.device, self.engine.P.flat.numel() * 4)
            self._pub = _AsyncPublisher(self.engine.P.flat, params)
        if self.metrics is None:
            from ..obs import Metrics
            self.metrics_local = Metrics([])
        else:
            self.metrics_local = self.metrics

    def _fit_capacity(self, want: int, cfg) -> int:
        """The replay lives in HBM: clamp the requested item count to what fits next to the engine's activations (75 %
        of the free memory). With every frame stored once the reference's default ``-s 100000`` - 100 000 SEQUENCES for
        R2D2 - is ~34 GB (full stacks: 315 GiB), and an MI355X holds ~20 M Ape-X transitions."""
        import warnings
        if self.device.type != "cuda":
            return want
        free, _ = torch.cuda.mem_get_info(self.device)
        if self.is_r2d2:                 # ~48 planes per sequence: neighbours share theirs (replay/device_seq.py)
            from ..replay.device_seq import planes_per_sequence
            per_item = int(planes_per_sequence(cfg) * 1.08 * 7056) + cfg.total_len * 20 + 2 * 4 * cfg.hidden + 64
        else:                            # ~1.3 planes per transition (replay/device_per.py) + tables
            per_item = int(float(os.environ.get("DRL_PLANES_PER_ITEM", "1.3")) * 7056) + 64
        fit = int(free * 0.75 // per_item)
        if want > fit:
            warnings.warn("replay size %d does not fit in HBM (%.0f GiB free): using %d" % (want, free / 2 ** 30, fit))
            return max(fit, cfg.batch_size)
        return want

    # ---- pieces -------------------------------------------------------------------------------------
    def ingest(self, max_blocks: Optional[int] = None) -> int:
        """One ingest turn between two learner steps; returns the number of items that became sampleable."""
        if self.pipe is not None:
            n = self.pipe.service()
        else:                           # synchronous path (CPU tests of this class)
            n = 0
            for blk in self.exp.drain(max_blocks or self.max_blocks):
                n += self.replay.ingest_block(blk)
        self.ingested += n
        return n

    def publish(self, wait: bool = False) -> int:
        self._last_pub = time.time()
        if self._pub is not None:
            return self._pub.publish(wait=wait)
        flat = self.engine.P.flat
        return self.params.publish2(flat.cpu().numpy() if getattr(self.params, "wants_host", True) else None,
                                    flat if getattr(self.params, "wants_device", False) else None)

    def close(self) -> None:
        if self.pipe is not None:
            self.pipe.close()
        if self._pub is not None:
            self._pub.close()

    def _replay_path(self, path: str) -> str:
        """Sidecar of this rank's replay shard: ``model_<t>.replay.pth`` for a single learner / rank 0,
        ``model_<t>.replay.r<k>.pth`` for the other data-parallel ranks (every rank owns a different shard)."""
        r = self.dist.rank if self.dist is not None else 0
        return path[:-4] + (".replay.pth" if r == 0 else ".replay.r%d.pth" % r)

    def save(self, save_dir: str) -> str:
        """Data-parallel learners call this TOGETHER: weights, optimizer moments and the step state are identical on
        every rank, so rank 0 alone writes them (the reference's ``model_<t>.pth``, learner.py:96-98); with
        ``--save-replay`` every rank writes its own shard."""
        os.makedirs(save_dir, exist_ok=True)
        path = os.path.join(save_dir, "model_%d.pth" % self.step)
        e = self.engine
        if self.dist is None or self.dist.rank == 0:                        # one writer
            torch.save(self.engine.state_dict(), path)                      # reference format (learner.py:98)
            aux = dict(step=self.step, engine="hip", algorithm=self.algorithm,
                       target_flat=(e.T if not self.is_r2d2 else e.Tg).flat.cpu(),
                       opt_flat=[t.flat.cpu() for t in ((e.SQ,) if not self.is_r2d2 else (e.M1, e.M2))],
                       step_state=e.st.cpu())
            torch.save(aux, path[:-4] + ".aux.pth")
        if self.save_replay:
            if self.pipe is not None:
                self.pipe.flush()                                           # blocks in flight become part of the snapshot
                # flush() turned the ingest machinery over without a learner step in between: a batch drawn before it may
                # point at slots that were retired and rewritten meanwhile, so the next step draws afresh
                if hasattr(e, "redraw"):
                    e.redraw()
            torch.save(self.replay.snapshot(), self._replay_path(path))
        return path

    def resume(self, path: str) -> int:
        from ..ops import hip as H
        e = self.engine
        e.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))
        aux_path = path[:-4] + ".aux.pth"
        if os.path.exists(aux_path):
            aux = torch.load(aux_path, map_location="cpu", weights_only=True)
            self.step = int(aux["step"])
            if aux.get("engine") == "hip":
                tgt = e.T if not self.is_r2d2 else e.Tg
                tgt.flat.copy_(aux["target_flat"])
                H.cast_bf16(tgt.flat, e.Tb.flat)
                for dst, src in zip(((e.SQ,) if not self.is_r2d2 else (e.M1, e.M2)), aux["opt_flat"]):
                    dst.flat.copy_(src)
                e.st.copy_(aux["step_state"])
            else:
                e.sync_target()
        else:
            e.sync_target()
        e.steps_done = self.step
        rp = self._replay_path(path)
        if os.path.exists(rp):                                              # plain tensors only: nothing is executed on load
            n = self.replay.restore(torch.load(rp, map_location="cpu", weights_only=True, mmap=True))
            self.ingested += n
        return self.step

    def wait_memory(self, n: int, timeout: Optional[float] = None, stop_event=None) -> bool:
        t0 = time.time()
        while len(self.replay) <= n:
            if stop_event is not None and stop_event.is_set():
                return False
            if self.ingest(64) == 0:
                time.sleep(0.001 if self.pipe is not None else 0.01)
            if timeout is not None and time.time() - t0 > timeout:
                return False
            if self.heartbeat is not None:
                self.heartbeat.tick(self.step)
        return True

    # ---- main loop -------------------------------------------------------------------------------------
    def optimize_loop(self, max_steps: Optional[int] = None, save_model_dir: str = "./models", stop_event=None,
                      warmup_timeout: Optional[float] = None) -> int:
        cfg = self.cfg
        if not self.wait_memory(max(cfg.batch_size, cfg.start_memory_size), warmup_timeout, stop_event):
            if stop_event is not None and stop_event.is_set():
                return self.step                                  # asked to stop before learning started
            raise TimeoutError("replay never reached the learn-start size")
        if self.dist is not None:
            # shards fill at different rates; the steps that follow contain collectives with bounded waits (the direct
            # xGMI all-reduce polls its peers for at most DRL_XGMI_TIMEOUT_S), so the ranks start them together
            self.dist.barrier()
        end = None if max_steps is None else self.step + max_steps
        t_last, n_last, i_last = time.time(), self.step, self.ingested
        inflight = deque()
        dp = self.dist is not None and self.dist.enabled
        stop_period = 200
        # K learner steps per loop turn where the engine has them as ONE graph replay (ApexEngine.run_steps): a quarter of
        # the launches, event records and ingest turns per step; everything a step needs from the one before it is on the
        # device. Only where no periodic duty (metrics, checkpoint, the agreed stop of data-parallel ranks) falls inside.
        K = int(getattr(self.engine, "steps_per_graph", 1)) if hasattr(self.engine, "run_steps") else 1
        periods = [cfg.fit_timing, cfg.save_timing] + ([stop_period] if dp else [])

        def clear_of_duties(t: int) -> bool:
            return all((t % p == 0 and K <= p) or (t % p) + K <= p for p in periods)
        # A loop that only ever waits for EVENTS lets the HIP runtime's host-side records of finished launches pile up:
        # 2.6 KB per R2D2 step (a graph of ~50 kernels on four streams), 1.6 MB/s, 5.6 GB per hour of resident memory
        # (scripts/debug/learner_rss.py; with a device-wide synchronize now and then: flat). Every SYNC_EVERY steps the loop
        # therefore drains the device once - the run-ahead of three loop turns, a few ms every ~7 s.
        SYNC_EVERY = self.sync_every
        next_sync = self.step + SYNC_EVERY
        while end is None or self.step < end:
            if self.step >= next_sync:
                torch.cuda.synchronize(self.device)
                next_sync = self.step + SYNC_EVERY
            if stop_event is not None:
                # data-parallel ranks leave the loop TOGETHER (the step contains collectives): the request is agreed on
                # every stop_period steps; a single learner looks at its own flag every step
                if dp:
                    if self.step % stop_period == 0 and self.dist.any_true(stop_event.is_set()):
                        break
                elif stop_event.is_set():
                    break
            t = self.step
            ph = self.metrics_local.phase
            with ph("learner/ingest"):
                self.ingest()
            k = K if (K > 1 and (end is None or t + K <= end) and clear_of_duties(t)) else 1
            with ph("learner/step"):
                if k > 1:
                    self.engine.run_steps(k)
                else:
                    self.engine.step()
                # bound the host's run-ahead: launches are asynchronous, and a loop that queues hundreds of graph replays
                # ahead of the GPU makes everything ordered behind "now" on the learner stream (the ingest DMA's start
                # event, a parameter snapshot, a stop request) wait for the whole queue
                ev = torch.cuda.Event()
                ev.record()
                inflight.append(ev)
                if len(inflight) > (self.max_ahead if k == 1 else max(1, self.max_ahead - 1)):
                    inflight.popleft().synchronize()
            pp = cfg.publish_period
            if (t % pp == 0 or (t % pp) + k > pp) and time.time() - self._last_pub >= self.publish_interval:
                with ph("learner/publish"):
                    self.publish()
            if t % cfg.fit_timing == 0:
                now = time.time()
                if self.metrics is not None:
                    st = self.engine.stats()                             # one small D2H every fit_timing steps
                    self.metrics.scalar("Q loss", t, st["loss"], force=True)
                    self.metrics.scalar("Memory size", t, len(self.replay), force=True)
                    if t > n_last:
                        self.metrics.scalar("Updates per second", t, (t - n_last) / max(1e-9, now - t_last), force=True)
                        self.metrics.scalar("Ingested items per second", t, (self.ingested - i_last) / max(1e-9, now - t_last),
                                            force=True)
                    self._report_runtime(t)
                t_last, n_last, i_last = now, t, self.ingested
                if self.heartbeat is not None:
                    self.heartbeat.tick(t)
            if t % cfg.save_timing == 0:
                with ph("learner/save"):
                    self.save(save_model_dir)
            self.step += k
        torch.cuda.synchronize(self.device)
        return self.step

    def _report_runtime(self, t: int) -> None:
        """Ring fill / rejects, ingest backlog and the wall-clock share of every loop phase -> metrics sinks (JSONL)."""
        m = self.metrics
        if hasattr(self.exp, "stats"):
            rs = self.exp.stats()
            m.scalar("Ring fill", t, rs["size"], force=True)
            m.scalar("Ring rejected pushes", t, rs["rejected"], force=True)
        if self.pipe is not None:
            m.scalar("Ingest blocks in flight", t, self.pipe.pending(), force=True)
        for name, rep in self.metrics_local.phase_report().items():
            m.scalar("Phase ms %s" % name, t, rep["mean_ms"], force=True)


def run_hip_learner(args, cfg, n_action, exp, par, metrics, hb, device, dist=None) -> int:
    """``dist``: the data-parallel learners' group (runtime/nodes.py builds it from the launcher's world or from
    torch.distributed.run's environment); None = a single learner."""
    seed_ = (args.seed or 0) + (dist.rank if dist is not None else 0)      # every shard samples its own stream
    learner = HipLearner(args.algorithm, n_action, cfg, exp, par, metrics, args.replaysize, device, heartbeat=hb, dist=dist,
                         seed=seed_, save_replay=getattr(args, "save_replay", False),
                         reset_channels=not getattr(args, "attach", False))
    if dist is not None:                                                   # identical initial weights on every rank
        dist.broadcast_flat(learner.engine.P.flat)
        from ..ops import hip as H
        H.cast_bf16(learner.engine.P.flat, learner.engine.Pb.flat)
        learner.engine.sync_target()
    if args.resume:
        learner.resume(args.resume)
    learner.publish(wait=True)
    # SIGTERM (the launcher's teardown, --timeout): leave the loop at a step boundary - agreed on by all data-parallel
    # ranks - instead of dying inside a step whose peers then wait for this rank's part of a collective
    import signal
    stop = threading.Event()
    try:
        signal.signal(signal.SIGTERM, lambda *_: stop.set())
    except ValueError:                                                     # not the main thread (tests)
        pass
    rc = 0
    try:
        learner.optimize_loop(max_steps=args.max_steps, save_model_dir=args.save_dir, stop_event=stop)
    except KeyboardInterrupt:
        pass
    except RuntimeError as e:
        if "discarded" not in str(e):
            raise
        rc = _guard_exit(learner, args.save_dir, e)
    finally:
        learner.close()
    return rc


def _guard_exit(learner: "HipLearner", save_dir: str, err: BaseException) -> int:
    """The step guard fired: the poisoned step was discarded on the device (engine/guard.py), so what is in HBM is the state
    after the last good step. Write it out, leave a note that names the path that timed out, and return EXIT_GUARD."""
    import json
    import sys
    print("[learner] %s" % err, file=sys.stderr, flush=True)
    note = {"step": learner.step, "error": str(err)[:500], "checkpoint": None}
    try:
        torch.cuda.synchronize(learner.device)
        note["checkpoint"] = learner.save(save_dir)
    except Exception as e2:  # noqa: BLE001 - the note still says what happened; the launcher falls back to an older checkpoint
        note["save_error"] = repr(e2)[:300]
    try:
        os.makedirs(save_dir, exist_ok=True)
        with open(os.path.join(save_dir, FAULT_NOTE), "w") as fh:
            json.dump(note, fh)
    except OSError:
        pass
    return EXIT_GUARD
