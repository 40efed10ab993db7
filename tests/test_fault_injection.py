"""Fault injection (the reference has only `trap kill` + `wait -n`, run.sh:3,44): a producer killed in the
middle of a ring write, stale heartbeat rows, an actor SIGKILLed under --restart-actors, an actor SIGSTOPped under
--kill-stalled. Real processes, CPU only."""
import os
import re
import signal
import subprocess
import sys
import threading
import time

import numpy as np
import pytest

from distributed_rl_amd.runtime.heartbeat import HeartbeatTable
from distributed_rl_amd.transport import ShmExperience, codec, shm_names

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _block(n=2, seed=0):
    rng = np.random.default_rng(seed)
    return dict(state=rng.integers(0, 256, (n, 84, 84, 4), dtype=np.uint8),
                next_state=rng.integers(0, 256, (n, 84, 84, 4), dtype=np.uint8),
                action=np.arange(n, dtype=np.int32), reward=np.ones(n, np.float32), done=np.zeros(n, np.float32),
                prio=np.ones(n, np.float32))


def test_ring_skips_reservation_of_dead_producer(run_id):
    """A producer that dies between reserve and commit must not wedge the single-consumer ring: the consumer
    skips the slot once its owner is gone; blocks committed behind it arrive; a LIVE slow producer is never skipped."""
    name = shm_names(run_id)[0]
    ring = ShmExperience(name, create=True, n_slots=4, slot_bytes=1 << 20)
    try:
        ring.set_abandon_grace(0.2)
        code = ("import ctypes,os,sys; sys.path.insert(0,%r)\n"
                "from distributed_rl_amd.transport import ShmExperience\n"
                "r = ShmExperience(%r)\n"
                "t = ctypes.c_uint64()\n"
                "assert r._lib.drl_ring_try_reserve(r._h, ctypes.byref(t))\n"
                "os._exit(0)\n") % (ROOT, name)
        subprocess.run([sys.executable, "-c", code], check=True, timeout=60)
        assert ring.size() == 1 and ring.drain() == []                 # reserved, never committed
        assert ring.push(_block(seed=1), codec.KIND_APEX, 7, 0)        # a healthy producer queues behind it
        assert ring.drain() == []                                      # first look: the stall clock starts
        time.sleep(0.3)
        got = ring.drain()
        assert len(got) == 1 and np.array_equal(got[0]["state"], _block(seed=1)["state"])
        assert ring.stats()["abandoned"] == 1
        # a live producer that is merely slow keeps its slot
        import ctypes
        t = ctypes.c_uint64()
        assert ring._lib.drl_ring_try_reserve(ring._h, ctypes.byref(t))
        assert ring.drain() == []
        time.sleep(0.3)
        assert ring.drain() == [] and ring.stats()["abandoned"] == 1
        assert ring._lib.drl_ring_commit(ring._h, t.value, 0, 0) == 1  # ... and may still commit (an empty block here)
        # the ring keeps working for a full lap afterwards
        for k in range(6):
            assert ring.push(_block(seed=k), codec.KIND_APEX, 1, k)
            while True:
                got = ring.acquire()
                if got is None:
                    break
                ring.release(got[0])
        assert ring.size() == 0
        # a commit that comes AFTER the consumer gave the reservation up fails cleanly (it must not overwrite the sequence
        # number of a slot that belongs to a later lap): reserved by a process that exits, skipped, then committed "late"
        code2 = code.replace("os._exit(0)", "print(t.value, flush=True); os._exit(0)")
        late = int(subprocess.run([sys.executable, "-c", code2], check=True, timeout=60, capture_output=True, text=True).stdout.split()[-1])
        assert ring.drain() == []
        time.sleep(0.3)
        assert ring.drain() == [] and ring.stats()["abandoned"] == 2
        assert ring._lib.drl_ring_commit(ring._h, late, 0, 0) == 0
        for k in range(6):                                              # every slot is still reservable
            assert ring.push(_block(seed=k), codec.KIND_APEX, 1, k)
            assert len(ring.drain()) == 1
    finally:
        ring.close()


def test_heartbeat_rows_of_dead_children_are_freed(run_id):
    tab = HeartbeatTable(shm_names(run_id)[2], create=True)
    try:
        hb = tab.register("actor", 3)
        hb.tick(10)
        rows = tab.snapshot()
        assert len(rows) == 1 and rows[0]["pid"] == os.getpid() and rows[0]["role"] == "actor" and rows[0]["ident"] == 3
        assert tab.free_pid(os.getpid()) == 1 and tab.snapshot() == []
        assert tab.free_pid(os.getpid()) == 0
        again = tab.register("learner", 0)                              # the slot is reusable
        assert again.slot == hb.slot and tab.snapshot()[0]["pid"] == os.getpid()
    finally:
        tab.close()


def _launch(run_id, tmp_path, extra, learner_steps=40):
    save = tmp_path / "models"
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1", DRL_NUM_ENVS="2")
    cmd = [sys.executable, "-u", "-m", "distributed_rl_amd.runtime.launcher", "--num-actors", "2", "--conf", "config/all.conf",
           "--run-id", run_id, "--timeout", "240",
           "--learner-args", "--max-steps %d --batch-size 16 --start-memory-size 600 --save-timing 1000 --save-dir %s --quiet "
                             "--device cpu" % (learner_steps, save),
           "--actor-args", "--quiet --device cpu --actor-batch 16"] + extra
    p = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    lines = []
    th = threading.Thread(target=lambda: [lines.append(ln) for ln in p.stdout], daemon=True)
    th.start()
    return p, lines, th


def _actor_pid(run_id, ident, timeout=120, exclude=None):
    """pid of actor ``ident`` from the heartbeat table, once it has started ticking."""
    t0 = time.time()
    while time.time() - t0 < timeout:
        try:
            tab = HeartbeatTable(shm_names(run_id)[2], create=False)
        except OSError:
            time.sleep(0.2)
            continue
        try:
            for row in tab.snapshot():
                if row["role"] == "actor" and row["ident"] == ident and row["beats"] > 0 and row["pid"] != exclude:
                    return row["pid"]
        finally:
            tab._lib.drl_beat_close(tab._h, b"")
            tab._h = None
        time.sleep(0.2)
    raise AssertionError("actor %d never registered a heartbeat" % ident)


def test_sigkill_actor_is_restarted_and_learner_finishes(tmp_path, run_id):
    p, lines, th = _launch(run_id, tmp_path, ["--restart-actors"])
    try:
        pid = _actor_pid(run_id, 2)
        os.kill(pid, signal.SIGKILL)                                    # fault: an actor dies mid-run
        rc = p.wait(timeout=300)
    finally:
        if p.poll() is None:
            p.kill()
    th.join(10)
    out = "".join(lines)
    assert re.search(r"actor_2 exited with -9, restarting", out), out[-3000:]
    assert "learner exited with 0" in out and rc == 0, out[-3000:]       # the learner reached its step budget regardless


def test_sigstop_actor_is_killed_as_stalled_and_restarted(tmp_path, run_id):
    p, lines, th = _launch(run_id, tmp_path, ["--restart-actors", "--kill-stalled", "--stall-seconds", "3", "--watch-period", "1"],
                           learner_steps=200)
    try:
        pid = _actor_pid(run_id, 1)
        os.kill(pid, signal.SIGSTOP)                                    # fault: an actor hangs (no more heartbeats)
        t0 = time.time()
        while time.time() - t0 < 120 and not any("terminating stalled actor_1" in ln for ln in lines):
            time.sleep(0.2)
        new_pid = _actor_pid(run_id, 1, exclude=pid)                                 # a fresh actor_1 is ticking again
        rc = p.wait(timeout=300)
    finally:
        if p.poll() is None:
            p.kill()
        try:
            os.kill(pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
    th.join(10)
    out = "".join(lines)
    assert "terminating stalled actor_1" in out, out[-3000:]
    assert re.search(r"actor_1 exited with -?\d+, restarting", out), out[-3000:]
    assert new_pid != pid
    assert "learner exited with 0" in out and rc == 0, out[-3000:]


def test_heal_command_from_fault_note(tmp_path):
    """What the launcher starts in place of a learner that left with status 75: attached, resumed from the checkpoint the note
    names (or the newest one), the paths the fired guards name switched off (all of them when there is no note)."""
    import json
    from distributed_rl_amd.runtime.launcher import heal_learner, newest_checkpoint
    d = str(tmp_path)
    for t in (0, 10000, 200):
        open(os.path.join(d, "model_%d.pth" % t), "w").close()
    open(os.path.join(d, "model_200.aux.pth"), "w").close()
    assert newest_checkpoint(d).endswith("model_10000.pth")
    cmd, env, why = heal_learner(["python", "learner_node.py", "--resume", "old.pth", "-a", "r2d2"], {"A": "1"}, d)
    assert cmd[-3:] == ["--attach", "--resume", os.path.join(d, "model_10000.pth")] and "old.pth" not in cmd
    assert env == {"A": "1", "DRL_PERSISTENT_LSTM": "0", "DRL_PERSISTENT_BPTT": "0", "DRL_XGMI_AR": "0"}
    json.dump({"error": "learner step discarded: a bounded in-kernel wait timed out (direct xGMI all-reduce)",
               "checkpoint": os.path.join(d, "model_200.pth")}, open(os.path.join(d, "learner_fault.json"), "w"))
    cmd, env, why = heal_learner(["python", "learner_node.py"], {}, d)
    assert cmd[-1].endswith("model_200.pth") and env == {"DRL_XGMI_AR": "0"} and "xGMI" in why


def test_learner_whose_guard_fired_is_restarted_from_its_checkpoint_next_to_running_actors(tmp_path, run_id):
    """Self-healing: the learner's step guard fires at its 10th step (injected; on a GPU it is a persistent
    kernel's bounded wait), it writes its state + a fault note and leaves with status 75; the launcher starts a fresh learner
    that ATTACHES to the live ring / register with the offending path switched off and resumes from that checkpoint. The
    actors are never restarted and the run ends with the healed learner's exit status 0."""
    os.environ["DRL_INJECT_GUARD_AT_STEP"] = "10"
    try:
        p, lines, th = _launch(run_id, tmp_path, [], learner_steps=40)
    finally:
        del os.environ["DRL_INJECT_GUARD_AT_STEP"]
    try:
        pid_before = _actor_pid(run_id, 1)
        rc = p.wait(timeout=300)
    finally:
        if p.poll() is None:
            p.kill()
    th.join(10)
    out = "".join(lines)
    assert re.search(r"learner guard fired .*persistent LSTM.*restarting from .*model_10\.pth with DRL_PERSISTENT_LSTM=0 \(heal 1 of 3\)",
                     out), out[-3000:]
    assert "learner exited with 0" in out and rc == 0, out[-3000:]
    assert not re.search(r"actor_\d exited with .*restarting", out), out[-3000:]
    import glob
    saved = sorted(os.path.basename(f) for f in glob.glob(str(tmp_path / "models" / "model_*.pth")) if ".aux" not in f)
    assert "model_10.pth" in saved and os.path.exists(str(tmp_path / "models" / "learner_fault.json"))
    assert pid_before > 0
