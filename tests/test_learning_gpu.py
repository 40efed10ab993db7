"""Does the whole loop LEARN (VERDICT r1 #6b)? SyntheticCatch, vectorised actor with the HIP inference engine -> experience
channel -> graph-captured HIP learner -> parameter channel -> actor, both algorithms (scripts/learn_catch.py). The greedy
policy's mean reward per ball must rise from the untrained level (about -0.6) to clearly positive within 3 000 updates / 61 K
environment steps, and end where the PyTorch oracle learner (runtime/learner.py on the same GPU, same actor, same schedule)
ends. Measured (MI355X, r2): Ape-X HIP -0.67 -> +0.88, torch -> +0.97; R2D2 HIP -0.63 -> +0.94."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))

_results = {}


def _run(algo, kind):
    import learn_catch
    if (algo, kind) not in _results:
        _results[(algo, kind)] = learn_catch.train(algo, kind, rounds=30, log=print)
    return _results[(algo, kind)]


def _late(r):
    """Score at the end of the run = the better of the last two evaluations (rounds 25 and 30). Runs differ in when ingested
    blocks become visible to the sampler (an ingest thread and copy events), and single evaluations of one run swing by
    +-0.15 around its trend: six runs of the same build ended at 0.60 .. 0.86 (round 3, scripts/debug/learn_ab.py)."""
    return max(s for _, s in r["curve"][-2:])


@pytest.mark.timeout(300)
@pytest.mark.parametrize("algo", ["ape_x", "r2d2"])
def test_catch_return_rises_with_the_hip_learner(algo):
    r = _run(algo, "hip")
    print(r)
    assert r["start"] < -0.3                                  # untrained: misses most balls
    assert _late(r) > 0.5 and _late(r) - r["start"] > 1.0
    assert max(s for _, s in r["curve"][:2]) < _late(r)       # it rose during the run, not before it


@pytest.mark.timeout(300)
@pytest.mark.parametrize("algo", ["ape_x", "r2d2"])
def test_hip_learner_tracks_the_torch_learner_on_catch(algo):
    hip, ref = _run(algo, "hip"), _run(algo, "torch")
    print(hip, ref)
    assert ref["final"] > 0.5                                  # the oracle learner learns the task with this schedule ...
    assert _late(hip) > _late(ref) - 0.4                       # ... and the HIP learner ends where it does
