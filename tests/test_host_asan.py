"""Host library under AddressSanitizer + UBSan (host code only: sanitizers do not run on the GPU here): the shm ring /
parameter register stress of tests/test_host_tsan.py once more, and the native plane log pass (drl_planelog_observe) on a
ring that wraps every few steps."""
import os
import shutil
import subprocess

import pytest

HOST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "distributed_rl_amd", "csrc", "host")
FLAGS = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"]


def _build(tmp_path, driver):
    exe = tmp_path / driver[:-4]
    cc = subprocess.run(["g++"] + FLAGS + [os.path.join(HOST, "drl_host.cpp"), os.path.join(HOST, driver), "-o", str(exe), "-lrt",
                                            "-pthread"], capture_output=True, text=True, timeout=300)
    if cc.returncode != 0 and "asan" in (cc.stderr or "").lower():
        pytest.skip("libasan not installed: %s" % cc.stderr[-200:])
    assert cc.returncode == 0, cc.stderr[-2000:]
    return exe


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_plane_log_pass_is_asan_clean(tmp_path):
    run = subprocess.run([str(_build(tmp_path, "stress_asan.cpp"))], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and run.stdout.startswith("ok "), (run.stdout[-500:], run.stderr[-2000:])


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_shm_runtime_is_asan_clean(tmp_path):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")        # segments are unlinked, mappings live until exit
    run = subprocess.run([str(_build(tmp_path, "stress_tsan.cpp")), "1"], capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0 and "stress ok" in run.stdout, (run.stdout[-500:], run.stderr[-2000:])
