"""ThreadSanitizer build of the shared-memory runtime (csrc/host/drl_host.cpp) under a multi-threaded stress harness
(SURVEY 5.2). Host code only: GPU sanitizers are not available on this pool. The one intentional race - the seqlock
copy of the parameter register, validated after the fact by its `writing` word - is suppressed BY NAME; any other
report fails the test."""
import os
import shutil
import subprocess

import pytest

HOST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "distributed_rl_amd", "csrc", "host")


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_shm_runtime_is_tsan_clean(tmp_path):
    exe = tmp_path / "stress_tsan"
    cc = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", os.path.join(HOST, "drl_host.cpp"),
                         os.path.join(HOST, "stress_tsan.cpp"), "-o", str(exe), "-lrt", "-pthread"],
                        capture_output=True, text=True, timeout=300)
    if cc.returncode != 0 and "tsan" in (cc.stderr or "").lower():
        pytest.skip("libtsan not installed: %s" % cc.stderr[-200:])
    assert cc.returncode == 0, cc.stderr[-2000:]
    supp = tmp_path / "tsan.supp"
    supp.write_text("# seqlock: racy copy is detected by the `writing` word and retried (drl_host.cpp)\n"
                    "race:drl_param_publish\nrace:drl_param_read\n")
    env = dict(os.environ, TSAN_OPTIONS="suppressions=%s halt_on_error=0 exitcode=66" % supp)
    run = subprocess.run([str(exe), "1"], capture_output=True, text=True, timeout=600, env=env)
    assert "stress ok" in run.stdout, (run.stdout[-500:], run.stderr[-2000:])
    assert "WARNING: ThreadSanitizer" not in run.stderr, run.stderr[-3000:]
    assert run.returncode == 0, (run.returncode, run.stderr[-1000:])
