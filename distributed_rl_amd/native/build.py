"""In-tree build of the two native shared objects.

* ``libdrl_host.so`` — g++ -O2, host only (shm ring, param register, heartbeats).
* ``libdrl_hip.so``  — ``hipcc --offload-arch=gfx950`` over every ``csrc/hip/*.hip``; cross-compiles
  without a GPU. Plain C ABI (raw device pointers + ``hipStream_t``), no torch headers, so a file
  compiles in seconds and the ``.so`` is independent of the PyTorch C++ ABI.

Objects are cached per source by content hash under ``native/.build``; the ``.so`` files land next
to this module (git-ignored, but shipped to the GPU box by gpurun).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from typing import List

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
BUILD = os.path.join(HERE, ".build")
HOST_SO = os.path.join(HERE, "libdrl_host.so")
HIP_SO = os.path.join(HERE, "libdrl_hip.so")
ARCH = "gfx950"

HIPCC_FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=%s" % ARCH, "-ffast-math",
               "-Wno-unused-result", "-munsafe-fp-atomics"]
# The loss / priority / optimizer arithmetic (TD targets with the value rescale and its inverse, Huber, IS weights, the
# clip coefficient, RMSprop / Adam) is compiled WITHOUT -ffast-math: IEEE division and square root, no reassociation. These
# kernels are latency- or HBM-bound, so it costs nothing; the MFMA kernels keep it (their scalar code is address arithmetic
# and epilogue scaling). Where a kernel wants an approximate transcendental it says so (__powf, __expf).
PRECISE = {"td.hip", "optim.hip", "head.hip"}


def flags_for(src: str) -> List[str]:
    return [f for f in HIPCC_FLAGS if not (f == "-ffast-math" and os.path.basename(src) in PRECISE)]


def _hash(path: str, extra: str = "") -> str:
    h = hashlib.sha1(extra.encode())
    with open(path, "rb") as fh:
        h.update(fh.read())
    d = os.path.dirname(path)
    for inc in sorted(f for f in os.listdir(d) if f.endswith(".h")):     # every header: they are few and small
        with open(os.path.join(d, inc), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _run(cmd: List[str]) -> None:
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + "\n")
        raise RuntimeError("native build failed: %s" % os.path.basename(cmd[-1]))


def build_host(force: bool = False) -> str:
    src = os.path.join(CSRC, "host", "drl_host.cpp")
    os.makedirs(BUILD, exist_ok=True)
    stamp = os.path.join(BUILD, "host.stamp")
    tag = _hash(src)
    if not force and os.path.exists(HOST_SO) and os.path.exists(stamp) and open(stamp).read() == tag:
        return HOST_SO
    _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", HOST_SO + ".tmp", src, "-lrt", "-pthread"])
    os.replace(HOST_SO + ".tmp", HOST_SO)
    with open(stamp, "w") as fh:
        fh.write(tag)
    return HOST_SO


def hip_sources() -> List[str]:
    d = os.path.join(CSRC, "hip")
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(".hip"))


def build_hip(force: bool = False, jobs: int = 4, save_temps: bool = False) -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(BUILD, exist_ok=True)
    srcs = hip_sources()
    objs, todo = [], []
    for s in srcs:
        tag = _hash(s, " ".join(flags_for(s)))
        obj = os.path.join(BUILD, "%s.%s.o" % (os.path.basename(s)[:-4], tag))
        objs.append(obj)
        if force or not os.path.exists(obj):
            todo.append((s, obj))

    def compile_one(so):
        s, obj = so
        cmd = [hipcc] + flags_for(s) + ["-c", s, "-o", obj]
        if save_temps:
            cmd.insert(1, "-save-temps=obj")
        _run(cmd)

    if todo:
        with ThreadPoolExecutor(max_workers=jobs) as ex:
            list(ex.map(compile_one, todo))
    stamp = os.path.join(BUILD, "hip.stamp")
    tag = " ".join(os.path.basename(o) for o in objs)
    if todo or force or not os.path.exists(HIP_SO) or not os.path.exists(stamp) or open(stamp).read() != tag:
        _run([hipcc, "-shared", "-fPIC", "--offload-arch=%s" % ARCH, "-o", HIP_SO + ".tmp"] + objs)
        os.replace(HIP_SO + ".tmp", HIP_SO)
        with open(stamp, "w") as fh:
            fh.write(tag)
        # drop stale objects
        keep = set(objs)
        for f in os.listdir(BUILD):
            p = os.path.join(BUILD, f)
            if f.endswith(".o") and p not in keep:
                os.remove(p)
    return HIP_SO


def build_all(force: bool = False) -> None:
    build_host(force)
    build_hip(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
    print("built", HOST_SO, HIP_SO)
