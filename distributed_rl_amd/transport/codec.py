"""Wire format of experience blocks: a fixed 64-byte header + raw little-endian arrays.

Replaces ``lz4(pickle((list[Transition], list[float])))`` (ape_x/actor.py:97-98, utils.py:43-53):
frames are uint8 and already dense, so the block is laid out exactly as the replay store wants it
and can be written in place into a shared-memory slot and uploaded to HBM with one copy per array.

kind 1 — Ape-X transitions:   state u8[K,*obs], next_state u8[K,*obs], action i32[K], reward f32[K], done f32[K], prio f32[K]
kind 2 — R2D2 sequences:      frames u8[K,T,*obs], action i32[K,T], reward f32[K,T], done f32[K,T], length i32[K],
                              h0 f32[K,H], c0 f32[K,H], prio f32[K]
kind 3 — Ape-X transitions over a plane pool (every frame once): planes u8[M,obs0,obs1], win_s i32[K], win_s2 i32[K]
                              (index of the first of the obs2 consecutive planes that make up s / s'), action, reward, done,
                              prio as in kind 1. M travels in the header's T field.
kind 4 — R2D2 sequences over a plane pool (every frame once, ACROSS the overlapping sequences of an environment too):
                              planes u8[M,obs0,obs1], win i32[K,T] (pool index of the first of the obs2 consecutive planes of
                              every observation), action / reward / done / length / h0 / c0 / prio as in kind 2. M travels in
                              the header's reserved field.
"""
from __future__ import annotations

import struct
from typing import Dict, Tuple

import numpy as np

MAGIC = 0x31424C5844  # "DXLB1"
HEADER = struct.Struct("<QIIIIIIIIQQ")   # magic, kind, K, T, H, obs0, obs1, obs2, actor_id, env_steps, reserved
HEADER_BYTES = 64
KIND_APEX, KIND_R2D2, KIND_APEX_PLANES, KIND_R2D2_PLANES = 1, 2, 3, 4


def _al(n: int) -> int:
    return (n + 63) // 64 * 64


def apex_block_bytes(K: int, obs_shape: Tuple[int, int, int]) -> int:
    ob = int(np.prod(obs_shape))
    return HEADER_BYTES + 2 * _al(K * ob) + 4 * _al(4 * K)


def apex_planes_block_bytes(K: int, M: int, obs_shape: Tuple[int, int, int]) -> int:
    return HEADER_BYTES + _al(M * obs_shape[0] * obs_shape[1]) + 6 * _al(4 * K)


def r2d2_block_bytes(K: int, T: int, H: int, obs_shape: Tuple[int, int, int]) -> int:
    ob = int(np.prod(obs_shape))
    return HEADER_BYTES + _al(K * T * ob) + 3 * _al(4 * K * T) + _al(4 * K) + 2 * _al(4 * K * H) + _al(4 * K)


def r2d2_planes_block_bytes(K: int, T: int, H: int, M: int, obs_shape: Tuple[int, int, int]) -> int:
    return HEADER_BYTES + _al(M * obs_shape[0] * obs_shape[1]) + 4 * _al(4 * K * T) + _al(4 * K) + 2 * _al(4 * K * H) + _al(4 * K)


def block_bytes(block: Dict[str, np.ndarray], kind: int) -> int:
    """Encoded size of ``block``."""
    if kind == KIND_APEX:
        return apex_block_bytes(block["action"].shape[0], tuple(block["state"].shape[1:]))
    if kind == KIND_APEX_PLANES:
        return apex_planes_block_bytes(block["action"].shape[0], block["planes"].shape[0], tuple(block["planes"].shape[1:]) + (4,))
    if kind == KIND_R2D2_PLANES:
        return r2d2_planes_block_bytes(block["action"].shape[0], block["action"].shape[1], block["h0"].shape[1],
                                       block["planes"].shape[0], tuple(block["planes"].shape[1:]) + (4,))
    return r2d2_block_bytes(block["action"].shape[0], block["action"].shape[1], block["h0"].shape[1],
                            tuple(block["frames"].shape[2:]))


def _fields(kind: int, K: int, T: int, H: int, obs: Tuple[int, int, int], M: int = 0):
    if kind == KIND_APEX:
        return [("state", np.uint8, (K,) + obs), ("next_state", np.uint8, (K,) + obs), ("action", np.int32, (K,)),
                ("reward", np.float32, (K,)), ("done", np.float32, (K,)), ("prio", np.float32, (K,))]
    if kind == KIND_R2D2:
        return [("frames", np.uint8, (K, T) + obs), ("action", np.int32, (K, T)), ("reward", np.float32, (K, T)),
                ("done", np.float32, (K, T)), ("length", np.int32, (K,)), ("h0", np.float32, (K, H)),
                ("c0", np.float32, (K, H)), ("prio", np.float32, (K,))]
    if kind == KIND_APEX_PLANES:
        return [("planes", np.uint8, (T, obs[0], obs[1])), ("win_s", np.int32, (K,)), ("win_s2", np.int32, (K,)),
                ("action", np.int32, (K,)), ("reward", np.float32, (K,)), ("done", np.float32, (K,)), ("prio", np.float32, (K,))]
    if kind == KIND_R2D2_PLANES:
        return [("planes", np.uint8, (M, obs[0], obs[1])), ("win", np.int32, (K, T)), ("action", np.int32, (K, T)),
                ("reward", np.float32, (K, T)), ("done", np.float32, (K, T)), ("length", np.int32, (K,)),
                ("h0", np.float32, (K, H)), ("c0", np.float32, (K, H)), ("prio", np.float32, (K,))]
    raise ValueError("unknown block kind %d" % kind)


def encode_into(buf, block: Dict[str, np.ndarray], kind: int, actor_id: int = 0, env_steps: int = 0) -> int:
    """Write ``block`` into the writable buffer ``buf``; returns the number of bytes used."""
    mv = memoryview(buf).cast("B")
    M = 0
    if kind == KIND_APEX:
        K = int(block["action"].shape[0]); T = 0; H = 0
        obs = tuple(int(x) for x in block["state"].shape[1:])
    elif kind == KIND_APEX_PLANES:
        K = int(block["action"].shape[0]); T = int(block["planes"].shape[0]); H = 0
        obs = (int(block["planes"].shape[1]), int(block["planes"].shape[2]), int(block.get("stack", 4)))
    elif kind == KIND_R2D2_PLANES:
        K, T = (int(x) for x in block["action"].shape)
        H, M = int(block["h0"].shape[1]), int(block["planes"].shape[0])
        obs = (int(block["planes"].shape[1]), int(block["planes"].shape[2]), int(block.get("stack", 4)))
    else:
        K, T = (int(x) for x in block["action"].shape)
        H = int(block["h0"].shape[1])
        obs = tuple(int(x) for x in block["frames"].shape[2:])
    HEADER.pack_into(mv, 0, MAGIC, kind, K, T, H, obs[0], obs[1], obs[2], actor_id, env_steps, M)
    off = HEADER_BYTES
    for name, dt, shape in _fields(kind, K, T, H, obs, M):
        a = np.ascontiguousarray(block[name], dtype=dt).reshape(shape)
        nb = a.nbytes
        np.frombuffer(mv, dtype=np.uint8, count=nb, offset=off)[:] = a.reshape(-1).view(np.uint8)
        off += _al(nb)
    return off


def encode(block: Dict[str, np.ndarray], kind: int, actor_id: int = 0, env_steps: int = 0) -> bytearray:
    n = block_bytes(block, kind)
    out = bytearray(n)
    used = encode_into(out, block, kind, actor_id, env_steps)
    assert used == n
    return out


def layout(buf):
    """``(meta, [(name, dtype, shape, byte offset, nbytes), ...])`` of the block in ``buf`` (header only is read)."""
    mv = memoryview(buf).cast("B")
    magic, kind, K, T, H, o0, o1, o2, actor_id, env_steps, M = HEADER.unpack_from(mv, 0)
    if magic != MAGIC:
        raise ValueError("bad experience block magic")
    meta = {"kind": kind, "K": K, "M": T if kind == KIND_APEX_PLANES else M if kind == KIND_R2D2_PLANES else 0,
            "actor_id": actor_id, "env_steps": env_steps}
    off, out = HEADER_BYTES, []
    for name, dt, shape in _fields(kind, K, T, H, (o0, o1, o2), M):
        nb = int(np.prod(shape)) * np.dtype(dt).itemsize
        out.append((name, dt, shape, off, nb))
        off += _al(nb)
    meta["bytes"] = off
    return meta, out


def decode(buf, copy: bool = False) -> Dict[str, np.ndarray]:
    """Zero-copy views into ``buf`` (set ``copy=True`` when the buffer will be recycled before use)."""
    mv = memoryview(buf).cast("B")
    magic, kind, K, T, H, o0, o1, o2, actor_id, env_steps, M = HEADER.unpack_from(mv, 0)
    if magic != MAGIC:
        raise ValueError("bad experience block magic")
    out = {"kind": kind, "actor_id": actor_id, "env_steps": env_steps}
    off = HEADER_BYTES
    for name, dt, shape in _fields(kind, K, T, H, (o0, o1, o2), M):
        cnt = int(np.prod(shape))
        a = np.frombuffer(mv, dtype=dt, count=cnt, offset=off).reshape(shape)
        out[name] = a.copy() if copy else a
        off += _al(cnt * np.dtype(dt).itemsize)
    return out
