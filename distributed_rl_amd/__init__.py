"""distributed_rl_amd — MI355X-native distributed RL (Ape-X dueling double-DQN + R2D2).

Layer map:

  csrc/       HIP (gfx950) kernels + C++ host runtime (shm ring, param register, CPU sum-tree, vector env)
  native/     build + ctypes loaders for the two in-tree shared objects
  ops/        Python entry points of every kernel + the pure-PyTorch oracle they are tested against
  models/     DuelingDQN / DuelingLSTMDQN with reference-compatible ``state_dict``
  replay/     host (CPU) and device (HBM-resident) prioritized replay for transitions and sequences
  algo/       n-step / sequence assembly, Ape-X and R2D2 loss+priority steps
  engine/     graph-captured, hand-scheduled GPU learner / actor-inference engines
  parallel/   RCCL (torch.distributed "nccl") flat-buffer all-reduce, param broadcast, sharded-PER coordination
  transport/  experience queue + parameter register (in-proc, POSIX shm, optional redis)
  runtime/    actor loops (single env and vectorised), learner loop, launcher/supervisor
  envs/       synthetic Atari-shaped env, optional gym/ALE adapter, ``MultiFrame<Game>-v0`` registry
  obs/        metrics registry, timers, sinks (stdout / JSONL / optional visdom)
"""

__version__ = "0.1.0"
