"""Model family: DuelingDQN (Ape-X) and DuelingLSTMDQN (R2D2).

These ``nn.Module`` classes are the *parameter containers* and the fp32 oracle/CPU path; their
``state_dict`` keys and shapes are bit-compatible with the reference. The MI355X
engines (``distributed_rl_amd.engine``) read parameters through :mod:`.packing`, which converts
to the kernels' NHWC / bf16 layouts and back.
"""
from .dueling_dqn import DuelingDQN
from .dueling_lstm_dqn import DuelingLSTMDQN
from .packing import ParamPack, apex_layout, r2d2_layout

__all__ = ["DuelingDQN", "DuelingLSTMDQN", "ParamPack", "apex_layout", "r2d2_layout", "build_model"]


def build_model(algorithm: str, n_action: int, batch_size: int = 1, nstep: int = 5, burn_in: int = 40):
    if algorithm == "ape_x":
        return DuelingDQN(n_action)
    if algorithm == "r2d2":
        return DuelingLSTMDQN(n_action, batch_size, n_burn_in=burn_in, nstep_return=nstep)
    raise ValueError("Unknown the algorithm: %s." % algorithm)
