"""cn_chess_ai_amd — MI355X-native batched Xiangqi self-play + DQN training hot path.

Host-side mirror of the reference's ChessBoard / ChessAI / DQN interface (Qervas/cn_chess_ai) over the C ABI of
libxqhip.so (include/xq_capi.h).  All compute is hand-written HIP for gfx950; this package is plumbing.
"""
from . import _capi
from ._capi import XqError
from .vecenv import VecEnv, ReplayBuffer, StepResult, START_BOARD, eps_to_u32
from .dqn import DQN, Trainer, TrainerConfig

__all__ = ["VecEnv", "ReplayBuffer", "DQN", "Trainer", "TrainerConfig", "XqError", "StepResult", "START_BOARD",
           "eps_to_u32"]
