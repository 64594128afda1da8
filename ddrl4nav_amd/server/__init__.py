"""ddrl4nav_amd.server -- the learner's consumer loop (mirror of USTC_lab/server/backward.py:168-217)."""
from ddrl4nav_amd.server.backward import BackwardTrainer

__all__ = ["BackwardTrainer"]
