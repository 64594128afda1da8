"""ddrl4nav_amd.server -- the learner's consumer loop (mirror of USTC_lab/server/backward.py:30-62,145-151,168-217)."""
from ddrl4nav_amd.server.backward import BackwardQueue, BackwardTrainer, batch_logger, decode_train_blob

__all__ = ["BackwardQueue", "BackwardTrainer", "batch_logger", "decode_train_blob"]
