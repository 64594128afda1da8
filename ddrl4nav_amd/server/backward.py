"""BackwardTrainer: the "gather B, TRAINING_ITER_TIME x iterate, publish" loop around ``net.learn``.

Mirror of ``BackwardTrainThread.run`` (USTC_lab/server/backward.py:168-217) without its thread / Redis-queue plumbing
(out of scope): the caller hands over batches (``consume``); everything the reference does per
batch and per yielded iteration is kept, with the same config constants:

  * ``LOAD_CHECKPOINT`` / ``LOAD_CHECKPOINT_PATH`` / ``LOAD_EPISODE``: state_dict loaded before the first batch, and every
    ``update_time`` offset by ``LOAD_EPISODE`` (backward.py:131-135,175-176,190);
  * the weights are published once at start and then whenever ``last and update_time % MODEL_TO_REDIS_FREQUENCY == 0``
    (backward.py:179,196-197; ``nn2redis`` = one blob SET + one INCR of the update tag, nn/base.py:60-66);
  * losses are averaged between log points, ``update_time % LOG_LOSS_FREQUENCY == 0`` (backward.py:202-206);
  * ``torch.save(net.state_dict(), SAVE_MODEL_PATH + "_<update_time>.pt")`` whenever
    ``last and SAVE_MODELS and update_time % SAVE_FREQUENCY == 0`` (backward.py:207-209);
  * the train lock key is cleared after every batch (backward.py:212-213);
  * ``TEST`` skips training (backward.py:188).
``pipe`` is anything with ``set / incr / execute`` (a redis pipeline, or an in-process stand-in when Forward and Backward
are co-located); ``log`` receives ``(key, value, update_time)``."""
import time
from collections import defaultdict

import torch


class BackwardTrainer:
    def __init__(self, net, config, config_nn, pipe=None, log=None):
        self.net, self.config, self.config_nn = net, config, config_nn
        self.pipe, self.log = pipe, log
        self.update_tag = config.TASK_NAME + config.UPDATE_TAG_KEY
        self.train_lock_key = config.TASK_NAME + config.TRAIN_LOCK_KEY
        self.log_loss_freq = config.LOG_LOSS_FREQUENCY
        self.save_model, self.save_freq = config.SAVE_MODELS, config.SAVE_FREQUENCY
        self.save_model_path = config.SAVE_MODEL_PATH
        self.model2redis_freq = config_nn.MODEL_TO_REDIS_FREQUENCY
        self.load_checkpoint_path, self.load_checkpoint_start = None, 0
        if config.LOAD_CHECKPOINT:
            self.load_checkpoint_path = config.LOAD_CHECKPOINT_PATH
            self.load_checkpoint_start += config.LOAD_EPISODE
        self.test = getattr(config, "TEST", False)
        self.device = net.device
        self.tensortype = config_nn.MODULE_TENSOR_DTYPE
        self.data_len = 0
        self.published = self.saved = 0
        self._loss = defaultdict(list)
        self._started = False

    def _publish(self):
        if self.pipe is not None:
            self.net.nn2redis(self.pipe, self.update_tag)
        self.published += 1

    def start(self):
        if self.load_checkpoint_path:
            self.net.load_state_dict(torch.load(self.load_checkpoint_path, map_location=self.device))
        self._publish()
        self._started = True

    def consume(self, train_data, dict_logger=None):
        """One pass of the while-loop body; returns the (shifted) update_time of the last iteration."""
        if not self._started:
            self.start()
        t0 = time.time()
        train_data.to_tensor(dtype=self.tensortype, device=self.device)
        self.data_len += len(train_data)
        if dict_logger and self.log:
            for k, v in dict_logger.items():
                self.log(k, {"mean": v}, self.data_len)
        update_time = self.load_checkpoint_start
        if not self.test:
            for loss_items, update_time, last in self.net.learn(train_data):
                update_time += self.load_checkpoint_start
                for k, v in loss_items.items():
                    self._loss[k].append(v)
                if last and update_time % self.model2redis_freq == 0:
                    self._publish()
                if update_time % self.log_loss_freq == 0:
                    for k, vals in self._loss.items():
                        if vals:
                            if self.log:
                                self.log(k, sum(vals) / len(vals), update_time)
                            vals.clear()
                if last and self.save_model and update_time % self.save_freq == 0:
                    torch.save(self.net.state_dict(), self.save_model_path + "_" + str(update_time) + ".pt")
                    self.saved += 1
        if self.pipe is not None:
            self.pipe.set(self.train_lock_key, 0)
            self.pipe.execute()
        self.last_batch_seconds = time.time() - t0
        return update_time
