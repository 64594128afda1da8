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
are co-located); ``log`` receives ``(key, value, update_time)``.

The learner's side of the Redis drop-in, upstream of ``consume`` (round 6): ``decode_train_blob`` is the body of
``BackwardGetDataThread.get_train_data`` (backward.py:145-151) after the BRPOP -- one ``encode_backward_data`` blob (the >= 128-sample
pieces ``TrainingProcess.run`` pushes, agent/multiqueue.py:83-105) -> ``(Experience, dict_logger)``; ``BackwardQueue.get(batch_size)``
is backward.py:48-62 -- pop items until >= ``TRAINING_MIN_BATCH`` samples, ``Experience.batch_data`` them, average the logger dicts
(``batch_logger``, backward.py:30-39); ``BackwardTrainer.train_from_queue`` is one turn of the trainer's while-loop (backward.py:184-187)."""
import queue
import time
from collections import defaultdict
from typing import Dict, List, Tuple

import numpy as np
import torch

from ddrl4nav_amd.data.experience import Experience


def batch_logger(p: List[Dict]) -> Dict:
    """Key-wise mean of the env-statistics dicts that arrived with the gathered blobs (backward.py:30-39; keys of the first)."""
    if len(p) == 0:
        return {}
    return {key: float(np.mean([j[key] for j in p])) for key in p[0]}


def decode_train_blob(easy_bytes, batch_bytes) -> Tuple[Experience, Dict]:
    """backward.py:148-149: states arrays + [advs, actions, old_logps, values] + the marshalled logger dict of one blob."""
    list_np_states, list_np_other4, dict_logger = easy_bytes.decode_backward_data(batch_bytes)
    return Experience(list_np_states, *list_np_other4), dict_logger


class BackwardQueue:
    """backward.py:42-65.  The reference's MultiQueue is a multiprocessing queue between its getter and trainer THREADS of one process
    (manager/trainer_manager.py:19-41); a thread-safe ``queue.Queue`` serves the same two threads here."""

    def __init__(self, maxsize=0):
        self.q = queue.Queue(maxsize)
        self._partial = ([], [], 0)      # pieces a timed-out get() had already taken off the queue

    def get(self, batch_size, *args) -> Tuple[Experience, Dict]:
        """Blocks (``*args`` = ``queue.Queue.get``'s block / timeout; a timeout raises ``queue.Empty``) until the gathered pieces hold at
        least ``batch_size`` samples: never fewer, possibly more (whole pieces only), in arrival order.  One deviation from the
        reference, whose gather list is a local of ``get`` (backward.py:50-52) and dies with an exception: pieces already taken when a
        timeout strikes are kept and head the next call's batch -- no sample is ever dropped."""
        list_exp, list_dict, cur_size = self._partial
        self._partial = ([], [], 0)
        try:
            while cur_size < batch_size:
                data, dict_logger = self.q.get(*args)
                assert isinstance(data, Experience)
                assert isinstance(dict_logger, dict)
                cur_size += len(data)
                list_exp.append(data)
                if len(dict_logger):
                    list_dict.append(dict_logger)
        except queue.Empty:
            self._partial = (list_exp, list_dict, cur_size)
            raise
        return Experience.batch_data(list_exp), batch_logger(list_dict)

    def put(self, data: Tuple[Experience, Dict], *args) -> None:
        self.q.put(data, *args)

    def put_blob(self, easy_bytes, batch_bytes, *args) -> None:
        """get_train_data (backward.py:145-151) without the BRPOP / lock SET around it."""
        self.put(decode_train_blob(easy_bytes, batch_bytes), *args)


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
        self.min_batch_size = config_nn.TRAINING_MIN_BATCH
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

    def train_from_queue(self, training_data_queue, *args):
        """One turn of the while-loop (backward.py:184-187): gather >= TRAINING_MIN_BATCH samples from the queue, then ``consume``."""
        train_data, dict_logger = training_data_queue.get(self.min_batch_size, *args)
        return self.consume(train_data, dict_logger)

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
