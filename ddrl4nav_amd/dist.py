"""Data-parallel plumbing: one process per GPU, envs sharded, ONE all-reduce of the flat fp32
gradient arena (+ loss tail) per PPO iteration -- RCCL over xGMI on the GPU box (backend "nccl"),
gloo in the CPU tests.  The reference has no multi-GPU path (`# TODO support mutil GPU CARD`,
USTC_lab/server/backward.py:167); the partitioning is SURVEY.md section 8e.

Because every rank scales its loss terms and gradients by 1/B_global inside the kernels
(ddrl_ppo_iter), a plain SUM makes every rank hold the full-batch mean gradient; clip + Adam then
run identically everywhere and the replicas stay in step without a parameter broadcast."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).
    Returns (rank, world, local_rank).  No-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # DDRL_DIST_BACKEND=gloo lets several ranks share one GPU (single-GPU test boxes)
            backend = os.environ.get("DDRL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def shard_envs(n_envs_total, world, rank):
    """Contiguous env shard [lo, hi) of rank `rank` (GPU r owns envs [r*256, (r+1)*256) in the
    2048-env config).  Uneven totals give the first ranks one extra env."""
    base, extra = divmod(int(n_envs_total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def global_batch(local_batch, group=None):
    """Sum of the per-rank batch sizes (B_global of ddrl_ppo_iter)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return int(local_batch)
    t = torch.tensor([int(local_batch)], dtype=torch.int64,
                     device="cuda" if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t.item())


def allreduce_flat(flat, group=None):
    """In-place SUM all-reduce of the flat gradient arena (13,487,388 B + 32 B tail per call)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if flat.is_cuda and dist.get_backend(group) == "gloo":
            # several ranks sharing one GPU (single-GPU test boxes, DDRL_DIST_BACKEND=gloo): reduce on the host
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            flat.copy_(host)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def broadcast_params(flat, src=0, group=None):
    """Make the replicas bit-identical at start-up (weights come from rank 0)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if flat.is_cuda and dist.get_backend(group) == "gloo":
            host = flat.cpu()
            dist.broadcast(host, src=src, group=group)
            flat.copy_(host)
        else:
            dist.broadcast(flat, src=src, group=group)
    return flat


class RcclComm:
    """The C-ABI communicator (include/ddrl.h: ddrl_comm_*): RCCL without torch.distributed on the data path.  A
    non-Python host exchanges the 128-byte id out of band; here rank 0's id travels through the already initialised
    torch.distributed group (any backend) or is passed in.  `DDRL_ALLREDUCE=rccl` makes HotPath.allreduce_grads use it."""

    def __init__(self, rank=0, world=1, unique_id=None, group=None):
        from ctypes import byref, c_void_p, create_string_buffer
        from . import _lib
        self.lib, self.rank, self.world = _lib.load(), int(rank), int(world)
        if unique_id is None:
            buf = create_string_buffer(128)
            if self.rank == 0:
                _lib.check(self.lib.ddrl_comm_unique_id(buf))
            ids = [bytes(buf.raw)]
            if self.world > 1:
                dist.broadcast_object_list(ids, src=0, group=group)
            unique_id = ids[0]
        assert len(unique_id) == 128
        self.h = c_void_p()
        _lib.check(self.lib.ddrl_comm_create(create_string_buffer(unique_id, 128), self.rank, self.world, byref(self.h)))

    def allreduce(self, flat):
        from ctypes import c_void_p
        from . import _lib
        assert flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()
        _lib.check(self.lib.ddrl_allreduce_f32(self.h, c_void_p(flat.data_ptr()), flat.numel(),
                                               c_void_p(torch.cuda.current_stream().cuda_stream)))
        return flat

    def broadcast(self, flat, root=0):
        from ctypes import c_void_p
        from . import _lib
        _lib.check(self.lib.ddrl_broadcast_f32(self.h, c_void_p(flat.data_ptr()), flat.numel(), int(root),
                                               c_void_p(torch.cuda.current_stream().cuda_stream)))
        return flat

    def close(self):
        if getattr(self, "h", None):
            self.lib.ddrl_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
