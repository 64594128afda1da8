"""One rank of the multi-rank learner tests (tests/test_dist_gpu.py): a fresh process that initialises
torch.distributed from the environment, builds the net through the drop-in surface (create_net), runs
PPO.learn on ITS shard of a fixture batch and writes what it saw.  Several ranks may share one GPU
(DDRL_DIST_BACKEND=gloo, dist.py); on a multi-GPU node the same script runs one rank per GPU over RCCL.

usage: python tests/dist_worker.py <outdir> <mode> <bounds>      e.g.  ... default 0,40,64
env:   RANK WORLD_SIZE MASTER_ADDR MASTER_PORT [LOCAL_RANK] [DDRL_DIST_BACKEND]
"""
import hashlib
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def nccl_one_rank(outdir):
    """The path N > 1 RCCL ranks take by default -- torch.distributed's nccl backend, gradient all-reduce in layer buckets on a second
    stream behind the backward's events (engine.py _allreduce_overlapped) -- on the one GPU of a test box: a ONE-rank nccl group (the
    reductions are identities, the stream / event choreography is the real one).  Writes the parameters after three iterations with
    and without it."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from ddrl4nav_amd.engine import HotPath
    from ddrl4nav_amd.utils.recipe import flatten, make_weights
    import parity_util as P
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % os.environ["MASTER_PORT"], rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    frames, actions, old_logps, advs, rets = P.mode_batch("default")
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    args = (d(frames), d(actions), d(old_logps), d(advs), d(rets))
    out = {}
    for overlap in (False, True):
        h = HotPath(max_batch=64)
        h.set_params(flatten(make_weights(0)))
        assert h.comm is None and not h._overlap  # world 1: nothing is enabled by itself
        if overlap:
            h.enable_overlap()
        for _ in range(3):
            h.ppo_iter(*args)
            h.allreduce_grads()
            h.clip_adam_step()
        out["params_%d" % overlap] = h.params.cpu().numpy().copy()
        out["loss_%d" % overlap] = np.float64(h.stats()["PpoTotalLoss"])
        h.close()
    np.savez(os.path.join(outdir, "rank0.npz"), **out)
    dist.destroy_process_group()


def bucket_ranks(outdir, bounds):
    """N gloo ranks sharing the GPU: the layer-bucket path (bucket events, communication stream, ranges) with REAL partners --
    every bucket's ranges are reduced host-staged on the communication stream -- against the flat reduction of the same
    gradients: bit-identical arenas, ranges that tile the arena exactly once, and the stale-event rule (a second reduction
    without a new ppo_iter must still see everything the compute stream wrote)."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from ddrl4nav_amd.dist import allreduce_flat, init_from_env
    from ddrl4nav_amd.engine import HotPath
    from ddrl4nav_amd.utils.recipe import flatten, make_weights
    import parity_util as P
    rank, world, _ = init_from_env()
    frames, actions, old_logps, advs, rets = P.mode_batch("default")
    lo, hi = bounds[rank], bounds[rank + 1]
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a[lo:hi])).cuda()
    args = (d(frames), d(actions), d(old_logps), d(advs), d(rets))
    h = HotPath(max_batch=64)
    h.set_params(flatten(make_weights(0)))
    assert not h._overlap          # off by default (DDRL_ALLREDUCE has no "overlap" option)
    h.ppo_iter(*args, b_global=64)
    local = h.grads.clone()
    flat = allreduce_flat(local.clone())
    h.enable_overlap()
    covered = np.zeros(h.grads.numel(), np.int32)
    for ranges in h.grad_buckets():
        for off, cnt in ranges:
            covered[off:off + cnt] += 1
    assert (covered == 1).all()    # the buckets tile the arena + loss tail exactly once
    h.ppo_iter(*args, b_global=64)  # records the bucket events
    assert torch.equal(h.grads, local)
    h.allreduce_grads()
    torch.cuda.synchronize()
    bucketed = h.grads.clone()
    # stale events: new gradients written by "another producer" on the compute stream, no ppo_iter in between
    h.grads.copy_(local * 2.0)
    h.allreduce_grads()
    torch.cuda.synchronize()
    stale = h.grads.clone()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), flat=flat.cpu().numpy(), bucketed=bucketed.cpu().numpy(),
             stale=stale.cpu().numpy(), n_buckets=len(h.grad_buckets()))
    h.close()
    dist.barrier()
    dist.destroy_process_group()


def rccl_duplicate(outdir):
    """DDRL_ALLREDUCE=rccl with two ranks on ONE device must be refused with an error, not reach ncclCommInitRank (which would
    hang or abort on duplicate devices)."""
    from ddrl4nav_amd import _lib
    from ddrl4nav_amd.dist import init_from_env
    from ddrl4nav_amd.engine import HotPath
    rank, world, _ = init_from_env()
    try:
        HotPath(max_batch=8)
    except _lib.DdrlError as e:
        print("REFUSED:", e)
        sys.exit(3)
    sys.exit(0)


def main():
    outdir, mode, bounds = sys.argv[1], sys.argv[2], [int(t) for t in sys.argv[3].split(",")]
    if mode == "nccl1":
        return nccl_one_rank(outdir)
    if mode == "buckets":
        return bucket_ranks(outdir, bounds)
    if mode == "rccl_dup":
        return rccl_duplicate(outdir)
    import numpy as np
    import torch
    import torch.distributed as dist

    from ddrl4nav_amd.config import BaseConfig, ConfigNN
    from ddrl4nav_amd.data import Experience
    from ddrl4nav_amd.dist import init_from_env
    from ddrl4nav_amd.runner import create_net
    from ddrl4nav_amd.utils.recipe import make_weights
    import parity_util as P

    rank, world, _ = init_from_env()
    assert world == len(bounds) - 1
    if mode.startswith("gail:"):
        return gail_rank(outdir, mode[5:], bounds, rank, world)
    _, _, _, shared, smooth = P.MODES[mode]
    env = {"env_type": "gym", "env_name": "PongNoFrameskip-v4", "env_num": 8, "int_frame_stack": 4, "discrete_action": True,
           "discrete_actions": list(range(6)), "agent_num_per_env": 1, "batch_num_per_env": 8}
    cfg_nn = ConfigNN(env)
    cfg_nn.SHARE_CNN_NET, cfg_nn.SMOOTH_L1_LOSS = shared, smooth
    configs = {"config": BaseConfig(types.SimpleNamespace(task="dist", ip="127.0.0.1"), env), "config_nn": cfg_nn, "config_env": env}
    net = create_net(configs, max_batch=64)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in make_weights(0, shared=shared).items()})
    frames, actions, old_logps, advs, rets = P.mode_batch(mode)
    lo, hi = bounds[rank], bounds[rank + 1]
    exp = Experience(states=[frames[lo:hi]], advs=advs[lo:hi], actions=actions[lo:hi], old_logps=old_logps[lo:hi],
                     values=rets[lo:hi].reshape(1, -1))
    exp.to_tensor(dtype=torch.float32, device="cuda")
    losses, digests, keep = [], {}, {}
    for it, (ld, update_time, last) in enumerate(net.learn(exp), 1):
        assert update_time == it and last is True
        losses.append([ld["PpoTotalLoss"], ld["ActorLoss"], ld["VLoss"], ld["EntLoss"]])
        if it in (1, 10):
            flat = net.hot_path.params.cpu().numpy()
            digests[it] = hashlib.sha256(flat.tobytes()).hexdigest()
            if rank == 0:
                keep["params_it%d" % it] = flat
    st = net.hot_path.stats()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), losses=np.asarray(losses, np.float64), digest1=digests[1],
             digest10=digests[10], gradnorm=st["GradNorm"], local_batch=hi - lo, **keep)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def gail_rank(outdir, name, bounds, rank, world):
    """One rank of a data-parallel GAIL.learn (BASELINE config 5): its shard of the policy batch AND of the expert batch; the
    discriminator's two WGAN means run over the union of the shards (nn/gail.py)."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from ddrl4nav_amd.config import BaseConfig, ConfigNN
    from ddrl4nav_amd.data import Experience
    from ddrl4nav_amd.runner import create_net
    from ddrl4nav_amd.utils.recipe import hash_weights
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    task = None
    if name == "f16_gail_classical":
        env = {"env_type": "gym", "env_name": "CartPole-v1", "env_num": 8, "discrete_action": True, "discrete_actions": [0, 1], "input_dim": 4}
        states, seed = g["states"], 16
    elif name == "f22_gail_navped":   # config 5's own encoder: robot_nav with a pedestrian map -> shared NavPedPreNet(1 + 3) (runner/utils.py:88-102)
        env = {"env_type": "robot_nav", "env_name": "robot_nav", "env_num": 8, "discrete_action": True, "discrete_actions": list(range(5)),
               "image_batch": 1, "ped_sim": {"total": 3}}
        states, seed, task = [g["state0"], g["state1"], g["state2"]], 22, "robot_nav"
    else:
        env = {"env_type": "gym", "env_name": "PongNoFrameskip-v4", "env_num": 8, "int_frame_stack": 4, "discrete_action": True,
               "discrete_actions": list(range(6))}
        states, seed = np.load(os.path.join(HERE, "golden", "f3_loss.npz"))["frames"], 17
    cfg = BaseConfig(types.SimpleNamespace(task="gail", ip="127.0.0.1"), env)
    if task:
        cfg.TASK_TYPE = task
    cfg_nn = ConfigNN(env)
    cfg_nn.NETWORK_TYPE, cfg_nn.SHARE_CNN_NET = "gail", True
    hidden = int(g["d_mlp_hidden"])
    cfg.GAN_D_MLP_LIST = [(512 + cfg.ACTIONS_DIM, hidden, "relu"), (hidden, 1, None)]
    # bounds are given in 64ths of a batch: the policy batch (B samples) and the expert batch (E samples) are cut at the same fractions
    B, E = len(g["actions"]), len(g["expert_actions"])
    lo, hi = bounds[rank] * B // bounds[-1], bounds[rank + 1] * B // bounds[-1]
    elo, ehi = bounds[rank] * E // bounds[-1], bounds[rank + 1] * E // bounds[-1]
    if isinstance(states, list):     # F22 stores its expert batch (a LIST of state components)
        expert = [([g["expert_state%d" % i][elo:ehi] for i in range(3)], g["expert_actions"][elo:ehi])]
        shard = [s[lo:hi] for s in states]
    else:
        ex_states = states[g["expert_index"]][::-1].copy()
        expert = [(ex_states[elo:ehi][None], g["expert_actions"][elo:ehi])]
        shard = [states[lo:hi]]
    net = create_net({"config": cfg, "config_nn": cfg_nn, "config_env": env}, max_batch=256, expert_data=expert)
    w = hash_weights([(k, tuple(p.shape)) for k, p in net.named_parameters()], seed)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()}, strict=False)
    exp = Experience(states=shard, advs=g["advs"][lo:hi], actions=g["actions"][lo:hi], old_logps=g["old_logps"][lo:hi],
                     values=g["rets"][:, lo:hi])
    d_loss, losses, keep, digests = [], [], {}, {}
    for item, ut, last in net.learn(exp):
        params = {k: p.detach().cpu().numpy().copy() for k, p in net.named_parameters()}
        dig = hashlib.sha256(b"".join(params[k].tobytes() for k in sorted(params))).hexdigest()
        if not last:
            d_loss.append(item["Gail[D]Loss"])
            digests["D1"] = dig
            if rank == 0:
                keep.update({"D1/" + k: v for k, v in params.items()})
        else:
            losses.append([item[k] for k in ("PpoTotalLoss", "ActorLoss", "VLoss", "EntLoss")])
            if ut == 10:
                digests["it10"] = dig
                if rank == 0:
                    keep.update({"it10/" + k: v for k, v in params.items()})
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), d_loss=np.asarray(d_loss), losses=np.asarray(losses, np.float64),
             digest_d1=digests["D1"], digest_it10=digests["it10"], local_batch=hi - lo, **keep)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
