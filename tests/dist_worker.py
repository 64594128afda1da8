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


def main():
    outdir, mode, bounds = sys.argv[1], sys.argv[2], [int(t) for t in sys.argv[3].split(",")]
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


if __name__ == "__main__":
    main()
