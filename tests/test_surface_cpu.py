"""CPU tests of the host-side mirror of the reference's plug-in surface: ConfigNN contract,
Experience batching, weight-blob format, env sharding and the gloo all-reduce path."""
import json
import os
import types

import numpy as np
import pytest
import torch

from ddrl4nav_amd.utils.recipe import flatten, make_weights, param_specs

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_config_nn_contract_matches_reference():
    from ddrl4nav_amd.config import ConfigNN
    want = json.load(open(os.path.join(GOLDEN, "f8_config_nn.json")))
    cfg = ConfigNN({"discrete_action": True, "discrete_actions": list(range(6))})
    for k, v in want.items():
        assert getattr(cfg, k) == v, k
    from ddrl4nav_amd.nn import CategoricalActor
    assert cfg.ACTOR_CLASS is CategoricalActor
    from ddrl4nav_amd.nn import GaussionActor
    cont = ConfigNN({"discrete_action": False, "act_dim": 2})   # config_nn.py:15-17
    assert cont.ACTOR_CLASS is GaussionActor and cont.ACTION_OUTPUT_DIM == 2 and cont.ACTIONS_DIM == 2


def test_base_config_and_game_type():
    from ddrl4nav_amd.config import BaseConfig, game_type
    assert game_type("PongNoFrameskip-v4") == "atari" and game_type("CartPole-v1") == "classical"
    with pytest.raises(NameError):
        game_type("NoSuchGame")
    parse = types.SimpleNamespace(task="t", ip="localhost")
    c = BaseConfig(parse, {"env_type": "gym", "env_name": "PongNoFrameskip-v4", "env_num": 8})
    assert c.TASK_TYPE == "atari" and c.PREDICTING_MIN_BATCH == 4 and c.TIME_MAX == 256 and c.TASK_NAME == "t-127.0.0.1"


def test_experience_batch_data_golden(golden):
    from ddrl4nav_amd.data import Experience
    g = golden("f8_experience")
    exps = [Experience(states=[g["in%d/states" % i]], advs=g["in%d/advs" % i], actions=g["in%d/actions" % i],
                       old_logps=g["in%d/old_logps" % i], values=g["in%d/values" % i], is_clean=g["in%d/is_clean" % i])
            for i in range(5)]
    for tag, clean in (("all", True), ("clean", False)):
        b = Experience.batch_data(exps, clean=clean)
        assert np.array_equal(b.states[0], g[tag + "/states"])
        for k in ("advs", "actions", "old_logps", "values"):
            assert np.array_equal(getattr(b, k), g[tag + "/" + k]), (tag, k)
    assert len(Experience.batch_data(exps)) == 15
    chunks = list(Experience.batch_data_gene(exps * 30))  # 150 records -> 64 + 64 + 22
    assert [len(c.is_clean) > 0 for c in chunks] == [True] * 3
    # to_tensor keeps uint8 frames uint8 and casts the rest
    b = Experience.batch_data(exps)
    b.to_tensor(dtype=torch.float32, device="cpu")
    assert b.states[0].dtype == torch.uint8 and b.advs.dtype == torch.float32 and b.values.shape == (1, 15)


def test_weight_blob_format_golden(golden):
    """nn2redis blob: >I ndim, >I dims, raw fp32, named_parameters() order (reference base.py:38-66)."""
    from ddrl4nav_amd.nn.base import Basenn
    w = make_weights(0)
    names = [n for n, _, _ in param_specs()][8:12]
    enc = Basenn._encode_wb(None, w[names[0]])
    assert enc[:4] == b"\x00\x00\x00\x02" and enc[4:12] == b"\x00\x00\x00\x06\x00\x00\x02\x00"
    blob = b"".join(Basenn._encode_wb(None, w[n]) for n in names)
    assert np.array_equal(np.frombuffer(blob, np.uint8), golden("f7_codec")["blob_heads"])
    me = types.SimpleNamespace(model_dtype=np.float32, model_dtype_bytes=4, device="cpu")
    t, used = Basenn._decode_wb(me, blob)
    assert tuple(t.shape) == (6, 512) and used == 12 + 6 * 512 * 4
    assert np.array_equal(t.numpy(), w[names[0]])


def test_shard_envs():
    from ddrl4nav_amd.dist import shard_envs
    assert [shard_envs(2048, 8, r) for r in (0, 7)] == [(0, 256), (1792, 2048)]
    spans = [shard_envs(10, 4, r) for r in range(4)]
    assert spans == [(0, 3), (3, 6), (6, 8), (8, 10)]


def _ddp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from ddrl4nav_amd.dist import allreduce_flat, global_batch, init_from_env, shard_envs
    from oracle import ddrl_oracle as O
    torch.set_num_threads(1)
    init_from_env(backend="gloo")
    g = np.load(os.path.join(GOLDEN, "f3_loss.npz"))
    B = 16
    lo, hi = shard_envs(B, world, rank)
    net = O.OraclePPO()
    net.load_weights(make_weights(0))
    x = O.frames_to_f32(g["frames"][lo:hi])
    t = lambda k: torch.from_numpy(g[k][lo:hi])
    bg = global_batch(hi - lo)
    # what ddrl_ppo_iter does on each rank: local sums scaled by 1/B_global
    _, al, vl, ent = O.ppo_losses(net, x, t("actions"), t("old_logps"), t("advs"), t("rets"))
    scale = (hi - lo) / bg
    (al * scale).backward()
    (vl * scale).backward()
    flat = torch.cat([p.grad.reshape(-1) for p in net.parameters()] +
                     [torch.stack([al.detach() * scale, vl.detach() * scale, ent.detach() * scale])])
    allreduce_flat(flat)
    q.put((rank, bg, flat.numpy()))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_gloo_world2_gradient_allreduce_equals_full_batch(world):
    """world_size 2 and 8 (BASELINE config 3's rank count) on CPU: shard the batch, scale by 1/B_global, SUM all-reduce the flat
    arena -> the full-batch gradient and loss means (the N>1 path of bench.py / PPO.learn)."""
    import torch.multiprocessing as mp
    from oracle import ddrl_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=300) for _ in procs], key=lambda o: o[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert outs[0][1] == 16 and all(np.array_equal(outs[0][2], o[2]) for o in outs[1:])  # replicas agree bit for bit
    g = np.load(os.path.join(GOLDEN, "f3_loss.npz"))
    torch.set_num_threads(1)
    net = O.OraclePPO()
    net.load_weights(make_weights(0))
    t = lambda k: torch.from_numpy(g[k][:16])
    _, al, vl, ent = O.ppo_losses(net, O.frames_to_f32(g["frames"][:16]), t("actions"), t("old_logps"), t("advs"), t("rets"))
    al.backward()
    vl.backward()
    full = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).numpy()
    got = outs[0][2]
    assert np.abs(got[:-3] - full).max() <= 2e-6 * np.abs(full).max()
    np.testing.assert_allclose(got[-3:], [al.item(), vl.item(), ent.item()], rtol=1e-5, atol=1e-7)


def test_generic_net_parameter_trees_match_reference(golden):
    """Module trees of the operator-composed nets (no GPU needed to build the parameter holders):
    names and order equal the reference's named_parameters() (fixtures f13-f15 store them)."""
    from ddrl4nav_amd.nn import (CategoricalActor, Critic, GaussionActor, MLPPreNet, NavPedPreNet, NavPreNet, NavPreNet1D)
    from torch import nn

    class Tree(nn.Module):  # PPO's registration order (ppo.py:26-28): prenet, actor, critic
        def __init__(self, prenet, actor, critic):
            super().__init__()
            self.prenet, self.actor, self.critic = prenet, actor, critic

    t13 = Tree(None, GaussionActor(action_output_dim=2, pre=NavPreNet1D(3)), Critic(pre=NavPreNet1D(3)))
    assert [k for k, _ in t13.named_parameters()] == list(golden("f13_nav1d_gauss")["names"])
    t14 = Tree(NavPedPreNet(4), CategoricalActor(action_output_dim=5), Critic())
    assert [k for k, _ in t14.named_parameters()] == list(golden("f14_navped_shared")["names"])
    t15 = Tree(None, CategoricalActor(action_output_dim=2, pre=MLPPreNet(4, 512)), Critic(pre=MLPPreNet(4, 512)))
    assert [k for k, _ in t15.named_parameters()] == list(golden("f15_mlp_classical")["names"])
    n = NavPreNet(2)
    assert n.fc0[0].in_features == 256 * 6 * 6 and n.fc1[0].in_features == 521 and n.conv1.in_channels == 2


def test_create_net_rejects_unknown_task_and_needs_gpu():
    import torch
    from ddrl4nav_amd import _lib
    from ddrl4nav_amd.config import BaseConfig, ConfigNN
    from ddrl4nav_amd.runner import create_net
    env = {"env_type": "gym", "env_name": "PongNoFrameskip-v4", "env_num": 8, "discrete_action": True,
           "discrete_actions": [0, 1], "input_dim": 4}
    cfg = BaseConfig(types.SimpleNamespace(task="t", ip="127.0.0.1"), env)
    cfg.TASK_TYPE = "no_such_task"
    with pytest.raises(NotImplementedError):
        create_net({"config": cfg, "config_nn": ConfigNN(env), "config_env": env})
    if not torch.cuda.is_available():
        cfg.TASK_TYPE = "classical"
        with pytest.raises(_lib.DdrlError):   # no CPU fallback for the generic nets either
            create_net({"config": cfg, "config_nn": ConfigNN(env), "config_env": env})


@pytest.mark.parametrize("kind", ["classical", "atari"])
def test_mimic_dataset_reader_and_batch_order_match_reference(golden, tmp_path, kind):
    """Expert data sets written by the reference's MimicExpWriter (fixture F19 carries the files) read back identically, and
    the batches follow torch's RandomSampler exactly as the reference's DataLoader(shuffle=True) does (GAIL.py:49-57)."""
    from ddrl4nav_amd.data.mimic_exp import MimicExpFactory, batches
    g = golden("f19_mimic")
    d = tmp_path / kind
    d.mkdir()
    for f in g[kind + "/files"]:
        (d / str(f)).write_bytes(g["%s/file/%s" % (kind, f)].tobytes())
    ds = MimicExpFactory().mimic_reader(kind, str(d) + "/")
    assert len(ds) == int(g[kind + "/len"])
    xs, ys = zip(*[ds[i] for i in range(len(ds))])
    assert np.array_equal(np.stack(xs), g[kind + "/x"])
    assert np.array_equal(np.stack([np.asarray(y, np.float32) for y in ys]), g[kind + "/y"])
    torch.manual_seed(190)
    loader = batches(ds, 8)
    assert len(loader) == 4
    for epoch in range(2):
        for bi, batch in enumerate(loader):
            assert np.array_equal(batch[0].numpy(), g["%s/epoch%d/batch%d/x" % (kind, epoch, bi)]), (epoch, bi)
            assert np.array_equal(batch[1].numpy(), g["%s/epoch%d/batch%d/y" % (kind, epoch, bi)])
            if bi == 1:
                break


def test_bench_gpus_n_refuses_to_pretend(tmp_path):
    """`python bench.py --gpus 2` without a launcher starts the ranks itself; on a host with fewer GPUs it must fail loudly
    (exit code 2) instead of running one GPU and printing n_gpus: 1."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import torch
    if torch.cuda.device_count() >= 2:
        return  # a multi-GPU host runs the real thing (tests/test_dist_gpu.py covers the one-GPU rehearsal)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 2 and "--gpus 2" in out.stderr and not out.stdout.strip()
    # a launcher that disagrees with --gpus is refused as well
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1"],
                         env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 2 and "WORLD_SIZE=2" in out.stderr


def test_cost_model_reads_the_committed_profiles():
    """tools/cost_model.py: three constants fitted to the committed microbenchmark lines, applied to the committed
    PMC counts -- the newest profile set must parse, the constants must sit where the microbenchmark puts them, and the model's sum over
    the eleven GEMM kernels must stay within 25 % of the measured sum (it is 10 % for r03_v3); bench.py's roofline.mix_model uses the same
    code."""
    import contextlib
    import glob
    import io
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import cost_model
    mix = sorted(glob.glob(os.path.join(root, "profiles", "*_mfma16_mix.txt")))[-1]
    rate, c_valu, c_byte, rows = cost_model.fit(mix)
    assert 1400 < rate < 1800 and 0.04 < c_valu < 0.15 and 0.002 < c_byte < 0.006, (rate, c_valu, c_byte)
    assert rows["+ 3 VALU, 128 B read per MFMA"] < rows["+ 3 VALU per MFMA"] < rows["MFMA + LDS operands only"]
    buf = io.StringIO()
    argv = sys.argv
    sys.argv = ["cost_model.py"]
    try:
        with contextlib.redirect_stdout(buf):
            cost_model.main()
    finally:
        sys.argv = argv
    last = buf.getvalue().strip().splitlines()[-1].split()
    model, measured = float(last[-3]), float(last[-2])
    assert last[0] == "sum" and 0.8 < measured / model < 1.25, buf.getvalue()
