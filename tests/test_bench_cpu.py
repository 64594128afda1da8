"""bench.py's host-side bookkeeping that needs no GPU: the lookup of the committed PMC summaries (profiles/*_pmc_traffic.json).
Round 4's driver line carried null `roofline.traffic` / `hbm_frac` because the newest file BY NAME was a nav profile."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_every_training_kernel_resolves_in_the_newest_pong_profile():
    srcs = set()
    for k in bench.TRAIN_KERNELS + ("heads_loss", "clip_adam"):
        t = bench.pmc_traffic(k)
        assert t is not None, k
        assert t["hbm_bytes_per_launch_corrected"] > 0 and t["fetch_bytes"] > 0
        assert "nav" not in t["source"]
        srcs.add(t["source"])
    assert len(srcs) == 1, srcs      # one profile serves the whole iteration
    newest = os.path.basename(bench._pmc_files("atari")[-1])
    assert srcs == {newest}
    it = bench.iteration_traffic()
    assert it and it > 28296 * 65536           # never below the algorithmic bytes of an iteration
    for k in bench.TRAIN_KERNELS:
        assert bench.pmc_traffic(k)["executed_over_algorithmic"] >= 0.99, k


def test_families_do_not_cross(tmp_path, monkeypatch):
    """A nav profile that sorts after the Pong one must not shadow it, and shared kernel names (clip_adam) stay per family."""
    prof = tmp_path / "profiles"
    prof.mkdir()
    ent = {"hbm_bytes": 3.0, "fetch_bytes": 1.0, "write_bytes": 1.0, "mfma_busy_frac": 0.5, "clock_ghz": 2.0}
    (prof / "r07_v2_pmc_traffic.json").write_text(json.dumps({"batch": 65536, "kernels": {"conv_dgrad2_both": ent, "clip_adam": dict(ent, hbm_bytes=7.0)}}))
    (prof / "r07_v10_pmc_traffic.json").write_text(json.dumps({"batch": 65536, "kernels": {"clip_adam": dict(ent, hbm_bytes=9.0)}}))
    (prof / "r07_nav9_pmc_traffic.json").write_text(json.dumps({"batch": 4096, "kernels": {"clip_adam": dict(ent, hbm_bytes=11.0)}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "_PMC_DOCS", {})
    assert bench.pmc_traffic("ConvDgrad2")["source"] == "r07_v2_pmc_traffic.json"      # newest file that HAS the kernel
    assert bench.pmc_traffic("clip_adam")["hbm_bytes_per_launch"] == 9.0               # v10 after v2 (integers, not text)
    assert bench._pmc_lookup("clip_adam", "nav")[1]["hbm_bytes"] == 11.0
    assert bench.pmc_traffic("ConvWgrad2") is None


def test_traffic_evidence_is_dated_against_the_running_source(tmp_path, monkeypatch):
    """`roofline.traffic` is a committed PMC summary, not a measurement of the run that prints it -- so the line says
    which build the summary is of and whether the kernel's source file has changed since (bench.evidence_age).  Summaries written from
    round 6 on carry the SHA-1 of every csrc file (tools/pmc_to_profiles.py `sources`): the comparison needs no .git (GPU boxes have
    none).  Older summaries fall back to git ancestry of the last commit that touched the file, or None where that is undecidable."""
    import hashlib
    prof = tmp_path / "profiles"
    prof.mkdir()
    ent = {"hbm_bytes": 5.0, "fetch_bytes": 2.0, "write_bytes": 3.0, "mfma_busy_frac": 0.5, "clock_ghz": 1.8}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sha = lambda f: hashlib.sha1(open(os.path.join(root, "ddrl4nav_amd", "csrc", f), "rb").read()).hexdigest()
    doc = {"batch": 65536, "build": "abc1234 (r09_v1)", "kernels": {"conv_dgrad2_both": ent, "conv_wgrad2_pipe": ent},
           "sources": {"conv2.hip": sha("conv2.hip"), "wgrad2.hip": "0" * 40}}
    (prof / "r09_v1_pmc_traffic.json").write_text(json.dumps(doc))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (tmp_path / "ddrl4nav_amd").mkdir()
    os.symlink(os.path.join(root, "ddrl4nav_amd", "csrc"), tmp_path / "ddrl4nav_amd" / "csrc")
    bench._PMC_DOCS.clear()
    fresh, stale = bench.pmc_traffic("ConvDgrad2"), bench.pmc_traffic("ConvWgrad2")
    assert fresh["traffic_stale"] is False and fresh["traffic_build"] == "abc1234 (r09_v1)" and fresh["traffic_source_file"] == "conv2.hip"
    assert stale["traffic_stale"] is True and stale["traffic_source_file"] == "wgrad2.hip"      # the pipe kernel resolves through its alias
    # a summary without hashes and no .git beside it: undecidable, said so
    del doc["sources"]
    (prof / "r09_v1_pmc_traffic.json").write_text(json.dumps(doc))
    bench._PMC_DOCS.clear()
    assert bench.pmc_traffic("ConvDgrad2")["traffic_stale"] is None
    bench._PMC_DOCS.clear()


def test_committed_summaries_date_themselves():
    """On the real tree every kernel of the headline family answers the staleness question with True or False (this container has .git;
    round-6 summaries carry hashes), never silently."""
    bench._PMC_DOCS.clear()
    for k in bench.TRAIN_KERNELS:
        t = bench.pmc_traffic(k)
        assert t is not None and t["traffic_build"] and t["traffic_source_file"], k
        assert t["traffic_stale"] in (True, False) or not os.path.isdir(os.path.join(bench.ROOT, ".git")), k


def test_preflight_static_needs_no_gpu_and_names_the_rccl_in_use():
    """the pre-collective half of `bench.py --preflight` (also written to stderr by every N > 1 bench run before its
    first collective): visible devices, peer-access matrix, the RCCL torch links and the librccl csrc/comm.cpp resolved."""
    st = bench.preflight_static()
    assert st["world"] == 1 and st["rank"] == 0 and "device_count" in st and isinstance(st["env"], dict)
    r = st["ddrl_comm_rccl"]
    assert r["status"] == 0 and r["path"].endswith(".so") or ".so." in r["path"], r
    assert r["version_code"] > 20000, r                      # ncclGetVersion: major * 10000 + minor * 100 + patch
    assert len(st.get("peer_access", [])) == st["device_count"]


def test_power_evidence_resolves_from_the_committed_hwmon_summary():
    """Round 6: the bench line names what binds the update -- the socket's power cap -- from the newest committed hwmon summary."""
    p = bench.power_evidence()
    assert p is not None and p["source"].endswith("_hwmon_summary.json")
    assert p["cap_W"] >= 1000 and 0.9 * p["cap_W"] <= p["update_power_W"] <= p["cap_W"] and 1.0 < p["update_sclk_GHz"] < p["idle_sclk_GHz"]
