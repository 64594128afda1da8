"""The lane-exchange reductions of csrc/ppo_math.h (round 6: heads_loss's butterflies without ds_bpermute, and the transposing reduction
of a turn's 28 dot products) against the shuffle butterfly they replace, lane by lane and bit by bit, on the GPU: tools/wave_butterfly.hip
is compiled for gfx950 and run in a child process (512 waves x 32 registers of distinct random values)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_lane_exchange_reductions_are_bit_identical_to_the_shuffle_butterfly(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "wave_butterfly")
    build = subprocess.run([hipcc, "-O3", "-w", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "ddrl4nav_amd", "csrc"),
                            os.path.join(ROOT, "tools", "wave_butterfly.hip"), "-o", exe], capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "wave_sum mismatching lanes 0, wave_max 0 of 32768; transposing reduction mismatching values 0 of 16384" in run.stdout
