"""Host-side C++ behind the C ABI under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only).  `make asan` builds
csrc/easybytes.cpp (the parser of untrusted wire bytes, USTC_lab/data/easybytes.py:18-75) and csrc/comm.cpp with
g++ -fsanitize=address,undefined -fno-sanitize-recover=all and links them with the fuzz driver tests/c/eb_fuzz.cpp (round trips,
truncations, byte mutations, hostile 32-bit dims / 64-bit lengths).  Everything runs in child processes: a finding is a failed test, not a
dead pytest."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ddrl4nav_amd", "csrc")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++ with libasan / libubsan")


def _build(dirname, src_dir=CSRC):
    r = subprocess.run(["make", "-C", src_dir, "asan", "ASAN_DIR=%s" % dirname], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return os.path.join(dirname if os.path.isabs(dirname) else os.path.join(src_dir, dirname), "eb_fuzz")


def test_codec_and_comm_host_code_clean_under_asan_ubsan():
    exe = _build("_asan")
    for seed in (1, 20261004, 977):
        r = subprocess.run([exe, str(seed), "30000"], capture_output=True, text=True, timeout=600, env=ENV)
        assert r.returncode == 0, (seed, r.stdout[-500:], r.stderr[-4000:])
        assert "eb_fuzz OK: 30000 iterations" in r.stdout


def test_a_reintroduced_overflow_fails_the_fuzz(tmp_path):
    """The target has teeth: with round 5's order restored in ddrl_eb_scan (multiply the 32-bit dims first, compare afterwards) the same
    driver stops on UBSan's `signed integer overflow`."""
    src = open(os.path.join(CSRC, "easybytes.cpp")).read()
    guard = "      if (a.dims[d] != 0 && prod > 0xFFFFFFFFll / a.dims[d]) return DDRL_ERR_INVALID_ARG;\n      prod *= a.dims[d];\n"
    assert src.count(guard) == 1, "ddrl_eb_scan's dim-product guard moved: update this test"
    old = "      prod *= a.dims[d];\n      if (prod > 0xFFFFFFFFll) return DDRL_ERR_INVALID_ARG;\n"
    work = tmp_path / "ddrl4nav_amd" / "csrc"
    work.mkdir(parents=True)
    (work / "easybytes.cpp").write_text(src.replace(guard, old))
    for f in ("comm.cpp", "Makefile"):
        shutil.copy(os.path.join(CSRC, f), work / f)
    for d in ("include", os.path.join("tests", "c")):
        shutil.copytree(os.path.join(ROOT, d), tmp_path / d)
    exe = _build(str(tmp_path / "out"), src_dir=str(work))
    r = subprocess.run([exe, "1", "30000"], capture_output=True, text=True, timeout=600, env=ENV)
    assert r.returncode != 0, "the sanitizer build did not notice the overflow"
    assert "signed integer overflow" in r.stderr and "easybytes.cpp" in r.stderr, r.stderr[-2000:]
