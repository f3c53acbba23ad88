"""AddressSanitizer + UBSan over the index-heavy HOST code, on the CPU box (the reference's
Debug flags, /root/reference/CMakeLists.txt:8-12).  tools/asan builds, with g++/gcc -fsanitize=address,undefined,
  * the library's host-only planners (sigma_amd/csrc/sgm_plan_host.hpp: the statements libsigma_hip.so runs) behind the same C ABI,
  * the oracle (oracle/sigma_oracle.c),
  * the stand-in transport (tests/mock_rccl/mock_rccl.cpp, host memory for device buffers) with a driver of forked ranks,
and this file runs tests/test_dist_cpu.py, the planner tests of tests/test_cabi_cpu.py and tests/test_oracle_golden.py against
them (LD_PRELOAD=libasan.so; tools/asan/pyhook routes the symbols in the pytest process and in the gloo ranks it spawns).  A
report of either sanitizer fails the test.  Never on a GPU: no GPU sanitizer run exists on this pool."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASAN = os.path.join(ROOT, "tools", "asan")
REPORT = re.compile(r"ERROR: AddressSanitizer|runtime error:|ERROR: LeakSanitizer|SUMMARY: (Address|UndefinedBehavior)Sanitizer")


def _libasan():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.fixture(scope="module")
def asan_build():
    if _libasan() is None:
        pytest.skip("gcc has no shared AddressSanitizer runtime here")
    subprocess.check_call(["make", "-C", ASAN], stdout=subprocess.DEVNULL)
    return ASAN


def _san_env():
    env = dict(os.environ)
    env["ASAN_OPTIONS"] = "detect_leaks=0:halt_on_error=1:abort_on_error=0:alloc_dealloc_mismatch=0:new_delete_type_mismatch=0:detect_odr_violation=0"
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    return env


@pytest.mark.parametrize("ranks,rounds", [(1, 2), (3, 4), (8, 6), (16, 2)])
def test_stand_in_transport_under_asan_ubsan(asan_build, ranks, rounds):
    """Rings (messages beyond the ring length, several per peer, crossing pairs), barrier, all-reduce, all-gather and the mixed
    group CG posts, as `ranks` forked processes on host memory.  (First run of this: UBSan found the reduction reading doubles
    from a slot at offset 76 of the shared segment; the slots are 64-byte aligned since.)"""
    env = _san_env()
    env["ASAN_OPTIONS"] = env["ASAN_OPTIONS"].replace("detect_leaks=0", "detect_leaks=1")
    p = subprocess.run([os.path.join(asan_build, "mock_rccl_asan"), str(ranks), str(rounds)], env=env, capture_output=True, text=True,
                       timeout=600)
    out = p.stdout + p.stderr
    assert p.returncode == 0 and not REPORT.search(out), out[-4000:]
    assert f"{ranks} ranks x {rounds} rounds, 0 failed" in out


def test_cpu_suites_against_the_sanitizer_builds(asan_build, tmp_path):
    """tests/test_dist_cpu.py (world_size 2 / 3 gloo over the product's planners), the planner and slice-schedule tests of
    tests/test_cabi_cpu.py and tests/test_oracle_golden.py (every reference fixture through the oracle), all against the
    sanitizer builds.  The log is kept under profiles/ by tools/asan/run.sh."""
    env = _san_env()
    env.update({"LD_PRELOAD": _libasan(), "SGM_ASAN_HOOK": "1",
                "PYTHONPATH": os.pathsep.join([os.path.join(ASAN, "pyhook"), ROOT, env.get("PYTHONPATH", "")])})
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_dist_cpu.py"), os.path.join(ROOT, "tests", "test_oracle_golden.py"),
           os.path.join(ROOT, "tests", "test_cabi_cpu.py"), "-k",
           "not test_cabi_cpu or halo_plan or slice_schedule"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    out = p.stdout + p.stderr
    (tmp_path / "asan_suites.log").write_text(out)
    assert "[asan hook]" in out, out[-3000:]
    assert not REPORT.search(out), out[-6000:]
    assert p.returncode == 0, out[-6000:]
    assert re.search(r"\d+ passed", out)
