"""The CPU checker under the sanitizers (the reference's Debug build uses -fsanitize=address/undefined,
CMakeLists.txt:134).  oracle/Makefile `san` builds oracle/san_driver.cpp + pbf_oracle.cpp twice:
  ASan + UBSan (OpenMP build)          — heap / bounds / UB over predict, sort, table, diffuse, lambda, delta-p,
                                         finalise, wells, obstacles, XSPH / vorticity and marching cubes;
  TSan (Jacobi mode, 4 threads)        — the same static partition on std::thread (libgomp's barriers are invisible to
                                         TSan); must report no race: the Jacobi restatement is what the GPU implements.
Both print the same checksums: the result does not depend on the threading runtime or thread count."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle")


def run(binary, env=None):
    r = subprocess.run([os.path.join(ORACLE, "_san", binary)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, OMP_NUM_THREADS="4", **(env or {})))
    return r.returncode, r.stdout + r.stderr


def test_oracle_clean_under_asan_ubsan_and_tsan():
    b = subprocess.run(["make", "-C", ORACLE, "san"], capture_output=True, text=True, timeout=900)
    assert b.returncode == 0, b.stdout[-2000:] + b.stderr[-2000:]
    rc_a, out_a = run("pbf_oracle_asan", {"ASAN_OPTIONS": "detect_leaks=1", "UBSAN_OPTIONS": "halt_on_error=1"})
    assert rc_a == 0 and "ERROR: AddressSanitizer" not in out_a and "runtime error" not in out_a, out_a[-3000:]
    rc_t, out_t = run("pbf_oracle_tsan")
    assert rc_t == 0 and "WARNING: ThreadSanitizer" not in out_t, out_t[-3000:]
    sums_a = re.search(r"sanitizer run ok: (.*)", out_a).group(1)
    sums_t = re.search(r"sanitizer run ok: (.*)", out_t).group(1)
    assert sums_a == sums_t  # OpenMP (4 threads) == std::thread partition, bit for bit
