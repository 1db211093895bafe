"""AddressSanitizer + UBSan over the host-side code that reads untrusted bytes (CPU build only -- GPU sanitizers are not available on
the pool): marker parser, host entropy decoder, and the host emulations of the GPU entropy stage, i.e. the kernels' own decode
routines (huffman_gpu_core.h, progressive_gpu_core.h), fed with the goldens and thousands of mutated copies.  The harness
(tests/sanitizers/host_fuzz.cpp) also cross-checks the two decoders on every stream both accept.  A longer campaign (300,000 mutated
streams, 136,623 of them parsed, 123,471 through the emulation) ran clean on the final code of round 2."""
import glob
import os
import shutil
import subprocess

import pytest

from conftest import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "nvimagecodec_amd", "csrc")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_parser_and_entropy_decoders_are_clean_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "host_fuzz")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           "-I" + os.path.join(ROOT, "include"), "-I" + SRC, os.path.join(ROOT, "tests", "sanitizers", "host_fuzz.cpp")]
    cmd += [os.path.join(SRC, f) for f in ("jpeg_syntax.cpp", "entropy_decode.cpp", "gpu_huffman_host.cpp", "progressive_gpu_host.cpp")]
    build = subprocess.run(cmd + ["-o", exe], capture_output=True, text=True, timeout=600)
    if build.returncode != 0 and "asan" in build.stderr.lower() and "cannot find" in build.stderr.lower():
        pytest.skip("no sanitizer runtime in this toolchain")
    assert build.returncode == 0, build.stderr[-2000:]
    seeds = [p for p in sorted(glob.glob(os.path.join(GOLDEN, "decode", "*.jpg"))) if os.path.getsize(p) < 40000]
    seeds += sorted(glob.glob(os.path.join(GOLDEN, "cmyk", "*.jpg")))[:8]
    assert len(seeds) > 100
    run = subprocess.run([exe, "6000", "20261004"] + seeds, capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert run.returncode == 0, (run.stdout + run.stderr)[-4000:]
    assert "0 coefficient mismatches" in run.stdout
    parsed = int(run.stdout.split("host_fuzz:")[1].split("parsed")[0])
    assert parsed > 1500  # the mutations are not all rejected by the first check


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_fork_join_pool_takes_concurrent_callers_under_tsan(tmp_path):
    """resolve() on the plugin's completion thread and plan() on the caller's thread use ONE ForkJoinPool at the same time.
    tests/sanitizers/pool_race.cpp: three threads, 400 parallel_for calls each, every index exactly once, exceptions to their own caller."""
    exe = str(tmp_path / "pool_race")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-I" + SRC, os.path.join(ROOT, "tests", "sanitizers", "pool_race.cpp"),
                            "-o", exe, "-lpthread"], capture_output=True, text=True, timeout=600)
    if build.returncode != 0 and "tsan" in build.stderr.lower() and "cannot find" in build.stderr.lower():
        pytest.skip("no ThreadSanitizer runtime in this toolchain")
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    if "FATAL: ThreadSanitizer" in run.stderr and "unexpected memory mapping" in run.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow memory in this container")
    assert run.returncode == 0, (run.stdout + run.stderr)[-4000:]
    assert "0 violations" in run.stdout
