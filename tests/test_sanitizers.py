"""Host code under AddressSanitizer + UBSan (CPU build only; GPU ASan is not available on this pool).

tests/san/builder_fuzz.cpp links csrc/automaton.cpp -- the whole automaton builder -- with the oracle and runs
seeded random pattern sets through it: 1 / 255 / 256 used byte values, 1,024- and 1,025-byte patterns, empty
patterns, duplicate storms, tiny LDS tiles, the compact form forced on.  Any sanitizer report aborts the run."""
import os
import shutil
import subprocess

import pytest

from helpers import ROOT

SAN = os.path.join(ROOT, "tests", "san")
EXE = os.path.join(SAN, "builder_fuzz")
SRCS = [os.path.join(SAN, "builder_fuzz.cpp"), os.path.join(ROOT, "php_aho_corasick_amd", "csrc", "automaton.cpp"),
        os.path.join(ROOT, "php_aho_corasick_amd", "csrc", "automaton.hpp"), os.path.join(ROOT, "oracle", "ac_oracle.c")]


@pytest.fixture(scope="module")
def fuzz():
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(s) for s in SRCS):
        obj = os.path.join(SAN, "ac_oracle.san.o")
        flags = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"]
        subprocess.run(["gcc", *flags, "-c", SRCS[3], "-o", obj], check=True)
        subprocess.run([gxx, "-std=c++17", *flags, SRCS[0], SRCS[1], obj, "-o", EXE], check=True)
    return EXE


@pytest.mark.parametrize("seed0", [1, 100001])
def test_builder_fuzz_under_asan_ubsan(fuzz, seed0):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([fuzz, "600", str(seed0)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.strip().endswith("600 cases, 0 mismatches")
