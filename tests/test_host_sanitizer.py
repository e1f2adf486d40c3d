"""Host sanitizers over the C-ABI shim (CPU-only, no GPU needed).

`make asan` builds mcmc_amd/csrc with AddressSanitizer + UndefinedBehaviorSanitizer on the host code and links
tests/abi_check.cpp against it: that driver calls every extern "C" entry point of include/mcmc_hip.h with a NULL context, with a
device-less dry context and one bad argument after the other, and with the empty-input fast paths, and walks the layout
queries over their whole range.  The test fails on a wrong return code and on any sanitizer report."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mcmc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mcmc_[a-z0-9_]+)\s*\(", text)))


def test_driver_names_every_entry_point():
    src = open(os.path.join(ROOT, "tests", "abi_check.cpp")).read()
    missing = [name for name in _declared_symbols() if not re.search(r"\b%s\s*\(" % name, src)]
    assert not missing, f"tests/abi_check.cpp does not call {missing}"


@pytest.mark.timeout(900)
def test_abi_argument_validation_under_asan_ubsan():
    if shutil.which("make") is None or not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("needs make and hipcc (the build container)")
    build = subprocess.run(["make", "-j8", "-C", ROOT, "asan"], capture_output=True, text=True, timeout=850)
    assert build.returncode == 0, build.stdout[-2000:] + build.stderr[-4000:]
    exe = os.path.join(ROOT, "build", "asan", "abi_check")
    # instrumented: the sanitizer runtimes are linked in
    syms = subprocess.run(["nm", exe], capture_output=True, text=True).stdout
    assert "__asan_init" in syms and "__ubsan_handle" in syms
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=env)
    out = run.stdout + run.stderr
    assert "AddressSanitizer" not in out and "runtime error" not in out and "LeakSanitizer" not in out, out[-4000:]
    assert run.returncode == 0, out[-4000:]
    m = re.search(r"abi_check: (\d+) checks, 0 failed", run.stdout)
    assert m and int(m.group(1)) > 1000, out[-2000:]
