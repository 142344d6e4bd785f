"""CPU: `make asan` -- the library's host code (cost model, ufv_gemm dispatch, split-K flag ring across its wrap-around and from two threads, error word, no-device
path, the whole-stage calls' workspace carving) compiled with -fsanitize=address,undefined against a stand-in HIP runtime (tests/asan/hip_stub.cpp) and driven by
tests/asan/host_logic.cpp.  GPU AddressSanitizer is not available on the pool; the kernels' own memory discipline is what the parity tests and the ISA audits cover."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not installed")
def test_host_logic_is_clean_under_address_and_undefined_behaviour_sanitizers():
    r = subprocess.run(["make", "-C", ROOT, "-j4", "asan"], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout[-2500:], r.stderr[-2500:])
    assert "0 failed checks" in r.stdout and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, (r.stdout[-800:], r.stderr[-2000:])
    # ... and the instrumentation is real: a deliberate one-element heap overrun in the same binary is caught
    exe = os.path.join(ROOT, "build", "asan", "host_logic")
    s = subprocess.run([exe], env=dict(os.environ, UFV_ASAN_SELFTEST="1"), capture_output=True, text=True, timeout=120)
    assert s.returncode != 0 and "AddressSanitizer" in s.stderr and "heap-buffer-overflow" in s.stderr and "unnoticed" not in s.stdout
