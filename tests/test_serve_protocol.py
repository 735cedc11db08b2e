"""The query servers' ticket protocol under ThreadSanitizer and AddressSanitizer + UBSan, on host threads.

tests/native/serve_sim.cpp runs the PRODUCT's protocol code — the caller side (csrc/jv_serve_host.h, what jv_abi.cpp's
serve_query calls) and the grid side (csrc/jv_serve_claim.h, what the resident kernels run, compiled for the host through a
shim) — with 256 caller threads issuing one query per call (the reference's pattern,
T/index/engine/JVectorConcurrentQueryTests.java:78-138), next to server pauses, grids that idle out and are restarted,
failing launches (abandoned tickets) and filtered / unfiltered slots.  Every answered call is verified; a deadlock trips an alarm.
Round 5: this test found a race the GPU stress tests never hit (a slot marked abandoned only after the launch lock was dropped
could be answered by the next caller's grid while it was already handed on); fixed in jv_serve_host.h."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "serve_sim.cpp")
INC = os.path.join(ROOT, "opensearch-jvector_amd", "csrc")


def _build(tag, flags):
    out = os.path.join(ROOT, "build", f"serve_sim_{tag}")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-pthread", "-I", INC] + flags + [SRC, "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
@pytest.mark.parametrize("tag,flags,env", [
    ("tsan", ["-fsanitize=thread"], {"TSAN_OPTIONS": "halt_on_error=0 exitcode=66"}),
    ("asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"], {"ASAN_OPTIONS": "detect_leaks=1"}),
])
def test_ticket_protocol_is_clean_under_the_sanitizers(tag, flags, env):
    exe = _build(tag, flags)
    for args in (["256", "40", "8", "37"], ["64", "120", "3", "7"], ["96", "80", "5", "0"]):
        r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
        out = r.stdout + r.stderr
        assert "ThreadSanitizer" not in out and "AddressSanitizer" not in out and "runtime error" not in out, out[-4000:]
        assert r.returncode == 0, out[-2000:]
        assert "wrong 0" in r.stdout, r.stdout
