"""Round 4: the host-side C/C++ (legacy C ABI in smm_legacy.cpp, the host parts of smm_api.hip, the oracle's C
restatement) is built with AddressSanitizer + UndefinedBehaviorSanitizer and driven by the CPU test files
(scripts/sanitize_cpu.sh).  The reference has no sanitizer build at all; GPU sanitizers are not
available on this pool, so this is the host pass only.  ~50 s, most of it hipcc compiling the (unsanitized) device code."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(os.environ.get("SMM_LIB_PATH", "").endswith("_san.so"), reason="already inside the sanitizer job")
@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_host_code_is_clean_under_asan_and_ubsan(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "SMM_LIB_PATH", "SMM_ORACLE_LIB")}
    out = subprocess.run(["bash", os.path.join(ROOT, "scripts", "sanitize_cpu.sh"), str(tmp_path)], env=env,
                         capture_output=True, text=True, timeout=900)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0, tail
    assert "sanitizers: clean" in out.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail
