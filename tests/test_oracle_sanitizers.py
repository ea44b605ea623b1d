"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (gcc; GPU sanitizers are not available on the
pool): oracle/selftest.c drives one reset, one actuated control interval and every getter of the 2D and 3D oracles."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "selftest")
    src = [os.path.join(ROOT, "oracle", f) for f in ("selftest.c", "rbc_oracle.c", "rbc_oracle3d.c")]
    build = subprocess.run(["gcc", "-std=c11", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
                            "-fno-sanitize-recover=undefined", "-ffp-contract=off", "-I", os.path.join(ROOT, "oracle"), *src, "-lm", "-o", exe],
                           capture_output=True, text=True, timeout=300)
    if build.returncode != 0 and ("asan" in build.stderr.lower() or "sanitize" in build.stderr.lower()):
        pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600,
                         env={**os.environ, "ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"})
    assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
    assert "selftest: ok" in run.stdout and "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr
