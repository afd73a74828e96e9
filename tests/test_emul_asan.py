"""Memory-safety check of the kernels and their host launch code: the shared parity cases on an
AddressSanitizer build of the host-emulated kernel sources (the GPU-less stand-in for a compute
sanitizer).  Skipped when the clang ASan runtime is not installed."""
import glob
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_parity_cases_are_asan_clean():
    rts = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not rts or not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("clang AddressSanitizer runtime not available")
    subprocess.check_call([os.path.join(HERE, "emul", "build_asan.sh")])
    env = dict(os.environ, LD_PRELOAD=rts[0], ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0")
    p = subprocess.run([sys.executable, os.path.join(HERE, "emul", "asan_run.py")], env=env,
                       capture_output=True, text=True, timeout=1500)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0 and "ASAN_CLEAN" in p.stdout and "AddressSanitizer" not in p.stderr, tail
