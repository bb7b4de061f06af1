"""Sanitizer run of the library's HOST code: `make asan` builds libudaseg_hip_asan.so with AddressSanitizer +
UBSan on the host side; tools/asan_host_check.py then drives ~24 000 entry-point calls (argument validation, tap / parity
tables, K-slice and split-K plans, fused-input and split-output descriptions) through it.  CPU only: the GPU pool has no
sanitizer support, and every call stops at its first HIP call for want of a device -- after the host code under test ran.
(First catch: an empty batch reached the weight-gradient split-K planner and divided by zero.)"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "uda_aerial_semantic_segmentation_research_amd", "csrc")
CLANG = "/opt/rocm/lib/llvm/bin/clang"


@pytest.mark.skipif(shutil.which("make") is None or not os.path.exists(CLANG), reason="ROCm toolchain not present")
def test_host_code_is_clean_under_asan_and_ubsan():
    import torch
    if torch.cuda.is_available():
        pytest.skip("sanitizer check is for the CPU box (on a GPU box the calls would launch kernels on fake pointers)")
    subprocess.run(["make", "-C", CSRC, "asan", "-j8"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    rt = subprocess.run([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], check=True, capture_output=True, text=True).stdout.strip()
    assert os.path.exists(rt), rt
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               UDASEG_LIB=os.path.join(ROOT, "uda_aerial_semantic_segmentation_research_amd", "libudaseg_hip_asan.so"))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asan_host_check.py")], env=env, capture_output=True, text=True,
                       timeout=600)
    report = p.stdout[-2000:] + p.stderr[-4000:]
    assert p.returncode == 0, report
    assert "asan host check ok" in p.stdout, report
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, report
