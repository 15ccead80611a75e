"""The sanitizer build variant of the host side: `make sanitize` compiles libopenvoice_amd's
x86 code -- argument validation, weight packers, dispatch, launch wrappers -- with ASan + UBSan into its own library,
and scripts/sanitize_host.sh runs the CPU ABI tests against it under the ASan runtime.  No GPU needed."""
import glob
import os
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_host_side_is_clean_under_asan_and_ubsan():
    if not glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"):
        pytest.skip("ASan runtime not installed")
    res = subprocess.run(["bash", os.path.join(REPO, "scripts", "sanitize_host.sh")], capture_output=True, text=True,
                         timeout=850, cwd=REPO)
    tail = (res.stdout + res.stderr)[-3000:]
    assert res.returncode == 0, tail
    assert " passed" in res.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
