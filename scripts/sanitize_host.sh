#!/bin/bash
# ASan + UBSan run of the host side of libopenvoice_amd.so (a sanitizer build variant of the host
# binding).  Needs no GPU: the CPU ABI tests exercise argument validation, the fp32 / bf16 weight packers and every
# exported symbol; the kernels themselves are device code, which the sanitizers do not instrument.
#   bash scripts/sanitize_host.sh
set -eu
cd "$(dirname "$0")/.."
make -C openvoice_amd/csrc -j"$(nproc)" sanitize > /tmp/ov_sanitize_build.log 2>&1 || { tail -20 /tmp/ov_sanitize_build.log; exit 1; }
RT=$(find /opt/rocm/lib/llvm/lib/clang -name 'libclang_rt.asan-x86_64.so' | head -1)
[ -n "$RT" ] || { echo "ASan runtime not found under /opt/rocm/lib/llvm"; exit 1; }
export LD_PRELOAD="$RT"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1
export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
export OPENVOICE_AMD_LIB="$PWD/openvoice_amd/libopenvoice_amd_san.so"
python -m pytest tests/test_abi_cpu.py tests/test_host_algebra_cpu.py -q -p no:cacheprovider "$@"
