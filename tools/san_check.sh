#!/bin/bash
# The HOST sources of libtcmi.so under AddressSanitizer + UBSan (CPU only: GPU sanitizers are not available on this pool).
#   make -C trueconsense_amd/csrc SAN=1  ->  trueconsense_amd/lib/san/libtcmi.so (the .hip files carry no instrumentation)
# then the non-GPU tests and a loop of randomly damaged files through tcmi_bam_load, with the runtime preloaded into python.
#   tools/san_check.sh [damaged files = 400]
set -euo pipefail
cd "$(dirname "$0")/.."
make -C trueconsense_amd/csrc SAN=1 -j8 >/dev/null
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
export TCMI_LIB=$PWD/trueconsense_amd/lib/san/libtcmi.so
# (python itself is not instrumented: its interned allocations are not leaks of ours; UBSan reports are made fatal)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
LD_PRELOAD=$RT python tools/san_damaged_loop.py "${1:-400}"
LD_PRELOAD=$RT python -m pytest tests -q -m "not gpu" -p no:cacheprovider \
    --deselect tests/test_abi.py::test_rccl_hook_library_exports_what_its_header_declares \
    --deselect tests/test_c_abi_from_c.py
echo "san_check: clean"
