#!/bin/bash
# AddressSanitizer + UBSan build of the HOST side of libscfgp_hip.so (device code is not instrumented: GPU ASan is not
# available on this pool) and a run of every entry point that works without a GPU: argument checks, error paths of
# scfgp_create on a GPU-less box, scfgp_selftest_row_splits over many shapes.  Usage: bash tools/asan_host.sh
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
make -C $root/scfgp_amd/csrc -j4 VARIANT=_asan EXTRA="-fsanitize=address,undefined -fno-gpu-sanitize -g -fno-omit-frame-pointer" > /dev/null
rt=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 LD_PRELOAD=$rt python3 $root/tools/asan_host_driver.py
