#!/bin/bash
# Host-side sanitizer run (no GPU needed): builds the library with AddressSanitizer + UBSan on its host code (make san) and runs the
# CPU tests that execute host C++ through the C ABI -- the HDF5 / hickle reader (incl. the malformed-file cases), the GeoTIFF writer,
# the export table -- against it.  Python itself is not instrumented: leak checking is off, the ASan runtime is preloaded.
set -e
cd "$(dirname "$0")/.."
OUT=${SAN_OUT:-/tmp/ttc_san}
make -C sentinel-tree-cover_amd/csrc -j"${JOBS:-8}" san SAN_OUT="$OUT" > "$OUT.build.log" 2>&1 || { tail -20 "$OUT.build.log"; exit 1; }
ASAN=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
export LD_PRELOAD="$ASAN" TTC_LIB="$OUT/libttc_hip_san.so"
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1" UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"
python -m pytest tests/test_hkl_reader.py tests/test_geotiff.py tests/test_lib_exports.py -q -x -p no:cacheprovider "$@"
