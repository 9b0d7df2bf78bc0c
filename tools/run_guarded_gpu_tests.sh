#!/bin/bash
# Device-side memory check of the HIP kernels without a sanitizer (GPU AddressSanitizer / xnack are refused on the target pool):
# the GPU suite with TTC_GUARD=<KiB> -- every device buffer the library owns sits between two 0xA5-filled guard zones, scanned when its context
# closes (ttc_debug_check_guards).  Catches out-of-bounds WRITES within the zone of any owned buffer; reads are not seen.
#   usage (GPU box, repo root): bash tools/run_guarded_gpu_tests.sh [KiB] [pytest args...]   -> gpurun_out/device_guard_report.txt
K=${1:-64}; shift
mkdir -p gpurun_out; rm -f gpurun_out/device_guard_report.txt
TTC_GUARD=$K python -m pytest tests -m gpu -q "$@" 2>&1 | grep -v Warning | tail -25
cat gpurun_out/device_guard_report.txt
