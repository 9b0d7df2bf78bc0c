#!/bin/bash
# usage: tools/gpu_pmc.sh <tag> "<counters>" -- <python args>   (runs on the GPU box from the repo root)
TAG=$1; CNT=$2; shift 3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CNT -d $R/gpurun_out/pmc_$TAG -o $TAG -- python "$@" > $R/gpurun_out/pmc_$TAG.log 2>&1
tail -2 $R/gpurun_out/pmc_$TAG.log
