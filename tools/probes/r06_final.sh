#!/bin/bash
# round-6 closing run (GPU box, repo root): the fp32 legs' kernel stats after the last kernel change, then the driver-style bench lines
cd $GRAFT_REPO_ROOT
LEGS="w172_l4_fp32 w168_l12_fp32" ONLY=stats bash tools/gpu_profiles.sh r06_i > gpurun_out/r06_i_profiles.log 2>&1
python bench.py > gpurun_out/r06_i_bench.json 2> gpurun_out/r06_i_bench.err
python bench.py --precision fp16 --no-cpu-baseline --no-alt > gpurun_out/r06_i_bench_fp16.json 2>> gpurun_out/r06_i_bench.err
python bench.py --preprocess-only --tiles 256 --no-cpu-baseline > gpurun_out/r06_i_bench_preprocess_only.json 2>> gpurun_out/r06_i_bench.err
ls gpurun_out | grep r06_i
