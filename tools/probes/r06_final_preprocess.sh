set -x
cd $GRAFT_REPO_ROOT
ONLY=pmc_pre_total,pmc_pre bash tools/gpu_profiles.sh r06_f > gpurun_out/r06_f_profiles.log 2>&1
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $O/prof_r06_f_pre -o pre -- python $R/bench.py --preprocess-only --tiles 64 --inflight 1 --no-cpu-baseline > $O/r06_f_preprocess_profiled.json 2> $O/r06_f_preprocess.err
f=$(find $O/prof_r06_f_pre -name "*results.db" | head -1)
(cd $R && python tools/rocpd_stats.py $f > $O/r06_f_preprocess_kernel_stats.md)
rm -rf $O/prof_r06_f_pre
cd $R
python bench.py --preprocess-only --tiles 256 --no-cpu-baseline > gpurun_out/r06_f_bench_preprocess_only.json 2>> gpurun_out/r06_f_bench.err
ls gpurun_out | grep r06_f
