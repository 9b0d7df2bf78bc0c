cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rocprofv3 --kernel-trace --memory-copy-trace -d $O/prof_job -o job -- python $R/bench.py --job-level-only > $O/r05_job_traced.json 2> $O/r05_job_traced.err
f=$(find $O/prof_job -name "*results.db" | head -1)
cd $R && python tools/probes/job_loop_gaps.py $f > $O/r05_job_timeline.txt 2>&1
rm -rf $O/prof_job
cat $O/r05_job_timeline.txt; tail -c 1500 $O/r05_job_traced.json
