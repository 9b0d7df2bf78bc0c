cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_gapfill.py tests/test_gpu_process_tile.py -m gpu -x -q 2>&1 | tail -2
python bench.py --preprocess-only --tiles 252 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('preprocess', round(d['value']/1e6,1), d['roofline']['frac'])"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rocprofv3 --kernel-trace --stats -d $O/prof_m -o m -- python $R/bench.py --preprocess-only --tiles 30 --inflight 1 --no-cpu-baseline > /dev/null 2>&1
f=$(find $O/prof_m -name "*results.db" | head -1)
cd $R && python tools/rocpd_stats.py $f | grep -E "fix_missing|missing_counts" | cut -c1-70,100-175
rm -rf $O/prof_m
