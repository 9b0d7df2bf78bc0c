cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for n in 3 4; do
rocprofv3 --kernel-trace -d $O/prof_h -o h -- python $R/bench.py --no-dprob --no-alt --no-cpu-baseline --steps 60 --warmup 5 --inflight $n > $O/r05_head_traced.json 2> $O/r05_head_traced.err
f=$(find $O/prof_h -name "*results.db" | head -1)
echo "== inflight $n"; python -c "
import json; d=json.load(open('$O/r05_head_traced.json')); print(d['value']/1e6, d['ms_per_step'])"
cd $R && python tools/probes/job_loop_timeline.py $f 0.45 0.85 2>&1 | grep -v "table"
rm -rf $O/prof_h; cd /tmp
done
