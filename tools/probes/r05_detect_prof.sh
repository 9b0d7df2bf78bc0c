cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rocprofv3 --kernel-trace --stats -d $O/prof_d -o d -- python $R/bench.py --detect --inflight 1 --no-dprob --no-alt --no-cpu-baseline --steps 10 > $O/r05_detect_profiled.json 2> $O/r05_detect.err
f=$(find $O/prof_d -name "*results.db" | head -1)
cd $R && python tools/rocpd_stats.py $f > $O/r05_detect_kernel_stats.md
rm -rf $O/prof_d
grep -E "k_cd_|cd_|clouds|k_c[a-z]*_" $O/r05_detect_kernel_stats.md | head -50 | cut -c1-90,130-200
python -c "
import json; d=json.load(open('$O/r05_detect_profiled.json')); print(d['value']/1e6, d['ms_per_step'])"
