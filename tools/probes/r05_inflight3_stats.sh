cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rocprofv3 --kernel-trace --stats -d $O/prof_i3 -o i3 -- python $R/bench.py --no-alt --no-dprob --no-cpu-baseline > $O/r05_g_fp32_inflight3_profiled.json 2> $O/r05_g_i3.err
f=$(find $O/prof_i3 -name "*results.db" | head -1)
cd $R && python tools/rocpd_stats.py $f > $O/r05_g_fp32_inflight3_kernel_stats.md
rm -rf $O/prof_i3
head -12 $O/r05_g_fp32_inflight3_kernel_stats.md | cut -c1-60,120-200
python -c "
import json; d=json.load(open('$O/r05_g_fp32_inflight3_profiled.json')); r=d['roofline']; print(d['value']/1e6, r['launch_ms'], r['frac'], r.get('isolated_launch_ms'))"
