#!/bin/bash
# usage (GPU box, repo root): bash tools/gpu_profiles.sh <tag>   -> gpurun_out/<tag>_*: bench lines, rocprofv3 kernel stats, PMC passes
TAG=${1:-r02_c}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
prof() {  # name, bench args...
    local name=$1; shift
    rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}_$name -o $name -- python $R/bench.py "$@" > $O/${TAG}_${name}_profiled.json 2> $O/${TAG}_${name}.err
    f=$(find $O/prof_${TAG}_$name -name "*results.db" | head -1)
    (cd $R && python tools/rocpd_stats.py $f > $O/${TAG}_${name}_kernel_stats.md)
    rm -rf $O/prof_${TAG}_$name
}
prof bench                                   # the default command, as the driver runs it
prof fp16 --precision fp16 --no-alt --no-cpu-baseline
prof preprocess --preprocess-only --tiles 64 --no-cpu-baseline
cd $R
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
    t=${TAG}_pmc_$(echo $c | cut -d' ' -f1)
    bash tools/gpu_pmc.sh $t "$c" -- $R/tools/gpu_probe.py 172 4 36 fp16 > /dev/null 2>&1
    f=$(find gpurun_out/pmc_$t -name "*results.db" | head -1)
    echo "== $c"; python tools/rocpd_pmc.py $f "conv3x3_h16<0, 3, 2, 0, 0>"
    rm -rf gpurun_out/pmc_$t
done > $O/${TAG}_pmc_h16_gates.txt 2>&1
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python bench.py --precision fp16 --no-cpu-baseline > $O/${TAG}_bench_fp16.json 2>> $O/${TAG}_bench.err
python bench.py --precision bf16 --no-alt --no-cpu-baseline > $O/${TAG}_bench_bf16.json 2>> $O/${TAG}_bench.err
python bench.py --preprocess-only --tiles 256 --no-cpu-baseline > $O/${TAG}_bench_preprocess_only.json 2>> $O/${TAG}_bench.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/%s_bench*.json" % os.environ.get("TAGX", ""))):
    pass
PY
for f in $O/${TAG}_bench.json $O/${TAG}_bench_fp16.json $O/${TAG}_bench_bf16.json $O/${TAG}_bench_preprocess_only.json $O/${TAG}_bench_profiled.json; do
    python -c "
import json,sys
d=json.load(open('$f')); r=d['roofline']
print('$f'.split('/')[-1], round(d['value']/1e6,2),'Mpx/s', round(d['ms_per_step'],2),'ms/step dprob',d.get('max_dprob'),'| roofline', r.get('launch_ms'), r.get('frac'), r.get('isolated_launch_ms'), '| alt', (d.get('alt_precision') or {}).get('value'))
"
done
cat $O/${TAG}_pmc_h16_gates.txt
