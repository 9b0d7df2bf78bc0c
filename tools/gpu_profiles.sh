#!/bin/bash
# usage (GPU box, repo root): bash tools/gpu_profiles.sh <tag>   -> gpurun_out/<tag>_*: bench lines, rocprofv3 kernel stats, PMC passes.
# Kernel-stats runs use --inflight 1 --no-dprob --no-alt --no-cpu-baseline so that a kernel's average is not a mix of live
# (two tiles in flight), isolated, warm-up and 3-window parity launches.  PMC passes are their own runs (--kernel-trace --pmc only).
# ONLY=<section>[,<section>] restricts the run to stats | pmc_f32 | pmc_h16 | pmc_pre | bench.
TAG=${1:-r05_e}
want() { [ -z "$ONLY" ] || [[ ",$ONLY," == *",$1,"* ]]; }
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
# round 6: one bench LEG per rocprofv3 run (bench.py --profile-leg), so that a kernel's average in a stats file is the average of that leg's
# launches; tools/profile_ref.py turns the summaries into profiles/<round>_bench_profile.json, which bench.py checks its own event times against.
# LEGS="w172_l4_fp32 w172_l4_fp16 ..." selects the legs (default: the headline's three engines + BASELINE's literal geometry)
if want stats; then
for key in ${LEGS:-w172_l4_fp32 w172_l4_fp16 w172_l4_bf16 w168_l12_fp32 w168_l12_fp16}; do
    win=$(echo $key | sed 's/w\([0-9]*\)_l.*/\1/'); len=$(echo $key | sed 's/.*_l\([0-9]*\)_.*/\1/'); prec=${key##*_}
    prof isolated_$key --profile-leg isolated --steps 10 --win $win --length $len --precision $prec
    prof live_$key --profile-leg live --steps 20 --warmup 3 --win $win --length $len --precision $prec
done
prof preprocess --preprocess-only --tiles 64 --inflight 1 --no-cpu-baseline
fi
cd $R
pmc() {  # out-file, kernel substring(s) separated by |, counters, command...
    local out=$1 subs=$2 cnt=$3; shift 3
    local t=${TAG}_pmc_$(echo $cnt | cut -d' ' -f1)_$(basename $out .txt)
    bash tools/gpu_pmc.sh $t "$cnt" -- "$@" > /dev/null 2>&1
    f=$(find gpurun_out/pmc_$t -name "*results.db" | head -1)
    echo "== $cnt" >> $out
    IFS='|' read -ra SS <<< "$subs"
    for sname in "${SS[@]}"; do python tools/rocpd_pmc.py $f "$sname" >> $out; done
    rm -rf gpurun_out/pmc_$t
}
# the ConvGRU gates launch of every engine / geometry bench.py prices (tools/pmc_json.py -> profiles/<round>_pmc_gates_<precision>_w<win>_l<len>.json)
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
    want pmc_f32 && pmc $O/${TAG}_pmc_gates_fp32_w172_l4.txt "conv3x3_wino4<0|conv3x3_wino4<2|conv3x3_wino<1, 1" "$c" $R/tools/gpu_probe.py 172 4 36 fp32
    want pmc_h16 && pmc $O/${TAG}_pmc_gates_fp16_w172_l4.txt "conv3x3_h16<0, 3, 2, 0, 1|k_gru_apply2_b16|k_gru_apply1_b16" "$c" $R/tools/gpu_probe.py 172 4 36 fp16
    want pmc_b16 && pmc $O/${TAG}_pmc_gates_bf16_w172_l4.txt "conv3x3_h16<1, 3, 2, 0, 1" "$c" $R/tools/gpu_probe.py 172 4 36 bf16
    want pmc_l12 && pmc $O/${TAG}_pmc_gates_fp32_w168_l12.txt "conv3x3_wino4<0" "$c" $R/tools/gpu_probe.py 168 12 36 fp32
    want pmc_l12 && pmc $O/${TAG}_pmc_gates_fp16_w168_l12.txt "conv3x3_h16<0, 3, 2, 0, 1" "$c" $R/tools/gpu_probe.py 168 12 36 fp16
done
for c in FETCH_SIZE WRITE_SIZE; do
    want pmc_pre && pmc $O/${TAG}_pmc_preprocess.txt "k_med_count|k_med_final|k_med_bracket|k_med_sample|k_tile_temporal|k_assemble|k_ref_all|k_gram_all|k_gram_snow|k_accum_final_all|k_decode_upsample" "$c" $R/bench.py --preprocess-only --tiles 4 --inflight 1 --warmup 1 --no-cpu-baseline
done
# the WHOLE preprocessing chain's HBM traffic per tile (VERDICT r5 #4d): each counter in its own pass, summed over every dispatch of 1 warm-up + 8 tiles
if want pmc_pre_total; then
for c in FETCH_SIZE WRITE_SIZE; do
    t=${TAG}_pmctot_$c
    bash tools/gpu_pmc.sh $t "$c" -- $R/bench.py --preprocess-only --tiles 8 --inflight 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
    f=$(find gpurun_out/pmc_$t -name "*results.db" | head -1)
    python tools/rocpd_pmc_total.py $f $c 9 > $O/${TAG}_pmc_preprocess_total_$c.json
    rm -rf gpurun_out/pmc_$t
done
fi
want bench || exit 0
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python bench.py --precision fp16 --no-cpu-baseline --no-alt > $O/${TAG}_bench_fp16.json 2>> $O/${TAG}_bench.err
python bench.py --preprocess-only --tiles 256 --no-cpu-baseline > $O/${TAG}_bench_preprocess_only.json 2>> $O/${TAG}_bench.err
for f in $O/${TAG}_bench.json $O/${TAG}_bench_fp16.json $O/${TAG}_bench_preprocess_only.json $O/${TAG}_fp32_profiled.json $O/${TAG}_fp16_profiled.json; do
    python -c "
import json,sys
d=json.load(open('$f')); r=d['roofline']
print('$f'.split('/')[-1], round(d['value']/1e6,2),'Mpx/s', round(d['ms_per_step'],2),'ms/step dprob',d.get('max_dprob'),'| roofline', r.get('launch_ms'), r.get('frac'), r.get('isolated_launch_ms'))
"
done
