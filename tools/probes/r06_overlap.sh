#!/bin/bash
# Overlap probes of round 6: the fp32 headline step under grid-size / stagger / in-flight knobs; one line per setting.
# usage (GPU box, repo root): bash tools/probes/r06_overlap.sh > gpurun_out/r06_overlap.txt
run() {  # label, env..., -- bench args
    local label=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    env "${envs[@]}" python bench.py --profile-leg live --steps 20 --warmup 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['conv_families']
print('%-44s %6.2f ms/step  %6.2f Mpx/s  gates live %.3f ms' % ('$label', d['ms_per_step'], d['inflight']*618*618/d['ms_per_step']/1e3, t['conv_gates']['launch_ms']))"
}
run "default (3 in flight)" X=1 --
run "default again" X=1 --
for ms in 3 5 7; do run "stagger $ms ms" TTC_BENCH_STAGGER_MS=$ms --; done
run "2 in flight" X=1 -- --inflight 2
run "2 in flight stagger 7" TTC_BENCH_STAGGER_MS=7 -- --inflight 2
run "4 in flight stagger 3.5" TTC_BENCH_STAGGER_MS=3.5 -- --inflight 4
for g in 240 224 208 192; do run "wino4 grid $g + wino2 resident $((2*g))" TTC_WINO4_GRID=$g TTC_WINO_PERSIST=$((2*g)) --; done
for g in 224 192; do run "grid $g + stagger 5" TTC_WINO4_GRID=$g TTC_WINO_PERSIST=$((2*g)) TTC_BENCH_STAGGER_MS=5 --; done
run "prio stream 0" TTC_BENCH_PRIO=1 --
