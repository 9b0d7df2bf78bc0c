cd $GRAFT_REPO_ROOT
nproc
for cfg in "4 8" "6 8" "8 4" "8 8" "12 4" "6 16"; do set -- $cfg
TTC_IO_THREADS=$2 python bench.py --job-level-only --job-readers $1 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)['job_level']; print('readers $1 inflate $2: job', round(d['value']/1e6,2), 'given mask', round(d['given_mask']['value']/1e6,2), 'next_tile', d['host_seconds_in_loop']['next_tile_host_s'], d['given_mask']['host_seconds_in_loop']['next_tile_host_s'], 'read ms', d['ms_per_stage_serial']['read_hkl'])"
done
