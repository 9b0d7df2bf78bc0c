cd $GRAFT_REPO_ROOT
for q in 4 8 16; do
echo "== GPU_MAX_HW_QUEUES=$q"
GPU_MAX_HW_QUEUES=$q python bench.py --job-level-only 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)['job_level']; print('job', round(d['value']/1e6,2), 'given mask', round(d['given_mask']['value']/1e6,2), d['host_seconds_in_loop'], d['given_mask']['host_seconds_in_loop'])"
GPU_MAX_HW_QUEUES=$q python bench.py --no-dprob --no-alt --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('headline', round(d['value']/1e6,2), 'sustained', d.get('sustained',{}).get('value'))"
done
