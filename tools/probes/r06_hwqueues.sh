#!/bin/bash
# probe: does the HIP runtime's hardware-queue count (GPU_MAX_HW_QUEUES, default 4) cap the concurrency of the preprocessing chain's small kernels?
for q in 4 8; do for n in 3 6 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py --preprocess-only --tiles 192 --inflight $n --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('GPU_MAX_HW_QUEUES=$q inflight=$n: %.1f Mpx/s  %.3f ms/tile' % (d['value']/1e6, d['config']['ms_per_tile']))"
done; done
for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py --profile-leg live --steps 20 --warmup 3 --inflight 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('headline GPU_MAX_HW_QUEUES=$q inflight=4: %.2f ms/tile' % (d['ms_per_step']/4))"
done
