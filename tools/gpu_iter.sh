#!/bin/bash
# Development loop on the GPU box: search/step parity tests, cycle stamps, short bench.
# Usage (through gpurun): bash tools/gpu_iter.sh <tag>
tag=${1:-x}
cd "$(dirname "$0")/.."
timeout -k 10 300 python -u -m pytest tests/test_gpu_search.py tests/test_gpu_step.py -x -q --timeout 120 --timeout-method thread > gpurun_out/it_${tag}_t.log 2>&1 || { tail -5 gpurun_out/it_${tag}_t.log; exit 1; }
tail -1 gpurun_out/it_${tag}_t.log
timeout -k 10 120 python tools/stamp_profile.py > gpurun_out/it_${tag}_s.log 2>&1 || { tail -5 gpurun_out/it_${tag}_s.log; exit 1; }
tail -9 gpurun_out/it_${tag}_s.log
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/it_${tag}_b.log 2>&1 || { tail -5 gpurun_out/it_${tag}_b.log; exit 1; }
python3 -c "
import json,sys
d=json.loads(open('gpurun_out/it_${tag}_b.log').read().strip().splitlines()[-1])
print('value %.1fM  launch %.3f ms  frac %.3f' % (d['value']/1e6, d['roofline']['avg_launch_ms'], d['roofline']['frac']))
"
