#!/bin/bash
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_ops.py tests/test_gpu_f16.py tests/test_gpu_bounds.py tests/test_gpu_configs.py -x -q -m gpu -k "loss or cross_entropy or softmax or seg or pair or config" 2>&1 | tail -4 || exit 1
bash tools/dev/quick_stats.sh train-b32 > gpurun_out/quick_stats_train.txt 2>&1 || exit 1
bash tools/dev/quick_stats.sh highres-fp16 > gpurun_out/quick_stats_f16.txt 2>&1 || exit 1
grep -i "seg_\|finish\|kernel time" gpurun_out/quick_stats_train.txt gpurun_out/quick_stats_f16.txt
cat gpurun_out/quick_train-b32/bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
cat gpurun_out/quick_highres-fp16/bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
