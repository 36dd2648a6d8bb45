#!/bin/bash
# round 4, GPU call 18: the solve / update bodies entered by the working waves only: kernel time against the build before, WRITE_SIZE, parity
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
timeout 600 python tools/ab_timing.py ab/base.so ab/split.so mr > gpurun_out/r04/ab18.txt 2>&1; cat gpurun_out/r04/ab18.txt
timeout 900 python tools/pmc_traffic.py r04 > gpurun_out/r04_pmc_traffic.log 2>&1
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04_pmc_traffic.json'))
print({k: d[k] for k in ('fetch_size_bytes', 'write_size_bytes', 'bytes_lo', 'bytes_hi') if k in d})
PY
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref.py -m gpu -q --maxfail=10 > gpurun_out/r04/pytest18.log 2>&1; tail -2 gpurun_out/r04/pytest18.log
