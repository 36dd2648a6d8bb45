#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
timeout 900 python tools/pmc_traffic.py r04 > gpurun_out/r04_pmc_traffic.log 2>&1; tail -5 gpurun_out/r04_pmc_traffic.log
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04_pmc_traffic.json'))
print({k: d[k] for k in ('fetch_size_bytes', 'write_size_bytes', 'bytes_lo', 'bytes_hi') if k in d})
PY
