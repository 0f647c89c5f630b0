#!/bin/bash
# bench lines of the other BASELINE configs -> gpurun_out/cfg_<name>.json
for w in c1_test3k c2_10kx3k c3_framelaff_10kx3k; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/cfg_$w.json
done
timeout 900 python bench.py --workload c5_ml_100kx30k --precision bf16 --no-cpu-baseline --steps 10 2>/dev/null | tail -1 > gpurun_out/cfg_c5_bf16.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/cfg_*.json')):
    try:
        d=json.loads(open(f).read())
    except Exception as e:
        print(f, 'unparsable', e); continue
    print(f, 'ms/step %.4f' % d['ms_per_step'], {k: round(v['ms_per_step'],4) for k,v in d.get('kernels',{}).items()}, d.get('stages_ms_eager_pass'))
PY
