#!/bin/bash
# round 6, first GPU call: the parity suite on the round's first edits, the two microbenchmarks, a baseline of the headline
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r6_first
( cd tools/microbench && timeout 300 ./mfma_batch 2000 ) > gpurun_out/r6_first/mfma_batch.txt 2>&1
( cd tools/microbench && timeout 300 ./valu_rate ) > gpurun_out/r6_first/valu_rate.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6_first/pytest.txt 2>&1
for i in 1 2 3; do timeout 200 python bench.py --cpu-seconds 0 > gpurun_out/r6_first/bench_$i.json 2> gpurun_out/r6_first/bench_$i.err; done
timeout 300 python bench.py > gpurun_out/r6_first/bench_full.json 2> gpurun_out/r6_first/bench_full.err
timeout 200 python bench.py --workload 2uuy --cpu-seconds 0 > gpurun_out/r6_first/bench_2uuy.json 2>&1
tail -3 gpurun_out/r6_first/pytest.txt; cat gpurun_out/r6_first/mfma_batch.txt; tail -16 gpurun_out/r6_first/valu_rate.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6_first/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, '%.0f'%d['value'], d['ms_per_step'], d.get('ms_per_step_min'), d.get('ms_per_step_median'), d.get('ms_per_step_max'), d['roofline']['kernel_ms'])
    except Exception as e: print(f, 'FAILED', e)
PY
