#!/bin/bash
# round 6: the compaction's chunk loop with an early exit (variant chunkexit against diagbase, both -DLD_DIAG_BUILD), mu0 again; the headline complex at larger batches
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_thirteenth; mkdir -p $O
bash tools/ab6.sh 3 > $O/ab_1k4c.txt 2>&1; cat $O/ab_1k4c.txt
bash tools/ab6.sh 1 --workload 1ppe > $O/ab_1ppe.txt 2>&1; cat $O/ab_1ppe.txt
bash tools/ab6.sh 1 --workload 2uuy > $O/ab_2uuy.txt 2>&1; cat $O/ab_2uuy.txt
for b in 4096 16384 19456 32768 65536; do echo "batch $b: $(timeout 200 python3 bench.py --cpu-seconds 0 --batch $b 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f evals/s step %.4f ms kernel %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))")"; done > $O/batches.txt 2>&1; cat $O/batches.txt
