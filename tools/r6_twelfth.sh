#!/bin/bash
# round 6: the refined atom order -- wave timers against the previous library, the pairing-aware refinement off (variant mu0), kernel trace
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_twelfth; mkdir -p $O
bash tools/ab6.sh 3 > $O/ab_1k4c.txt 2>&1; cat $O/ab_1k4c.txt
bash tools/ab6.sh 1 --workload 1ppe > $O/ab_1ppe.txt 2>&1; cat $O/ab_1ppe.txt
bash tools/ab6.sh 1 --workload 2uuy > $O/ab_2uuy.txt 2>&1; cat $O/ab_2uuy.txt
echo "== installed" > $O/waves.txt; timeout 120 python tools/bm_wave_times.py >> $O/waves.txt 2>&1
bash tools/ab_wave_times.sh >> $O/waves.txt 2>&1; cat $O/waves.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-stats --cpu-seconds 0 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(ls $O/trace/*/*kernel_stats.csv | head -1); head -12 $f | cut -c1-200
