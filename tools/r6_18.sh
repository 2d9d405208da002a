#!/bin/bash
# round 6: the new culling kernel at six waves per SIMD (workgroups of 6 / 8 waves) against four waves a workgroup (five per SIMD); kernel trace per variant
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_18; mkdir -p $O
bash tools/ab6.sh 2 > $O/ab_1k4c.txt 2>&1; cat $O/ab_1k4c.txt
bash tools/ab6.sh 1 --workload 1ppe > $O/ab_1ppe.txt 2>&1; cat $O/ab_1ppe.txt
bash tools/ab6.sh 1 --workload gso-1ppe --swarms 128 > $O/ab_gso1ppe128.txt 2>&1; cat $O/ab_gso1ppe128.txt
bash tools/trace_variants.sh > $O/trace.txt 2>&1; grep -E "^==|cull<false" $O/trace.txt
