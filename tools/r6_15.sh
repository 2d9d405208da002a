#!/bin/bash
# round 6: the launch-shape constants re-swept on the refined atom order (culling item size, P factor, the census's cost model)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_15; mkdir -p $O
bash tools/ab6.sh 2 > $O/ab_1k4c.txt 2>&1; cat $O/ab_1k4c.txt
bash tools/ab6.sh 1 --workload 1ppe > $O/ab_1ppe.txt 2>&1; cat $O/ab_1ppe.txt
bash tools/ab6.sh 1 --workload gso-1ppe --swarms 128 > $O/ab_gso128.txt 2>&1; cat $O/ab_gso128.txt
