#!/bin/bash
# round 6, sixth GPU call: what the atomics still cost; culling items of 4 / 16 poses (run length of the rows a batch's lanes hold)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_sixth; mkdir -p $O
for v in cull16 cull4; do
LIGHTDOCK_HIP_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pose_energies or variants_agree or frame_edges or full_size or larger_than_one or nothing_moves" > $O/pytest_$v.txt 2>&1
echo $v; tail -2 $O/pytest_$v.txt
done
bash tools/ab6.sh 3 > $O/ab_1k4c.txt 2>&1; cat $O/ab_1k4c.txt
bash tools/ab6.sh 1 --workload 1ppe > $O/ab_1ppe.txt 2>&1; cat $O/ab_1ppe.txt
bash tools/ab6.sh 1 --workload gso-1ppe --swarms 128 > $O/ab_gso1ppe128.txt 2>&1; cat $O/ab_gso1ppe128.txt
