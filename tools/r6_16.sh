#!/bin/bash
# round 6: the culling kernel's rigid form boxes an item's eight poses at once, lane = (pose, ligand subtile): parity subset, A/B against the build before it (variant prev), kernel trace
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_16; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pose_energies or variants_agree or random_molecules or frame_edges or wild or tiny_molecules or full_size or larger_than_one or block_count or receptor_larger or outside_the_f32 or gso_steps or nothing_moves or pass_of_more or pass_smaller or device_batch or zero_rows or odd_sizes or degenerate" > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
bash tools/ab6.sh 3 > $O/ab_1k4c.txt 2>&1; cat $O/ab_1k4c.txt
bash tools/ab6.sh 2 --workload 1ppe > $O/ab_1ppe.txt 2>&1; cat $O/ab_1ppe.txt
bash tools/ab6.sh 1 --workload 2uuy > $O/ab_2uuy.txt 2>&1; cat $O/ab_2uuy.txt
bash tools/ab6.sh 1 --workload gso-1k4c > $O/ab_gso1k4c.txt 2>&1; cat $O/ab_gso1k4c.txt
bash tools/ab6.sh 1 --workload gso-1ppe > $O/ab_gso1ppe.txt 2>&1; cat $O/ab_gso1ppe.txt
bash tools/ab6.sh 1 --workload gso-1ppe --swarms 128 > $O/ab_gso1ppe128.txt 2>&1; cat $O/ab_gso1ppe128.txt
