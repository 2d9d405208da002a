#!/bin/bash
# round 6, second GPU call: parity of the restructured batch loop, A/B against round 5's library, the microbenchmarks again
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_second; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pose_energies or variants_agree or gso_steps or random_molecules or frame_edges or wild or tiny_molecules or full_size or larger_than_one or zero_for_the_whole or nothing_moves or one_rigid" > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
bash tools/ab6.sh 3 > $O/ab_1k4c.txt 2>&1; cat $O/ab_1k4c.txt
bash tools/ab6.sh 2 --workload 1ppe > $O/ab_1ppe.txt 2>&1; cat $O/ab_1ppe.txt
bash tools/ab6.sh 2 --workload 2uuy > $O/ab_2uuy.txt 2>&1; cat $O/ab_2uuy.txt
bash tools/ab6.sh 1 --workload gso-1k4c > $O/ab_gso1k4c.txt 2>&1; cat $O/ab_gso1k4c.txt
( cd tools/microbench && timeout 300 ./mfma_batch 2000 10 26 ) > $O/mfma_batch_10_26.txt 2>&1; cat $O/mfma_batch_10_26.txt
( cd tools/microbench && timeout 300 ./mfma_batch 2000 13 40 ) > $O/mfma_batch_13_40.txt 2>&1; cat $O/mfma_batch_13_40.txt
( cd tools/microbench && timeout 300 ./valu_rate ) > $O/valu_rate.txt 2>&1; grep -v "^{" $O/valu_rate.txt | cut -c1-150
