#!/bin/bash
# round 6: per-receptor-subtile reach (zero rows of the potential): the new tests, the parity subset, A/B against the build before it
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_14; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "zero_rows" > $O/pytest_new.txt 2>&1; tail -15 $O/pytest_new.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pose_energies or variants_agree or random_molecules or frame_edges or wild or tiny_molecules or full_size or larger_than_one or block_count or receptor_larger or outside_the_f32 or gso_steps or nothing_moves or pass_of_more or pass_smaller or device_batch or bins_that_are_zero or anm" > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
bash tools/ab6.sh 3 > $O/ab_1k4c.txt 2>&1; cat $O/ab_1k4c.txt
bash tools/ab6.sh 2 --workload 2uuy > $O/ab_2uuy.txt 2>&1; cat $O/ab_2uuy.txt
bash tools/ab6.sh 1 --workload 1ppe > $O/ab_1ppe.txt 2>&1; cat $O/ab_1ppe.txt
