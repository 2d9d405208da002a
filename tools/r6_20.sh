#!/bin/bash
# round 6: the culling kernel's step loop over a lane-distributed list of the surviving tiles (variant list) against the scalar bit walk (e1: boxes read as two ds_read_b128) and the build before (prev)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_20; mkdir -p $O
for v in prev e1 list; do LIGHTDOCK_HIP_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pose_energies or variants_agree or random_molecules or tiny_molecules or block_count or receptor_larger or zero_rows or device_batch" > $O/pytest_$v.txt 2>&1; echo "$v: $(tail -1 $O/pytest_$v.txt)"; done
bash tools/trace_variants.sh > $O/trace.txt 2>&1; grep -E "^==|cull<false|pairs<false" $O/trace.txt
bash tools/trace_variants.sh --workload 1ppe > $O/trace_1ppe.txt 2>&1; grep -E "^==|cull<false" $O/trace_1ppe.txt
bash tools/trace_variants.sh --workload 2uuy > $O/trace_2uuy.txt 2>&1; grep -E "^==|cull<false" $O/trace_2uuy.txt
bash tools/ab6.sh 2 > $O/ab_1k4c.txt 2>&1; cat $O/ab_1k4c.txt
