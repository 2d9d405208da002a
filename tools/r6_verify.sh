#!/bin/bash
# round 6, long checks on the final sources: 400 GSO steps of 1ppe and 200 of 2uuy (ANM form) against the oracle, 300 random rigid complexes
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_verify; mkdir -p $O
timeout 900 python3 tools/long_run_check.py > $O/long_run_1ppe.txt 2>&1; tail -3 $O/long_run_1ppe.txt
timeout 900 python3 tools/long_run_check_anm.py 200 > $O/long_run_2uuy_anm.txt 2>&1; tail -3 $O/long_run_2uuy_anm.txt
timeout 900 python3 tools/fuzz_parity.py 300 1000 > $O/fuzz_rigid.txt 2>&1; tail -2 $O/fuzz_rigid.txt
