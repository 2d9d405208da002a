#!/bin/bash
# round 6 (VERDICT r05 item 5 i): the ANM form with W = 16 A and the Cauchy-Schwarz WILD test: parity, fuzz, bench, wave times
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_anm; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "anm or wild or 2uuy or ab_icode or normal_modes or which_kernel or variants_agree or pose_energies or launcher or 1czy or receptor_anm" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 900 python3 tools/fuzz_parity.py 120 1 anm > $O/fuzz_anm.txt 2>&1; tail -3 $O/fuzz_anm.txt
bash tools/ab6.sh 3 --workload 2uuy > $O/ab_2uuy.txt 2>&1; cat $O/ab_2uuy.txt
timeout 300 python3 tools/bm_wave_times.py --workload 2uuy > $O/bm_wave_times_2uuy.txt 2>&1; cat $O/bm_wave_times_2uuy.txt
