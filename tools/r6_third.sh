#!/bin/bash
# round 6, third GPU call: culling kernel without VCC selects; timing prototypes (no posing / no final atomics / exact path 8 pairs a lane);
# the three-waves-per-SIMD microbenchmark; kernel trace of the sequence
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_third; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pose_energies or variants_agree or random_molecules or frame_edges or wild or tiny_molecules or full_size or larger_than_one or block_count or receptor_larger or outside_the_f32" > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
( cd tools/microbench && timeout 300 ./mfma_batch 2000 10 26 ) > $O/mfma_batch.txt 2>&1; tail -8 $O/mfma_batch.txt
bash tools/ab6.sh 3 > $O/ab_1k4c.txt 2>&1; cat $O/ab_1k4c.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 10 --warmup 3 --cpu-seconds 0 --no-stats > $O/bench_traced.json 2> $O/trace.log
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs cat | cut -d, -f1-6 | head -14
LIGHTDOCK_HIP_VARIANT=r5 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_r5 -- python3 bench.py --steps 10 --warmup 3 --cpu-seconds 0 --no-stats > $O/bench_traced_r5.json 2> $O/trace_r5.log
find $O/trace_r5 -name "*kernel_stats.csv" | head -1 | xargs cat | cut -d, -f1-6 | head -14
