#!/bin/bash
# round 5: the block-major ANM form -- parity subset, then the 2uuy bench (both kernels)
export TMPDIR=/tmp
mkdir -p gpurun_out/r05anm
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "2uuy or anm or icode or 1czy or outside or variants_agree or zero_for_the_whole" \
  --deselect tests/test_gpu_parity.py::test_committed_profile_matches_the_kernel_sources > gpurun_out/r05anm/pytest.txt 2>&1
tail -30 gpurun_out/r05anm/pytest.txt
timeout 300 python bench.py --workload 2uuy --steps 20 --warmup 3 > gpurun_out/r05anm/bench_2uuy_bm.json 2> gpurun_out/r05anm/bench_2uuy_bm.err
LIGHTDOCK_BM_ANM=0 timeout 300 python bench.py --workload 2uuy --steps 20 --warmup 3 > gpurun_out/r05anm/bench_2uuy_packed.json 2> gpurun_out/r05anm/bench_2uuy_packed.err
tail -3 gpurun_out/r05anm/bench_2uuy_bm.err; cat gpurun_out/r05anm/bench_2uuy_bm.json gpurun_out/r05anm/bench_2uuy_packed.json
