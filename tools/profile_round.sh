# Collects the rocprofv3 evidence for the bench line of this round (run on the GPU box):
#   1. --kernel-trace --stats of the default bench command (no counters)
#   2. separate --pmc passes (gpurun refuses --pmc combined with tracing domains)
# Usage: bash tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag; mkdir -p $out
B="python3 bench.py --steps 10 --warmup 3 --cpu-seconds 0"
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B > $out/bench_traced.json 2> $out/trace.log
P="python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0"
timeout 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $P > /dev/null 2> $out/fetch.log
timeout 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- $P > /dev/null 2> $out/write.log
timeout 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $out/tcc -- $P > /dev/null 2> $out/tcc.log
timeout 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $out/sq1 -- $P > /dev/null 2> $out/sq1.log
timeout 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS --output-format csv -d $out/sq2 -- $P > /dev/null 2> $out/sq2.log
timeout 240 python3 bench.py --steps 10 --warmup 3 > $out/bench.json 2> $out/bench.err
python3 tools/summarize_profile.py $out > $out/summary.txt
cat $out/summary.txt
