# Collects the rocprofv3 evidence of a round (run on the GPU box):
#   per workload: --kernel-trace --stats of the bench command (no counters), then separate --pmc passes
#   (gpurun refuses --pmc combined with tracing domains), then the un-profiled bench line.  The profiled runs skip bench.py's
#   counting launch (--no-stats): every kernel of a sequence then runs the same number of times, and a mean per launch is one population.
# Usage: bash tools/profile_round.sh <tag> [workload ...]     -> gpurun_out/prof_<tag>/<workload>/
tag=${1:-r03}; shift
wls=${@:-1k4c 1ppe 1azp-dna gso-1ppe gso-1k4c 2uuy}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for w in $wls; do
  out=gpurun_out/prof_$tag/$w; mkdir -p $out
  B="python3 bench.py --workload $w --steps 10 --warmup 3 --cpu-seconds 0 --no-stats"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B > $out/bench_traced.json 2> $out/trace.log
  P="python3 bench.py --workload $w --steps 3 --warmup 1 --cpu-seconds 0 --no-stats"
  i=0
  while read -r set; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --output-format csv -d $out/pmc$i -- $P > /dev/null 2> $out/pmc$i.log
  done <<'SETS'
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
SETS
  timeout 300 python3 bench.py --workload $w --steps 10 --warmup 3 > $out/bench.json 2> $out/bench.err
  python3 tools/summarize_profile.py $out > $out/summary.txt
  cat $out/summary.txt
done
