# SQ / LDS / TA / TCP counters of the packed DFIRE kernel (own --pmc passes, no tracing).
# Usage (GPU box): bash tools/pmc_packed.sh <tag> [bench args]   -> gpurun_out/pmc_<tag>/
tag=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/pmc_$tag; mkdir -p $out
B="python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 $*"
rocprofv3 --list-avail > $out/avail.txt 2>&1
i=0
while read -r set; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $out/p$i -- $B > /dev/null 2> $out/e$i.log
done <<'SETS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS
SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_FLAT
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_COALESCED_READ_CYCLES_sum
TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_UTCL1_REQUEST_sum
TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_LATENCY_sum
GRBM_GUI_ACTIVE GRBM_COUNT
SETS
python3 - $out <<'P'
import csv,glob,collections,sys
out=sys.argv[1]
for p in sorted(glob.glob(out+'/p*/*/*counter_collection.csv')):
    rows=list(csv.DictReader(open(p)))
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k=r['Kernel_Name']
        if 'pairs<false' in k or 'pose_energy_pairs' in k:
            agg[k[:48]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,d in agg.items():
        for c,v in d.items(): print('%-48s %-40s n=%d mean=%.5g'%(k,c,len(v),sum(v)/len(v)))
P
grep -il "error\|invalid\|not found" $out/e*.log | head
