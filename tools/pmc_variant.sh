# GPU box: instruction counters of the pair kernel for one library variant.  usage: bash tools/pmc_variant.sh <variant>
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
shopt -s nullglob
L=lightdock-rust_amd/lib
cp $L/liblightdock_hip.so /tmp/keep2.so
trap 'cp /tmp/keep2.so $L/liblightdock_hip.so' EXIT INT TERM   # an interrupted run must not leave a variant installed; tools/ab6.sh never installs one
cp $L/variants/$1.so $L/liblightdock_hip.so
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmcv_$1 -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 > /dev/null 2>&1
cp /tmp/keep2.so $L/liblightdock_hip.so
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmcv_$1/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "dfire_packed_pairs<false" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print("$1", k, "%.4g" % (sum(v)/len(v)))
PY
