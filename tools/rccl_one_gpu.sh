# GPU box with ONE device: does RCCL accept two ranks on the same device?  (It normally refuses: "Duplicate GPU detected".)
# The answer is recorded in DESIGN.md section 7.   usage: bash tools/rccl_one_gpu.sh
cd $GRAFT_REPO_ROOT
LD_BENCH_FORCE_DEVICE=0 NCCL_DEBUG=WARN timeout 300 python bench.py --gpus 2 --backend nccl --workload gso-1ppe --swarms 8 --steps 2 --warmup 1 --cpu-seconds 0 2>&1 | grep -v "^$" | tail -25
