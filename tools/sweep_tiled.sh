# throughput of the tiled DFIRE kernel for several workgroup shapes (run on the GPU box)
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python3 bench.py --steps 5 --warmup 2 --batch ${BATCH:-8192} --workload ${WORKLOAD:-1k4c} --cpu-seconds 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('evals/s %.0f kernel_ms %.3f frac %.3f'%(d['value'], d['roofline']['kernel_ms'], d['roofline']['frac']))"; }
for w in ${WAVES:-1 2 4 5 6 8}; do run LIGHTDOCK_TILED_WAVES=$w; done
