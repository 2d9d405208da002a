cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python3 bench.py --steps 5 --warmup 2 --batch ${BATCH:-8192} --workload ${WORKLOAD:-1k4c} --cpu-seconds 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('evals/s %.0f kernel_ms %.3f'%(d['value'], d['roofline']['kernel_ms']))"; }
for cfg in ${CFGS:-"4 1" "4 2" "4 3" "8 2"}; do set -- $cfg; run LIGHTDOCK_TILED_WAVES=$1 LIGHTDOCK_TILED_SPLIT=$2; done
