cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python3 bench.py --steps 5 --warmup 2 --batch 8192 --cpu-seconds 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('evals/s %.0f kernel_ms %.3f'%(d['value'], d['roofline']['kernel_ms']))"; }
for pad in 0 9000 15000 23000 36000 63000; do run LIGHTDOCK_TILED_WAVES=4 LIGHTDOCK_TILED_LDS_PAD=$pad; done
