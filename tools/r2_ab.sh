cd $GRAFT_REPO_ROOT
timeout 300 python tools/debug_packed.py 1ppe 1k4c 2uuy 2>&1 | grep -v "tiled\|allpairs\|amdgpu.ids\|eps50"
for i in 1 2; do timeout 100 python bench.py --cpu-seconds 0 --steps 10 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f evals/s kernel %.3f ms' % (d['value'], d['roofline']['kernel_ms']))"; done
timeout 100 python bench.py --workload 1ppe --cpu-seconds 0 2>&1 | tail -1 | cut -c1-120
timeout 200 python bench.py --workload gso-1ppe --cpu-seconds 0 --steps 10 2>&1 | tail -1 | cut -c1-120
