cd $GRAFT_REPO_ROOT
timeout 300 python tools/debug_packed.py 1ppe 1k4c 2uuy 2>&1 | grep -v "tiled\|allpairs\|amdgpu.ids\|packed1"
bash tools/ab.sh --steps 10 --warmup 2
LIGHTDOCK_PACKED_CELLS=1 timeout 100 python bench.py --cpu-seconds 0 --steps 10 | tail -1 | cut -c1-100
