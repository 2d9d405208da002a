"""N > 1 plumbing on CPU: world-size-2 gloo processes shard swarms with no data-path
collective, and the merged result equals the single-process result.  The work function here
is the CPU oracle's GSO (tests may use the oracle; the GPU path needs an MI355X)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT, GOLDEN


def test_shard_is_a_partition(pkg):
    from lightdock_rust_amd import multi
    for n, w in ((1024, 8), (10, 4), (3, 8), (0, 2)):
        owned = [multi.shard(n, r, w) for r in range(w)]
        flat = sorted(i for o in owned for i in o)
        assert flat == list(range(n))
        sizes = [len(o) for o in owned]
        assert max(sizes) - min(sizes) <= 1


def _oracle_swarm_job(orc, pkg, n_swarms, steps):
    d = os.path.join(GOLDEN, "unit", "1azp")
    scorer = orc.Scorer("dna", os.path.join(d, "1azp_receptor.pdb"), os.path.join(d, "1azp_ligand.pdb"))

    class Batch:  # a stand-in with the shape of pkg.GSO over several swarms
        def __init__(self, ids):
            self.swarms = [orc.GSO(scorer, pkg.synth.swarm(6, seed=s), seed=324324) for s in ids]

        def step(self):
            for g in self.swarms:
                g.step()

    def summarize(gso, k, s):
        st = gso.swarms[k].state()
        return (s, float(st["scoring"].min()), st["n_neighbors"].tolist())

    return Batch, summarize


def _worker(rank, world_size, port, out_path):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import __graft_entry__ as ge
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    pkg, orc = ge.package(), ge.oracle()
    from lightdock_rust_amd import multi
    Batch, summarize = _oracle_swarm_job(orc, pkg, 5, 2)
    holder = {}

    def job():
        holder["res"] = multi.run_swarms(Batch, 5, 2, dist, summarize)

    elapsed = multi.timed_region(job, dist)
    # bench.py's whole-job evaluation count: a plain sum over the ranks
    assert multi.sum_over_ranks(10 * (rank + 1), dist) == 10 * sum(range(1, world_size + 1))
    if rank == 0:
        np.save(out_path, np.array([elapsed] + [r[1] for r in holder["res"]]))
        assert [r[0] for r in holder["res"]] == list(range(5))
    else:
        assert holder["res"] is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_equal_one_process(pkg, orc, tmp_path):
    from lightdock_rust_amd import multi
    Batch, summarize = _oracle_swarm_job(orc, pkg, 5, 2)
    single = multi.run_swarms(Batch, 5, 2, None, summarize)
    out = str(tmp_path / "rank0.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    assert got[0] > 0.0
    assert np.array_equal(got[1:], np.array([r[1] for r in single]))


def test_gather_detects_missing_and_duplicate_swarms(pkg):
    from lightdock_rust_amd import multi
    with pytest.raises(RuntimeError, match="not evaluated"):
        multi.gather_by_swarm({0: "a"}, 2, None)
    assert multi.gather_by_swarm({1: "b", 0: "a"}, 2, None) == ["a", "b"]


def test_rank_to_device_mapping(pkg, monkeypatch):
    """bench.py and launch.py map ranks to devices through one function: one process per GPU, a rank that sees only
    its own GPU takes device 0, a forced device pins all ranks (dry runs), and more ranks than devices without that
    is an error -- never N ranks silently sharing one GPU."""
    from lightdock_rust_amd import multi
    for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(k, raising=False)
    assert [multi.device_of_rank(r, 8) for r in range(8)] == list(range(8))
    assert multi.device_of_rank(0, 1) == 0
    with pytest.raises(ValueError):
        multi.device_of_rank(1, 1)                       # two ranks, one visible device, nothing says it is the rank's own
    with pytest.raises(ValueError):
        multi.device_of_rank(8, 8)
    with pytest.raises(ValueError):
        multi.device_of_rank(0, 0)
    assert [multi.device_of_rank(r, 1, forced="0") for r in range(4)] == [0, 0, 0, 0]
    assert multi.device_of_rank(3, 8, forced="5") == 5
    with pytest.raises(ValueError):
        multi.device_of_rank(0, 2, forced="2")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")        # a job-wide mask: still N ranks on one GPU
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "4")
    with pytest.raises(ValueError):
        multi.device_of_rank(3, 1)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "3")        # per-rank visibility, and the launcher says so: the one device it sees is its own
    monkeypatch.setenv("LD_RANK_OWNS_DEVICE", "1")
    assert multi.device_of_rank(3, 1) == 0
    monkeypatch.delenv("LD_RANK_OWNS_DEVICE")
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "1")            # a job of one local rank
    assert multi.device_of_rank(0, 1) == 0
    with pytest.raises(ValueError):
        multi.device_of_rank(1, 1)                       # ... which can only be local rank 0


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without a launcher starts two ranks itself (the driver's scaling run would otherwise
    report 1-GPU numbers at every N).  Here (no GPU): the parent refuses N > visible devices; past that check both
    children start with the torchrun environment and stop at "needs an MI355X", and the parent reports the failure."""
    import subprocess
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("LD_BENCH_FORCE_DEVICE", None)
    bench = os.path.join(ROOT, "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "device(s) visible" in r.stderr
    env["LD_BENCH_NO_GPU_CHECK"] = "1"
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert r.stderr.count("needs an MI355X") == 2           # both ranks ran
    assert "rank(s) failed: 0 (exit 1), 1 (exit 1)" in r.stderr
    assert r.stdout.strip() == ""                            # and no JSON line pretends otherwise
