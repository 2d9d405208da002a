"""Multi-swarm launcher: what `ant_thony.py --cores N task.list` does for the reference
(example/1czy/execution.sh:20-24: one `lightdock-rust setup.json initial_positions_i.dat steps
method` process per swarm), done as ONE batched GSO per GPU.

    python lightdock-rust_amd/launch.py <setup.json> <steps> <dfire|dna|pydock> --swarms 0-51 [--init-dir init]

Run it under `python -m torch.distributed.run --nproc-per-node N ...` to shard the swarms over
N GPUs (rank r takes swarms r, r+N, ...; no collective touches the data path).  Rank -> device: local rank r computes
on device r.  A launcher that masks the devices PER RANK (SLURM --gpus-per-task, a wrapper exporting
HIP_VISIBLE_DEVICES=$LOCAL_RANK) must also export LD_RANK_OWNS_DEVICE=1: one visible device next to local rank > 0 is
otherwise refused (a job-wide mask would put N ranks on one GPU silently).  LIGHTDOCK_DEVICE=<id> pins every rank to one
device (dry runs of the N-rank path on a 1-GPU box).  Path rules
follow src/bin/lightdock-rust.rs:158-333: PDBs next to setup.json with the "lightdock_" prefix;
swarm_<i>/, rec_nm.npy, lig_nm.npy and $LIGHTDOCK_DATA|data/DCparams relative to the CWD.  Where the
flattened rec_nm.npy / lig_nm.npy (lgd_flatten.py, example/1czy/execution.sh:10-11) are missing, the
(modes, atoms, 3) files lightdock3_setup.py writes -- lightdock_rec.nm.npy / lightdock_lig.nm.npy, in the
CWD or next to setup.json -- are read instead: same numbers, same order.
"""
import argparse
import json
import os
import sys

import numpy as np

DEFAULT_SEED = 324324  # src/constants.rs:2


def parse_swarm_list(text):
    ids = []
    for part in text.split(","):
        if "-" in part:
            a, b = part.split("-")
            ids.extend(range(int(a), int(b) + 1))
        else:
            ids.append(int(part))
    return ids


def read_positions(path, pose_len, use_anm):
    rows = np.array([[float(v) for v in line.split(" ")] for line in open(path).read().splitlines()])
    if rows.shape[1] < 7 or (use_anm and rows.shape[1] != pose_len):
        raise ValueError("%s: %d columns, expected %d" % (path, rows.shape[1], pose_len))
    return rows[:, :pose_len]


def load_nmodes(side, sim):
    """rec_nm.npy / lig_nm.npy of the CWD (src/bin/lightdock-rust.rs:231-232), else the unflattened
    lightdock_<side>.nm.npy; C-order flattening of (modes, atoms, 3) is what lgd_flatten.py does."""
    for path in ("%s_nm.npy" % side, "lightdock_%s.nm.npy" % side, os.path.join(sim, "lightdock_%s.nm.npy" % side)):
        if os.path.exists(path):
            return np.ascontiguousarray(np.load(path), dtype=np.float64).reshape(-1)
    raise FileNotFoundError("neither %s_nm.npy nor lightdock_%s.nm.npy found" % (side, side))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("setup")
    ap.add_argument("steps", type=int)
    ap.add_argument("method")
    ap.add_argument("--swarms", required=True, help="e.g. 0-51 or 0,3,7")
    ap.add_argument("--init-dir", default=None, help="directory of initial_positions_<i>.dat (default: next to setup.json)")
    args = ap.parse_args(argv)

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    import __graft_entry__ as ge
    pkg = ge.package()

    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    # one process per GPU; LIGHTDOCK_DEVICE pins every rank to one device (several ranks on one GPU for a dry run)
    local = pkg.multi.device_of_rank(local, pkg.device_count(), os.environ.get("LIGHTDOCK_DEVICE"))
    pkg.init(local)
    setup = json.load(open(args.setup))
    sim = os.path.dirname(os.path.abspath(args.setup))
    method = args.method.lower()
    use_anm = bool(setup["use_anm"])
    kw = dict(use_anm=use_anm, rec_num_anm=int(setup["anm_rec"]), lig_num_anm=int(setup["anm_lig"]))
    for side, key in (("rec", "receptor_restraints"), ("lig", "ligand_restraints")):
        r = setup.get(key)
        if r is not None:
            kw[side + "_active"], kw[side + "_passive"] = r["active"], r["passive"]
    if use_anm:
        if kw["rec_num_anm"] > 0:
            kw["rec_nmodes"] = load_nmodes("rec", sim)
        if kw["lig_num_anm"] > 0:
            kw["lig_nmodes"] = load_nmodes("lig", sim)
    if method == "dfire":
        kw["potential"] = pkg.load_dcparams(os.path.join(os.environ.get("LIGHTDOCK_DATA", "data"), "DCparams"))
    scorer = pkg.Scorer.from_pdb(method, os.path.join(sim, "lightdock_" + setup["receptor_pdb"]),
                                 os.path.join(sim, "lightdock_" + setup["ligand_pdb"]), **kw)

    swarms = parse_swarm_list(args.swarms)
    mine = [swarms[k] for k in pkg.multi.shard(len(swarms), rank, world)]
    init_dir = args.init_dir or sim
    if mine:
        pos = np.stack([read_positions(os.path.join(init_dir, "initial_positions_%d.dat" % s), scorer.pose_len, use_anm)
                        for s in mine])
        seed = int(setup["seed"]) if setup.get("seed") is not None else DEFAULT_SEED
        gso = pkg.GSO(scorer, pos, seeds=[seed] * len(mine))
        for s in mine:
            os.makedirs("swarm_%d" % s, exist_ok=True)
        done = 0
        while done < args.steps:                      # GSO::run, src/lib.rs:46-58
            nxt = 1 if done == 0 else min(args.steps, (done // 10 + 1) * 10)
            gso.run(nxt - done)
            done = nxt
            if done % 10 == 0 or done == 1:
                gso.save_many(range(len(mine)), done, ["swarm_%d" % s for s in mine])
        best = {s: float(gso.read(k)["scoring"].max()) for k, s in enumerate(mine)}
    else:
        best = {}
    print("rank %d/%d: %d swarms, best scoring per swarm: %s" % (rank, world, len(mine), best))
    return 0


if __name__ == "__main__":
    sys.exit(main())
