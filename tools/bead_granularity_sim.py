"""VERDICT r04 item 1a, the cost model: how many pair slots a 1k4c pose walks if the membrane beads (453 atoms of one type, 6.4 A apart:
their 8-atom subtiles are ~20 A across) are culled at their own granularity -- groups of 8 (today), 4, 2, 1 beads with their own
boxes, 8 ligand atoms x g beads a block.  Replays the example poses through the library's tile order on the CPU; the block test
is the kernels' (gap between two boxes <= 15 A).  Usage: python tools/bead_granularity_sim.py"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg, orc = ge.package(), ge.oracle()
g = os.path.join(ge.GOLDEN, "1k4c")
rec = pkg.model_from_pdb("dfire", os.path.join(g, "lightdock_receptor_membrane.pdb"))
lig = pkg.model_from_pdb("dfire", os.path.join(g, "lightdock_ligand.pdb"))
pos = orc.parse_positions(os.path.join(g, "initial_positions_0.dat"))[:, :7]
def order(m):
    o, perm = pkg.dfire_tile_layout(m["coordinates"], m["dfire_types"])
    pad = o == 0xFFFFFFFF
    idx = np.where(pad, 0, o).astype(np.int64)
    return m["coordinates"][idx].copy(), ~pad, np.asarray(m["dfire_types"])[idx]
rc, rv, rt = order(rec); lc0, lv, _ = order(lig)
bead_type = np.bincount(rt[rv]).argmax() if False else None
# beads: the atoms named BJ of residue MMB -- in the model they are the `membrane` list; fall back on the most isolated type
mem = set(int(i) for i in rec.get("membrane", []))
o, _ = pkg.dfire_tile_layout(rec["coordinates"], rec["dfire_types"])
is_bead = np.array([(int(a) in mem) if a != 0xFFFFFFFF else False for a in o])
print("receptor atoms %d, beads %d, subtiles with a bead %d of %d" % (rv.sum(), is_bead.sum(), (is_bead.reshape(-1, 8).any(1)).sum(), len(rv) // 8))
def rotmat(q):
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1-2*(y*y+z*z), 2*(x*y-z*w), 2*(x*z+y*w)], [2*(x*y+z*w), 1-2*(x*x+z*z), 2*(y*z-x*w)], [2*(x*z-y*w), 2*(y*z+x*w), 1-2*(x*x+y*y)]])
def boxes_of(groups, c):   # groups: list of index arrays
    lo = np.array([c[ix].min(0) for ix in groups]); hi = np.array([c[ix].max(0) for ix in groups])
    return lo, hi
def near(alo, ahi, blo, bhi):
    gap = np.maximum(0, np.maximum(alo[:, None] - bhi[None], blo[None] - ahi[:, None]))
    return (gap ** 2).sum(-1) <= 225.0
lig_groups = [np.arange(s * 8, s * 8 + 8)[lv[s * 8:s * 8 + 8]] for s in range(len(lv) // 8)]
lig_groups = [ix for ix in lig_groups if len(ix)]
results = {}
for gsize in (8, 4, 2, 1):
    rec_groups = []   # (indices, slots per ligand subtile = 8 x len)
    for s in range(len(rv) // 8):
        ix = np.arange(s * 8, s * 8 + 8)[rv[s * 8:s * 8 + 8]]
        if not len(ix):
            continue
        b = ix[is_bead[ix]]; p = ix[~is_bead[ix]]
        if gsize == 8 or not len(b):
            rec_groups.append(ix)
            continue
        if len(p):
            rec_groups.append(p)
        for k in range(0, len(b), gsize):
            rec_groups.append(b[k:k + gsize])
    width = np.array([8 if gsize == 8 else (len(ix) if is_bead[ix].all() else 8) for ix in rec_groups])   # a protein (part-)subtile still costs a full 8-wide block
    rlo, rhi = boxes_of(rec_groups, rc)
    slots = []
    for p in pos[::10]:
        R = rotmat(p[3:7]); l = lc0 @ R.T + p[:3]
        llo, lhi = boxes_of(lig_groups, l)
        hit = near(llo, lhi, rlo, rhi)
        slots.append((hit * (8 * width)[None]).sum())
    results[gsize] = np.mean(slots)
    print("bead groups of %d: %8.0f pair slots per pose (%.3f of today's), %d receptor groups" % (gsize, results[gsize], results[gsize] / results[8], len(rec_groups)))
