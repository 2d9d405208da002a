"""How many DISTINCT table lines the in-cutoff pairs of one pose (and of groups of poses of a swarm) touch, against
the lines the kernel fills (0.49 per pair): the reuse a block-major order over poses could tap.  CPU only; 1k4c."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
from scipy.spatial import cKDTree
pkg, orc = ge.package(), ge.oracle()
g = os.path.join(ge.GOLDEN, "1k4c")
rec = pkg.model_from_pdb("dfire", os.path.join(g, "lightdock_receptor_membrane.pdb"))
lig = pkg.model_from_pdb("dfire", os.path.join(g, "lightdock_ligand.pdb"))
pos = orc.parse_positions(os.path.join(g, "initial_positions_0.dat"))[:, :7]
lut, steps, iface = pkg.dfire_bin_lut()
_, permr = pkg.dfire_tile_layout(rec["coordinates"], rec["dfire_types"])
_, perml = pkg.dfire_tile_layout(lig["coordinates"], lig["dfire_types"])
rt = permr.astype(np.int64)[rec["dfire_types"].astype(np.int64)]; lt = perml.astype(np.int64)[lig["dfire_types"].astype(np.int64)]
def rotmat(q):
    w,x,y,z = q/np.linalg.norm(q)
    return np.array([[1-2*(y*y+z*z),2*(x*y-z*w),2*(x*z+y*w)],[2*(x*y+z*w),1-2*(x*x+z*z),2*(y*z-x*w)],[2*(x*z-y*w),2*(y*z+x*w),1-2*(x*x+y*y)]])
tree = cKDTree(rec["coordinates"])
def lines_of(p):
    R = rotmat(p[3:7]); l = lig["coordinates"] @ R.T + p[:3]
    pairs = cKDTree(l).query_ball_tree(tree, 15.0)
    li = np.concatenate([np.full(len(v), i) for i, v in enumerate(pairs) if v]); ri = np.concatenate([np.array(v) for v in pairs if v])
    d2 = ((l[li] - rec["coordinates"][ri])**2).sum(1)
    cell = np.minimum((4*d2).astype(np.int64), 900); b = (lut[cell] & 31).astype(np.int64); b = b + (d2 >= steps[np.minimum(b+1, 20)])
    key = ((lt[li]//2)*200 + rt[ri]//2)*8 + b//4
    return key
sets = [lines_of(p) for p in pos[:64]]
hits = sum(len(k) for k in sets); uniq_each = sum(len(np.unique(k)) for k in sets)
print("per pose: hits %.0f, distinct lines %.0f (%.3f per hit)" % (hits/64, uniq_each/64, uniq_each/hits))
for P in (2, 4, 8, 16, 64):
    tot = 0
    for s in range(0, 64, P):
        tot += len(np.unique(np.concatenate(sets[s:s+P])))
    print("groups of %2d poses: distinct lines per hit %.3f" % (P, tot/hits))
