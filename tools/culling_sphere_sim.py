"""Round 6: would ligand subtile SPHERES (rotation invariant: posed as one point each, no per-atom posing and no box reductions in the
culling kernel) let through as few 8 x 8 blocks as the boxes of the posed atoms do?  Replays example poses through the library's tile
order on the CPU.  Usage: culling_sphere_sim.py [1k4c|1ppe]"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg, orc = ge.package(), ge.oracle()
name = sys.argv[1] if len(sys.argv) > 1 else "1k4c"
files = {"1k4c": ("lightdock_receptor_membrane.pdb", "lightdock_ligand.pdb"), "1ppe": ("lightdock_1ppe_e.pdb", "lightdock_1ppe_i.pdb")}[name]
g = os.path.join(ge.GOLDEN, name)
rec = pkg.model_from_pdb("dfire", os.path.join(g, files[0]))
lig = pkg.model_from_pdb("dfire", os.path.join(g, files[1]))
pos = orc.parse_positions(os.path.join(g, "initial_positions_0.dat"))[:, :7]
def order(m, far):
    o, perm = pkg.dfire_tile_layout(m["coordinates"], m["dfire_types"])
    pad = o == 0xFFFFFFFF
    c = m["coordinates"][np.where(pad, 0, o).astype(np.int64)].copy()
    c[pad] = far
    return c, ~pad
rc, rv = order(rec, 1e9); lc0, lv = order(lig, -1e9)
def rotmat(q):
    w,x,y,z = q/np.linalg.norm(q)
    return np.array([[1-2*(y*y+z*z),2*(x*y-z*w),2*(x*z+y*w)],[2*(x*y+z*w),1-2*(x*x+z*z),2*(y*z-x*w)],[2*(x*z-y*w),2*(y*z+x*w),1-2*(x*x+y*y)]])
def boxes(c, v, T):
    n = len(c)//T
    cc = c.reshape(n, T, 3); vv = v.reshape(n, T)
    return np.where(vv[..., None], cc, np.inf).min(1), np.where(vv[..., None], cc, -np.inf).max(1)
def ritter(points):
    """a small enclosing sphere (Ritter's, then shrunk by a few passes of the 'move towards the farthest' iteration)"""
    p = points
    c = p.mean(0)
    for _ in range(60):
        d = np.linalg.norm(p - c, axis=1); k = d.argmax()
        c = c + (p[k] - c) * 0.05
    return c, np.linalg.norm(p - c, axis=1).max()
def spheres(c, v, T):
    n = len(c)//T
    ctr = np.zeros((n, 3)); r = np.zeros(n); ok = np.zeros(n, bool)
    for k in range(n):
        pts = c[k*T:(k+1)*T][v[k*T:(k+1)*T]]
        if len(pts):
            ctr[k], r[k] = ritter(pts); ok[k] = True
    return ctr, r, ok
lsc, lsr, lsok = spheres(lc0, lv, 8)      # ligand subtiles, local frame
ltc, ltr, ltok = spheres(lc0, lv, 64)     # ligand tiles
rl8, rh8 = boxes(rc, rv, 8); rl64, rh64 = boxes(rc, rv, 64)
print("ligand subtile sphere radius: mean %.2f max %.2f A; tile: mean %.2f max %.2f" % (lsr[lsok].mean(), lsr.max(), ltr[ltok].mean(), ltr.max()))
def gap_point_box(pt, lo, hi):   # [n,3] x [m,3] -> [n,m]
    g = np.maximum(0, np.maximum(lo[None] - pt[:, None], pt[:, None] - hi[None]))
    return np.sqrt((g**2).sum(-1))
tot = dict(box=0, sph=0, both=0, tp_box=0, tp_sph=0)
sample = pos[::10]
for p in sample:
    R = rotmat(p[3:7]); l = lc0 @ R.T + p[:3]; l[~lv] = -1e9
    # current: boxes of the posed atoms
    tl, th = boxes(l, lv, 64)
    gap = np.maximum(0, np.maximum(tl[:, None]-rh64[None], rl64[None]-th[:, None]))
    tact = (gap**2).sum(-1) <= 225.0
    ll, lh = boxes(l, lv, 8)
    gap = np.maximum(0, np.maximum(ll[:, None]-rh8[None], rl8[None]-lh[:, None]))
    act_box = ((gap**2).sum(-1) <= 225.0) & np.repeat(np.repeat(tact, 8, axis=0), 8, axis=1)
    # spheres: posed centres
    tc = ltc @ R.T + p[:3]
    tact_s = (gap_point_box(tc, rl64, rh64) <= 15.0 + ltr[:, None]) & ltok[:, None]
    sc = lsc @ R.T + p[:3]
    act_sph = (gap_point_box(sc, rl8, rh8) <= 15.0 + lsr[:, None]) & lsok[:, None] & np.repeat(np.repeat(tact_s, 8, axis=0), 8, axis=1)
    tot["box"] += int(act_box.sum()); tot["sph"] += int(act_sph.sum()); tot["both"] += int((act_box & act_sph).sum())
    tot["tp_box"] += int(tact.sum()); tot["tp_sph"] += int(tact_s.sum())
n = len(sample)
print(name, "per pose: tile pairs box %.0f sphere %.0f | 8x8 blocks: boxes of posed atoms %.0f | subtile spheres %.0f (%+.1f %%) | both %.0f (%+.1f %%)" % (
    tot["tp_box"]/n, tot["tp_sph"]/n, tot["box"]/n, tot["sph"]/n, 100.0*(tot["sph"]/tot["box"]-1), tot["both"]/n, 100.0*(tot["both"]/tot["box"]-1)))
