"""How many atom pairs the box culling lets through for different block shapes, replaying the
example poses through the library's tile order on the CPU.  A block = a ligand atoms x b receptor
atoms (consecutive atoms of the tile order); a block is evaluated when the gap between its two
bounding boxes is within the cutoff.  Usage: culling_sim.py [1k4c|1ppe]"""
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
shapes = [(8, 8), (4, 8), (8, 4), (4, 4), (8, 16), (16, 8), (2, 8), (8, 2), (4, 16), (2, 16)]
stats = {s: [0, 0] for s in shapes}
hits = 0; tilepairs = 0
sample = pos[::20]
for p in sample:
    R = rotmat(p[3:7]); l = lc0 @ R.T + p[:3]; l[~lv] = -1e9
    # tile level (64 x 64) first, like the kernel
    tl, th = boxes(l, lv, 64); rl, rh = boxes(rc, rv, 64)
    gap = np.maximum(0, np.maximum(tl[:, None]-rh[None], rl[None]-th[:, None]))
    tact = (gap**2).sum(-1) <= 225.0
    tilepairs += tact.sum()
    from scipy.spatial import cKDTree
    hits += cKDTree(l[lv]).count_neighbors(cKDTree(rc[rv]), 15.0)
    for (a, b) in shapes:
        ll, lh = boxes(l, lv, a); rl2, rh2 = boxes(rc, rv, b)
        gap = np.maximum(0, np.maximum(ll[:, None]-rh2[None], rl2[None]-lh[:, None]))
        act = (gap**2).sum(-1) <= 225.0
        # only inside surviving tile pairs
        tmask = np.repeat(np.repeat(tact, 64//a, axis=0), 64//b, axis=1)
        n = int((act & tmask).sum())
        stats[(a, b)][0] += n; stats[(a, b)][1] += n*a*b
n = len(sample)
print(name, "in-cutoff pairs/pose %.0f, surviving 64x64 tile pairs/pose %.0f" % (hits/n, tilepairs/n))
for s in shapes:
    b, pr = stats[s]
    print("  block %2d lig x %2d rec: %7.0f blocks/pose, %8.0f pair tests/pose, hit rate %.1f %%, box tests/pose %.0f" % (s[0], s[1], b/n, pr/n, 100*hits/pr, tilepairs/n*(64//s[0])*(64//s[1])))

# second-level tests on the 8 x 8 blocks that pass the box test: what share do they remove?
from itertools import product
diag4 = np.array([[1, 1, 1], [1, 1, -1], [1, -1, 1], [-1, 1, 1]]) / np.sqrt(3.0)
face6 = np.array([[1, 1, 0], [1, -1, 0], [1, 0, 1], [1, 0, -1], [0, 1, 1], [0, 1, -1]]) / np.sqrt(2.0)
def proj_boxes(c, v, T, axes):
    n = len(c)//T
    pr = (c @ axes.T).reshape(n, T, -1); vv = v.reshape(n, T)
    return np.where(vv[..., None], pr, np.inf).min(1), np.where(vv[..., None], pr, -np.inf).max(1)
def spheres(c, v, T):
    n = len(c)//T
    cc = c.reshape(n, T, 3); vv = v.reshape(n, T)
    lo = np.where(vv[..., None], cc, np.inf).min(1); hi = np.where(vv[..., None], cc, -np.inf).max(1)
    ctr = 0.5*(lo+hi)
    r = np.where(vv, np.linalg.norm(cc - ctr[:, None], axis=-1), 0).max(1)
    return ctr, r
tot = dict(aabb=0, sphere=0, d4=0, d10=0, sphere_d4=0, nonempty=0)
for p in sample:
    R = rotmat(p[3:7]); l = lc0 @ R.T + p[:3]; l[~lv] = -1e9
    ll, lh = boxes(l, lv, 8); rl2, rh2 = boxes(rc, rv, 8)
    gap = np.maximum(0, np.maximum(ll[:, None]-rh2[None], rl2[None]-lh[:, None]))
    act = (gap**2).sum(-1) <= 225.0
    lvv = lv.reshape(-1, 8).any(1); rvv = rv.reshape(-1, 8).any(1)
    act &= lvv[:, None] & rvv[None]
    lc_, lr_ = spheres(l, lv, 8); rc_, rr_ = spheres(rc, rv, 8)
    dc = np.linalg.norm(lc_[:, None]-rc_[None], axis=-1)
    sph = dc - lr_[:, None] - rr_[None] <= 15.0
    def kdop(axes):
        a_lo, a_hi = proj_boxes(l, lv, 8, axes); b_lo, b_hi = proj_boxes(rc, rv, 8, axes)
        g = np.maximum(a_lo[:, None]-b_hi[None], b_lo[None]-a_hi[:, None]).max(-1)
        return g <= 15.0
    k4 = kdop(diag4); k10 = kdop(np.vstack([diag4, face6]))
    # truly non-empty blocks
    idx = np.argwhere(act)
    L8 = l.reshape(-1, 8, 3); R8 = rc.reshape(-1, 8, 3)
    ne = 0
    for ch in np.array_split(idx, max(1, len(idx)//4000)):
        d2 = ((L8[ch[:, 0]][:, :, None] - R8[ch[:, 1]][:, None])**2).sum(-1)
        ne += int((d2.reshape(len(ch), -1).min(1) <= 225.0).sum())
    tot["aabb"] += int(act.sum()); tot["sphere"] += int((act & sph).sum()); tot["d4"] += int((act & k4).sum())
    tot["d10"] += int((act & k10).sum()); tot["sphere_d4"] += int((act & sph & k4).sum()); tot["nonempty"] += ne
print("8x8 blocks per pose after: box test %.0f | + spheres %.0f | + 4 cube diagonals %.0f | + 10 diagonals %.0f | + spheres + 4 diagonals %.0f | holding a pair in cutoff %.0f"
      % tuple(tot[k]/n for k in ("aabb", "sphere", "d4", "d10", "sphere_d4", "nonempty")))
