"""Counts distinct table lines per in-cutoff pair for candidate layouts of the DFIRE potential
(patches of a ligand types x b receptor types x c bins), replaying the example poses through the
kernel's own tile order and 8x8 box culling on the CPU.  Usage: table_layout_sim.py [1k4c|1ppe]"""
import sys, os, numpy as np, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg, orc = ge.package(), ge.oracle()
name = sys.argv[1] if len(sys.argv) > 1 else "1k4c"
files = {"1k4c": ("lightdock_receptor_membrane.pdb", "lightdock_ligand.pdb"), "1ppe": ("lightdock_1ppe_e.pdb", "lightdock_1ppe_i.pdb")}[name]
g = os.path.join(ge.GOLDEN, name)
rec = pkg.model_from_pdb("dfire", os.path.join(g, files[0]))
lig = pkg.model_from_pdb("dfire", os.path.join(g, files[1]))
pos = orc.parse_positions(os.path.join(g, "initial_positions_0.dat"))[:, :7]
lut, steps, iface = pkg.dfire_bin_lut()
def order(m):
    o, perm = pkg.dfire_tile_layout(m["coordinates"], m["dfire_types"])     # what the scorer uses
    pad = o == 0xFFFFFFFF
    idx = np.where(pad, 0, o).astype(np.int64)
    c = m["coordinates"][idx].copy(); t = m["dfire_types"][idx].astype(np.int64)
    c[pad] = 1e9 if m is rec else -1e9
    PERMS[id(m)] = perm.astype(np.int64)
    return c, t, ~pad
PERMS = {}
rc, rt, rv = order(rec); lc0, lt, lv = order(lig)
def rotmat(q):
    w,x,y,z = q/np.linalg.norm(q)
    return np.array([[1-2*(y*y+z*z),2*(x*y-z*w),2*(x*z+y*w)],[2*(x*y+z*w),1-2*(x*x+z*z),2*(y*z-x*w)],[2*(x*z-y*w),2*(y*z+x*w),1-2*(x*x+y*y)]])
def boxes(c, v, T):
    n = len(c)//T
    cc = c.reshape(n, T, 3); vv = v.reshape(n, T)
    lo = np.where(vv[..., None], cc, np.inf).min(1); hi = np.where(vv[..., None], cc, -np.inf).max(1)
    return lo, hi
rlo, rhi = boxes(rc, rv, 8)
def bin_of(d2):
    cell = np.minimum((4*d2).astype(np.int64), 900)
    b = (lut[cell] & 31).astype(np.int64)
    b = b + (d2 >= steps[np.minimum(b+1, 20)])   # exact step
    return b
numberings = {"reference type numbers": (lt, rt), "types paired per molecule (library)": (PERMS[id(lig)][lt], PERMS[id(rec)][rt])}
shapes = [(1,1,16),(1,16,1),(1,8,2),(1,4,4),(2,8,1),(4,4,1),(2,4,2),(4,2,2),(2,2,4),(4,1,4),(8,1,2),(2,1,8),(1,2,8),(1,8,1),(2,2,2),(1,1,8),(2,4,4),(4,4,4)]
counts = {(k, sh): 0 for k in numberings for sh in shapes}
hits_total = 0; blocks_total = 0
for p in pos[::25]:
    R = rotmat(p[3:7]); l = lc0 @ R.T + p[:3]; l[~lv] = -1e9
    llo, lhi = boxes(l, lv, 8)
    gap = np.maximum(0, np.maximum(llo[:, None, :]-rhi[None, :, :], rlo[None, :, :]-lhi[:, None, :]))
    act = np.argwhere((gap**2).sum(-1) <= 225.0)
    blocks_total += len(act)
    L = l.reshape(-1, 8, 3); Rr = rc.reshape(-1, 8, 3)
    for chunk in np.array_split(act, max(1, len(act)//2000)):
        ls, rs = chunk[:, 0], chunk[:, 1]
        d2 = ((L[ls][:, :, None, :] - Rr[rs][:, None, :, :])**2).sum(-1)
        hit = d2 <= 225.0
        bins = bin_of(np.where(hit, d2, 0.0))
        hits_total += hit.sum()
        blk = np.broadcast_to(np.arange(len(chunk))[:, None, None], d2.shape)
        for k, (ltm, rtm) in numberings.items():
            lt_ = np.broadcast_to(ltm.reshape(-1, 8)[ls][:, :, None], d2.shape); rt_ = np.broadcast_to(rtm.reshape(-1, 8)[rs][:, None, :], d2.shape)
            for (a, b, c) in shapes:
                key = ((blk*200 + lt_//a)*200 + rt_//b)*32 + bins//c
                counts[(k, (a, b, c))] += len(np.unique(key[hit]))
n_p = len(pos[::25])
print(name, "blocks/pose %.0f hits/pose %.0f hits/block %.1f" % (blocks_total/n_p, hits_total/n_p, hits_total/blocks_total))
for k in numberings:
    print(k)
    for sh in sorted(shapes, key=lambda s_: (s_[0]*s_[1]*s_[2], counts[(k, s_)])):
        print("  patch = %d lig x %d rec x %d bins (%3d B): %.3f lines/hit" % (sh[0], sh[1], sh[2], 8*sh[0]*sh[1]*sh[2], counts[(k, sh)]/hits_total))
