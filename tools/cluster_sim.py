"""How many 8 x 8 blocks the box culling lets through for DIFFERENT ATOM ORDERS (the library's
median splits + window swaps against globally refined subtiles / rotation-invariant ligand costs),
replaying example poses on the CPU like culling_sim.py.  Usage: cluster_sim.py [1k4c|1ppe|2uuy]"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg, orc = ge.package(), ge.oracle()
name = sys.argv[1] if len(sys.argv) > 1 else "1k4c"
files = {"1k4c": ("lightdock_receptor_membrane.pdb", "lightdock_ligand.pdb"), "1ppe": ("lightdock_1ppe_e.pdb", "lightdock_1ppe_i.pdb"),
         "2uuy": ("lightdock_2UUY_rec.pdb", "lightdock_2UUY_lig.pdb")}[name]
g = os.path.join(ge.GOLDEN, name)
rec = pkg.model_from_pdb("dfire", os.path.join(g, files[0]))
lig = pkg.model_from_pdb("dfire", os.path.join(g, files[1]))
pos = orc.parse_positions(os.path.join(g, "initial_positions_0.dat"))[:, :7]
GROW = 36.0

def lib_order(m):
    o, _ = pkg.dfire_tile_layout(m["coordinates"], m["dfire_types"])
    return o.astype(np.int64)

def rotmat(q):
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1-2*(y*y+z*z), 2*(x*y-z*w), 2*(x*z+y*w)], [2*(x*y+z*w), 1-2*(x*x+z*z), 2*(y*z-x*w)], [2*(x*z-y*w), 2*(y*z+x*w), 1-2*(x*x+y*y)]])

def boxes(c, v, T):
    n = len(c) // T
    cc = c.reshape(n, T, 3); vv = v.reshape(n, T)
    return np.where(vv[..., None], cc, np.inf).min(1), np.where(vv[..., None], cc, -np.inf).max(1)

def evaluate(ro, lo, label, sample):
    PAD = 0xFFFFFFFF
    def coords(m, o, far):
        pad = o == PAD
        c = m["coordinates"][np.where(pad, 0, o)].copy(); c[pad] = far
        return c, ~pad
    rc, rv = coords(rec, ro, 1e9); lc0, lv = coords(lig, lo, -1e9)
    nb = nt = ne = 0
    for p in sample:
        R = rotmat(p[3:7]); l = lc0 @ R.T + p[:3]; l[~lv] = -1e9
        tl, th = boxes(l, lv, 64); rl, rh = boxes(rc, rv, 64)
        gap = np.maximum(0, np.maximum(tl[:, None]-rh[None], rl[None]-th[:, None]))
        tact = (gap**2).sum(-1) <= 225.0
        nt += tact.sum()
        ll, lh = boxes(l, lv, 8); rl2, rh2 = boxes(rc, rv, 8)
        gap = np.maximum(0, np.maximum(ll[:, None]-rh2[None], rl2[None]-lh[:, None]))
        act = (gap**2).sum(-1) <= 225.0
        act &= np.repeat(np.repeat(tact, 8, axis=0), 8, axis=1)
        nb += act.sum()
    n = len(sample)
    print("%-60s blocks/pose %7.0f  tile pairs/pose %5.0f" % (label, nb/n, nt/n))
    return nb / n

# ---- costs of a group of 8 atoms -------------------------------------------------------------
def cost_aabb(P):          # P [..., 8, 3]
    e = P.max(-2) - P.min(-2)
    return np.prod(e + GROW, -1)

rng = np.random.default_rng(1)
def rand_rots(k):
    q = rng.normal(size=(k, 4)); return np.stack([rotmat(x) for x in q])
ROTS = rand_rots(12)
def cost_rotavg(P):        # mean over fixed random rotations of the AABB cost: what a posed ligand subtile costs
    Q = np.einsum('kab,...ib->k...ia', ROTS, P)
    e = Q.max(-2) - Q.min(-2)
    return np.prod(e + GROW, -1).mean(0)

def refine_global(xyz, order, cost, knn=10, passes=4):
    """Swap atoms between a subtile and its knn nearest subtiles (by centroid) while the summed cost falls."""
    PAD = 0xFFFFFFFF
    o = order.copy()
    ns = len(o) // 8
    full = np.array([np.all(o[8*s:8*s+8] != PAD) for s in range(ns)])
    for ps in range(passes):
        S = o.reshape(ns, 8)
        cen = np.array([xyz[S[s][S[s] != PAD]].mean(0) if np.any(S[s] != PAD) else np.full(3, 1e9) for s in range(ns)])
        cst = np.array([cost(xyz[S[s]]) if full[s] else 0.0 for s in range(ns)])
        improved = 0
        for a in range(ns):
            if not full[a]: continue
            d = ((cen - cen[a])**2).sum(1); d[~full] = 1e300; d[a] = 1e300
            for b in np.argsort(d)[:knn]:
                if b < a or not full[b]: continue
                for i in range(8):
                    # all 8 swaps of S[a][i] with S[b][j] at once
                    A = np.repeat(xyz[S[a]][None], 8, 0); B = np.repeat(xyz[S[b]][None], 8, 0)
                    ai = xyz[S[a][i]].copy()
                    A[:, i] = xyz[S[b]]                 # variant j: a's atom i replaced by b's atom j
                    B[np.arange(8), np.arange(8)] = ai  # and b's atom j by a's atom i
                    tot = cost(A) + cost(B)
                    j = int(np.argmin(tot))
                    if tot[j] < cst[a] + cst[b] - 1e-9:
                        S[a][i], S[b][j] = S[b][j], S[a][i]
                        cst[a] = cost(xyz[S[a]]); cst[b] = cost(xyz[S[b]])
                        improved += 1
        o = S.reshape(-1)
        if not improved: break
    return o

def retile(xyz, order):
    """Group subtiles into tiles of 8 by median splits of the subtile centroids (unit = 8 subtiles)."""
    PAD = 0xFFFFFFFF
    ns = len(order) // 8
    S = order.reshape(ns, 8)
    fullm = np.array([np.all(S[s] != PAD) for s in range(ns)])
    fulls = np.where(fullm)[0]; rest = np.where(~fullm)[0]
    cen = np.array([xyz[S[s]].mean(0) for s in fulls])
    ids = list(range(len(fulls)))
    def split(ids):
        n = len(ids)
        if n <= 8: return ids
        c = cen[ids]; ax = int(np.argmax(c.max(0) - c.min(0)))
        ids = [ids[k] for k in np.argsort(c[:, ax], kind='stable')]
        half = ((n // 2 + 4) // 8) * 8
        half = max(8, half)
        if half >= n: half = (n - 1) // 8 * 8
        return split(ids[:half]) + split(ids[half:])
    ids = split(ids)
    out = np.concatenate([S[fulls[i]] for i in ids] + [S[r] for r in rest])
    return out

if __name__ == "__main__":
    sample = pos[::10]
    ro, lo = lib_order(rec), lib_order(lig)
    base = evaluate(ro, lo, "library order", sample)
    rx, lx = rec["coordinates"], lig["coordinates"]
    import time
    t = time.time(); ro2 = refine_global(rx, ro, cost_aabb); print("rec refine %.0f s" % (time.time() - t))
    evaluate(ro2, lo, "receptor: global swaps (aabb cost), tiles kept", sample)
    ro3 = retile(rx, ro2)
    evaluate(ro3, lo, "receptor: global swaps + retiled", sample)
    t = time.time(); lo2 = refine_global(lx, lo, cost_rotavg); print("lig refine %.0f s" % (time.time() - t))
    evaluate(ro, lo2, "ligand: global swaps (rotation-averaged cost), tiles kept", sample)
    lo3 = retile(lx, lo2)
    evaluate(ro, lo3, "ligand: global swaps + retiled", sample)
    evaluate(ro3, lo3, "both", sample)
    np.save("/tmp/cluster_sim_%s_rec.npy" % name, ro3); np.save("/tmp/cluster_sim_%s_lig.npy" % name, lo3)
