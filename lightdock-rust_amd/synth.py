"""Deterministic synthetic inputs shared by tests and bench.py.

* ``dcparams()``: the DFIRE table data/DCparams is not in the reference mount, so parity and
  throughput runs use a seeded stand-in in the same format: 169*169*20 values, each what
  ``"%.9f" % (u*4 - 2)`` parses back to, u from SplitMix64(0x4C44 + k); bin 0 of every
  (type_a, type_b) row is 10.0 like the real table's repulsive core (src/dfire.rs:376).
* ``swarm()``: starting poses shaped like the shipped initial_positions_0.dat files
  (200 translations in a 10 A ball around a centre 23.4 A from the origin + random unit
  quaternions), rounded to 9 decimals like the .dat text.
* ``jitter()``: replicate real starting poses into a large batch with seeded translation noise.
"""
import numpy as np

TABLE_LEN = 169 * 169 * 20
_M64 = (1 << 64) - 1


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def dcparams(seed=0x4C44):
    with np.errstate(over="ignore"):
        k = np.arange(TABLE_LEN, dtype=np.uint64) + np.uint64(seed)
        z = _splitmix64(k)
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    text = ["%.9f" % v for v in (u * 4.0 - 2.0)]
    table = np.array([float(t) for t in text], dtype=np.float64)
    table[0::20] = 10.0
    return table


def write_dcparams(path, table=None):
    table = dcparams() if table is None else table
    with open(path, "w") as f:
        f.write("\n".join("%.9f" % v for v in table))
        f.write("\n")


def swarm(n_glowworms=200, seed=0, extra_cols=0):
    rng = np.random.Generator(np.random.PCG64(324324 + seed))
    z = 2.0 * rng.random() - 1.0
    phi = 2.0 * np.pi * rng.random()
    centre = 23.4 * np.array([np.sqrt(1 - z * z) * np.cos(phi), np.sqrt(1 - z * z) * np.sin(phi), z])
    pts = []
    while len(pts) < n_glowworms:
        p = 2.0 * rng.random(3) - 1.0
        if p @ p <= 1.0:
            pts.append(p)
    t = np.array(pts) * 10.0 + centre
    u = rng.random((n_glowworms, 3))
    q = np.stack([np.sqrt(1 - u[:, 0]) * np.sin(2 * np.pi * u[:, 1]), np.sqrt(1 - u[:, 0]) * np.cos(2 * np.pi * u[:, 1]),
                  np.sqrt(u[:, 0]) * np.sin(2 * np.pi * u[:, 2]), np.sqrt(u[:, 0]) * np.cos(2 * np.pi * u[:, 2])], axis=1)
    cols = [t, q]
    if extra_cols:
        cols.append(rng.random((n_glowworms, extra_cols)) * 9.0)
    return np.round(np.concatenate(cols, axis=1), 9)


def jitter(poses, n, seed=1, sigma=0.25):
    """n poses: the given rows cycled, translations perturbed by N(0, sigma) angstrom."""
    poses = np.asarray(poses, dtype=np.float64)
    rng = np.random.Generator(np.random.PCG64(seed))
    out = poses[np.arange(n) % poses.shape[0]].copy()
    out[:, :3] += rng.normal(0.0, sigma, size=(n, 3))
    return out
