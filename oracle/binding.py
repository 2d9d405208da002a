"""ctypes view of the CPU oracle (oracle/libld_oracle.so).

TEST INFRASTRUCTURE: importable only from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libld_oracle.so")
CLI_PATH = os.path.join(HERE, "ld_oracle_cli")
TABLE_LEN = 169 * 169 * 20
METHODS = {"dfire": 0, "dna": 1, "pydock": 2}

_lib = None


def build():
    subprocess.run(["make", "-C", HERE, "-s"], check=True, capture_output=True)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    L = C.CDLL(LIB_PATH)
    vp, dbl, sz = C.c_void_p, C.c_double, C.c_size_t
    L.orc_last_error.restype = C.c_char_p
    for name in ("orc_q_dot", "orc_q_norm2", "orc_q_norm", "orc_q_distance", "orc_rng_f64", "orc_scorer_energy",
                 "orc_scorer_energy_ex"):
        getattr(L, name).restype = dbl
    L.orc_q_lerp.argtypes = [vp, vp, dbl, vp]
    L.orc_q_slerp.argtypes = [vp, vp, dbl, vp]
    L.orc_rng_new.restype = vp
    L.orc_rng_new.argtypes = [C.c_uint64]
    L.orc_rng_free.argtypes = [vp]
    L.orc_rng_next_u64.restype = C.c_uint64
    L.orc_rng_next_u64.argtypes = [vp]
    L.orc_rng_f64.argtypes = [vp]
    L.orc_q_random.argtypes = [vp, vp]
    L.orc_load_dcparams.argtypes = [C.c_char_p, vp]
    L.orc_dfire_bin.argtypes = [dbl]
    L.orc_scorer_new.restype = vp
    L.orc_scorer_new.argtypes = [C.c_int, C.c_char_p, C.c_char_p, vp, C.c_int, vp, C.c_int, vp, sz, C.c_int,
                                 vp, C.c_int, vp, C.c_int, vp, sz, C.c_int, C.c_int, vp]
    L.orc_scorer_free.argtypes = [vp]
    L.orc_scorer_energy.argtypes = [vp, vp, vp, vp, vp]
    L.orc_scorer_energy_ex.argtypes = [vp, vp, vp, vp, vp, vp]
    L.orc_scorer_energy_rows_mt.argtypes = [vp, vp, sz, sz, C.c_int, vp]
    L.orc_scorer_num_atoms.restype = sz
    L.orc_scorer_num_atoms.argtypes = [vp, C.c_int]
    for name in ("orc_scorer_coordinates", "orc_scorer_dfire_types", "orc_scorer_ele_charges", "orc_scorer_vdw_charges",
                 "orc_scorer_vdw_radii", "orc_scorer_membrane", "orc_scorer_restraint_offsets", "orc_scorer_restraint_atoms"):
        getattr(L, name).restype = vp
        getattr(L, name).argtypes = [vp, C.c_int]
    for name in ("orc_scorer_num_membrane", "orc_scorer_num_restraint_groups"):
        getattr(L, name).restype = sz
        getattr(L, name).argtypes = [vp, C.c_int]
    L.orc_gso_new.restype = vp
    L.orc_gso_new.argtypes = [vp, C.c_int, C.c_int, C.c_uint64, vp, C.c_int, C.c_int, C.c_int]
    L.orc_gso_free.argtypes = [vp]
    L.orc_gso_step.argtypes = [vp]
    L.orc_gso_save.argtypes = [vp, C.c_int, C.c_char_p]
    L.orc_gso_run.argtypes = [vp, C.c_int, C.c_char_p]
    L.orc_gso_num_evals.restype = C.c_uint64
    L.orc_gso_num_evals.argtypes = [vp]
    L.orc_gso_state.argtypes = [vp] * 8
    L.orc_gso_neighbors.argtypes = [vp, C.c_int, vp, C.c_int]
    L.orc_parse_positions.restype = vp
    L.orc_parse_positions.argtypes = [C.c_char_p, vp, vp]
    L.orc_read_npy_f64.restype = vp
    L.orc_read_npy_f64.argtypes = [C.c_char_p, vp]
    L.orc_free.argtypes = [vp]
    L.orc_cli_main.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
    _lib = L
    return L


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _strs(items):
    items = [s.encode() for s in (items or [])]
    arr = (C.c_char_p * max(1, len(items)))(*items)
    return C.cast(arr, C.c_void_p), len(items), arr


# ---- quaternions (w, x, y, z) ------------------------------------------------------
def _q_unary(name, q):
    out = np.empty(4)
    getattr(lib(), name)(_p(_f64(q)), _p(out))
    return out


def q_conjugate(q):
    return _q_unary("orc_q_conjugate", q)


def q_inverse(q):
    return _q_unary("orc_q_inverse", q)


def q_normalize(q):
    q = _f64(q).copy()
    lib().orc_q_normalize(_p(q))
    return q


def q_dot(a, b):
    return lib().orc_q_dot(_p(_f64(a)), _p(_f64(b)))


def q_norm(q):
    return lib().orc_q_norm(_p(_f64(q)))


def q_norm2(q):
    return lib().orc_q_norm2(_p(_f64(q)))


def q_distance(a, b):
    return lib().orc_q_distance(_p(_f64(a)), _p(_f64(b)))


def q_mul(a, b):
    out = np.empty(4)
    lib().orc_q_mul(_p(_f64(a)), _p(_f64(b)), _p(out))
    return out


def q_rotate(q, v):
    out = np.empty(3)
    lib().orc_q_rotate(_p(_f64(q)), _p(_f64(v)), _p(out))
    return out


def q_lerp(a, b, t):
    out = np.empty(4)
    lib().orc_q_lerp(_p(_f64(a)), _p(_f64(b)), float(t), _p(out))
    return out


def q_slerp(a, b, t):
    out = np.empty(4)
    lib().orc_q_slerp(_p(_f64(a)), _p(_f64(b)), float(t), _p(out))
    return out


class Rng:
    def __init__(self, seed):
        self._h = C.c_void_p(lib().orc_rng_new(seed))

    def __del__(self):
        try:  # module globals may already be gone at interpreter shutdown
            if getattr(self, "_h", None):
                lib().orc_rng_free(self._h)
                self._h = None
        except Exception:
            pass

    def next_u64(self):
        return lib().orc_rng_next_u64(self._h)

    def f64(self):
        return lib().orc_rng_f64(self._h)

    def quaternion(self):
        out = np.empty(4)
        lib().orc_q_random(self._h, _p(out))
        return out


def load_dcparams(path):
    out = np.empty(TABLE_LEN)
    if lib().orc_load_dcparams(os.fsencode(path), _p(out)) != 0:
        raise RuntimeError(lib().orc_last_error().decode())
    return out


def dfire_bin(d2):
    return lib().orc_dfire_bin(float(d2))


class Scorer:
    def __init__(self, method, receptor_pdb, ligand_pdb, rec_active=(), rec_passive=(), rec_nmodes=None, rec_num_anm=0,
                 lig_active=(), lig_passive=(), lig_nmodes=None, lig_num_anm=0, use_anm=False, potential=None):
        L = lib()
        self.method = METHODS.get(method, method)
        ra, nra, k1 = _strs(rec_active)
        rp, nrp, k2 = _strs(rec_passive)
        la, nla, k3 = _strs(lig_active)
        lp, nlp, k4 = _strs(lig_passive)
        rnm = None if rec_nmodes is None else _f64(rec_nmodes).ravel()
        lnm = None if lig_nmodes is None else _f64(lig_nmodes).ravel()
        pot = None if potential is None else _f64(potential)
        h = L.orc_scorer_new(self.method, os.fsencode(receptor_pdb), os.fsencode(ligand_pdb), ra, nra, rp, nrp, _p(rnm),
                             0 if rnm is None else rnm.size, rec_num_anm, la, nla, lp, nlp, _p(lnm),
                             0 if lnm is None else lnm.size, lig_num_anm, 1 if use_anm else 0, _p(pot))
        if not h:
            raise RuntimeError(L.orc_last_error().decode())
        self._h = C.c_void_p(h)
        self.use_anm = bool(use_anm)
        self.anm_rec = rec_num_anm if use_anm else 0
        self.anm_lig = lig_num_anm if use_anm else 0

    def __del__(self):
        try:  # module globals may already be gone at interpreter shutdown
            if getattr(self, "_h", None):
                lib().orc_scorer_free(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def pose_len(self):
        return 7 + self.anm_rec + self.anm_lig

    def _split(self, row):
        row = _f64(row)
        rec = _f64(row[7:7 + self.anm_rec]) if self.anm_rec else None
        lig = _f64(row[7 + self.anm_rec:7 + self.anm_rec + self.anm_lig]) if self.anm_lig else None
        return _f64(row[:3]), _f64(row[3:7]), rec, lig

    def energy(self, translation, rotation, rec_nm=None, lig_nm=None):
        rn = None if rec_nm is None or len(rec_nm) == 0 else _f64(rec_nm)
        ln = None if lig_nm is None or len(lig_nm) == 0 else _f64(lig_nm)
        return lib().orc_scorer_energy(self._h, _p(_f64(translation)), _p(_f64(rotation)), _p(rn), _p(ln))

    def energy_row(self, row):
        t, q, rn, ln = self._split(row)
        return lib().orc_scorer_energy(self._h, _p(t), _p(q), _p(rn), _p(ln))

    def energy_rows(self, rows):
        return np.array([self.energy_row(r) for r in np.asarray(rows)])

    def energy_rows_mt(self, rows, threads):
        """All rows on `threads` host threads inside the library (the bench's CPU baseline)."""
        rows = _f64(rows)
        out = np.empty(rows.shape[0])
        if lib().orc_scorer_energy_rows_mt(self._h, _p(rows), rows.shape[0], rows.shape[1], int(threads), _p(out)) != 0:
            raise ValueError("orc_scorer_energy_rows_mt: bad arguments")
        return out

    def energy_ex_row(self, row):
        t, q, rn, ln = self._split(row)
        stats = np.empty(8)
        e = lib().orc_scorer_energy_ex(self._h, _p(t), _p(q), _p(rn), _p(ln), _p(stats))
        return e, stats

    def num_atoms(self, side):
        return lib().orc_scorer_num_atoms(self._h, side)

    def _arr(self, fn, side, n, dtype):
        ptr = getattr(lib(), fn)(self._h, side)
        if not ptr or n == 0:
            return np.zeros(0, dtype=dtype)
        ct = C.c_double if dtype == np.float64 else C.c_uint32
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,)).copy()

    def model(self, side):
        """The ld_molecule-shaped dict of one side (what DockingModel::new produced)."""
        n = self.num_atoms(side)
        m = {"coordinates": self._arr("orc_scorer_coordinates", side, 3 * n, np.float64).reshape(n, 3)}
        if self.method == 0:
            m["dfire_types"] = self._arr("orc_scorer_dfire_types", side, n, np.uint32)
        else:
            for k in ("ele_charges", "vdw_charges", "vdw_radii"):
                m[k] = self._arr("orc_scorer_" + k, side, n, np.float64)
        nm = lib().orc_scorer_num_membrane(self._h, side)
        m["membrane"] = self._arr("orc_scorer_membrane", side, nm, np.uint32)
        ng = lib().orc_scorer_num_restraint_groups(self._h, side)
        offs = self._arr("orc_scorer_restraint_offsets", side, ng + 1, np.uint32)
        m["restraint_offsets"] = offs if ng else np.zeros(1, dtype=np.uint32)
        m["restraint_atoms"] = self._arr("orc_scorer_restraint_atoms", side, int(offs[-1]) if ng else 0, np.uint32)
        return m


class GSO:
    def __init__(self, scorer, positions, seed=324324):
        positions = _f64(positions)
        self.scorer = scorer
        self.n, self.row_len = positions.shape
        h = lib().orc_gso_new(_p(positions), self.n, self.row_len, seed, scorer._h, 1 if scorer.use_anm else 0,
                              scorer.anm_rec, scorer.anm_lig)
        if not h:
            raise RuntimeError(lib().orc_last_error().decode())
        self._h = C.c_void_p(h)

    def __del__(self):
        try:  # module globals may already be gone at interpreter shutdown
            if getattr(self, "_h", None):
                lib().orc_gso_free(self._h)
                self._h = None
        except Exception:
            pass

    def step(self):
        lib().orc_gso_step(self._h)

    def run(self, steps, directory):
        if lib().orc_gso_run(self._h, steps, os.fsencode(directory)) != 0:
            raise RuntimeError(lib().orc_last_error().decode())

    def save(self, step, directory):
        if lib().orc_gso_save(self._h, step, os.fsencode(directory)) != 0:
            raise RuntimeError(lib().orc_last_error().decode())

    @property
    def num_evals(self):
        return lib().orc_gso_num_evals(self._h)

    def state(self):
        n = self.n
        st = {"poses": np.empty((n, self.row_len)), "luciferin": np.empty(n), "vision_range": np.empty(n),
              "scoring": np.empty(n), "n_neighbors": np.empty(n, dtype=np.int32), "moved": np.empty(n, dtype=np.int32),
              "target": np.empty(n, dtype=np.int32)}
        lib().orc_gso_state(self._h, _p(st["poses"]), _p(st["luciferin"]), _p(st["vision_range"]), _p(st["scoring"]),
                            _p(st["n_neighbors"]), _p(st["moved"]), _p(st["target"]))
        return st

    def neighbors(self, i):
        buf = np.empty(self.n, dtype=np.int32)
        k = lib().orc_gso_neighbors(self._h, i, _p(buf), self.n)
        return buf[:k].copy()


def parse_positions(path):
    rows, cols = C.c_int(), C.c_int()
    ptr = lib().orc_parse_positions(os.fsencode(path), C.byref(rows), C.byref(cols))
    if not ptr:
        raise RuntimeError(lib().orc_last_error().decode())
    arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(rows.value * cols.value,)).copy()
    lib().orc_free(ptr)
    return arr.reshape(rows.value, cols.value)


def read_npy(path):
    n = C.c_size_t()
    ptr = lib().orc_read_npy_f64(os.fsencode(path), C.byref(n))
    if not ptr:
        raise RuntimeError(lib().orc_last_error().decode())
    arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(n.value,)).copy()
    lib().orc_free(ptr)
    return arr
