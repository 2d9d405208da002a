"""lightdock-rust_amd -- Python view of the C ABI in include/lightdock_hip.h.

This package is plumbing for tests, bench.py and the multi-GPU launcher: it loads
``lib/liblightdock_hip.so`` (hand-written HIP kernels for gfx950 + the C++ host side) with
ctypes and mirrors the reference's operator interface for the GSO + DFIRE/DNA path
(``Score::energy``, ``GSO::new/run``; lightdock-rust src/scoring.rs:11-19, src/lib.rs:27-58).

There is no CPU fallback anywhere in here: if the shared library is missing, or no
MI355X is visible when a scorer is created, the call raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "liblightdock_hip.so")
# A/B tooling (tools/ab6.sh): LIGHTDOCK_HIP_VARIANT=<name> loads lib/variants/<name>.so -- a build of the same sources with extra
# flags, tools/build_variant.sh -- instead of the installed library, so that no script has to copy a variant over it (a diagnostic
# build left installed by an interrupted script would give wrong sums silently).  A variant that does not exist is an error.
if os.environ.get("LIGHTDOCK_HIP_VARIANT"):
    LIB_PATH = os.path.join(_HERE, "lib", "variants", os.environ["LIGHTDOCK_HIP_VARIANT"] + ".so")
CLI_PATH = os.path.join(_HERE, "bin", "lightdock-hip")
INCLUDE_DIR = os.path.normpath(os.path.join(_HERE, "..", "include"))

METHOD_DFIRE = 0
METHOD_DNA = 1
METHOD_PYDOCK = 2
DFIRE_TABLE_LEN = 169 * 169 * 20
METHODS = {"dfire": METHOD_DFIRE, "dna": METHOD_DNA, "pydock": METHOD_PYDOCK}


class LightdockError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("lightdock_hip status %d: %s" % (status, message))
        self.status = status


class _KernelInfo(C.Structure):
    _fields_ = [("pair_kernel_name", C.c_char_p), ("block_threads", C.c_uint32), ("receptor_chunks", C.c_uint32),
                ("lds_bytes", C.c_uint32), ("pair_tests_per_pose", C.c_uint64), ("stream_bytes_per_pose", C.c_uint64)]


class _Molecule(C.Structure):
    _fields_ = [("n_atoms", C.c_size_t), ("coordinates", C.c_void_p), ("dfire_types", C.c_void_p),
                ("ele_charges", C.c_void_p), ("vdw_charges", C.c_void_p), ("vdw_radii", C.c_void_p),
                ("n_membrane", C.c_size_t), ("membrane", C.c_void_p), ("n_restraint_groups", C.c_size_t),
                ("restraint_offsets", C.c_void_p), ("restraint_atoms", C.c_void_p), ("num_anm", C.c_size_t),
                ("nmodes", C.c_void_p)]


class _ScorerDesc(C.Structure):
    _fields_ = [("method", C.c_int), ("use_anm", C.c_int), ("receptor", _Molecule), ("ligand", _Molecule),
                ("potential", C.c_void_p)]


_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 and ask for
    it by the name "libamdhip64.so"; this library asks for the soname "libamdhip64.so.7".  If
    torch is imported first both resolve to torch's copy, but the other way round the process
    would end up with two runtimes (and torch then reports no GPU).  So when a torch wheel with
    a bundled runtime is installed and not loaded yet, load that copy first; without torch the
    system runtime under /opt/rocm is used, as by the lightdock-hip binary."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
        if spec is not None and spec.origin:
            bundled = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
            if os.path.exists(bundled):
                C.CDLL(bundled, mode=C.RTLD_GLOBAL)
    except (ImportError, OSError, ValueError):
        pass


def load_library():
    """dlopen the in-tree HIP library; fail loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                          "there is no CPU fallback for the pose-energy path" % LIB_PATH)
    _share_hip_runtime_with_torch()
    lib = C.CDLL(LIB_PATH)
    vp, sz, dp = C.c_void_p, C.c_size_t, C.POINTER(C.c_double)
    lib.ld_last_error.restype = C.c_char_p
    lib.ld_version.restype = C.c_char_p
    lib.ld_init.argtypes = [C.c_int]
    lib.ld_scorer_create.restype = vp
    lib.ld_scorer_create.argtypes = [C.POINTER(_ScorerDesc)]
    lib.ld_scorer_create_from_pdb.restype = vp
    lib.ld_scorer_create_from_pdb.argtypes = [C.c_int, C.c_char_p, C.c_char_p,
                                              vp, sz, vp, sz, vp, sz, sz,
                                              vp, sz, vp, sz, vp, sz, sz, C.c_int, vp]
    lib.ld_scorer_destroy.argtypes = [vp]
    lib.ld_load_dcparams.argtypes = [C.c_char_p, vp]
    lib.ld_scorer_num_atoms.restype = sz
    lib.ld_scorer_num_atoms.argtypes = [vp, C.c_int]
    lib.ld_scorer_pose_len.restype = sz
    lib.ld_scorer_pose_len.argtypes = [vp]
    lib.ld_scorer_method.argtypes = [vp]
    lib.ld_scorer_model_arrays.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp]
    lib.ld_scorer_set_stream.argtypes = [vp, vp]
    lib.ld_scorer_energy.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.ld_scorer_energy_batch.argtypes = [vp, sz, vp, sz, vp]
    lib.ld_scorer_energy_batch_device.argtypes = [vp, sz, vp, sz, vp, vp, vp]
    lib.ld_scorer_kernel_info.argtypes = [vp, C.POINTER(_KernelInfo)]
    lib.ld_scorer_last_block_counts.argtypes = [vp, sz, vp]
    lib.ld_scorer_bm_quiet_subtiles.argtypes = [vp, vp]
    lib.ld_scorer_enable_timing.argtypes = [vp, C.c_int]
    lib.ld_scorer_pair_kernel_time.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    lib.ld_gso_create.restype = vp
    lib.ld_gso_create.argtypes = [vp, sz, sz, vp, vp]
    lib.ld_gso_destroy.argtypes = [vp]
    lib.ld_gso_step.argtypes = [vp]
    lib.ld_gso_run.argtypes = [vp, C.c_uint32]
    lib.ld_gso_steps_done.restype = C.c_uint32
    lib.ld_gso_steps_done.argtypes = [vp]
    lib.ld_gso_num_evals.restype = C.c_uint64
    lib.ld_gso_num_evals.argtypes = [vp]
    lib.ld_gso_read.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, vp]
    lib.ld_gso_save.argtypes = [vp, sz, C.c_uint32, C.c_char_p]
    lib.ld_gso_save_many.argtypes = [vp, sz, vp, vp, C.c_uint32]
    lib.ld_cli_main.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
    lib.ld_model_from_pdb.restype = vp
    lib.ld_model_from_pdb.argtypes = [C.c_int, C.c_char_p, vp, sz, vp, sz, vp, sz, sz]
    lib.ld_model_view.argtypes = [vp, C.POINTER(_Molecule)]
    lib.ld_model_destroy.argtypes = [vp]
    lib.ld_dfire_bin_lut.argtypes = [vp, vp, C.POINTER(C.c_double)]
    lib.ld_dfire_packed_lut.argtypes = [C.c_int, C.c_double, vp, C.POINTER(C.c_double)]
    lib.ld_dfire_bm_lut.argtypes = [C.c_double, C.c_double, vp, C.POINTER(C.c_double)]
    lib.ld_stdrng_key.argtypes = [C.c_uint64, vp]
    lib.ld_spatial_tile_order.restype = sz
    lib.ld_spatial_tile_order.argtypes = [vp, sz, vp]
    _lib = lib
    return lib


def _check(status):
    if status != 0:
        raise LightdockError(status, load_library().ld_last_error().decode())


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _strings(items):
    items = [s.encode() for s in (items or [])]
    arr = (C.c_char_p * max(1, len(items)))(*items)
    return C.cast(arr, C.c_void_p), len(items), arr


def init(device=-1):
    _check(load_library().ld_init(device))


def device_count():
    return load_library().ld_device_count()


def load_dcparams(path):
    out = np.empty(DFIRE_TABLE_LEN, dtype=np.float64)
    _check(load_library().ld_load_dcparams(os.fsencode(path), _ptr(out)))
    return out


def _np_from(ptr, n, ctype, dtype):
    if not ptr or n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,)).copy()


def model_from_pdb(method, pdb_path, active=(), passive=(), nmodes=None, num_anm=0):
    """Host-side DockingModel::new (src/dfire.rs:115-190, src/dna.rs:249-364); no GPU involved.
    Returns the ld_molecule fields as a dict of numpy arrays (usable with Scorer.from_arrays)."""
    lib = load_library()
    method = METHODS.get(method, method)
    a, na, k1 = _strings(active)
    p, npas, k2 = _strings(passive)
    nm = None if nmodes is None else _f64(nmodes).ravel()
    h = lib.ld_model_from_pdb(method, os.fsencode(pdb_path), a, na, p, npas, _ptr(nm), 0 if nm is None else nm.size, num_anm)
    if not h:
        raise LightdockError(-1, lib.ld_last_error().decode())
    h = C.c_void_p(h)
    try:
        v = _Molecule()
        _check(lib.ld_model_view(h, C.byref(v)))
        n = v.n_atoms
        out = {"coordinates": _np_from(v.coordinates, 3 * n, C.c_double, np.float64).reshape(n, 3),
               "membrane": _np_from(v.membrane, v.n_membrane, C.c_uint32, np.uint32), "num_anm": int(v.num_anm)}
        offs = _np_from(v.restraint_offsets, v.n_restraint_groups + 1, C.c_uint32, np.uint32)
        out["restraint_offsets"] = offs
        out["restraint_atoms"] = _np_from(v.restraint_atoms, int(offs[-1]) if len(offs) else 0, C.c_uint32, np.uint32)
        if v.dfire_types:
            out["dfire_types"] = _np_from(v.dfire_types, n, C.c_uint32, np.uint32)
        for k in ("ele_charges", "vdw_charges", "vdw_radii"):
            if getattr(v, k):
                out[k] = _np_from(getattr(v, k), n, C.c_double, np.float64)
        if v.nmodes:
            out["nmodes"] = _np_from(v.nmodes, int(v.num_anm) * n * 3, C.c_double, np.float64)
        return out
    finally:
        lib.ld_model_destroy(h)


def dfire_bin_lut():
    lut = np.zeros(901, dtype=np.uint8)
    steps = np.zeros(21, dtype=np.float64)
    d2 = C.c_double()
    _check(load_library().ld_dfire_bin_lut(_ptr(lut), _ptr(steps), C.byref(d2)))
    return lut, steps, d2.value


def dfire_packed_lut(cells_per_unit=2, ubound=256.0):
    """(words, eps) of the default DFIRE kernel's cell LUT, see ld_dfire_packed_lut in the header."""
    words = np.zeros(1028 * cells_per_unit, dtype=np.uint32)
    eps = C.c_double()
    _check(load_library().ld_dfire_packed_lut(cells_per_unit, ubound, _ptr(words), C.byref(eps)))
    return words, eps.value


def dfire_bm_lut(ubound=1024.0, lig_extent=45.0):
    """(codes, eps in LUT cells) of the block-major DFIRE kernel's cell LUT, see ld_dfire_bm_lut in the header."""
    codes = np.zeros(14592, dtype=np.uint8)
    eps = C.c_double()
    _check(load_library().ld_dfire_bm_lut(C.c_double(ubound), C.c_double(lig_extent), _ptr(codes), C.byref(eps)))
    return codes, eps.value


def dfire_bm_fix_scale(rec_xyz, reach, table_vmax):
    """(reach count, extra bits, scale) of the block-major kernel's fixed-point sums, see ld_dfire_bm_fix_scale in the header."""
    xyz = _f64(rec_xyz).reshape(-1, 3)
    count, extra, scale = C.c_uint64(), C.c_int(), C.c_double()
    lib = load_library()
    lib.ld_dfire_bm_fix_scale.argtypes = [C.c_void_p, C.c_size_t, C.c_double, C.c_double, C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.POINTER(C.c_double)]
    _check(lib.ld_dfire_bm_fix_scale(_ptr(xyz), xyz.shape[0], C.c_double(reach), C.c_double(table_vmax), C.byref(count), C.byref(extra), C.byref(scale)))
    return count.value, extra.value, scale.value


def spatial_tile_order(xyz):
    """Tile order of the DFIRE kernel: slot -> atom index (UINT32_MAX = padding)."""
    xyz = _f64(xyz).reshape(-1, 3)
    n = xyz.shape[0]
    out = np.zeros((n + 63) // 64 * 64, dtype=np.uint32)
    got = load_library().ld_spatial_tile_order(_ptr(xyz), n, _ptr(out))
    if got != out.size:
        raise LightdockError(-1, load_library().ld_last_error().decode())
    return out


def dfire_tile_layout(xyz, dfire_types):
    """(order, type_perm) the DFIRE scorer uses for one molecule: order[slot] = atom index
    (UINT32_MAX = padding), type_perm[type] = number in the patch layout of the potential."""
    xyz = _f64(xyz).reshape(-1, 3)
    types = np.ascontiguousarray(dfire_types, dtype=np.uint32)
    n = xyz.shape[0]
    order = np.zeros((n + 63) // 64 * 64, dtype=np.uint32)
    perm = np.zeros(169, dtype=np.uint32)
    lib = load_library()
    lib.ld_dfire_tile_layout.restype = C.c_size_t
    lib.ld_dfire_tile_layout.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    got = lib.ld_dfire_tile_layout(_ptr(xyz), _ptr(types), n, _ptr(order), _ptr(perm))
    if got != order.size:
        raise LightdockError(-1, lib.ld_last_error().decode())
    return order, perm


def stdrng_key(seed):
    key = np.zeros(8, dtype=np.uint32)
    load_library().ld_stdrng_key(seed, _ptr(key))
    return key


class Scorer:
    """A `Box<dyn Score>`: DFIRE::new / DNA::new (src/dfire.rs:201-234, src/dna.rs:375-408)."""

    def __init__(self, handle, keep=()):
        if not handle:
            raise LightdockError(-1, load_library().ld_last_error().decode())
        self._h = C.c_void_p(handle)
        self._keep = keep
        self.lib = load_library()

    @classmethod
    def from_pdb(cls, method, receptor_pdb, ligand_pdb, rec_active=(), rec_passive=(), rec_nmodes=None, rec_num_anm=0,
                 lig_active=(), lig_passive=(), lig_nmodes=None, lig_num_anm=0, use_anm=False, potential=None):
        lib = load_library()
        method = METHODS.get(method, method)
        ra, nra, k1 = _strings(rec_active)
        rp, nrp, k2 = _strings(rec_passive)
        la, nla, k3 = _strings(lig_active)
        lp, nlp, k4 = _strings(lig_passive)
        rnm = None if rec_nmodes is None else _f64(rec_nmodes).ravel()
        lnm = None if lig_nmodes is None else _f64(lig_nmodes).ravel()
        pot = None if potential is None else _f64(potential)
        if pot is not None and pot.size != DFIRE_TABLE_LEN:
            raise ValueError("DFIRE potential must have %d values" % DFIRE_TABLE_LEN)
        h = lib.ld_scorer_create_from_pdb(method, os.fsencode(receptor_pdb), os.fsencode(ligand_pdb),
                                          ra, nra, rp, nrp, _ptr(rnm), 0 if rnm is None else rnm.size, rec_num_anm,
                                          la, nla, lp, nlp, _ptr(lnm), 0 if lnm is None else lnm.size, lig_num_anm,
                                          1 if use_anm else 0, _ptr(pot))
        return cls(h, keep=(k1, k2, k3, k4))

    @classmethod
    def from_arrays(cls, method, receptor, ligand, use_anm=False, potential=None):
        """receptor / ligand: dicts with the fields of ld_molecule (numpy arrays)."""
        lib = load_library()
        method = METHODS.get(method, method)
        keep = []

        def mol(d):
            m = _Molecule()
            coords = _f64(d["coordinates"]).reshape(-1, 3)
            keep.append(coords)
            m.n_atoms = coords.shape[0]
            m.coordinates = _ptr(coords)
            for name, dt in (("dfire_types", np.uint32), ("ele_charges", np.float64), ("vdw_charges", np.float64),
                             ("vdw_radii", np.float64), ("membrane", np.uint32), ("restraint_offsets", np.uint32),
                             ("restraint_atoms", np.uint32), ("nmodes", np.float64)):
                v = d.get(name)
                if v is not None:
                    v = np.ascontiguousarray(v, dtype=dt).ravel()
                    keep.append(v)
                    setattr(m, name, _ptr(v))
            m.n_membrane = 0 if d.get("membrane") is None else len(d["membrane"])
            offs = d.get("restraint_offsets")
            m.n_restraint_groups = 0 if offs is None else len(offs) - 1
            m.num_anm = int(d.get("num_anm", 0))
            return m

        desc = _ScorerDesc()
        desc.method = method
        desc.use_anm = 1 if use_anm else 0
        desc.receptor = mol(receptor)
        desc.ligand = mol(ligand)
        if potential is not None:
            pot = _f64(potential)
            keep.append(pot)
            desc.potential = _ptr(pot)
        return cls(lib.ld_scorer_create(C.byref(desc)), keep=tuple(keep))

    def close(self):
        if self._h:
            self.lib.ld_scorer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    @property
    def pose_len(self):
        return self.lib.ld_scorer_pose_len(self._h)

    def num_atoms(self, side):
        return self.lib.ld_scorer_num_atoms(self._h, side)

    def model_arrays(self, side):
        n = self.num_atoms(side)
        out = {"coordinates": np.zeros((n, 3))}
        if self.lib.ld_scorer_method(self._h) == METHOD_DFIRE:
            out["dfire_types"] = np.zeros(n, dtype=np.uint32)
            _check(self.lib.ld_scorer_model_arrays(self._h, side, _ptr(out["coordinates"]), _ptr(out["dfire_types"]),
                                                   None, None, None))
        else:
            for k in ("ele_charges", "vdw_charges", "vdw_radii"):
                out[k] = np.zeros(n)
            _check(self.lib.ld_scorer_model_arrays(self._h, side, _ptr(out["coordinates"]), None,
                                                   _ptr(out["ele_charges"]), _ptr(out["vdw_charges"]), _ptr(out["vdw_radii"])))
        return out

    def set_stream(self, hip_stream):
        _check(self.lib.ld_scorer_set_stream(self._h, C.c_void_p(hip_stream)))

    def energy(self, translation, rotation, rec_nmodes=None, lig_nmodes=None):
        """Score::energy (src/scoring.rs:11-19); rotation = (w, x, y, z)."""
        t, q = _f64(translation), _f64(rotation)
        rn = None if rec_nmodes is None or len(rec_nmodes) == 0 else _f64(rec_nmodes)
        ln = None if lig_nmodes is None or len(lig_nmodes) == 0 else _f64(lig_nmodes)
        out = C.c_double()
        _check(self.lib.ld_scorer_energy(self._h, _ptr(t), _ptr(q), _ptr(rn), _ptr(ln), C.byref(out)))
        return out.value

    def energy_batch(self, poses):
        poses = _f64(poses)
        if poses.ndim != 2:
            raise ValueError("poses must be (n, pose_len)")
        out = np.empty(poses.shape[0], dtype=np.float64)
        _check(self.lib.ld_scorer_energy_batch(self._h, poses.shape[0], _ptr(poses), poses.shape[1], _ptr(out)))
        return out

    def energy_batch_device(self, n, d_poses, stride, d_energies, d_active=None, d_pair_counts=None):
        """Raw device pointers (ints), asynchronous on the scorer's stream."""
        _check(self.lib.ld_scorer_energy_batch_device(self._h, n, C.c_void_p(d_poses), stride,
                                                      C.c_void_p(d_active) if d_active else None, C.c_void_p(d_energies),
                                                      C.c_void_p(d_pair_counts) if d_pair_counts else None))

    def enable_timing(self, on=True):
        _check(self.lib.ld_scorer_enable_timing(self._h, 1 if on else 0))

    def pair_kernel_time(self):
        """(total ms, launches) of the pair kernel since the last call; HIP events on the scorer's stream."""
        ms, n = C.c_double(), C.c_uint64()
        _check(self.lib.ld_scorer_pair_kernel_time(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def last_block_counts(self, n):
        """8x8 atom-pair blocks evaluated per pose in the last counting launch (tiled DFIRE kernel)."""
        out = np.zeros(n, dtype=np.uint32)
        _check(self.lib.ld_scorer_last_block_counts(self._h, n, _ptr(out)))
        return out

    def bm_quiet_subtiles(self):
        """Receptor subtiles whose atoms' rows of the potential are zero against every ligand type (block-major DFIRE path)."""
        n = C.c_uint32()
        _check(self.lib.ld_scorer_bm_quiet_subtiles(self._h, C.byref(n)))
        return n.value

    def kernel_info(self):
        info = _KernelInfo()
        _check(self.lib.ld_scorer_kernel_info(self._h, C.byref(info)))
        return {"pair_kernel_name": info.pair_kernel_name.decode(), "block_threads": info.block_threads,
                "receptor_chunks": info.receptor_chunks, "lds_bytes": info.lds_bytes,
                "pair_tests_per_pose": info.pair_tests_per_pose, "stream_bytes_per_pose": info.stream_bytes_per_pose}


class GSO:
    """GSO::new / GSO::run (src/lib.rs:27-58) for a batch of independent swarms."""

    def __init__(self, scorer, positions, seeds=None):
        positions = _f64(positions)
        if positions.ndim == 2:
            positions = positions[None]
        if positions.ndim != 3 or positions.shape[2] != scorer.pose_len:
            raise ValueError("positions must be (swarms, glowworms, %d)" % scorer.pose_len)
        self.scorer = scorer
        self.lib = scorer.lib
        self.n_swarms, self.n_glowworms, self.pose_len = positions.shape
        sd = None if seeds is None else np.ascontiguousarray(seeds, dtype=np.uint64)
        if sd is not None and sd.size != self.n_swarms:
            raise ValueError("one seed per swarm")
        h = self.lib.ld_gso_create(scorer.handle, self.n_swarms, self.n_glowworms, _ptr(positions), _ptr(sd))
        if not h:
            raise LightdockError(-1, self.lib.ld_last_error().decode())
        self._h = C.c_void_p(h)

    def close(self):
        if self._h:
            self.lib.ld_gso_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self):
        _check(self.lib.ld_gso_step(self._h))

    def run(self, steps):
        _check(self.lib.ld_gso_run(self._h, steps))

    @property
    def steps_done(self):
        return self.lib.ld_gso_steps_done(self._h)

    @property
    def num_evals(self):
        return self.lib.ld_gso_num_evals(self._h)

    def read(self, swarm=0):
        n = self.n_glowworms
        st = {"poses": np.empty((n, self.pose_len)), "luciferin": np.empty(n), "vision_range": np.empty(n),
              "scoring": np.empty(n), "n_neighbors": np.empty(n, dtype=np.int32), "moved": np.empty(n, dtype=np.int32),
              "target": np.empty(n, dtype=np.int32)}
        _check(self.lib.ld_gso_read(self._h, swarm, _ptr(st["poses"]), _ptr(st["luciferin"]), _ptr(st["vision_range"]),
                                    _ptr(st["scoring"]), _ptr(st["n_neighbors"]), _ptr(st["moved"]), _ptr(st["target"])))
        return st

    def save(self, swarm, step, directory):
        _check(self.lib.ld_gso_save(self._h, swarm, step, os.fsencode(directory)))

    def save_many(self, swarms, step, directories):
        """gso_<step>.out of many swarms in one call (one device read, files written by threads)."""
        n = len(swarms)
        ids = (C.c_size_t * n)(*[int(s) for s in swarms])
        dirs = (C.c_char_p * n)(*[os.fsencode(d) for d in directories])
        _check(self.lib.ld_gso_save_many(self._h, n, ids, dirs, step))


def cli_main(argv):
    """The reference command line, in process (src/bin/lightdock-rust.rs:77-333)."""
    args = [os.fsencode(a) for a in argv]
    arr = (C.c_char_p * len(args))(*args)
    return load_library().ld_cli_main(len(args), arr)


from . import multi, synth  # noqa: E402,F401
