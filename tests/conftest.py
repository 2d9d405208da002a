import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _build_if_missing():
    import __graft_entry__ as ge
    ge.ensure_built()


@pytest.fixture(scope="session")
def orc():
    """CPU oracle binding (the checker)."""
    import __graft_entry__ as ge
    _build_if_missing()
    o = ge.oracle()
    o.lib()
    return o


@pytest.fixture(scope="session")
def pkg():
    """The product package, lightdock-rust_amd/ (C ABI via ctypes)."""
    import __graft_entry__ as ge
    _build_if_missing()
    # GPU tests borrow torch for device buffers; bring it up first so that every test sees the
    # same, already initialised runtime (the library shares torch's HIP runtime either way).
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    return ge.package()


@pytest.fixture(scope="session")
def table(pkg):
    """Synthetic DCparams (the real table is not in the reference mount)."""
    return pkg.synth.dcparams()


# name -> (method, receptor pdb, ligand pdb, setup facts)
CASES = {
    "1ppe": dict(method="dfire", rec="lightdock_1ppe_e.pdb", lig="lightdock_1ppe_i.pdb", rec_active=["E.ILE.16"],
                 lig_passive=["I.ARG.1"], use_anm=False, anm_rec=10, anm_lig=10),
    "1k4c": dict(method="dfire", rec="lightdock_receptor_membrane.pdb", lig="lightdock_ligand.pdb", use_anm=False,
                 anm_rec=10, anm_lig=10),
    "2uuy": dict(method="dfire", rec="lightdock_2UUY_rec.pdb", lig="lightdock_2UUY_lig.pdb", use_anm=True, anm_rec=10,
                 anm_lig=10),
    "ab_icode": dict(method="dfire", rec="lightdock_receptor.pdb", lig="lightdock_ligand.pdb",
                     rec_active=["H.ASP.52A", "H.LEU.82C"], use_anm=True, anm_rec=10, anm_lig=10),
    "1azp": dict(method="dna", rec="lightdock_protein.pdb", lig="lightdock_dna.pdb",
                 rec_active=["A.TRP.24", "A.VAL.26", "A.ARG.42"], lig_active=["B.DT.13"], use_anm=True, anm_rec=10,
                 anm_lig=10),
}


def case_paths(name):
    c = CASES[name]
    d = os.path.join(GOLDEN, name)
    return c, d, os.path.join(d, c["rec"]), os.path.join(d, c["lig"])


def case_kwargs(name, orc, table):
    """Constructor kwargs shared by oracle.Scorer and pkg.Scorer.from_pdb."""
    c, d, rec, lig = case_paths(name)
    kw = dict(rec_active=c.get("rec_active", []), rec_passive=c.get("rec_passive", []),
              lig_active=c.get("lig_active", []), lig_passive=c.get("lig_passive", []), use_anm=c["use_anm"],
              rec_num_anm=c["anm_rec"], lig_num_anm=c["anm_lig"])
    if c["use_anm"]:
        kw["rec_nmodes"] = orc.read_npy(os.path.join(d, "rec_nm.npy"))
        kw["lig_nmodes"] = orc.read_npy(os.path.join(d, "lig_nm.npy"))
    if c["method"] == "dfire":
        kw["potential"] = table
    return c["method"], rec, lig, kw


def case_positions(name, orc):
    c, d, _, _ = case_paths(name)
    rows = orc.parse_positions(os.path.join(d, "initial_positions_0.dat"))
    return rows if c["use_anm"] else rows[:, :7]


def parse_gso(path):
    """gso_N.out -> (coords (n, k), luciferin, n_neighbors, vision, scoring)."""
    coords, luc, nn, vis, sco = [], [], [], [], []
    with open(path) as f:
        header = f.readline()
        assert header.startswith("#Coordinates")
        for line in f:
            inner, rest = line[1:].split(")")
            coords.append([float(v) for v in inner.split(",")])
            parts = rest.split()
            luc.append(float(parts[2]))
            nn.append(int(parts[3]))
            vis.append(float(parts[4]))
            sco.append(float(parts[5]))
    return np.array(coords), np.array(luc), np.array(nn), np.array(vis), np.array(sco)


@pytest.fixture(scope="session")
def real_dcparams():
    """The real DFIRE table, if the user supplies one ($LIGHTDOCK_DATA/DCparams)."""
    d = os.environ.get("LIGHTDOCK_DATA")
    p = os.path.join(d, "DCparams") if d else None
    if not p or not os.path.exists(p):
        pytest.skip("real DCparams not available (stripped from the reference mount); set LIGHTDOCK_DATA")
    return p
