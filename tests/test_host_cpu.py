"""Host-side checks that run without a GPU: the C-ABI library loads and exports what
include/lightdock_hip.h declares, the host model builder agrees with the oracle, and the
product fails loudly (no CPU fallback) when no HIP device is present."""
import math
import os
import re
import shutil
import subprocess
import sys

import numpy as np
import pytest

from conftest import CASES, GOLDEN, ROOT, case_paths


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load_library()
    header = open(os.path.join(pkg.INCLUDE_DIR, "lightdock_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    names = set(re.findall(r"\b(ld_[a-z0-9_]+)\s*\(", header))
    assert len(names) >= 30
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, "declared but not exported: %s" % missing
    assert b"gfx950" in lib.ld_version()


def test_no_torch_types_in_the_abi(pkg):
    header = open(os.path.join(pkg.INCLUDE_DIR, "lightdock_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)   # comments may mention torch tensors
    assert "torch" not in header and "at::" not in header and "std::" not in header


def test_header_is_plain_c_and_a_c_program_links(pkg, orc, tmp_path):
    """The boundary is a C ABI: the header compiles as C99 and a C program links against the
    shared library (host-only entry points; nothing here touches a GPU)."""
    src = tmp_path / "link.c"
    src.write_text(
        '#include <stdio.h>\n#include "lightdock_hip.h"\n'
        "int main(void) {\n"
        "  uint32_t key[8]; uint8_t lut[901]; double steps[21], iface;\n"
        "  ld_stdrng_key(324324u, key);\n"
        "  if (ld_dfire_bin_lut(lut, steps, &iface) != LD_OK) return 2;\n"
        '  printf("%u %u %.17g %s\\n", (unsigned)key[0], (unsigned)lut[899], iface, ld_version());\n'
        "  return ld_scorer_create(NULL) == NULL && ld_last_error()[0] ? 0 : 3;\n"
        "}\n")
    lib_dir = os.path.dirname(pkg.LIB_PATH)
    exe = tmp_path / "link"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", pkg.INCLUDE_DIR, str(src),
                        "-L", lib_dir, "-llightdock_hip", "-Wl,-rpath," + lib_dir, "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    key0, last_bin, iface, version = r.stdout.split(None, 3)
    assert int(key0) == int(pkg.stdrng_key(324324)[0])
    assert int(last_bin) & 31 == 19 and float(iface) == pkg.dfire_bin_lut()[2]
    assert "gfx950" in version


@pytest.mark.parametrize("name", ["1ppe", "1k4c", "2uuy", "ab_icode", "1azp"])
def test_host_model_builder_matches_oracle(pkg, orc, table, name):
    c, d, rec, lig = case_paths(name)
    from conftest import case_kwargs
    method, rec, lig, kw = case_kwargs(name, orc, table)
    o = orc.Scorer(method, rec, lig, **kw)
    for side, path, active in ((0, rec, c.get("rec_active", [])), (1, lig, c.get("lig_active", []))):
        m = pkg.model_from_pdb(method, path, active=active)
        w = o.model(side)
        assert np.array_equal(m["coordinates"], w["coordinates"])
        for k in ("dfire_types", "ele_charges", "vdw_charges", "vdw_radii"):
            if k in w:
                assert np.array_equal(m[k], w[k]), k
        assert np.array_equal(m["membrane"], w["membrane"])
        assert np.array_equal(m["restraint_offsets"], w["restraint_offsets"])
        assert np.array_equal(m["restraint_atoms"], w["restraint_atoms"])


def test_insertion_code_restraints_ab_icode(pkg, orc):
    """Restraint ids carry the insertion code without a separator (src/dfire.rs:139-142):
    "H.ASP.52A" and "H.LEU.82C" of example/ab_icode must each find their residue's atoms, and
    not those of H.52 / H.82 (same serial, no or another insertion code)."""
    c, d, rec, lig = case_paths("ab_icode")
    m = pkg.model_from_pdb("dfire", rec, active=c["rec_active"])
    offs, atoms = m["restraint_offsets"], m["restraint_atoms"]
    assert len(offs) == 3 and offs[1] > 0 and offs[2] > offs[1]          # two groups, both non-empty
    lines = [l for l in open(rec) if l.startswith(("ATOM", "HETATM"))]
    want = {"52A": ("ASP", 8), "82C": ("LEU", 8)}
    groups = [set(atoms[offs[g]:offs[g + 1]].tolist()) for g in range(2)]
    for key, (resname, n_atoms) in want.items():
        idx = {i for i, l in enumerate(lines) if l[21] == "H" and l[22:27].strip() == key}
        assert len(idx) == n_atoms and all(lines[i][17:20] == resname for i in idx)
        assert idx in groups
    none = pkg.model_from_pdb("dfire", rec, active=["H.ASP.52", "H.LEU.82"])   # without the code: other residues or nothing
    assert set(none["restraint_atoms"].tolist()).isdisjoint(set(atoms.tolist()))


def test_pdb_walk_order_groups_like_pdbtbx(pkg, orc, table, tmp_path):
    """Atoms of one residue that are split in the file, and a chain that re-appears, are
    regrouped chain -> residue -> atom (first-appearance order), by both implementations."""
    lines = [
        "ATOM      1  N   ALA A   1       0.000   0.000   0.000  1.00  0.00           N",
        "ATOM      2  N   GLY B   1       5.000   0.000   0.000  1.00  0.00           N",
        "ATOM      3  CA  ALA A   1       1.000   0.000   0.000  1.00  0.00           C",
        "ATOM      4  N   SER A   2       2.000   0.000   0.000  1.00  0.00           N",
        "ATOM      5  C   ALA A   1       3.000   0.000   0.000  1.00  0.00           C",
        "ATOM      6  N   SER A   2A      4.000   0.000   0.000  1.00  0.00           N",
    ]
    p = tmp_path / "mix.pdb"
    p.write_text("\n".join(lines) + "\n")
    m = pkg.model_from_pdb("dfire", str(p), active=["A.SER.2A", "A.ALA.1"])
    assert list(m["coordinates"][:, 0]) == [0.0, 1.0, 3.0, 2.0, 4.0, 5.0]
    assert list(m["dfire_types"]) == [74, 75, 76, 90, 90, 79]
    assert list(m["restraint_offsets"]) == [0, 3, 4] and list(m["restraint_atoms"]) == [0, 1, 2, 4]
    good = os.path.join(GOLDEN, "1ppe", "lightdock_1ppe_i.pdb")
    o = orc.Scorer("dfire", str(p), good, rec_active=["A.SER.2A", "A.ALA.1"], potential=table)
    w = o.model(0)
    assert np.array_equal(w["coordinates"], m["coordinates"]) and np.array_equal(w["dfire_types"], m["dfire_types"])
    assert np.array_equal(w["restraint_atoms"], m["restraint_atoms"])


def test_host_unsupported_atoms_fail_with_reference_messages(pkg, tmp_path):
    bad = tmp_path / "bad.pdb"
    bad.write_text("ATOM      1  N   XYZ A   1      11.104  13.207   2.100  1.00  0.00           N\n")
    with pytest.raises(pkg.LightdockError, match="Residue name not supported in DFIRE"):
        pkg.model_from_pdb("dfire", str(bad))
    bad.write_text("ATOM      1  H1  ALA A   1      11.104  13.207   2.100  1.00  0.00           H\n")
    with pytest.raises(pkg.LightdockError, match="Not supported atom type"):
        pkg.model_from_pdb("dfire", str(bad))
    m = pkg.model_from_pdb("dna", str(bad))      # DNA: N-terminal H1 falls back to "ALA-H" (src/dna.rs:321-326)
    assert m["ele_charges"][0] == 0.2719 and m["vdw_radii"][0] == 0.6
    bad.write_text("ATOM      1  QQ  ALA A   1      11.104  13.207   2.100  1.00  0.00           H\n")
    with pytest.raises(pkg.LightdockError, match=r"DNA Error: Atom \[\"ALA-QQ\"\] not supported"):
        pkg.model_from_pdb("dna", str(bad))
    with pytest.raises(pkg.LightdockError, match="cannot open PDB"):
        pkg.model_from_pdb("dna", str(tmp_path / "missing.pdb"))


def test_dfire_bin_lut_equals_reference_formula(pkg, orc):
    """The kernel's sqrt-free binning must equal DIST_TO_BINS[(sqrt(d2)*2-1) as usize]-1
    (src/dfire.rs:336-337) for EVERY double in [0, 225].  Both sides are monotone step functions
    of d2, so agreement at both ends of each 0.25-wide cell plus on each exact step proves it."""
    lut, steps, iface = pkg.dfire_bin_lut()
    steps = np.append(steps, np.inf)

    def kernel_bin(d2):
        b = int(lut[int(d2 * 4.0)])
        return b + (1 if d2 >= steps[b + 1] else 0)

    assert kernel_bin(0.0) == 0 and kernel_bin(225.0) == 20 and kernel_bin(math.nextafter(225.0, 0)) == 19
    rounding_cases = 0
    for c in range(0, 900):
        lo, hi = c * 0.25, math.nextafter((c + 1) * 0.25, 0.0)
        assert orc.dfire_bin(lo) == kernel_bin(lo)
        assert orc.dfire_bin(hi) == kernel_bin(hi)
        rounding_cases += orc.dfire_bin(hi) != lut[c]
    assert rounding_cases > 0     # e.g. sqrt(pred(6.25)) rounds up to 2.5: a plain cell LUT would be wrong
    for b in range(1, 21):
        assert orc.dfire_bin(steps[b]) == b and orc.dfire_bin(math.nextafter(steps[b], 0.0)) == b - 1
    rng = np.random.default_rng(5)
    for d2 in rng.uniform(0.0, 225.0, size=20000):
        assert orc.dfire_bin(d2) == kernel_bin(d2)
    # interface threshold: d <= 3.9 (src/dfire.rs:339) as a d2 threshold
    assert math.sqrt(iface) * 2.0 - 1.0 <= 3.9 < math.sqrt(math.nextafter(iface, 1e9)) * 2.0 - 1.0
    assert abs(iface - 2.45 ** 2) < 1e-12


def test_stdrng_key_matches_oracle_stream(pkg, orc):
    """Host key expansion (PCG32) feeds the device ChaCha20; check it through the oracle's
    generator by re-deriving the first block in numpy."""
    key = pkg.stdrng_key(324324)

    def chacha_block(key, counter):
        def rotl(v, n):
            return ((v << n) & 0xffffffff) | (v >> (32 - n))
        s = [0x61707865, 0x3320646e, 0x79622d32, 0x6b206574] + [int(k) for k in key] + \
            [counter & 0xffffffff, counter >> 32, 0, 0]
        x = list(s)

        def qr(a, b, c, d):
            x[a] = (x[a] + x[b]) & 0xffffffff; x[d] = rotl(x[d] ^ x[a], 16)
            x[c] = (x[c] + x[d]) & 0xffffffff; x[b] = rotl(x[b] ^ x[c], 12)
            x[a] = (x[a] + x[b]) & 0xffffffff; x[d] = rotl(x[d] ^ x[a], 8)
            x[c] = (x[c] + x[d]) & 0xffffffff; x[b] = rotl(x[b] ^ x[c], 7)
        for _ in range(10):
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
        return [(a + b) & 0xffffffff for a, b in zip(x, s)]

    r = orc.Rng(324324)
    for blk in range(3):
        w = chacha_block(key, blk)
        for k in range(8):
            assert r.next_u64() == (w[2 * k + 1] << 32) | w[2 * k]


def test_synthetic_dcparams_format(pkg, orc, tmp_path):
    t = pkg.synth.dcparams()
    assert t.shape == (169 * 169 * 20,) and np.all(t[0::20] == 10.0)
    assert np.all(np.abs(np.delete(t, np.s_[0::20])) <= 2.0)
    p = tmp_path / "DCparams"
    pkg.synth.write_dcparams(str(p), t)
    assert np.array_equal(orc.load_dcparams(str(p)), t)      # text round trip is exact
    assert np.array_equal(pkg.load_dcparams(str(p)), t)
    with pytest.raises(pkg.LightdockError, match="DFIRE parameters"):
        short = tmp_path / "short"
        short.write_text("1.0\n2.0\n")
        pkg.load_dcparams(str(short))


def test_cli_usage_errors_match_reference(pkg, tmp_path):
    """src/bin/lightdock-rust.rs:101,112,121,142: message on stderr, exit status 0; these paths
    return before any GPU work."""
    cli = pkg.CLI_PATH
    r = subprocess.run([cli], capture_output=True, text=True)
    assert r.returncode == 0 and "Wrong command line. Usage:" in r.stderr
    r = subprocess.run([cli, "a", "b", "ten", "dfire"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stderr.strip() == "Error: steps argument must be a number"
    r = subprocess.run([cli, "a", "b", "10", "zrank"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stderr.strip() == "Error: method not supported"
    r = subprocess.run([cli, "nope.json", "initial_positions_0.dat", "1", "DNA"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0 and r.stderr.startswith('Error reading setup file ["nope.json"]')
    bad = tmp_path / "setup.json"
    bad.write_text('{"anm_seed": 1}')
    r = subprocess.run([cli, str(bad), "initial_positions_0.dat", "1", "dna"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0 and "missing field" in r.stderr
    # a swarm file name without an id is a panic in the reference (exit 101)
    good = os.path.join(GOLDEN, "1ppe", "setup.json")
    r = subprocess.run([cli, good, "positions.dat", "1", "dna"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 101 and "Could not parse swarm from swarm filename" in r.stderr


def test_product_has_no_cpu_fallback(pkg, table):
    """Without a HIP device the scorer must refuse to exist (never route through the oracle)."""
    if pkg.device_count() > 0:
        pytest.skip("a GPU is visible here")
    c, d, rec, lig = case_paths("1ppe")
    with pytest.raises(pkg.LightdockError, match="no HIP device|no CPU fallback"):
        pkg.Scorer.from_pdb("dfire", rec, lig, potential=table)
    src = []
    root = os.path.dirname(pkg.__file__)
    for dirpath, _, files in os.walk(root):
        if "build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                src.append(open(os.path.join(dirpath, f), errors="ignore").read())
    blob = "\n".join(src)
    assert "ld_oracle" not in blob and "oracle/" not in blob.replace("no oracle/_ref", "")


def test_pydock_host_builder_matches_oracle(pkg, orc, tmp_path):
    odd = tmp_path / "odd.pdb"
    odd.write_text("ATOM      1  CQ1 LIG A   1      11.104  13.207   2.100  1.00  0.00           C\n"
                   "ATOM      2  F7  LIG A   1      12.104  13.207   2.100  1.00  0.00           F\n"
                   "ATOM      3  S1  LIG A   1      12.104  14.207   2.100  1.00  0.00           S\n"
                   "ATOM      4  N2  LIG A   1      12.104  14.207   3.100  1.00  0.00           N\n"
                   "ATOM      5  O9  LIG A   1      12.104  15.207   3.100  1.00  0.00           O\n"
                   "ATOM      6  H77 LIG A   1      12.104  15.207   4.100  1.00  0.00           H\n"
                   "ATOM      7  CA  ALA A   2      13.104  15.207   4.100  1.00  0.00           C\n")
    lig = os.path.join(GOLDEN, "unit", "1azp", "1azp_ligand.pdb")
    w = orc.Scorer("pydock", str(odd), lig).model(0)
    m = pkg.model_from_pdb("pydock", str(odd))
    for k in ("ele_charges", "vdw_charges", "vdw_radii"):
        assert np.array_equal(m[k], w[k]), k
    with pytest.raises(pkg.LightdockError, match=r'DNA Error: Atom \["LIG-CQ1"\] not supported'):
        pkg.model_from_pdb("dna", str(odd))
    odd.write_text("ATOM      1  XX  LIG A   1      11.104  13.207   2.100  1.00  0.00           X\n")
    with pytest.raises(pkg.LightdockError, match=r'PYDOCK Error: Atom \["\*-X"\] not supported'):
        pkg.model_from_pdb("pydock", str(odd))
    # every atom of the real fixtures types identically under dna and pydock
    for f in ("1azp_receptor.pdb", "1azp_ligand.pdb"):
        p = os.path.join(GOLDEN, "unit", "1azp", f)
        a, b = pkg.model_from_pdb("dna", p), pkg.model_from_pdb("pydock", p)
        assert all(np.array_equal(a[k], b[k]) for k in ("ele_charges", "vdw_charges", "vdw_radii"))


def test_cli_panics_of_the_reference_exit_101(pkg, orc, tmp_path):
    """What panics in the reference's worker thread (exit status 101 via join().unwrap(),
    src/bin/lightdock-rust.rs:85) fails the same way in both CLIs, before any GPU work:
    unreadable positions (bin:62), unreadable PDB (bin:201), restraints map without the
    "active"/"passive" lists (bin:257-272), ANM file of the wrong size (bin:233)."""
    import json as _json
    src = os.path.join(GOLDEN, "1azp")
    setup = _json.load(open(os.path.join(src, "setup.json")))
    for cli in (pkg.CLI_PATH, orc.CLI_PATH):
        run = tmp_path / os.path.basename(cli)
        run.mkdir()
        r = subprocess.run([cli, os.path.join(src, "setup.json"), "initial_positions_7.dat", "1", "dna"], cwd=run,
                           capture_output=True, text=True)
        assert r.returncode == 101 and "initial_positions_7.dat" in r.stderr
        assert os.path.isdir(run / "swarm_7")                      # created before the positions are read, bin:176-188
        # a setup next to which there are no lightdock_*.pdb files
        lonely = run / "setup.json"
        lonely.write_text(_json.dumps(setup))
        r = subprocess.run([cli, str(lonely), os.path.join(src, "initial_positions_0.dat"), "1", "dna"], cwd=run,
                           capture_output=True, text=True)
        assert r.returncode == 101 and "lightdock_protein.pdb" in r.stderr
        # restraints without "passive"
        bad = dict(setup, receptor_restraints={"active": ["A.TRP.24"]})
        for f in ("lightdock_protein.pdb", "lightdock_dna.pdb", "rec_nm.npy", "lig_nm.npy"):
            shutil.copy(os.path.join(src, f), run)
        lonely.write_text(_json.dumps(bad))
        r = subprocess.run([cli, str(lonely), os.path.join(src, "initial_positions_0.dat"), "1", "dna"], cwd=run,
                           capture_output=True, text=True)
        assert r.returncode == 101 and "passive" in r.stderr
        # ANM array that does not match the atom count
        np.save(run / "rec_nm.npy", np.zeros(30))
        lonely.write_text(_json.dumps(setup))
        r = subprocess.run([cli, str(lonely), os.path.join(src, "initial_positions_0.dat"), "1", "dna"], cwd=run,
                           capture_output=True, text=True)
        assert r.returncode == 101 and "Number of read ANM in receptor does not correspond" in r.stderr


def test_spatial_tile_order(pkg, orc, table):
    """host/spatial_order.cpp: a permutation with padding only at the tail, whose 8-atom subtiles
    and 64-atom tiles are far more compact than file order or random order."""
    from conftest import case_kwargs
    method, rec, lig, kw = case_kwargs("1k4c", orc, table)
    xyz = orc.Scorer(method, rec, lig, **kw).model(0)["coordinates"]
    n = len(xyz)
    order = pkg.spatial_tile_order(xyz)
    assert order.size == (n + 63) // 64 * 64
    real = order[order != 0xFFFFFFFF]
    assert np.array_equal(np.sort(real), np.arange(n)) and np.all(order[:n] != 0xFFFFFFFF)

    def diagonals(idx, size):
        d = []
        for k in range(0, n - size + 1, size):
            p = xyz[idx[k:k + size]]
            d.append(np.linalg.norm(p.max(0) - p.min(0)))
        return np.array(d)

    rnd = np.random.default_rng(0).permutation(n)
    file_order = np.arange(n)
    # an 8-atom leaf is about one residue; no leaf or tile sprawls like the membrane beads do in file order
    assert np.median(diagonals(real, 8)) < 10.0 < np.median(diagonals(rnd, 8))
    assert diagonals(real, 8).max() < 0.5 * diagonals(file_order, 8).max()
    assert diagonals(real, 64).max() < 0.5 * diagonals(file_order, 64).max()
    assert diagonals(real, 8).mean() < 0.6 * diagonals(file_order, 8).mean()
    for m in (0, 1, 7, 8, 9, 63, 64, 65, 200):
        pts = np.random.default_rng(m).normal(size=(m, 3))
        o = pkg.spatial_tile_order(pts)
        assert o.size == (m + 63) // 64 * 64 and np.array_equal(np.sort(o[:m]), np.arange(m)) and np.all(o[m:] == 0xFFFFFFFF)


def test_dfire_tile_layout(pkg, orc, table):
    """ld_dfire_tile_layout: a permutation of the atoms with padding at the tail, a bijective
    renumbering of the 169 types, and more atoms whose patch partner type (number ^ 1) sits in their
    own subtile than with the reference's type numbers on the purely geometric order."""
    from conftest import case_kwargs
    method, rec, lig, kw = case_kwargs("1ppe", orc, table)
    m = orc.Scorer(method, rec, lig, **kw).model(0)
    xyz, types = m["coordinates"], m["dfire_types"]
    n = len(xyz)
    order, perm = pkg.dfire_tile_layout(xyz, types)
    assert order.size == (n + 63) // 64 * 64 and np.array_equal(np.sort(order[:n]), np.arange(n)) and np.all(order[n:] == 0xFFFFFFFF)
    assert np.array_equal(np.sort(perm), np.arange(169))

    def partnered(o, numbers):
        have = tot = 0
        for k in range(0, n - 7, 8):
            ids = numbers[types[o[k:k + 8]]]
            present = set(ids.tolist())
            have += sum((int(x) ^ 1) in present for x in ids)
            tot += 8
        return have / tot

    geometric = pkg.spatial_tile_order(xyz)
    assert partnered(order, perm) > partnered(geometric, np.arange(169)) + 0.15
    assert partnered(order, perm) > 0.6


def test_tile_order_keeps_the_blocks_a_pose_has_low(pkg, orc, table):
    """What the atom order is FOR: the number of 8 x 8 blocks (ligand subtile x receptor subtile) whose boxes come within the
    cutoff -- the block-major pair kernel's time is proportional to it.  The example poses of 1k4c replayed through the two
    box tests of dfire_bm_cull on the CPU (tools/cluster_sim.py is the long form): 5 872 blocks a pose with the median splits
    and window swaps of rounds 2-5, 5 310 with round 6's sweeps over all subtiles; the order is deterministic."""
    from conftest import case_kwargs, case_positions
    method, rec, lig, kw = case_kwargs("1k4c", orc, table)
    cpu = orc.Scorer(method, rec, lig, **kw)

    def ordered(m, far):
        o, _ = pkg.dfire_tile_layout(m["coordinates"], m["dfire_types"])
        o2, _ = pkg.dfire_tile_layout(m["coordinates"], m["dfire_types"])
        assert np.array_equal(o, o2)
        pad = o == 0xFFFFFFFF
        c = m["coordinates"][np.where(pad, 0, o).astype(np.int64)].copy()
        c[pad] = far
        return c, ~pad

    def boxes(c, v, size):
        cc, vv = c.reshape(-1, size, 3), v.reshape(-1, size)
        return np.where(vv[..., None], cc, np.inf).min(1), np.where(vv[..., None], cc, -np.inf).max(1)

    def near(lo_a, hi_a, lo_b, hi_b):
        gap = np.maximum(0, np.maximum(lo_a[:, None] - hi_b[None], lo_b[None] - hi_a[:, None]))
        return (gap ** 2).sum(-1) <= 225.0

    rc, rv = ordered(cpu.model(0), 1e9)
    lc, lv = ordered(cpu.model(1), -1e9)
    blocks = tiles = 0
    poses = case_positions("1k4c", orc)[::20]
    for p in poses:
        w, x, y, z = p[3:7] / np.linalg.norm(p[3:7])
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        l = lc @ R.T + p[:3]
        l[~lv] = -1e9
        tile = near(*boxes(l, lv, 64), *boxes(rc, rv, 64))
        sub = near(*boxes(l, lv, 8), *boxes(rc, rv, 8)) & np.repeat(np.repeat(tile, 8, axis=0), 8, axis=1)
        blocks += sub.sum()
        tiles += tile.sum()
    assert blocks / len(poses) < 5450, blocks / len(poses)
    assert tiles / len(poses) < 410, tiles / len(poses)


def test_dfire_tables_equal_reference():
    """tools/check_dfire_tables.py: the reference's r3_to_numerical / ATOMNUMBER / ATOMRES / DIST_TO_BINS
    literals (src/dfire.rs:18-101) against the oracle's restated tables and the product's closed-form
    host model builder, over every (residue, atom) key.  Needs the reference tree: build container only."""
    if not os.path.exists("/root/reference/src/dfire.rs"):
        pytest.skip("reference tree not present (GPU box)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_dfire_tables.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "equal the reference" in r.stdout


@pytest.mark.parametrize("cells", [1, 2])
def test_packed_kernel_cell_lut_decides_like_the_reference(pkg, orc, cells):
    """The default DFIRE kernel tests pairs in f32: D' = cells * 4 d2 + 1/2 with an error below eps/2
    (eps carries a factor 2), cell = floor(D').  Whatever f32 value a true f64 distance can turn into,
    the LUT word of its cell must give the reference's answer -- bin (src/dfire.rs:336-337), cutoff
    (:334), interface (:339) -- or send the pair to the exact f64 path.  This replays the kernel's
    decision logic (dfire_packed.hip, "lean" cells) on the host for dense samples around every step."""
    words, eps = pkg.dfire_packed_lut(cells, 256.0)
    _, steps, iface = pkg.dfire_bin_lut()
    assert 0.0 < eps * cells < 0.2
    MISS, SLOW = 0x00800000, 0x40000000

    def term(b):
        return 8 * (b + 12 * (b >> 2))

    def reference(d4):
        d2 = d4 / 4.0
        return orc.dfire_bin(d2) if d2 <= 225.0 else None

    rng = np.random.default_rng(11)
    cand = list(rng.uniform(0.0, 1100.0, 30000))
    for s in list(steps[1:]) + [iface]:
        cand += [4.0 * s + d for d in (-0.3, -0.1, -2 * eps, -eps, -eps / 2, -1e-9, 0.0, 1e-9, eps / 2, eps, 2 * eps, 0.1, 0.3)]
    exact = lean = plain = 0
    for d4 in cand:
        if d4 < 0.0:
            continue
        want = reference(d4)
        for err in (-eps / 2, -eps / 4, 0.0, eps / 4, eps / 2):
            dp = float(np.float32(cells * (d4 + err) + 0.5))
            if dp < 0.0:
                continue
            c = min(int(dp), 1024 * cells)
            w = int(words[c])
            if w < MISS:
                plain += 1
                assert want is not None and w == term(want) and not d4 / 4.0 <= iface, (d4, err, hex(w))
            elif w == MISS:
                assert want is None, (d4, err)
            else:
                assert w & SLOW
                code = (w >> 24) & 0x1F
                delta = (dp - int(dp)) - 0.5
                if code in (0x01, 0x11) and abs(delta) > eps * cells:      # lean: decided in f32
                    lean += 1
                    t = (w & 0xFFF) + (0 if delta < 0.0 else (w >> 12) & 0xFFF)
                    if t == term(21):                   # the zero slot behind the last bin: beyond the cutoff
                        assert want is None, (d4, err)
                    else:
                        assert want is not None and t == term(want), (d4, err, hex(w))
                        assert (d4 / 4.0 <= iface) == (code == 0x11), (d4, err)
                else:
                    exact += 1                          # exact f64 path: right by construction
    assert plain > 100000 and lean > 100 and exact > 10
    # the cells the pair loop handles without any branch hold one bin each, over their whole interval
    flagged = int(((words >> 30) & 1).sum())
    assert flagged < 40 * cells + 10


@pytest.mark.parametrize("ubound,extent", [(1024.0, 45.0), (512.0, 20.0), (4096.0, 120.0)])
def test_block_major_cell_lut_decides_like_the_reference(pkg, orc, ubound, extent):
    """The block-major DFIRE kernels test pairs in f32: E = 14583.5 - 64 d2 with an error below eps/2 LUT cells (eps
    carries a factor 2), cell = floor(E), everything below 0 reading cell 0.  Whatever f32 value a true f64 distance can
    turn into, the code of its cell must give the reference's answer -- bin (src/dfire.rs:336-337) or beyond the
    cutoff (:334) -- or send the pair to the exact f64 path (the marker's code); and every pair that can set an interface flag (:339)
    must carry a code of bins 0 / 1, whose slots a block with tracked atoms fills with markers."""
    codes, eps = pkg.dfire_bm_lut(ubound, extent)
    _, steps, iface = pkg.dfire_bin_lut()
    assert 0.0 < eps < 8.0
    FLAGGED, MISS = 168, 160
    assert codes[0] == MISS and len(codes) == 14592
    assert set(int(c) for c in np.unique(codes)) <= {FLAGGED, MISS} | {8 * b for b in range(20)}

    rng = np.random.default_rng(5)
    cand = list(rng.uniform(0.0, 1100.0, 30000)) + list(rng.uniform(0.0, 40.0, 4000))
    e4 = eps / 16.0     # eps in units of 4 d2
    for s in list(steps[1:]) + [iface, 225.0]:
        cand += [4.0 * s + d for d in (-0.3, -0.1, -2 * e4, -e4, -e4 / 2, -1e-9, 0.0, 1e-9, e4 / 2, e4, 2 * e4, 0.1, 0.3)]
    cand += [2000.0, 1e4, 1e6, 1e30, float("inf")]
    plain = flagged = miss = 0
    for d4 in cand:
        if d4 < 0.0:
            continue
        d2 = d4 / 4.0
        want = orc.dfire_bin(d2) if d2 <= 225.0 else None
        for err in (-eps / 2, -eps / 4, 0.0, eps / 4, eps / 2):
            E = np.float32(14583.5) - np.float32(16.0 * d4 + err) if np.isfinite(d4) else np.float32(-np.inf)
            c = int(E) if E >= 0 else 0              # v_cvt_u32_f32: negative and NaN -> 0
            code = int(codes[c])
            if code == FLAGGED:
                flagged += 1                        # exact f64 path: right by construction
            elif code == MISS:
                miss += 1
                assert want is None, (d4, err)
            else:
                plain += 1
                assert want is not None and want <= 19 and code == 8 * want, (d4, err, code)
            if d2 <= iface:                         # a pair that sets interface flags: bins 0 / 1, or the exact path already
                assert code in (0, 8, FLAGGED), (d4, err, code)
    assert plain > 100000 and flagged > 10 and miss > 1000
    # one flagged cell per bin step and at the cutoff while eps < 1/2 cell; a few more for larger frames
    n_flagged = int((codes == FLAGGED).sum())
    assert 20 <= n_flagged <= 21 * (1 + 2 * int(np.ceil(max(eps - 0.5, 0.0)))) + 2 * int(np.ceil(eps)) + 2      # (the last step is the cutoff)


def test_committed_profile_matches_the_kernel_sources():
    """bench.py reports on-chip counter fractions from profiles/traffic.json: an entry is only valid for the build it
    was taken from.  The default workload's entry must carry the hash of the kernel sources at HEAD -- change a kernel,
    re-run tools/profile_round.sh + tools/update_traffic.py (bench.py itself omits stale counters and says so)."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    keys = [k for k in t if k.startswith("1k4c:8192:") and t[k].get("source_hash") == bench.kernel_source_hash()]
    assert keys, "no profiles/traffic.json entry of the default workload carries the hash of the current kernel sources (%s)" % bench.kernel_source_hash()
    assert any(t[k].get("valu_insts_per_launch") for k in keys)


def test_block_major_fixed_point_scale_is_count_aware(pkg):
    """The block-major DFIRE path sums table values as 64-bit fixed point (src/dfire.rs:325-345's sum as integer adds).  A
    (pose, ligand tile) sum collects 64 ligand atoms x every receptor atom within reach of the tile: 64 K 2^(44 - x) must stay
    below 2^63, K = the receptor atoms one tile can reach.  The host derives an upper bound on K from the receptor's geometry
    (dfire_bm_reach_count, for receptors of 8192 atoms and more) and takes bits off the scale when it passes 2^13
    (the guard used to look at the table's magnitude only)."""
    rng = np.random.default_rng(5)
    # below 8192 atoms nothing can overflow: no search, the count is n itself, full scale (2^40 for a table whose largest value is 10)
    small = rng.uniform(-30, 30, size=(3413, 3))
    count, extra, scale = pkg.dfire_bm_fix_scale(small, 15.0 + 9.0, 10.0)
    assert (count, extra, scale) == (3413, 0, 2.0 ** 40)
    # a protein-like density (one atom per 17 A^3), 20 000 atoms: a ball of radius 24 A holds ~3 400 of them
    side = (20000 * 17.0) ** (1.0 / 3.0)
    big = rng.uniform(0, side, size=(20000, 3))
    count, extra, scale = pkg.dfire_bm_fix_scale(big, 24.0, 10.0)
    tree_bound = 0
    from scipy.spatial import cKDTree
    tree = cKDTree(big)
    probes = rng.uniform(0, side, size=(4000, 3))
    tree_bound = max(len(b) for b in tree.query_ball_point(probes, 24.0))
    assert tree_bound <= count < 8192, (tree_bound, count)        # an upper bound, and a usable one
    assert extra == 0 and scale == 2.0 ** 40
    # the same atoms squeezed into a third of the side: one ball holds them all -> 20 000 > 8191 -> two bits off the scale
    dense = big / 3.0
    count, extra, scale = pkg.dfire_bm_fix_scale(dense, 24.0, 10.0)
    assert count == 20000 and extra == 2 and scale == 2.0 ** 38
    assert 64 * ((count + 3) // 4) * 2 ** 44 < 2 ** 63                # what the two bits buy: the bound of the docstring
    # the table's own magnitude: 2^e >= vmax; beyond 1024 (or not finite) no scale -- such a scorer runs the pose-major kernels
    assert pkg.dfire_bm_fix_scale(small, 24.0, 1.0)[2] == 2.0 ** 44
    assert pkg.dfire_bm_fix_scale(small, 24.0, 1000.0)[2] == 2.0 ** 34
    assert pkg.dfire_bm_fix_scale(small, 24.0, 2000.0)[2] == 0.0
    assert pkg.dfire_bm_fix_scale(small, 24.0, float("inf"))[2] == 0.0
    # one outlier atom (a 9999.999 dummy coordinate on every axis) must not turn the search into 10^10 probes, nor may a NaN or an
    # infinite coordinate reach a cast: the trivial bound n comes back at once
    import time
    for bad in (9999.999, 1e300, float("nan"), float("inf")):
        out = big.copy()
        out[7] = bad
        t0 = time.perf_counter()
        count, extra, scale = pkg.dfire_bm_fix_scale(out, 24.0, 10.0)
        assert time.perf_counter() - t0 < 5.0
        assert count == 20000 and extra == 2 and scale == 2.0 ** 38


def test_shipped_library_has_no_diagnostic_switch(pkg):
    """VERDICT r05 item 4.  Every getenv() in csrc/ is either documented in INTEGRATION.md's knob table or compiled only under
    -DLD_DIAG_BUILD (tools/build_variant.sh); the LD_BM_DIAG_* timing experiments need that flag too; and the default library
    holds none of their names.  (The reference reads two environment inputs: src/dfire.rs:239, src/bin/lightdock-rust.rs:89.)"""
    import glob
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    table = set(re.findall(r"^\| `([A-Z0-9_]+)", doc, flags=re.M))
    diag_only = {"LIGHTDOCK_BM_DIAG_IGNORE_ANM", "LIGHTDOCK_BM_HALF_OCCUPANCY", "LIGHTDOCK_ALLOW_ANY_ARCH"}
    assert not (table & diag_only), "diagnostic switches are not knobs of the shipped library"
    seen = set()
    csrc = os.path.join(ROOT, "lightdock-rust_amd", "csrc")
    for path in glob.glob(os.path.join(csrc, "**", "*"), recursive=True):
        if not path.endswith((".cpp", ".hip", ".hpp", ".inc", ".h")):
            continue
        # the preprocessor's view of a default build: text inside `#ifdef LD_DIAG_BUILD ... #else / #endif` is not compiled
        depth, diag_at, compiled = 0, None, []
        for line in open(path, encoding="utf-8", errors="replace"):
            s = line.strip()
            if s.startswith(("#if", "#ifdef", "#ifndef")):
                depth += 1
                if diag_at is None and re.match(r"#\s*ifdef\s+LD_DIAG_BUILD\b", s):
                    diag_at = depth
                    continue
            elif s.startswith("#else") and diag_at == depth:
                diag_at = -depth          # the #else branch of the diagnostic block IS compiled
                continue
            elif s.startswith("#endif"):
                if diag_at is not None and abs(diag_at) == depth:
                    diag_at = None
                depth -= 1
                continue
            if diag_at is None or diag_at < 0:
                compiled.append(line)
        text = "".join(compiled)
        for name in re.findall(r'getenv\(\s*"([A-Za-z0-9_]+)"', text):
            seen.add(name)
            assert name in table, "%s reads $%s, which INTEGRATION.md's knob table does not list" % (os.path.relpath(path, ROOT), name)
        assert "getenv(" not in re.sub(r'getenv\(\s*"[A-Za-z0-9_]+"', "", text), "a getenv() whose name the test cannot read in " + path
    assert {"LIGHTDOCK_DATA", "LIGHTDOCK_DFIRE_KERNEL"} <= seen
    # the guard that keeps the compile-time experiments out of a default build
    bm = open(os.path.join(csrc, "kernels", "dfire_bm.hip")).read()
    used = set(re.findall(r"\bLD_BM_DIAG_[A-Z_]+", bm))
    guard = re.search(r"#if !defined\(LD_DIAG_BUILD\) && \((.*?)\)\n#error", bm, flags=re.S)
    assert guard and used and used <= set(re.findall(r"LD_BM_DIAG_[A-Z_]+", guard.group(1))), "every LD_BM_DIAG_* flag must be in the #error guard"
    blob = open(pkg.LIB_PATH, "rb").read()
    assert b"_DIAG_" not in blob
    for name in diag_only:
        assert name.encode() not in blob, name + " is in the shipped library"
