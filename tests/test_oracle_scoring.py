"""Oracle vs the reference's scorer known-answer tests and committed example outputs."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, case_kwargs, case_positions, parse_gso


def test_dna_1azp_known_answer(orc):
    """src/dna.rs:538-572: identity pose of tests/1azp -> -364.88126358158974, assert_eq!."""
    d = os.path.join(GOLDEN, "unit", "1azp")
    s = orc.Scorer("dna", os.path.join(d, "1azp_receptor.pdb"), os.path.join(d, "1azp_ligand.pdb"))
    assert s.num_atoms(0) == 1094 and s.num_atoms(1) == 506
    assert s.energy([0.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0]) == -364.88126358158974


def test_dna_gso1_energies_1azp(orc, table):
    """gso_1.out = the 200 starting poses with their energies (nobody moves at step 1)."""
    method, rec, lig, kw = case_kwargs("1azp", orc, table)
    s = orc.Scorer(method, rec, lig, **kw)
    poses = case_positions("1azp", orc)
    _, _, nn, vis, sco = parse_gso(os.path.join(GOLDEN, "1azp", "swarm_0", "gso_1.out"))
    got = s.energy_rows(poses[:40])
    assert np.all(np.abs(got - sco[:40]) <= 0.5000001e-8 + 1e-12 * np.abs(sco[:40]))
    assert np.all(nn == 0) and np.allclose(vis, 0.6)


@pytest.mark.timeout(600)
def test_gso_replay_1azp_files_identical(orc, tmp_path):
    """The oracle CLI reproduces the committed example/1azp/swarm_0/gso_{1,10,20}.out byte for byte
    (DNA + ANM + restraints + StdRng + GSO + output format).  20 steps keep the CPU suite short;
    the 100-step replay is also byte-identical."""
    src = os.path.join(GOLDEN, "1azp")
    for f in ("rec_nm.npy", "lig_nm.npy"):
        shutil.copy(os.path.join(src, f), tmp_path)
    orc.lib()
    r = subprocess.run([orc.CLI_PATH, os.path.join(src, "setup.json"), os.path.join(src, "initial_positions_0.dat"),
                        "20", "dna"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Creating GSO with 200 glowworms" in r.stdout and "Starting optimization (20 steps)" in r.stdout
    for step in (1, 10, 20):
        want = open(os.path.join(src, "swarm_0", "gso_%d.out" % step), "rb").read()
        got = open(os.path.join(tmp_path, "swarm_0", "gso_%d.out" % step), "rb").read()
        assert got == want, "gso_%d.out differs" % step


def test_cli_usage_errors_exit_zero(orc, tmp_path):
    """src/bin/lightdock-rust.rs:101,112,142: usage errors go to stderr, exit status 0."""
    orc.lib()
    r = subprocess.run([orc.CLI_PATH], capture_output=True, text=True)
    assert r.returncode == 0 and "Wrong command line" in r.stderr
    r = subprocess.run([orc.CLI_PATH, "a", "b", "x", "dfire"], capture_output=True, text=True)
    assert r.returncode == 0 and "steps argument must be a number" in r.stderr
    r = subprocess.run([orc.CLI_PATH, "a", "b", "1", "zrank"], capture_output=True, text=True)
    assert r.returncode == 0 and "method not supported" in r.stderr
    r = subprocess.run([orc.CLI_PATH, "nope.json", "initial_positions_0.dat", "1", "dna"], capture_output=True, text=True,
                       cwd=tmp_path)
    assert r.returncode == 0 and "Error reading setup file" in r.stderr


def test_dfire_bins_follow_dist_to_bins(orc):
    """src/dfire.rs:49-53,336-337 incl. the r = 15.0 A corner that reads bin 20."""
    assert orc.dfire_bin(0.0) == 0 and orc.dfire_bin(0.09 ** 2) == 0      # negative d saturates to index 0
    assert orc.dfire_bin(1.99 ** 2) == 0 and orc.dfire_bin(2.0 ** 2) == 1
    assert orc.dfire_bin(7.99 ** 2) == 12 and orc.dfire_bin(8.0 ** 2) == 13
    assert orc.dfire_bin(14.99 ** 2) == 19 and orc.dfire_bin(225.0) == 20


def test_dfire_typing_of_fixtures(orc, table):
    """Every atom of the DFIRE fixtures maps to a type 0..167; 1k4c's 453 MMB beads are type 167."""
    for name, n_rec, n_lig in (("1ppe", 1615, 221), ("1k4c", 3413, 3268), ("2uuy", None, None)):
        method, rec, lig, kw = case_kwargs(name, orc, table)
        s = orc.Scorer(method, rec, lig, **kw)
        if n_rec:
            assert (s.num_atoms(0), s.num_atoms(1)) == (n_rec, n_lig)
        for side in (0, 1):
            m = s.model(side)
            assert m["dfire_types"].max() <= 167
        if name == "1k4c":
            m = s.model(0)
            assert len(m["membrane"]) == 453 and np.all(m["dfire_types"][m["membrane"]] == 167)
        if name == "1ppe":
            m = s.model(0)
            assert len(m["restraint_offsets"]) == 2 and m["restraint_offsets"][1] == 8   # E.ILE.16: 8 heavy atoms


def test_dfire_synthetic_table_energy_is_deterministic(orc, table):
    method, rec, lig, kw = case_kwargs("1ppe", orc, table)
    s = orc.Scorer(method, rec, lig, **kw)
    poses = case_positions("1ppe", orc)
    e, stats = s.energy_ex_row(poses[0])
    assert e == s.energy_row(poses[0])
    assert 20000 < stats[5] < 45000            # in-cutoff pairs of a starting pose (mean 31 299)
    raw = stats[0]
    score = (raw * 0.0157 - 4.7) * -1.0
    assert e == score + stats[2] * score + stats[3] * score


def test_dfire_energy_against_a_vectorised_numpy_restatement(orc, table):
    """A second, differently written restatement of src/dfire.rs:265-362 (numpy, all pairs at
    once, the sqrt-based bin of the reference with DIST_TO_BINS spelled out from its rule) must
    agree with the C oracle on real fixtures: guards the oracle itself, since no DFIRE golden can
    be reproduced without the real table."""
    dist_to_bins = [1, 1, 1] + [i - 1 for i in range(3, 15)] + [14 + (i - 15) // 2 for i in range(15, 49)] + [31, 32]
    assert len(dist_to_bins) == 51 and dist_to_bins[29] == 21 and dist_to_bins[28] == 20      # src/dfire.rs:49-53
    for name in ("1ppe", "1k4c"):
        method, rec, lig, kw = case_kwargs(name, orc, table)
        s = orc.Scorer(method, rec, lig, **kw)
        mr, ml = s.model(0), s.model(1)
        poses = case_positions(name, orc)[:3]
        for p in poses:
            t, q = p[:3], p[3:7]
            w, x, y, z = q
            n2 = w * w + x * x + y * y + z * z                                   # rotate divides by |q|^2, src/qt.rs:48-61
            R = np.array([[w*w+x*x-y*y-z*z, 2*(x*y-w*z), 2*(x*z+w*y)],
                          [2*(x*y+w*z), w*w-x*x+y*y-z*z, 2*(y*z-w*x)],
                          [2*(x*z-w*y), 2*(y*z+w*x), w*w-x*x-y*y+z*z]]) / n2
            lc = ml["coordinates"] @ R.T + t
            d2 = ((mr["coordinates"][:, None, :] - lc[None, :, :]) ** 2).sum(-1)
            hit = d2 <= 225.0
            d = np.sqrt(d2[hit]) * 2.0 - 1.0
            idx = np.maximum(d, 0.0).astype(np.int64)                            # `d as usize` saturates below 0
            bins = np.array(dist_to_bins)[idx] - 1
            ti = np.broadcast_to(mr["dfire_types"][:, None], d2.shape)[hit].astype(np.int64)
            tj = np.broadcast_to(ml["dfire_types"][None, :], d2.shape)[hit].astype(np.int64)
            raw = table[ti * 169 * 20 + tj * 20 + bins].sum()                      # src/dfire.rs:338
            score = (raw * 0.0157 - 4.7) * -1.0
            iface = hit.copy()
            iface[hit] = d <= 3.9
            rec_if = set(np.flatnonzero(iface.any(1)).tolist())
            offs, atoms = mr["restraint_offsets"], mr["restraint_atoms"]
            groups = len(offs) - 1
            sat = sum(1 for g in range(groups) if rec_if & set(atoms[offs[g]:offs[g + 1]].tolist()))
            perc = sat / groups if groups else 0.0
            beads = mr["membrane"]
            pen = 999.0 * (len(rec_if & set(beads.tolist())) / len(beads)) if len(beads) else 0.0
            want = score + perc * score - pen
            got = s.energy_row(p)
            assert abs(got - want) <= 1e-9 * max(1.0, abs(want)), (name, got, want)


def test_unsupported_atoms_are_errors(orc, table, tmp_path):
    """src/dfire.rs:43,180 panics -> constructor errors."""
    bad = tmp_path / "bad.pdb"
    bad.write_text("ATOM      1  N   XYZ A   1      11.104  13.207   2.100  1.00  0.00           N\n")
    good = os.path.join(GOLDEN, "1ppe", "lightdock_1ppe_i.pdb")
    with pytest.raises(RuntimeError, match="Residue name not supported"):
        orc.Scorer("dfire", str(bad), good, potential=table)
    bad.write_text("ATOM      1  H1  ALA A   1      11.104  13.207   2.100  1.00  0.00           H\n")
    with pytest.raises(RuntimeError, match="Not supported atom type"):
        orc.Scorer("dfire", str(bad), good, potential=table)


def test_real_dcparams_goldens(orc, real_dcparams):
    """Opt-in: with the real table the oracle must hit src/dfire.rs:415 and the leaked values
    of src/dfire.rs:370-380."""
    t = orc.load_dcparams(real_dcparams)
    assert t[0] == 10.0 and t[2] == -0.624030868 and t[4998] == -0.0458685914 and t[168 * 168 * 20 - 1] == 0.0
    d = os.path.join(GOLDEN, "unit", "2oob")
    s = orc.Scorer("dfire", os.path.join(d, "2oob_receptor.pdb"), os.path.join(d, "2oob_ligand.pdb"), potential=t)
    assert s.energy([0.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0]) == 16.7540569503498
    # step-1 files of the DFIRE examples: the "Scoring" column is the energy of the initial poses
    # (src/lib.rs:51, src/swarm.rs:128-167), printed with 8 decimals
    from conftest import case_kwargs, case_positions, parse_gso
    for name in ("1ppe", "1k4c", "2uuy", "ab_icode"):
        method, rec, lig, kw = case_kwargs(name, orc, t)
        poses = case_positions(name, orc)
        want = parse_gso(os.path.join(GOLDEN, name, "swarm_0", "gso_1.out"))[4]
        got = orc.Scorer(method, rec, lig, **kw).energy_rows(poses[:40])
        assert np.all(np.abs(got - want[:40]) <= 1.01e-8 + 1e-9 * np.abs(want[:40])), name


def test_oracle_cli_on_the_multi_swarm_example_1czy(orc, pkg, tmp_path):
    """example/1czy (ten swarms, DFIRE + ANM + an active receptor restraint): the oracle CLI runs two of
    its swarms from init/initial_positions_<i>.dat; the step-1 file must carry the start poses moved once
    and the energies of the start poses, which the scorer API gives too.  With the real DCparams
    ($LIGHTDOCK_DATA) the files must also equal the reference's own swarm_<i>/gso_1.out."""
    src = os.path.join(GOLDEN, "1czy")
    (tmp_path / "data").mkdir()
    real = os.environ.get("LIGHTDOCK_DATA")
    real = os.path.join(real, "DCparams") if real else None
    if real and os.path.exists(real):
        shutil.copy(real, tmp_path / "data" / "DCparams")
        table = orc.load_dcparams(real)
    else:
        real = None
        pkg.synth.write_dcparams(str(tmp_path / "data" / "DCparams"))
        table = pkg.synth.dcparams()
    for f in ("rec_nm.npy", "lig_nm.npy"):
        shutil.copy(os.path.join(src, f), tmp_path)
    scorer = orc.Scorer("dfire", os.path.join(src, "lightdock_1czy_protein.pdb"), os.path.join(src, "lightdock_1czy_peptide.pdb"),
                        rec_active=["A.SER.467"], rec_nmodes=np.load(os.path.join(src, "rec_nm.npy")), rec_num_anm=10,
                        lig_nmodes=np.load(os.path.join(src, "lig_nm.npy")), lig_num_anm=10, use_anm=True, potential=table)
    assert scorer.num_atoms(0) == 1281 and scorer.num_atoms(1) == 53
    from conftest import parse_gso
    for s in (0, 7):
        init = os.path.join(src, "init", "initial_positions_%d.dat" % s)
        r = subprocess.run([orc.CLI_PATH, os.path.join(src, "setup.json"), init, "1", "dfire"], cwd=tmp_path, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        got = parse_gso(str(tmp_path / ("swarm_%d" % s) / "gso_1.out"))
        poses = orc.parse_positions(init)
        assert poses.shape == (200, 27)
        want = scorer.energy_rows(poses[:25])
        assert np.all(np.abs(got[4][:25] - want) <= 0.51e-8 + 1e-11 * np.abs(want))
        if real:
            ref = parse_gso(os.path.join(src, "swarm_%d" % s, "gso_1.out"))
            assert np.array_equal(got[2], ref[2])
            for x, y in zip(got, ref):
                assert np.allclose(x, y, rtol=0, atol=2e-7)


def test_pydock_known_answer_and_generic_fallback(orc, tmp_path):
    """src/pydock.rs:553-587: same golden as DNA; src/pydock.rs:332-345: unknown atoms are typed
    by the first letter of their name, H1/H2/H3 still only fall back to "<res>-H"."""
    d = os.path.join(GOLDEN, "unit", "1azp")
    rec, lig = os.path.join(d, "1azp_receptor.pdb"), os.path.join(d, "1azp_ligand.pdb")
    s = orc.Scorer("pydock", rec, lig)
    assert s.energy([0.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0]) == -364.88126358158974
    odd = tmp_path / "odd.pdb"
    odd.write_text("ATOM      1  CQ1 LIG A   1      11.104  13.207   2.100  1.00  0.00           C\n"
                   "ATOM      2  F7  LIG A   1      12.104  13.207   2.100  1.00  0.00           F\n"
                   "ATOM      3  S1  LIG A   1      12.104  14.207   2.100  1.00  0.00           S\n")
    s = orc.Scorer("pydock", str(odd), lig)
    m = s.model(0)
    assert list(m["ele_charges"]) == [0.5973, -0.342, -0.2737]
    assert list(m["vdw_charges"]) == [0.086, 0.061, 0.25] and list(m["vdw_radii"]) == [1.908, 1.75, 2.0]
    with pytest.raises(RuntimeError, match=r'DNA Error: Atom \["LIG-CQ1"\] not supported'):
        orc.Scorer("dna", str(odd), lig)
    odd.write_text("ATOM      1  XX  LIG A   1      11.104  13.207   2.100  1.00  0.00           X\n")
    with pytest.raises(RuntimeError, match=r'PYDOCK Error: Atom \["\*-X"\] not supported'):
        orc.Scorer("pydock", str(odd), lig)
    odd.write_text("ATOM      1  H2  LIG A   1      11.104  13.207   2.100  1.00  0.00           H\n")
    with pytest.raises(RuntimeError, match=r'PYDOCK Error: Atom \["LIG-H"\] not supported'):
        orc.Scorer("pydock", str(odd), lig)
