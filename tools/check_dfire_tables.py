#!/usr/bin/env python3
"""DFIRE typing and binning tables: the reference's literals against the oracle and the product.

Build container only (reads /root/reference/src/dfire.rs; nothing of it is copied).  The real
`data/DCparams` is missing from the reference mount, so DFIRE energies cannot be pinned against
the reference's goldens; what CAN be pinned is everything else the DFIRE model builder does:

  * `r3_to_numerical` (src/dfire.rs:18-46), `ATOMNUMBER` (:56-77), `ATOMRES` (:80-101):
    for EVERY "<RES><ATOM>" key of the reference the DFIRE atom type must equal
    ATOMRES[r3_to_numerical(RES)][ATOMNUMBER[key]] in (a) the oracle's model builder (which reads
    the literal tables restated in oracle/ld_oracle.c) and (b) the product's host model builder
    (`ld_model_from_pdb`, which types atoms in closed form);
  * `DIST_TO_BINS` (:49-53): the oracle's literal copy, the oracle's `dfire_bin` and the
    product's cell LUT + exact steps must give DIST_TO_BINS[idx] - 1 for every index a distance
    inside the cutoff can produce (idx 0..29), and the oracle's copy must equal all 51 literals.

Exit code 0 = all equal.  `tests/test_host_cpu.py::test_dfire_tables_equal_reference` runs this and
skips where /root/reference does not exist (the GPU box)."""
import os
import re
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/src/dfire.rs"


def parse_reference(path=REF):
    text = open(path).read()
    r3 = {m.group(1): int(m.group(2)) for m in re.finditer(r'"([A-Z]{3})"\s*=>\s*(\d+),', text[text.index("fn r3_to_numerical"):text.index("const DIST_TO_BINS")])}
    bins = [int(v) for v in re.search(r"const DIST_TO_BINS: &\[usize\] = &\[(.*?)\];", text, re.S).group(1).replace("\n", " ").split(",") if v.strip()]
    block = re.search(r"static ref ATOMNUMBER.*?hashmap!\[(.*?)\];", text, re.S).group(1)
    atomnumber = {k: int(v) for k, v in re.findall(r'"([A-Z0-9]+)"\s*=>\s*(\d+)', block)}
    rows = re.search(r"static ref ATOMRES: Vec<Vec<usize>> = vec!\[(.*?)\];\s*\}", text, re.S).group(1)
    atomres = [[int(v) for v in r.split(",")] for r in re.findall(r"vec!\[([0-9, ]+)\]", rows)]
    return r3, bins, atomnumber, atomres


def parse_oracle_literals():
    text = open(os.path.join(ROOT, "oracle", "ld_oracle.c")).read()
    bins = [int(v) for v in re.search(r"DIST_TO_BINS\[51\] = \{(.*?)\};", text, re.S).group(1).replace("\n", " ").split(",") if v.strip()]
    rows = re.search(r"DFIRE_ATOMRES\[22\]\[14\] = \{(.*?)\n\};", text, re.S).group(1)
    atomres = [[int(v) for v in r.split(",")] for r in re.findall(r"\{([0-9, ]+)\}", rows)]
    return bins, atomres


def one_atom_per_key_pdb(keys, r3):
    """A PDB with one residue per reference residue name holding exactly the atoms of the keys."""
    lines, serial, resseq = [], 1, 1
    by_res = {}
    for k in keys:
        res = k[:3]
        by_res.setdefault(res, []).append(k[3:])
    order = []
    for res in sorted(by_res, key=lambda r: (r3[r], r)):
        for atom in by_res[res]:
            name = (" " + atom) if len(atom) < 4 else atom
            lines.append("ATOM  %5d %-4s %3s A%4d    %8.3f%8.3f%8.3f  1.00  0.00           %s" %
                         (serial, name, res, resseq, 1.5 * serial, 0.0, 0.0, atom[0]))
            order.append(res + atom)
            serial += 1
        resseq += 1
    return "\n".join(lines) + "\nEND\n", order


def main():
    import numpy as np
    import __graft_entry__ as ge
    r3, bins, atomnumber, atomres = parse_reference()
    problems = []
    if len(bins) != 51 or len(atomres) != 22 or len(r3) != 22:
        problems.append("unexpected table sizes in the reference: %d bins, %d ATOMRES rows, %d residues" % (len(bins), len(atomres), len(r3)))
    o_bins, o_atomres = parse_oracle_literals()
    if o_bins != bins:
        problems.append("oracle DIST_TO_BINS literals differ from src/dfire.rs:49-53")
    if o_atomres != atomres:
        problems.append("oracle DFIRE_ATOMRES literals differ from src/dfire.rs:80-101")

    pdb, order = one_atom_per_key_pdb(sorted(atomnumber), r3)
    want = np.array([atomres[r3[k[:3]]][atomnumber[k]] for k in order])
    with tempfile.NamedTemporaryFile("w", suffix=".pdb", delete=False) as f:
        f.write(pdb)
        path = f.name
    try:
        orc, pkg = ge.oracle(), ge.package()
        cpu = orc.Scorer("dfire", path, path, potential=np.zeros(169 * 169 * 20))
        got_o = np.asarray(cpu.model(0)["dfire_types"])
        got_p = np.asarray(pkg.model_from_pdb("dfire", path)["dfire_types"])
    finally:
        os.unlink(path)
    for who, got in (("oracle model builder", got_o), ("ld_model_from_pdb", got_p)):
        if len(got) != len(want):
            problems.append("%s: %d atoms typed, %d keys" % (who, len(got), len(want)))
        else:
            for k, a, b in zip(order, got, want):
                if a != b:
                    problems.append("%s: %s typed %d, reference %d" % (who, k, a, b))

    lut, steps, _ = pkg.dfire_bin_lut()
    steps = np.append(steps, np.inf)
    for idx in range(30):           # d = sqrt(d2)*2-1 in [idx, idx+1): r = (idx + 1.5) / 2; idx 29 only at r = 15.0
        d2 = ((idx + 1.5) / 2.0) ** 2 if idx < 29 else 225.0
        ref = bins[idx] - 1
        if orc.dfire_bin(d2) != ref:
            problems.append("oracle dfire_bin(%g) = %d, DIST_TO_BINS[%d]-1 = %d" % (d2, orc.dfire_bin(d2), idx, ref))
        b = int(lut[int(d2 * 4.0)])
        b += 1 if d2 >= steps[b + 1] else 0
        if b != ref:
            problems.append("product LUT bin(%g) = %d, DIST_TO_BINS[%d]-1 = %d" % (d2, b, idx, ref))
    if problems:
        print("\n".join(problems))
        return 1
    print("DFIRE tables equal the reference: %d atom keys x 2 builders, 51 + 30 bin entries, %d x 14 ATOMRES" % (len(order), len(atomres)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
