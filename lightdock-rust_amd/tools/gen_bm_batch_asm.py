#!/usr/bin/env python3
"""Writes csrc/kernels/dfire_bm_batch.inc: the 64 pair slots of one batch of dfire_bm_pairs (kernels/dfire_bm.hip) as ONE
block of gfx950 instructions, software-pipelined by hand in 8 stages of 8 slots.

A slot = (ligand atom i, receptor atom j) of the block, lane = pose:
    E = (Rs_j - l2_i) + Rz_j lz_i + Ry_j ly_i + Rx_j lx_i        packed f32, two receptor atoms (a pair record) per instruction
    cell = (u32)E                                                  v_cvt_u32_f32 saturates: beyond the LUT -> cell 0 = "miss"
    code = lut[cell]                                               ds_read_u8, the LUT sits at LDS address 0
    value = cube[row(i, j)][code]                                  ds_read_b64; the row's LDS address is an instruction constant
    acc += value                                                   v_lshl_add_u64: 64-bit fixed point, exact and order-free
Stage h (receptor pair record g = h / 2, ligand atoms 4 (h % 2) .. + 3):  A(h) cells and LUT reads; C(h - 2) the adds of the
table values requested one stage ago; B(h - 1) the table reads of the codes requested one stage ago.  One s_waitcnt per
stage (LDS results return in order: when the table values of stage h - 2 are in, so are the codes of stage h - 1), instead
of the one per add the compiler's schedule had.  Written by hand because at two waves per SIMD every instruction a wave
issues, scalar or wait, costs it a turn.

Temporaries are fixed registers (clobbered): v[220:235] table values, v[236:243] / v[244:251]
cells and codes, v[252:255] the two packed E.  Operands: acc, acc1 (outputs, 64 bit: the sums over the pairs with the even / odd receptor atom of a record), Rs0..3 Rz0..3 Ry0..3 Rx0..3 (the block's receptor
records, wave-uniform: scalar register pairs), l2/lz/ly/lx0..3 = the values of the lane's ligand atoms (2p, 2p + 1), cube = LDS address of the wave's cube.
"""
import os

ROW = 176
T0 = 220   # first of the block's 36 fixed temporaries (tools/microbench/gen_mfma_batch.py moves them for its three-waves-per-SIMD form)


def stage_a(h):
    g, i0, cs = h // 2, 4 * (h % 2), T0 + 16 + 8 * (h % 2)
    e0, e1, e2, e3 = T0 + 32, T0 + 33, T0 + 34, T0 + 35
    ea, eb = "v[%d:%d]" % (e0, e1), "v[%d:%d]" % (e2, e3)
    out = []
    for t in range(2):
        i = i0 + 2 * t
        # ligand atoms i, i + 1 = the two halves of the pair registers l2/lz/ly/lx[i / 2]: op_sel broadcasts the half
        p = i // 2
        out += [
            "v_pk_add_f32 %s, %%[rs%d], %%[l2%d] op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" % (ea, g, p),
            "v_pk_add_f32 %s, %%[rs%d], %%[l2%d] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" % (eb, g, p),
            "v_pk_fma_f32 %s, %%[rz%d], %%[lz%d], %s op_sel:[0,0,0] op_sel_hi:[1,0,1]" % (ea, g, p, ea),
            "v_pk_fma_f32 %s, %%[rz%d], %%[lz%d], %s op_sel:[0,1,0] op_sel_hi:[1,1,1]" % (eb, g, p, eb),
            "v_pk_fma_f32 %s, %%[ry%d], %%[ly%d], %s op_sel:[0,0,0] op_sel_hi:[1,0,1]" % (ea, g, p, ea),
            "v_pk_fma_f32 %s, %%[ry%d], %%[ly%d], %s op_sel:[0,1,0] op_sel_hi:[1,1,1]" % (eb, g, p, eb),
            "v_pk_fma_f32 %s, %%[rx%d], %%[lx%d], %s op_sel:[0,0,0] op_sel_hi:[1,0,1]" % (ea, g, p, ea),
            "v_pk_fma_f32 %s, %%[rx%d], %%[lx%d], %s op_sel:[0,1,0] op_sel_hi:[1,1,1]" % (eb, g, p, eb),
        ]
        out += ["v_cvt_u32_f32 v%d, v%d" % (cs + 4 * t + k, e0 + k) for k in range(4)]
    out += ["ds_read_u8 v%d, v%d" % (cs + k, cs + k) for k in range(8)]
    return out


def stage_b(h):
    g, i0, cs, ts = h // 2, 4 * (h % 2), T0 + 16 + 8 * (h % 2), T0
    out = []
    for k in range(8):
        row = (i0 + k // 2) * 8 + 2 * g + (k % 2)
        out.append("ds_read_b64 v[%d:%d], v%d offset:%%c[cube]+%d" % (ts + 2 * k, ts + 2 * k + 1, cs + k, row * ROW))
    return out


def stage_c(h):
    # two running sums, taken in turn (acc: the pairs with receptor atom 2g, acc1: 2g + 1): an add depends on the one two before it,
    # and each sum covers 32 pairs only -- one more bit for the fixed point under the markers
    # (a sum's FIRST add takes 0 as its addend: the sums are outputs of the statement, no two v_mov_b64 in front of every batch)
    ts = T0   # (one set: the adds of stage h - 2 are issued before the reads of stage h - 1 that overwrite it)
    out = []
    for k in range(8):
        acc = "%[acc]" if k % 2 == 0 else "%[acc1]"
        out.append("v_lshl_add_u64 %s, v[%d:%d], 0, %s" % (acc, ts + 2 * k, ts + 2 * k + 1, "0" if h == 0 and k < 2 else acc))
    return out


def pose_block():
    """Two of the lane's 8 ligand atoms posed at a time: l = A x + t' (the affine map of bm_apply, same operations and nesting,
    so the culling kernel's boxes and the exact path's re-evaluation see the same bits), and |l|^2.  Operands: A0xy = {r00, r01},
    A0zw = {r02, tx - cx}, ... (the lane's map, the translation already relative to the block's centre); X, Y, Z = the local
    coordinates of atoms (2p, 2p + 1), wave-uniform; outputs LX, LY, LZ, L2 = the two atoms' values.  One statement per pair of
    atoms: all four at once had 68 registers live and spilled."""
    lines = []
    for c, row in (("x", 0), ("y", 1), ("z", 2)):
        d = "%%[l%s]" % c
        # t = fma(r?2, z, t');  t = fma(r?1, y, t);  t = fma(r?0, x, t)
        lines.append("v_pk_fma_f32 %s, %%[a%dzw], %%[z], %%[a%dzw] op_sel:[0,0,1] op_sel_hi:[0,1,1]" % (d, row, row))
        lines.append("v_pk_fma_f32 %s, %%[a%dxy], %%[y], %s op_sel:[1,0,0] op_sel_hi:[1,1,1]" % (d, row, d))
        lines.append("v_pk_fma_f32 %s, %%[a%dxy], %%[x], %s op_sel:[0,0,0] op_sel_hi:[0,1,1]" % (d, row, d))
    lines.append("v_pk_mul_f32 %[l2], %[lz], %[lz]")
    lines.append("v_pk_fma_f32 %[l2], %[ly], %[ly], %[l2]")
    lines.append("v_pk_fma_f32 %[l2], %[lx], %[lx], %[l2]")
    outs = ['[%s] "=&v"(%s)' % (name, arr) for name, arr in (("lx", "LX"), ("ly", "LY"), ("lz", "LZ"), ("l2", "L2"))]
    ins = ['[a%d%s] "v"(A%d%s)' % (r, h, r, h) for r in range(3) for h in ("xy", "zw")]
    # (X, Y, Z: wave-uniform, SCALAR register pairs -- read back from LDS per batch they were an LDS round trip at the head of every batch)
    ins += ['[%s] "s"(%s)' % (c, arr) for c, arr in (("x", "X"), ("y", "Y"), ("z", "Z"))]
    return lines, outs, ins


def pose_flex_block():
    """pose_block for molecules that flex (src/dfire.rs:288-301): the same three chains, then + (DX, DY, DZ) -- the deformation of the two
    atoms, formed by the caller -- and only then |l|^2.  (R x + t') + d: the order bm_recheck repeats."""
    lines, outs, ins = pose_block()
    at = [i for i, l in enumerate(lines) if l.startswith("v_pk_mul_f32 %[l2]")][0]
    lines = lines[:at] + ["v_pk_add_f32 %[lx], %[lx], %[dx]", "v_pk_add_f32 %[ly], %[ly], %[dy]", "v_pk_add_f32 %[lz], %[lz], %[dz]"] + lines[at:]
    ins = ins + ['[dx] "v"(DX)', '[dy] "v"(DY)', '[dz] "v"(DZ)']
    return lines, outs, ins


def flex_block():
    """The deformation of one subtile's 8 atoms for the lane's pose (src/dfire.rs:288-320): D[pc] = sum_k amplitude_k x mode_k of the
    atoms (2p, 2p + 1)'s coordinate c, pc = 3 p + c -- twelve packed sums of ten terms.  The subtile's modes lie in LDS in this
    order (BmModel: ((pc * 10 + k) * 2 + atom of the pair) floats), sixty 16-byte broadcast reads of two modes each; EIGHT are
    kept in flight in the fixed registers v[220:251] (the batch's temporaries, free here), every wait counted.  (As C++ the
    compiler, out of registers, issued each read, waited for it and only then multiplied: 120 LDS round trips a batch, and
    the ANM form ran at 6.4 us a batch against the rigid form's 2.4.)  The first term is a product, the others fused
    multiply-adds in mode order: bm_recheck repeats exactly this.  Operands: D0..D11 (outputs), A0..A4 (the molecule's
    amplitudes two by two), MODES (VGPR: the LDS address of the subtile's modes)."""
    RING, N = 8, 60
    reg = lambda i: "v[%d:%d]" % (220 + 4 * (i % RING), 223 + 4 * (i % RING))
    lo = lambda i: "v[%d:%d]" % (220 + 4 * (i % RING), 221 + 4 * (i % RING))
    hi = lambda i: "v[%d:%d]" % (222 + 4 * (i % RING), 223 + 4 * (i % RING))
    # (the counted waits below are right only if nothing else is outstanding on lgkmcnt at entry -- scalar loads return out of
    # order and share the counter; the block's scalar loads of rec_ops are not inputs of the FIRST statement: wait for all here)
    lines = ["s_waitcnt lgkmcnt(0)"] + ["ds_read_b128 %s, %%[modes] offset:%d" % (reg(i), 16 * i) for i in range(RING)]
    for i in range(N):
        issued = min(N, i + RING)
        lines.append("s_waitcnt lgkmcnt(%d)" % (issued - i - 1))
        pc, k2 = i // 5, i % 5
        d, a = "%%[d%d]" % pc, "%%[a%d]" % k2
        if k2 == 0:
            lines.append("v_pk_mul_f32 %s, %s, %s op_sel:[0,0] op_sel_hi:[0,1]" % (d, a, lo(i)))
        else:
            lines.append("v_pk_fma_f32 %s, %s, %s, %s op_sel:[0,0,0] op_sel_hi:[0,1,1]" % (d, a, lo(i), d))
        lines.append("v_pk_fma_f32 %s, %s, %s, %s op_sel:[1,0,0] op_sel_hi:[1,1,1]" % (d, a, hi(i), d))
        if i + RING < N:
            lines.append("ds_read_b128 %s, %%[modes] offset:%d" % (reg(i), 16 * (i + RING)))
    outs = ['[d%d] "=&v"(D[%d])' % (k, k) for k in range(12)]
    ins = ['[a%d] "v"(A[%d])' % (k, k) for k in range(5)] + ['[modes] "v"(MODES)']
    clobbers = ['"v%d"' % r for r in range(220, 252)]
    return lines, outs, ins, clobbers


def dma_block():
    """A block's 64 table rows L2 -> LDS: 13 LDS-DMA instructions of five rows each (55 lanes: row of the five, one of the row's
    11 pieces of 16 bytes).  Lane r of SRC holds the table offset of row r; the lane that copies (row, piece) of instruction t
    fetches it with ds_bpermute -- ALL THIRTEEN in flight, one wait, then the thirteen copies.  (Written as a loop in C++ the
    compiler made thirteen rounds of bpermute / s_waitcnt lgkmcnt(0) / copy, each a full LDS latency: 1.7 of a block set-up's
    2.5 us.)  Operands: ROWSEL = 4 * (row of the five), PIECE = 16 * piece (constants of the lane), SRC, TABLE (64-bit, uniform:
    the row table), CUBE (uniform: the LDS address of the wave's cube), M55 / M44 (the lanes of an instruction: 55, the last
    one's 44 -- its fifth row would be row 64).  m0 and exec are saved and restored inside the statement (m0 is a reserved register: naming it
    as a clobber is a warning and "may not be preserved"), scc -- set by s_add_u32 -- is a named clobber."""
    # (the bpermutes with every lane active: a lane that is switched off pushes nothing, and reading it returns 0)
    lines = []
    for t in range(13):
        lines.append("ds_bpermute_b32 %%[t%d], %%[rowsel], %%[src] offset:%d" % (t, 20 * t))
    lines += ["s_mov_b64 %[save], exec", "s_mov_b32 %[savem0], m0", "s_mov_b64 exec, %[m55]", "s_waitcnt lgkmcnt(0)"]
    for t in range(13):
        if t == 12:
            lines.append("s_mov_b64 exec, %[m44]")
        lines.append("v_add_u32 %%[t%d], %%[t%d], %%[piece]" % (t, t))
        lines.append("s_add_u32 m0, %%[cube], %d" % (880 * t))
        lines.append("s_nop 0")
        lines.append("global_load_lds_dwordx4 %%[t%d], %%[table]" % t)
    lines.append("s_mov_b64 exec, %[save]")
    lines.append("s_mov_b32 m0, %[savem0]")
    outs = ['[save] "=&s"(SAVE)', '[savem0] "=&s"(SAVEM0)'] + ['[t%d] "=&v"(TMP[%d])' % (t, t) for t in range(13)]
    ins = ['[rowsel] "v"(ROWSEL)', '[src] "v"(SRC)', '[piece] "v"(PIECE)', '[table] "s"(TABLE)', '[cube] "s"(CUBE)', '[m55] "s"(M55)', '[m44] "s"(M44)']
    return lines, outs, ins


def main():
    lines = []
    for h in range(10):
        if h < 8:
            lines += stage_a(h)
        if h >= 1:
            lines.append("s_waitcnt lgkmcnt(%d)" % (8 if h < 8 else 0))
        if h >= 2:
            lines += stage_c(h - 2)
        if 1 <= h <= 8:
            lines += stage_b(h - 1)
    body = " \\\n".join('    "%s\\n\\t"' % l for l in lines)
    def operands(r_constraint):
        ops = []
        for name in ("rs", "rz", "ry", "rx"):
            # the block's receptor operands are wave-uniform: SCALAR register pairs (a packed instruction takes one as its first source).
            # As vector registers the compiler kept the raw records in scalar registers and RECOMPUTED all sixteen operands in every
            # batch (36 packed instructions + the box centre: a ninth of a batch's vector instructions) rather than hold 32 registers.
            # (The form with vector registers, LD_BM_BATCH_ASM_V, is what a block-major kernel for molecules that flex per pose would
            # need -- receptor atoms differ per lane; it exists for the timing experiment LD_BM_DIAG_ANM_COST only.)
            ops += ['[%s%d] "%s"(%s[%d])' % (name, g, r_constraint, {"rs": "Rs", "rz": "Rz", "ry": "Ry", "rx": "Rx"}[name], g) for g in range(4)]
        for name, arr in (("l2", "L2"), ("lz", "LZ"), ("ly", "LY"), ("lx", "LX")):
            ops += ['[%s%d] "v"(%s[%d])' % (name, p, arr, p) for p in range(4)]
        return ops
    ops_in = operands("s")
    clobbers = ", ".join('"v%d"' % r for r in range(220, 256))
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(here, "..", "csrc", "kernels", "dfire_bm_batch.inc")
    with open(path, "w") as f:
        f.write("// GENERATED by lightdock-rust_amd/tools/gen_bm_batch_asm.py -- do not edit; the generator's docstring explains the schedule.\n")
        f.write("// %d instructions: %d vector, %d LDS, %d waits.\n" % (len(lines), sum(l.startswith("v_") for l in lines), sum(l.startswith("ds_") for l in lines), sum(l.startswith("s_waitcnt") for l in lines)))
        for macro, ops in (("LD_BM_BATCH_ASM", ops_in), ("LD_BM_BATCH_ASM_V", operands("v"))):
            f.write("#define %s(SUM0, SUM1, Rs, Rz, Ry, Rx, L2, LZ, LY, LX, CUBE) \\\n  asm volatile( \\\n" % macro + body + " \\\n")
            f.write('    : [acc] "=&v"(SUM0), [acc1] "=&v"(SUM1) \\\n    : ' + ", \\\n      ".join(ops) + ', \\\n      [cube] "n"(CUBE) \\\n')
            f.write("    : " + clobbers + ', "memory")\n\n')
        plines, pouts, pins = pose_block()
        f.write("\n// two of the lane's 8 ligand atoms posed: %d packed instructions\n" % len(plines))
        f.write("#define LD_BM_POSE_ASM(LX, LY, LZ, L2, A0xy, A0zw, A1xy, A1zw, A2xy, A2zw, X, Y, Z) \\\n  asm( \\\n")
        f.write(" \\\n".join('    "%s\\n\\t"' % l for l in plines) + " \\\n")
        f.write("    : " + ", \\\n      ".join(pouts) + " \\\n    : " + ", \\\n      ".join(pins) + ")\n")
        flines, fouts, fins = pose_flex_block()
        f.write("\n// the same for a molecule that flexes: + the two atoms' deformation, then |l|^2 (%d packed instructions)\n" % len(flines))
        f.write("#define LD_BM_POSE_FLEX_ASM(LX, LY, LZ, L2, A0xy, A0zw, A1xy, A1zw, A2xy, A2zw, X, Y, Z, DX, DY, DZ) \\\n  asm( \\\n")
        f.write(" \\\n".join('    "%s\\n\\t"' % l for l in flines) + " \\\n")
        f.write("    : " + ", \\\n      ".join(fouts) + " \\\n    : " + ", \\\n      ".join(fins) + ")\n")
        xlines, xouts, xins, xclob = flex_block()
        f.write("\n// a subtile's deformation for the lane's pose: 60 LDS reads, eight in flight, 120 packed multiply-adds (%d instructions)\n" % len(xlines))
        f.write("#define LD_BM_FLEX_ASM(D, A, MODES) \\\n  asm volatile( \\\n")
        f.write(" \\\n".join('    "%s\\n\\t"' % l for l in xlines) + " \\\n")
        f.write("    : " + ", \\\n      ".join(xouts) + " \\\n    : " + ", \\\n      ".join(xins) + " \\\n    : " + ", ".join(xclob) + ")\n")
        dlines, douts, dins = dma_block()
        f.write("\n// a block's 64 table rows -> the wave's cube: 13 ds_bpermute in flight, one wait, 13 LDS-DMA copies\n")
        f.write("#define LD_BM_DMA_ASM(SAVE, SAVEM0, TMP, ROWSEL, SRC, PIECE, TABLE, CUBE, M55, M44) \\\n  asm volatile( \\\n")
        f.write(" \\\n".join('    "%s\\n\\t"' % l for l in dlines) + " \\\n")
        # (s_add_u32 sets scc: named, or the compiler may keep a compare's result live across the statement; m0: saved and restored)
        f.write("    : " + ", \\\n      ".join(douts) + " \\\n    : " + ", \\\n      ".join(dins) +  ' \\\n    : "memory", "scc")\n')
    print("wrote", os.path.normpath(path), len(lines), "+", len(plines), "instructions")


if __name__ == "__main__":
    main()
