"""Where a wave of the tiled DFIRE kernel spends its time (s_memtime stamps summed over all waves).
Needs lib/variants/stamps.so: bash tools/build_variant.sh stamps -DLD_PACKED_STAMPS (here), then run this on the GPU box."""
import sys, os, ctypes as C, numpy as np, shutil
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, root)
L = os.path.join(root, "lightdock-rust_amd", "lib")
shutil.copy(os.path.join(L, "liblightdock_hip.so"), "/tmp/keep.so")
shutil.copy(os.path.join(L, "variants", "stamps.so"), os.path.join(L, "liblightdock_hip.so"))
try:
    import __graft_entry__ as ge
    import torch; torch.cuda.init()
    pkg, orc = ge.package(), ge.oracle(); pkg.init(0)
    g = os.path.join(ge.GOLDEN, "1k4c")
    s = pkg.Scorer.from_pdb("dfire", os.path.join(g, "lightdock_receptor_membrane.pdb"), os.path.join(g, "lightdock_ligand.pdb"), potential=pkg.synth.dcparams())
    base = orc.parse_positions(os.path.join(g, "initial_positions_0.dat"))[:, :7]
    poses = pkg.synth.jitter(base, 8192, seed=1000)
    lib = pkg.load_library()
    out = (C.c_ulonglong * 8)()
    s.energy_batch(poses)
    lib.ld_debug_stamps(out); a = np.array(out[:], dtype=np.float64)
    for _ in range(3): s.energy_batch(poses)
    lib.ld_debug_stamps(out); b = np.array(out[:], dtype=np.float64)
    d = (b - a) / 3
    waves = d[5]
    print("waves %d  cycles/wave total %.0f  setup %.0f (%.1f%%)  dma+subbox %.0f (%.1f%%)  block loop %.0f (%.1f%%)  other %.0f (%.1f%%); tiles/wave %.1f; per tile: dma %.0f loop %.0f" % (
        waves, d[0]/waves, d[1]/waves, 100*d[1]/d[0], d[2]/waves, 100*d[2]/d[0], d[3]/waves, 100*d[3]/d[0], (d[0]-d[1]-d[2]-d[3])/waves, 100*(d[0]-d[1]-d[2]-d[3])/d[0], d[4]/waves, d[2]/d[4], d[3]/d[4]))
    if d[7] > 0:
        print("balance inside a workgroup: sum of wave cycles / (slowest wave x waves) = %.3f" % (d[0] / d[7]))
    if d[6] > 0:
        print("trips/wave %.1f  trips/pose %.0f  loop cycles per trip %.0f" % (d[6]/waves, d[6]/8192, d[3]/d[6]))
finally:
    shutil.copy("/tmp/keep.so", os.path.join(L, "liblightdock_hip.so"))
