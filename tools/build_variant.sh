# builds lightdock-rust_amd/lib/variants/<name>.so with extra flags for the DFIRE kernels, next to the normal library
# usage: bash tools/build_variant.sh <name> [-DFLAG ...]
# Variants are DIAGNOSTIC builds: -DLD_DIAG_BUILD is always passed (it is what admits the LD_BM_DIAG_* timing experiments and the
# LIGHTDOCK_BM_DIAG_IGNORE_ANM / LIGHTDOCK_BM_HALF_OCCUPANCY / LIGHTDOCK_ALLOW_ANY_ARCH switches, none of which the shipped library has).
name=$1; shift
cd "$(dirname "$0")/../lightdock-rust_amd" && cp lib/liblightdock_hip.so /tmp/ld_keep.so \
 && touch csrc/kernels/dfire_bm.hip csrc/kernels/dfire_packed.hip csrc/kernels/dfire_tiled.hip csrc/kernels/pose_energy.hip csrc/scorer.cpp && make EXTRA_HIPFLAGS="-DLD_DIAG_BUILD $*" >/dev/null \
 && mkdir -p lib/variants && cp lib/liblightdock_hip.so lib/variants/$name.so \
 && touch csrc/kernels/dfire_bm.hip csrc/kernels/dfire_packed.hip csrc/kernels/dfire_tiled.hip csrc/kernels/pose_energy.hip csrc/scorer.cpp && make >/dev/null && echo built lib/variants/$name.so
