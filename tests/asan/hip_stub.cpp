// hip_stub.cpp -- TEST INFRASTRUCTURE for the sanitizer build of the host side (`make asan`).
//
// GPU AddressSanitizer is not available on this pool, so the host C++ (PDB / setup.json / npy /
// DCparams readers, docking-model builders, tile layout, scorer and GSO bookkeeping, both CLIs'
// error paths) is compiled with g++ -fsanitize=address,undefined and linked against THIS file
// instead of the HIP runtime and the kernels: device memory is host memory (so every upload,
// download and workspace size is checked by ASan), streams / events are no-ops, kernel launches do
// nothing, graph capture is refused (the eager path runs).  Energies are therefore meaningless in
// this build; nothing of it ships.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "kernels/dfire_bm.hpp"
#include "kernels/dfire_packed.hpp"
#include "kernels/dfire_tiled.hpp"
#include "kernels/gso_step.hpp"
#include "kernels/pose_energy.hpp"

extern "C" {
hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_t *p, int) {
    std::memset(p, 0, sizeof *p);
    std::strcpy(p->gcnArchName, "gfx950:sramecc+:xnack-");
    return hipSuccess;
}
hipError_t hipMalloc(void **p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemset(void *d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorNotSupported; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t *g) { *g = nullptr; return hipErrorNotSupported; }
hipError_t hipGraphInstantiate(hipGraphExec_t *, hipGraph_t, hipGraphNode_t *, char *, size_t) { return hipErrorNotSupported; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = nullptr; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = nullptr; return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "hip stub"; }
}

namespace ld {
// kernel launch entry points: the arguments are touched, nothing runs
size_t pair_kernel_lds_bytes(const PairLaunch &) { return 0; }
const char *pair_kernel_name(int method) { return method == 0 ? "pose_energy_pairs<0" : "pose_energy_pairs<1"; }
hipError_t launch_pair_kernel(const PairLaunch &p, hipStream_t) {
    if (p.n_poses && p.partial) p.partial[0] = 0.0;
    return hipSuccess;
}
hipError_t launch_finish_kernel(const FinishLaunch &f, hipStream_t) {
    for (size_t i = 0; i < f.n_poses; i++)
        if (!f.active || f.active[i]) f.energies[i] = 0.0;   // the output buffer really has n_poses doubles
    return hipSuccess;
}
size_t tiled_kernel_lds_bytes(const TiledLaunch &) { return 0; }
hipError_t launch_dfire_tiled(const TiledLaunch &t, hipStream_t) {
    if (t.n_poses && t.partial) t.partial[2 * (t.n_poses * (size_t)t.n_groups - 1) + 1] = 0.0;   // last slot of the workspace
    return hipSuccess;
}
hipError_t launch_prepare_receptor(const PrepareReceptorLaunch &p, hipStream_t) {
    if (p.n_poses && p.atoms_out) std::memset(p.atoms_out, 0, p.n_poses * (size_t)p.n_tiles * 64 * sizeof(TiledAtom));
    return hipSuccess;
}
size_t packed_kernel_lds_bytes(int) { return 0; }
hipError_t launch_dfire_packed(const PackedLaunch &t, hipStream_t) {
    if (t.n_poses && t.partial) t.partial[2 * (t.n_poses * (size_t)t.n_groups - 1) + 1] = 0.0;
    return hipSuccess;
}
size_t bm_pairs_lds_bytes() { return 0; }
hipError_t launch_bm_pose(const BmLaunch &t, hipStream_t) {
    if (t.n_poses && t.rt) t.rt[12 * t.n_poses - 1] = 0.f;   // last slot of the pass's affine maps (the workspace of a pass goes by row)
    if (t.n_poses && t.rt_exact) t.rt_exact[8 * t.n_poses - 1] = 0.0;
    return hipSuccess;
}
hipError_t launch_bm_cull(const BmLaunch &t, hipStream_t) {
    const size_t tile_pairs = (size_t)t.m.lig.n_tiles * t.m.rec_n_tiles;
    if (t.n_poses) {
        t.ent_row[tile_pairs * t.cap - 1] = 0;
        t.ent_mask[tile_pairs * t.cap - 1] = 0;
        t.tile_sum[t.n_poses * (size_t)t.m.lig.n_tiles - 1] = 0;
        t.exact_fix[t.n_poses - 1] = 0;
    }
    return hipSuccess;
}
hipError_t launch_bm_pairs(const BmLaunch &t, hipStream_t) {
    const size_t tile_pairs = (size_t)t.m.lig.n_tiles * t.m.rec_n_tiles;
    if (t.n_poses) {
        t.ent_partial[tile_pairs * kBmJobRows * t.cap - 1] = 0;
        t.queue[(size_t)t.pairs_groups * kBmWavesPerCu * kBmQueueCap - 1] = 0ull;
        t.job_order[tile_pairs * (t.cap / 64 + 1) * kBmJobRows - 1] = 0u;
    }
    return hipSuccess;
}
hipError_t launch_bm_gather(const BmLaunch &t, hipStream_t) {
    if (t.n_poses && !t.count_mode) t.partial[2 * (t.first + t.n_poses) - 1] = 0.0;
    if (t.n_poses && t.count_mode) t.count_partial[t.first + t.n_poses - 1] = 0u;
    return hipSuccess;
}
hipError_t launch_packed_prepare(const PackedPrepareLaunch &p, hipStream_t) {
    if (p.n_poses && p.pairs_out) std::memset(p.pairs_out, 0, p.n_poses * (size_t)p.n_tiles * 32 * sizeof(PackedRecPair));
    if (p.n_poses && p.sub_out) std::memset(p.sub_out, 0, p.n_poses * (size_t)p.n_tiles * 8 * sizeof(TiledBox));
    if (p.n_poses && p.tile_out) std::memset(p.tile_out, 0, p.n_poses * (size_t)p.n_tiles * sizeof(TiledBox));
    return hipSuccess;
}
size_t gso_kernel_lds_bytes(const GsoLaunch &) { return 0; }
hipError_t launch_gso_step(const GsoLaunch &g, hipStream_t) {
    const size_t total = (size_t)g.n_swarms * g.n_glowworms;
    std::memcpy(g.poses_out, g.poses_in, total * g.pose_len * sizeof(double));   // both pose buffers are that large
    for (size_t i = 0; i < total; i++) g.step[i]++;
    return hipSuccess;
}
}  // namespace ld
