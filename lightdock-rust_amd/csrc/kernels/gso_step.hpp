// gso_step.hpp -- launch interface of K2, the batched GSO movement kernel.
//
// `parts` workgroups per swarm (each stages the whole swarm's snapshot, moves its own share of
// the glowworms), one thread per glowworm.  Covers, for every swarm at once, the
// second half of Glowworm::compute_luciferin (src/glowworm.rs:70-71) and the whole of
// Swarm::movement_phase (src/swarm.rs:72-126).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace ld {

struct GsoLaunch {
    int n_swarms = 0;
    int n_glowworms = 0;  // per swarm
    int pose_len = 0;     // 7 + anm_rec + anm_lig
    int anm_rec = 0, anm_lig = 0;
    // state, indexed [swarm * n_glowworms + glowworm]
    const double *poses_in = nullptr;  // pre-move snapshot (rows of pose_len)
    double *poses_out = nullptr;       // post-move poses
    const double *luciferin_in = nullptr;  // luciferin before this step (double buffered like the poses)
    double *luciferin_out = nullptr;
    double *vision = nullptr;
    const double *scoring = nullptr;
    uint8_t *active = nullptr;        // out: moved flag == "re-score next step" (src/glowworm.rs:62)
    int32_t *n_neighbors = nullptr;
    int32_t *target = nullptr;
    uint32_t *step = nullptr;         // per glowworm: completed steps (Glowworm.step, src/glowworm.rs:71); advanced by the kernel
    int parts = 1;                    // workgroups per swarm; each moves a contiguous share of the glowworms
    const uint32_t *rng_key = nullptr;  // per swarm: 8 ChaCha key words (rand 0.7.3 StdRng)
    unsigned long long *evals = nullptr;  // running count of energy evaluations (adds #moved)
    // out: the glowworms that moved, compacted (any order): the next step's K1 evaluates exactly these
    // (src/glowworm.rs:62).  *moved_count must be zero at launch.
    uint32_t *moved_list = nullptr;
    uint32_t *moved_count = nullptr;
    uint32_t *zero_count = nullptr;   // the count this step's K1 read: set to zero for the step after next (instead of a memset launch per step)
};

constexpr size_t kGsoLdsLimit = 160 * 1024;   // a CU's LDS
size_t gso_kernel_lds_bytes(const GsoLaunch &g, bool phased);
bool gso_step_is_phased(const GsoLaunch &g);   // which of the two kernels a launch of this shape runs (LIGHTDOCK_GSO_K2 = single | phased forces one)
hipError_t launch_gso_step(const GsoLaunch &g, hipStream_t stream);

// rand_core 0.5 SeedableRng::seed_from_u64: PCG32 expansion of a u64 into the ChaCha key
void stdrng_key_from_seed(uint64_t seed, uint32_t key[8]);

}  // namespace ld
