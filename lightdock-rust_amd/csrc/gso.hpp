// gso.hpp -- the object behind an `ld_gso*`: device-resident state of a batch of
// independent swarms and the per-step launch sequence K1 (pose energies of the glowworms
// that moved) -> K2 (movement phase).  Mirrors GSO / Swarm / Glowworm of the reference
// (src/lib.rs:21-58, src/swarm.rs:9-167, src/glowworm.rs:6-58) as structure-of-arrays.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "kernels/gso_step.hpp"
#include "scorer.hpp"

namespace ld {

class Gso {
   public:
    Gso(Scorer &scorer, size_t n_swarms, size_t n_glowworms, const double *positions, const uint64_t *seeds);
    ~Gso();
    Gso(const Gso &) = delete;
    Gso &operator=(const Gso &) = delete;

    void step();
    void run(uint32_t steps);
    uint32_t steps_done() const { return steps_done_; }
    uint64_t num_evals();
    size_t n_swarms() const { return n_swarms_; }
    size_t n_glowworms() const { return n_glowworms_; }
    size_t pose_len() const { return pose_len_; }
    void read(size_t swarm, double *poses, double *luciferin, double *vision, double *scoring, int32_t *n_neighbors,
              int32_t *moved, int32_t *target);
    void save(size_t swarm, uint32_t step, const std::string &dir);  // Swarm::save, src/swarm.rs:128-167
    void save_many(const std::vector<size_t> &swarms, uint32_t step, const std::vector<std::string> &dirs);

   private:
    Scorer &scorer_;
    size_t n_swarms_, n_glowworms_, pose_len_;
    uint32_t steps_done_ = 0;
    // host copy of the printed state of all swarms, refreshed once per step by save()
    void refresh_mirror();
    void write_swarm(size_t swarm, uint32_t step, const std::string &dir) const;  // from the mirror
    std::vector<double> mirror_poses_, mirror_luc_, mirror_vis_, mirror_sco_;
    std::vector<int32_t> mirror_nn_;
    uint32_t mirror_step_ = 0;
    bool mirror_valid_ = false;
    DeviceArena arena_;
    double *poses_[2] = {nullptr, nullptr};
    int cur_ = 0;
    double *luciferin_[2] = {nullptr, nullptr};  // ping-pong with the poses
    double *vision_ = nullptr, *scoring_ = nullptr;
    int parts_ = 1;  // K2 workgroups per swarm
    uint8_t *active_ = nullptr;
    int32_t *n_neighbors_ = nullptr, *target_ = nullptr;
    uint32_t *step_ = nullptr, *rng_key_ = nullptr;
    unsigned long long *evals_ = nullptr;
    // K2 -> next K1: the glowworms to score, compacted.  Two lists and two counts, alternating like the pose buffers: K1 of a step reads
    // list / count [cur_], K2 of the same step fills [cur_ ^ 1] and zeroes count [cur_] for its next filling -- no memset launch per step
    uint32_t *moved_list_[2] = {nullptr, nullptr}, *moved_count_ = nullptr;
    hipGraphExec_t graph_exec_ = nullptr;  // two captured steps (even + odd pose buffer)
    int graph_cur_ = 0;
    uint64_t graph_generation_ = 0;  // scorer workspace generation the graph was captured against
};

}  // namespace ld

struct ld_gso {
    ld::Gso impl;
    ld_gso(ld::Scorer &s, size_t n_swarms, size_t n_glowworms, const double *positions, const uint64_t *seeds)
        : impl(s, n_swarms, n_glowworms, positions, seeds) {}
};
