#include "gso.hpp"

#include <thread>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace ld {

// rand_core 0.5 SeedableRng::seed_from_u64 (what StdRng::seed_from_u64 of src/lib.rs:38 runs): PCG32
// expansion of the u64 into the eight ChaCha key words
void stdrng_key_from_seed(uint64_t seed, uint32_t key[8]) {
    uint64_t state = seed;
    for (int i = 0; i < 8; i++) {
        state = state * 6364136223846793005ULL + 11634580027462260723ULL;
        const uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
        const uint32_t rot = (uint32_t)(state >> 59);
        key[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    }
}


Gso::Gso(Scorer &scorer, size_t n_swarms, size_t n_glowworms, const double *positions, const uint64_t *seeds)
    : scorer_(scorer), n_swarms_(n_swarms), n_glowworms_(n_glowworms), pose_len_(scorer.pose_len()) {
    if (n_swarms == 0 || n_glowworms == 0) throw Error(LD_ERR_INVALID, "ld_gso_create: empty swarm");
    if (!positions) throw Error(LD_ERR_INVALID, "ld_gso_create: positions missing");
    if (n_glowworms > 4096) throw Error(LD_ERR_INVALID, "ld_gso_create: more than 4096 glowworms per swarm");
    const size_t total = n_swarms * n_glowworms;
    if (total > (size_t)1 << 28) throw Error(LD_ERR_INVALID, "ld_gso_create: too many glowworms");
    if (pose_len_ > 7 + 2 * 64) throw Error(LD_ERR_INVALID, "ld_gso_create: pose row too long");

    std::vector<double> rows(positions, positions + total * pose_len_);
    poses_[0] = arena_.upload(rows);
    poses_[1] = arena_.upload(rows);
    // Glowworm::new defaults, src/glowworm.rs:45-57
    luciferin_[0] = arena_.upload(std::vector<double>(total, 5.0));
    luciferin_[1] = arena_.upload(std::vector<double>(total, 5.0));
    vision_ = arena_.upload(std::vector<double>(total, 0.2));
    scoring_ = arena_.upload(std::vector<double>(total, 0.0));
    active_ = arena_.upload(std::vector<uint8_t>(total, 1));  // step == 0: everybody is scored
    n_neighbors_ = arena_.upload(std::vector<int32_t>(total, 0));
    std::vector<int32_t> self(total);
    for (size_t i = 0; i < total; i++) self[i] = (int32_t)(i % n_glowworms);
    target_ = arena_.upload(self);
    step_ = arena_.upload(std::vector<uint32_t>(total, 0));
    // K2 workgroups per swarm: aim at >= 512 workgroups, at least 64 glowworms each
    parts_ = (int)std::min<size_t>((n_glowworms + 63) / 64, std::max<size_t>(1, (512 + n_swarms - 1) / n_swarms));
    std::vector<uint32_t> keys(8 * n_swarms);
    for (size_t s = 0; s < n_swarms; s++) stdrng_key_from_seed(seeds ? seeds[s] : 324324ULL, &keys[8 * s]);  // src/lib.rs:38
    rng_key_ = arena_.upload(keys);
    evals_ = arena_.upload(std::vector<unsigned long long>(1, (unsigned long long)total));
    std::vector<uint32_t> everybody(total);  // step 0 scores every glowworm (src/glowworm.rs:62)
    for (size_t i = 0; i < total; i++) everybody[i] = (uint32_t)i;
    moved_list_[0] = arena_.upload(everybody);
    moved_list_[1] = arena_.upload(everybody);
    moved_count_ = arena_.upload(std::vector<uint32_t>{(uint32_t)total, 0u});
}

Gso::~Gso() {
    if (graph_exec_) (void)hipGraphExecDestroy(graph_exec_);
}

void Gso::step() {
    // Swarm::update_luciferin (src/swarm.rs:66-70): energies only for glowworms that moved
    // (or all of them at step 0); the luciferin arithmetic itself is folded into K2.
    scorer_.energy_batch_device(n_swarms_ * n_glowworms_, poses_[cur_], pose_len_, active_, scoring_, nullptr, moved_list_[cur_],
                                moved_count_ + cur_);
    GsoLaunch g;
    g.n_swarms = (int)n_swarms_;
    g.n_glowworms = (int)n_glowworms_;
    g.pose_len = (int)pose_len_;
    g.anm_rec = (int)scorer_.anm_rec();
    g.anm_lig = (int)scorer_.anm_lig();
    g.poses_in = poses_[cur_];
    g.poses_out = poses_[cur_ ^ 1];
    g.luciferin_in = luciferin_[cur_];
    g.luciferin_out = luciferin_[cur_ ^ 1];
    g.parts = parts_;
    g.vision = vision_;
    g.scoring = scoring_;
    g.active = active_;
    g.n_neighbors = n_neighbors_;
    g.target = target_;
    g.step = step_;
    g.rng_key = rng_key_;
    g.evals = evals_;
    g.moved_list = moved_list_[cur_ ^ 1];
    g.moved_count = moved_count_ + (cur_ ^ 1);
    g.zero_count = moved_count_ + cur_;   // (K1 of this step has read it; its next writer is K2 of the next step)
    hip_check(launch_gso_step(g, scorer_.stream()), "launch gso_movement_phase");
    cur_ ^= 1;
    steps_done_++;
}

void Gso::run(uint32_t steps) {
    // A step is a handful of asynchronous launches (memsets, K1, tail, K2) with no host synchronisation in
    // between, so the stream stays ahead of the GPU.  LIGHTDOCK_GSO_GRAPH=1 replays two captured steps
    // (the pose buffers ping-pong) as one hipGraph instead; measured on MI355X / ROCm 7.2 it is equal to
    // 2 % slower for 1 to 1024 swarms, so plain launches are the default.
    const char *env = std::getenv("LIGHTDOCK_GSO_GRAPH");
    const bool want_graph = env && std::strcmp(env, "1") == 0;
    if (want_graph && steps >= 6 && !(scorer_.use_anm() && scorer_.anm_rec() > 0)) {
        // The captured launches carry the addresses of the scorer's shared workspaces.  Another user
        // of the same scorer (a larger pose batch, a larger GSO) may have reallocated them since.
        scorer_.prepare_batch(n_swarms_ * n_glowworms_);
        if (graph_exec_ && graph_generation_ != scorer_.workspace_generation()) {
            (void)hipGraphExecDestroy(graph_exec_);
            graph_exec_ = nullptr;
        }
        if (!graph_exec_) {
            step();  // eager once: every workspace reaches its final size
            steps--;
            hipStream_t st = scorer_.stream();
            hipGraph_t graph = nullptr;
            scorer_.set_capturing(true);
            bool ok = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess;
            if (ok) {
                const uint32_t before = steps_done_;
                try {
                    step();
                    step();
                } catch (const Error &) {
                    ok = false;
                }
                steps_done_ = before;  // nothing ran yet
                if (hipStreamEndCapture(st, &graph) != hipSuccess) ok = false;
            }
            scorer_.set_capturing(false);
            if (ok && graph && hipGraphInstantiate(&graph_exec_, graph, nullptr, nullptr, 0) != hipSuccess) graph_exec_ = nullptr;
            graph_cur_ = cur_;  // the captured pair of steps starts from this pose buffer
            graph_generation_ = scorer_.workspace_generation();
            if (graph) (void)hipGraphDestroy(graph);
            (void)hipGetLastError();
        }
        if (graph_exec_ && cur_ != graph_cur_ && steps > 0) {  // an odd number of eager steps since the capture
            step();
            steps--;
        }
        if (graph_exec_) {
            while (steps >= 2) {
                hip_check(hipGraphLaunch(graph_exec_, scorer_.stream()), "hipGraphLaunch");
                steps_done_ += 2;
                steps -= 2;
            }
        }
    }
    for (uint32_t s = 0; s < steps; s++) step();
}

uint64_t Gso::num_evals() {
    // evals_ starts at S*N (the step-0 evaluation of everybody) and K2 adds the glowworms it
    // moved, i.e. the evaluations of the NEXT step; subtract the ones not yet performed.
    unsigned long long total = 0;
    hip_check(hipStreamSynchronize(scorer_.stream()), "hipStreamSynchronize");
    hip_check(hipMemcpy(&total, evals_, sizeof total, hipMemcpyDeviceToHost), "D2H evals");
    if (steps_done_ == 0) return 0;
    std::vector<uint8_t> pending(n_swarms_ * n_glowworms_);
    hip_check(hipMemcpy(pending.data(), active_, pending.size(), hipMemcpyDeviceToHost), "D2H active");
    unsigned long long not_yet = 0;
    for (uint8_t a : pending) not_yet += a;
    return total - not_yet;
}

void Gso::read(size_t swarm, double *poses, double *luciferin, double *vision, double *scoring, int32_t *n_neighbors,
               int32_t *moved, int32_t *target) {
    if (swarm >= n_swarms_) throw Error(LD_ERR_INVALID, "ld_gso_read: swarm index out of range");
    hip_check(hipStreamSynchronize(scorer_.stream()), "hipStreamSynchronize");
    const size_t n = n_glowworms_, off = swarm * n;
    auto pull = [&](void *dst, const void *src, size_t bytes) {
        if (dst) hip_check(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost), "D2H state");
    };
    pull(poses, poses_[cur_] + off * pose_len_, n * pose_len_ * sizeof(double));
    pull(luciferin, luciferin_[cur_] + off, n * sizeof(double));
    pull(vision, vision_ + off, n * sizeof(double));
    pull(scoring, scoring_ + off, n * sizeof(double));
    pull(n_neighbors, n_neighbors_ + off, n * sizeof(int32_t));
    pull(target, target_ + off, n * sizeof(int32_t));
    if (moved) {
        std::vector<uint8_t> a(n);
        pull(a.data(), active_ + off, n);
        // before the first step `active` means "score me", not "moved" (src/glowworm.rs:55)
        for (size_t i = 0; i < n; i++) moved[i] = steps_done_ == 0 ? 0 : a[i];
    }
}

namespace {
// Rust `{:.N}` == C "%.Nf" (exact decimal expansion, ties to even) except for non-finite values.
std::string fixed(double v, int prec) {
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    char buf[400];
    std::snprintf(buf, sizeof buf, "%.*f", prec, v);
    return buf;
}
}  // namespace

// Host copy of everything save() prints, for all swarms at once: a launcher saves every swarm
// after the same step, and 5 copies per step beat 5 small copies per swarm (1024 swarms: 5 s -> 1 s
// of file output per run).
void Gso::refresh_mirror() {
    if (mirror_valid_ && mirror_step_ == steps_done_) return;
    hip_check(hipStreamSynchronize(scorer_.stream()), "hipStreamSynchronize");
    const size_t total = n_swarms_ * n_glowworms_;
    mirror_poses_.resize(total * pose_len_);
    mirror_luc_.resize(total);
    mirror_vis_.resize(total);
    mirror_sco_.resize(total);
    mirror_nn_.resize(total);
    hip_check(hipMemcpy(mirror_poses_.data(), poses_[cur_], mirror_poses_.size() * sizeof(double), hipMemcpyDeviceToHost), "D2H state");
    hip_check(hipMemcpy(mirror_luc_.data(), luciferin_[cur_], total * sizeof(double), hipMemcpyDeviceToHost), "D2H state");
    hip_check(hipMemcpy(mirror_vis_.data(), vision_, total * sizeof(double), hipMemcpyDeviceToHost), "D2H state");
    hip_check(hipMemcpy(mirror_sco_.data(), scoring_, total * sizeof(double), hipMemcpyDeviceToHost), "D2H state");
    hip_check(hipMemcpy(mirror_nn_.data(), n_neighbors_, total * sizeof(int32_t), hipMemcpyDeviceToHost), "D2H state");
    mirror_step_ = steps_done_;
    mirror_valid_ = true;
}

void Gso::save(size_t swarm, uint32_t step, const std::string &dir) {
    if (swarm >= n_swarms_) throw Error(LD_ERR_INVALID, "ld_gso_save: swarm index out of range");
    refresh_mirror();
    write_swarm(swarm, step, dir);
}

void Gso::save_many(const std::vector<size_t> &swarms, uint32_t step, const std::vector<std::string> &dirs) {
    for (size_t s : swarms)
        if (s >= n_swarms_) throw Error(LD_ERR_INVALID, "ld_gso_save_many: swarm index out of range");
    refresh_mirror();
    // formatting 200 lines of 12+ numbers per swarm is the cost; a few threads, each its own files
    const size_t n_threads = std::min<size_t>(std::max<size_t>(1, std::min<size_t>(std::thread::hardware_concurrency(), 16)), swarms.size());
    std::vector<std::string> errors(n_threads);
    std::vector<std::thread> pool;
    for (size_t t = 0; t < n_threads; t++)
        pool.emplace_back([&, t] {
            try {
                for (size_t k = t; k < swarms.size(); k += n_threads) write_swarm(swarms[k], step, dirs[k]);
            } catch (const std::exception &e) {
                errors[t] = e.what();
            }
        });
    for (std::thread &th : pool) th.join();
    for (const std::string &e : errors)
        if (!e.empty()) throw Error(LD_ERR_IO, e);
}

void Gso::write_swarm(size_t swarm, uint32_t step, const std::string &dir) const {
    const size_t n = n_glowworms_, off = swarm * n;
    const double *poses = mirror_poses_.data() + off * pose_len_;
    const double *luc = mirror_luc_.data() + off, *vis = mirror_vis_.data() + off, *sco = mirror_sco_.data() + off;
    const int32_t *nn = mirror_nn_.data() + off;
    const std::string path = dir + "/gso_" + std::to_string(step) + ".out";
    std::FILE *f = std::fopen(path.c_str(), "w");
    if (!f) throw Error(LD_ERR_IO, "Error saving GSO output: " + path + ": " + std::strerror(errno));
    std::fputs("#Coordinates  RecID  LigID  Luciferin  Neighbor's number  Vision Range  Scoring\n", f);
    for (size_t i = 0; i < n; i++) {
        const double *row = &poses[i * pose_len_];
        std::string line = "(";
        // 7 pose columns, then the ANM extents when the run uses them (src/swarm.rs:136-158)
        for (size_t c = 0; c < pose_len_; c++) {
            if (c) line += ", ";
            line += fixed(row[c], 7);
        }
        line += ")    0    0   " + fixed(luc[i], 8) + "  " + std::to_string(nn[i]) + " " + fixed(vis[i], 3) + " " +
                fixed(sco[i], 8) + "\n";
        std::fputs(line.c_str(), f);
    }
    if (std::fclose(f) != 0) throw Error(LD_ERR_IO, "Error saving GSO output: " + path);
}

}  // namespace ld
