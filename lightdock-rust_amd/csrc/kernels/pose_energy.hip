// pose_energy.hip -- K1: batched all-pairs pose-energy kernels for gfx950 (MI355X).
//
// What one workgroup does (256 threads = 4 wave64, one (pose, receptor chunk) each):
//   1. the pose row (translation, quaternion, ANM extents) is wave-uniform -> SGPRs;
//   2. the receptor chunk is read from HBM as coalesced SoA columns, ANM-deformed on the
//      fly (src/dfire.rs:304-320) and staged in LDS as 32-byte (DFIRE) / 64-byte (DNA)
//      records;
//   3. every lane owns one ligand atom of a 64-atom ligand group: it applies the pose
//      (q v q^-1 + t, then ANM; src/dfire.rs:282-302, src/qt.rs:57-61) in registers;
//   4. the wave walks the LDS records -- every lane reads the SAME record, which the LDS
//      serves as a broadcast -- and evaluates its pair: d2, cutoff test, then DFIRE's
//      distance-binned table gather (src/dfire.rs:325-345) or DNA's Coulomb + 12-6 terms
//      (src/dna.rs:471-512);
//   5. per-lane f64 partial sums are folded with a wave64 shuffle reduction, then across
//      the 4 waves through LDS, into one partial per (pose, chunk).
// Interface flags (src/dfire.rs:339-342) are only kept for the atoms the tail needs
// (restraint residues, membrane beads); they are rare events and go straight to a
// per-pose bit set in global memory with integer atomics (deterministic).
//
// Numerics: all geometry is IEEE f64 in the reference's operation order; this file is
// compiled with -ffp-contract=off so d2, the bin and the interface test are bit-identical
// to the CPU.  Only the ORDER of the += over pairs differs (parallel tree vs sequential).
// No MFMA: this is a lookup/reduction, not a contraction.
#include "pose_energy.hpp"

namespace ld {

namespace {

struct Quat {
    double w, x, y, z;
};

// Hamilton product in the reference's term order, src/qt.rs:174-185
__device__ __forceinline__ Quat qmul(const Quat &a, const Quat &b) {
    Quat r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
    r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
    return r;
}

// conj(q) / |q|^2, src/qt.rs:48-50 (the quaternion is NOT assumed to be unit)
__device__ __forceinline__ Quat qinverse(const Quat &q) {
    const double n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
    Quat r;
    r.w = q.w / n2;
    r.x = -q.x / n2;
    r.y = -q.y / n2;
    r.z = -q.z / n2;
    return r;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

struct alignas(16) DfireRec {
    double x, y, z;
    uint32_t trow;  // type * 3380
    int32_t slot;
};
static_assert(sizeof(DfireRec) == 32, "DfireRec must be 32 bytes");

struct alignas(16) DnaRec {
    double x, y, z;
    double charge, well_depth, radius;
    int32_t slot;
    int32_t pad[3];
};
static_assert(sizeof(DnaRec) == 64, "DnaRec must be 64 bytes");

// DNA constants, src/dna.rs:15-25
constexpr double kElecCutoff2 = 30.0 * 30.0;
constexpr double kVdwCutoff2 = 10.0 * 10.0;
constexpr double kElecMax = 1.0 * 4.0 / 332.0;
constexpr double kElecMin = -1.0 * 4.0 / 332.0;
constexpr double kVdwMax = 1.0;

__device__ __forceinline__ size_t round16(size_t v) { return (v + 15) & ~size_t(15); }

// Pose applied to one ligand atom: rotate, translate, then ANM (src/dfire.rs:282-302).
__device__ __forceinline__ void pose_ligand_atom(const DeviceMolecule &lig, int atom, const Quat &q, const Quat &qinv,
                                                 double tx, double ty, double tz, bool anm, const double *lig_nm,
                                                 double &ox, double &oy, double &oz) {
    const Quat v{0.0, lig.x[atom], lig.y[atom], lig.z[atom]};
    const Quat r = qmul(qmul(q, v), qinv);
    double px = r.x + tx, py = r.y + ty, pz = r.z + tz;
    if (anm) {
        const size_t np = (size_t)lig.n_pad;
        for (int k = 0; k < lig.num_anm; k++) {
            const double c = lig_nm[k];
            const double *m = lig.modes + (size_t)k * 3 * np;
            px += m[atom] * c;
            py += m[np + atom] * c;
            pz += m[2 * np + atom] * c;
        }
    }
    ox = px;
    oy = py;
    oz = pz;
}

template <int METHOD, bool COUNT>
__global__ __launch_bounds__(kBlockThreads) void pose_energy_pairs(const PairLaunch P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using Rec = typename std::conditional<METHOD == 0, DfireRec, DnaRec>::type;
    Rec *rec = reinterpret_cast<Rec *>(smem);
    const size_t rec_bytes = round16((size_t)P.chunk_atoms * sizeof(Rec));
    uint8_t *lut = smem + rec_bytes;  // DFIRE only
    const size_t lut_bytes = METHOD == 0 ? round16(kDfireLutCells) : 0;
    double *bin_step = reinterpret_cast<double *>(smem + rec_bytes + lut_bytes);  // DFIRE only
    const size_t step_bytes = METHOD == 0 ? kDfireSteps * sizeof(double) : 0;
    double *red = reinterpret_cast<double *>(smem + rec_bytes + lut_bytes + step_bytes);  // [kWaves][2]
    uint32_t *red_cnt = reinterpret_cast<uint32_t *>(red + 2 * kWaves);       // [kWaves]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t pose = blockIdx.x / (unsigned)P.n_chunks;
    const int chunk = blockIdx.x % (unsigned)P.n_chunks;
    if (P.active != nullptr && P.active[pose] == 0) return;

    const double *row = P.poses + pose * P.stride;
    const double tx = row[0], ty = row[1], tz = row[2];
    const Quat q{row[3], row[4], row[5], row[6]};
    const Quat qinv = qinverse(q);
    const bool anm_rec = P.use_anm && P.rec.num_anm > 0;
    const bool anm_lig = P.use_anm && P.lig.num_anm > 0;
    const double *rec_nm = row + 7;
    const double *lig_nm = row + 7 + (P.use_anm ? P.rec.num_anm : 0);

    // ---- stage the receptor chunk: coalesced SoA -> LDS records --------------------------
    const int r0 = chunk * P.chunk_atoms;
    const int rn = min(P.chunk_atoms, P.rec.n - r0);
    for (int i = tid; i < rn; i += kBlockThreads) {
        const int a = r0 + i;
        double x = P.rec.x[a], y = P.rec.y[a], z = P.rec.z[a];
        if (anm_rec) {  // src/dfire.rs:304-320
            const size_t np = (size_t)P.rec.n_pad;
            for (int k = 0; k < P.rec.num_anm; k++) {
                const double c = rec_nm[k];
                const double *m = P.rec.modes + (size_t)k * 3 * np;
                x += m[a] * c;
                y += m[np + a] * c;
                z += m[2 * np + a] * c;
            }
        }
        Rec r;
        r.x = x;
        r.y = y;
        r.z = z;
        r.slot = P.rec.slot[a];
        if constexpr (METHOD == 0) {
            r.trow = P.rec.tindex[a];
        } else {
            r.charge = P.rec.charge[a];
            r.well_depth = P.rec.well_depth[a];
            r.radius = P.rec.radius[a];
            r.pad[0] = r.pad[1] = r.pad[2] = 0;
        }
        rec[i] = r;
    }
    if constexpr (METHOD == 0) {
        for (int i = tid; i < kDfireLutCells / 4; i += kBlockThreads)
            reinterpret_cast<uint32_t *>(lut)[i] = reinterpret_cast<const uint32_t *>(P.lut)[i];
        if (tid < kDfireSteps) bin_step[tid] = P.bin_step[tid];
    }
    __syncthreads();

    // ---- which (ligand group, record range) pairs this wave owns --------------------------
    const int n_groups = (P.lig.n + 63) >> 6;
    int g_begin, g_step, j_begin, j_end;
    if (P.split_j) {
        const int quarter = (rn + kWaves - 1) / kWaves;
        g_begin = 0;
        g_step = 1;
        j_begin = min(rn, wave * quarter);
        j_end = min(rn, j_begin + quarter);
    } else {
        g_begin = wave;
        g_step = kWaves;
        j_begin = 0;
        j_end = rn;
    }

    uint32_t *pose_flags = P.flags + pose * (size_t)(P.rec.flag_words + P.lig.flag_words);
    double acc0 = 0.0, acc1 = 0.0;  // DFIRE: table sum | DNA: elec, vdw
    double pending = 0.0;           // DFIRE: the gather issued last, added one hit later
    uint32_t cnt = 0;

    for (int g = g_begin; g < n_groups; g += g_step) {
        const int atom = (g << 6) + lane;
        const bool valid = atom < P.lig.n;
        const int la = valid ? atom : P.lig.n - 1;
        double lx, ly, lz;
        pose_ligand_atom(P.lig, la, q, qinv, tx, ty, tz, anm_lig, lig_nm, lx, ly, lz);
        if (!valid) lx = 1.0e30;  // padded lane: never inside any cutoff
        const int lslot = P.lig.slot[la];
        bool lflag = false;

        if constexpr (METHOD == 0) {
            const uint32_t ltype20 = P.lig.tindex[la];
            const double *tab = P.table + ltype20;
            for (int j = j_begin; j < j_end; j++) {
                const DfireRec a = rec[j];
                // (x1 - la[0])^2 + (y1 - la[1])^2 + (z1 - la[2])^2, src/dfire.rs:331-333
                const double dx = a.x - lx, dy = a.y - ly, dz = a.z - lz;
                const double d2 = dx * dx + dy * dy + dz * dz;
                if (d2 <= 225.0) {
                    // DIST_TO_BINS[(sqrt(d2)*2-1) as usize] - 1 (src/dfire.rs:336-337) without
                    // the sqrt: the bin steps sit at d2 = (k/2)^2, multiples of 0.25, so the
                    // 0.25-wide cell gives the bin up to the exact position of the step, which
                    // bin_step[] holds to the last bit (DESIGN.md "bin LUT").
                    const int cell = (int)(d2 * 4.0);
                    uint32_t bin = lut[cell];
                    bin += d2 >= bin_step[bin + 1] ? 1u : 0u;
                    acc0 += pending;
                    pending = tab[a.trow + bin];
                    if (COUNT) cnt++;
                    if (d2 <= P.iface_d2) {  // d <= 3.9, src/dfire.rs:339
                        if (a.slot >= 0) atomicOr(&pose_flags[a.slot >> 5], 1u << (a.slot & 31));
                        lflag = true;
                    }
                }
            }
        } else {
            const double lq = P.lig.charge[la], le = P.lig.well_depth[la], lr = P.lig.radius[la];
            double closest = 1.0;  // smallest d2 seen: (almost) coincident atoms can make the reference's score NaN, see below
            for (int j = j_begin; j < j_end; j++) {
                const DnaRec a = rec[j];
                const double dx = a.x - lx, dy = a.y - ly, dz = a.z - lz;
                const double d2 = dx * dx + dy * dy + dz * dz;  // src/dna.rs:476-478
                closest = fmin(closest, d2);
                // (The scalar pipes of this kernel are as busy as the vector pipes -- 0.7 scalar instructions per vector
                // instruction, the saveexec / cbranch / exec-restore of these nested cutoffs, profiles/r03_1azp_dna_summary.txt --
                // but they are not what binds it: with the electrostatic term branch-free (select instead of branch: a
                // third fewer scalar instructions, the reciprocal also for the 34 % of pairs beyond 30 A) the kernel is
                // 9 % SLOWER, 2.24 against 2.46 M evaluations/s.  The vector pipe decides; the scalar work overlaps.)
                if (d2 <= kElecCutoff2) {                         // src/dna.rs:481-491
                    // 1/d2 by v_rcp_f64 + one Newton step instead of two correctly rounded f64 divisions.  The ISA
                    // promises v_rcp_f64 only about 2^-24 relative, so one step guarantees about 2^-47 (tens of ulps)
                    // per term, not 1 ulp.  The energy is continuous in these terms (every cutoff test above/below uses
                    // the exact d2) and the tests hold it to 1e-9 against the oracle (observed ~1e-13 on the 200
                    // 1azp poses, whose electrostatic sum cancels to a few per cent of its terms) and to the 8
                    // printed decimals of the reference's gso files -- at half the instruction count.  (A second
                    // step changes no printed digit and costs 5 %.)
                    const double r0 = __builtin_amdgcn_rcp(d2);
                    const double inv = __builtin_fma(r0, __builtin_fma(-d2, r0, 1.0), r0);
                    double e = (a.charge * lq) * inv;
                    e = fmin(e, kElecMax);
                    e = fmax(e, kElecMin);
                    acc0 += e;
                    if (COUNT) cnt++;
                    if (d2 <= kVdwCutoff2) {  // src/dna.rs:494-504
                        const double vdw_energy = a.well_depth * le;  // sqrt(eps_i) * sqrt(eps_j), roots taken on the host
                        const double rr = a.radius + lr;
                        const double rr2 = rr * rr;
                        const double rr6 = rr2 * (rr2 * rr2);
                        const double p6 = rr6 * (inv * inv * inv);
                        double k = vdw_energy * (p6 * p6 - 2.0 * p6);
                        k = fmin(k, kVdwMax);
                        acc1 += k;
                        if (d2 <= P.iface_d2) {  // src/dna.rs:507-510
                            if (a.slot >= 0) atomicOr(&pose_flags[a.slot >> 5], 1u << (a.slot & 31));
                            lflag = true;
                        }
                    }
                }
            }
            // Two atoms on (almost) the same spot: the reference's p6 = R^6 / d2^3 (src/dna.rs:498) is inf when d2^3
            // underflows or the quotient overflows, and k = e * (inf - inf) = NaN, which its ordered `k > VDW_CUTOFF`
            // keeps (:499-503): the score is NaN.  A p6 that is huge but finite gives k = inf, which it clamps.  fmin
            // above turns both into the clamp value: for such a lane (none in any real pose) redo the reference's own
            // division and keep the NaN where it has one.
            if (__builtin_expect(closest < 1.0e-90, 0)) {
                for (int j = j_begin; j < j_end; j++) {
                    const DnaRec a = rec[j];
                    const double dx = a.x - lx, dy = a.y - ly, dz = a.z - lz;
                    const double d2 = dx * dx + dy * dy + dz * dz;
                    if (!(d2 < 1.0e-90)) continue;
                    const double rr = a.radius + lr;
                    const double rr2 = rr * rr;
                    const double p6_ref = (rr2 * (rr2 * rr2)) / (d2 * (d2 * d2));
                    if (!(p6_ref <= 1.7976931348623157e308)) acc1 = __builtin_nan("");
                }
            }
        }
        if (lflag && lslot >= 0) atomicOr(&pose_flags[P.rec.flag_words + (lslot >> 5)], 1u << (lslot & 31));
    }
    acc0 += pending;

    // ---- wave64 shuffle reduction, then the 4 waves through LDS ---------------------------
    acc0 = wave_sum(acc0);
    acc1 = wave_sum(acc1);
    if (COUNT) cnt = wave_sum_u32(cnt);
    if (lane == 0) {
        red[2 * wave] = acc0;
        red[2 * wave + 1] = acc1;
        if (COUNT) red_cnt[wave] = cnt;
    }
    __syncthreads();
    if (tid == 0) {
        double s0 = 0.0, s1 = 0.0;
        uint32_t c = 0;
        for (int w = 0; w < kWaves; w++) {
            s0 += red[2 * w];
            s1 += red[2 * w + 1];
            if (COUNT) c += red_cnt[w];
        }
        const size_t slot = (pose * (size_t)P.n_chunks + chunk);
        P.partial[2 * slot] = s0;
        P.partial[2 * slot + 1] = s1;
        if (COUNT) P.count_partial[slot] = c;
    }
}

__device__ __forceinline__ bool flag_set(const uint32_t *words, uint32_t slot) {
    return (words[slot >> 5] >> (slot & 31)) & 1u;
}

// scoring.rs:21-36 over flag slots
__device__ double satisfied_fraction(const uint32_t *words, int n_groups, const uint32_t *offsets,
                                     const uint32_t *slots) {
    if (n_groups == 0) return 0.0;
    int hit = 0;
    for (int g = 0; g < n_groups; g++) {
        for (uint32_t k = offsets[g]; k < offsets[g + 1]; k++)
            if (flag_set(words, slots[k])) {
                hit++;
                break;
            }
    }
    return (double)hit / (double)n_groups;
}

// Eight lanes per pose: lane k folds the partials k, k + 8, ... in that order, the eight sums are folded as a
// fixed tree (the same for every launch), lane 0 applies the tail.
constexpr int kFinishLanes = 8;
__global__ __launch_bounds__(kBlockThreads) void pose_energy_finish(const FinishLaunch F) {
    const size_t t = (size_t)blockIdx.x * kBlockThreads + threadIdx.x;
    const size_t pose = t / kFinishLanes;
    const int sub = (int)(t % kFinishLanes);
    const bool live = pose < F.n_poses && !(F.active != nullptr && F.active[pose] == 0);
    double s0 = 0.0, s1 = 0.0;
    uint32_t cnt = 0;
    if (live) {
        for (int c = sub; c < F.n_chunks; c += kFinishLanes) {
            const size_t slot = pose * (size_t)F.n_chunks + c;
            const double2 v = *reinterpret_cast<const double2 *>(F.partial + 2 * slot);
            s0 += v.x;
            s1 += v.y;
            if (F.count_partial) cnt += F.count_partial[slot];
        }
    }
    for (int d = 1; d < kFinishLanes; d <<= 1) {  // every lane of the wave takes part: dead poses carry zeros
        s0 += __shfl_xor(s0, d);
        s1 += __shfl_xor(s1, d);
        cnt += __shfl_xor(cnt, d);
    }
    // membrane beads (src/scoring.rs:38-47): the pose's eight lanes share the scan, the count is an integer
    const uint32_t *rwords = F.flags + (live ? pose : 0) * (size_t)(F.rec_flag_words + F.lig_flag_words);
    int beads = 0;
    if (live)
        for (int k = sub; k < F.tail.n_membrane; k += kFinishLanes) beads += flag_set(rwords, F.tail.membrane_slots[k]) ? 1 : 0;
    if (F.tail.n_membrane > 0)
        for (int d = 1; d < kFinishLanes; d <<= 1) beads += __shfl_xor(beads, d);
    if (!live || sub != 0) return;
    double score;
    if (F.method == 0) {
        score = (s0 * 0.0157 - 4.7) * -1.0;  // src/dfire.rs:347
    } else {
        const double total_elec = s0 * 332.0 / 4.0;  // src/dna.rs:513
        score = (total_elec + s1) * -1.0;            // src/dna.rs:514
    }
    const uint32_t *lwords = rwords + F.rec_flag_words;
    const double pr = satisfied_fraction(rwords, F.tail.n_rec_groups, F.tail.rec_group_offsets, F.tail.rec_group_slots);
    const double pl = satisfied_fraction(lwords, F.tail.n_lig_groups, F.tail.lig_group_offsets, F.tail.lig_group_slots);
    double penalty = 0.0;
    if (F.tail.n_membrane > 0) {  // src/scoring.rs:38-47, src/dfire.rs:355-359
        const double intersection = (double)beads / (double)F.tail.n_membrane;
        if (intersection > 0.0) penalty = 999.0 * intersection;
    }
    F.energies[pose] = score + pr * score + pl * score - penalty;  // src/dfire.rs:361
    if (F.pair_counts) F.pair_counts[pose] = cnt;
}

}  // namespace

size_t pair_kernel_lds_bytes(const PairLaunch &p) {
    const size_t rec = ((size_t)p.chunk_atoms * (p.method == 0 ? sizeof(DfireRec) : sizeof(DnaRec)) + 15) & ~size_t(15);
    const size_t lut = p.method == 0 ? ((kDfireLutCells + 15) & ~15) + kDfireSteps * sizeof(double) : 0;
    return rec + lut + 2 * kWaves * sizeof(double) + kWaves * sizeof(uint32_t) + 16;
}

const char *pair_kernel_name(int method) { return method == 0 ? "pose_energy_pairs<0" : "pose_energy_pairs<1"; }

hipError_t launch_pair_kernel(const PairLaunch &p, hipStream_t stream) {
    if (p.n_poses == 0) return hipSuccess;
    const size_t blocks = p.n_poses * (size_t)p.n_chunks;
    if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks), block(kBlockThreads);
    const size_t lds = pair_kernel_lds_bytes(p);
    const bool count = p.count_partial != nullptr;
    if (p.method == 0) {
        if (count) hipLaunchKernelGGL((pose_energy_pairs<0, true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((pose_energy_pairs<0, false>), grid, block, lds, stream, p);
    } else {
        if (count) hipLaunchKernelGGL((pose_energy_pairs<1, true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((pose_energy_pairs<1, false>), grid, block, lds, stream, p);
    }
    return hipGetLastError();
}

hipError_t launch_finish_kernel(const FinishLaunch &f, hipStream_t stream) {
    if (f.n_poses == 0) return hipSuccess;
    const dim3 grid((unsigned)((f.n_poses * kFinishLanes + kBlockThreads - 1) / kBlockThreads)), block(kBlockThreads);
    hipLaunchKernelGGL(pose_energy_finish, grid, block, 0, stream, f);
    return hipGetLastError();
}

}  // namespace ld
