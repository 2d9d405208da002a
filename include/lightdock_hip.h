/*
 * lightdock_hip.h -- C ABI of the MI355X (gfx950) GSO + DFIRE/DNA pose-energy engine.
 *
 * This is the drop-in boundary for ONE path of lightdock-rust v0.3.2: the scoring
 * functions behind `trait Score` and the GSO step that drives them.  Every entry point
 * names the reference interface it replaces (paths relative to the reference tree).
 * Plain pointers and sizes only; all floating point is IEEE f64 like the reference.
 *
 * Conventions
 *  - Functions returning `int` return LD_OK (0) or a negative ld_status; functions
 *    returning a handle return NULL on failure.  ld_last_error() gives the message
 *    (thread local).  Where the reference panics (exit 101) this library fails the call.
 *  - The caller keeps ownership of every input array (copied at create); outputs are
 *    written into caller-provided buffers.
 *  - A handle is bound to the HIP device current at create time and to one stream
 *    (ld_scorer_set_stream); it is thread-compatible, not thread-safe, like a
 *    `&Box<dyn Score>` used from the reference's single worker thread.
 *  - There is NO CPU fallback: without a usable HIP device create fails loudly.
 *
 * Pose row layout (== one line of initial_positions_N.dat, src/swarm.rs:36-52):
 *    [tx ty tz qw qx qy qz | rec_nm[anm_rec] | lig_nm[anm_lig]]     (f64, row-major)
 */
#ifndef LIGHTDOCK_HIP_H
#define LIGHTDOCK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum ld_status {
    LD_OK = 0,
    LD_ERR_INVALID = -1,     /* bad argument / inconsistent sizes */
    LD_ERR_UNSUPPORTED = -2, /* residue/atom/method the scoring function does not know */
    LD_ERR_IO = -3,          /* file missing / unreadable / malformed */
    LD_ERR_DEVICE = -4,      /* HIP error or no gfx950 device */
    LD_ERR_NOMEM = -5
} ld_status;

/* src/scoring.rs:5-9 `enum Method` */
/* PYDOCK (src/pydock.rs) is the DNA energy (src/pydock.rs:425-545 == src/dna.rs:411-529) behind a
 * model builder with a generic-element fallback for unknown atoms (src/pydock.rs:332-345). */
typedef enum ld_method { LD_METHOD_DFIRE = 0, LD_METHOD_DNA = 1, LD_METHOD_PYDOCK = 2 } ld_method;

#define LD_DFIRE_TABLE_LEN (169 * 169 * 20) /* src/dfire.rs:216,251 */

const char *ld_last_error(void);
const char *ld_version(void);

/* Select the HIP device for handles created afterwards on this thread (hipSetDevice).
 * No counterpart in the reference (CPU only).  device < 0: $LIGHTDOCK_DEVICE or 0. */
int ld_init(int device);
int ld_device_count(void);

/* ------------------------------------------------------------------------------------
 * Scorer construction from arrays: what a Rust `impl Score` shim hands over after it has
 * run its own model builder.  Replaces DFIRE::new / DNA::new
 * (src/dfire.rs:201-234, src/dna.rs:375-408) minus the PDB walk.
 * ---------------------------------------------------------------------------------- */
typedef struct ld_molecule {
    size_t n_atoms;
    const double *coordinates;        /* n_atoms x 3, == DockingModel.coordinates (src/dfire.rs:106) */
    const uint32_t *dfire_types;      /* DFIRE: DockingModel.atoms, 0..167 (src/dfire.rs:105); else NULL */
    const double *ele_charges;        /* DNA: src/dna.rs:245; else NULL */
    const double *vdw_charges;        /* DNA: src/dna.rs:244 */
    const double *vdw_radii;          /* DNA: src/dna.rs:243 */
    size_t n_membrane;                /* DockingModel.membrane (src/dfire.rs:107): atom indices of MMB.BJ beads */
    const uint32_t *membrane;
    size_t n_restraint_groups;        /* DockingModel.active_restraints (src/dfire.rs:108) as CSR: */
    const uint32_t *restraint_offsets;/*   n_restraint_groups + 1 offsets into restraint_atoms */
    const uint32_t *restraint_atoms;  /*   atom indices, one group per restraint residue found in the PDB */
    size_t num_anm;                   /* DockingModel.num_anm */
    const double *nmodes;             /* num_anm x n_atoms x 3, C order (src/dfire.rs:292-299); NULL if num_anm == 0 */
} ld_molecule;

typedef struct ld_scorer_desc {
    int method;              /* ld_method */
    int use_anm;             /* DFIRE.use_anm / DNA.use_anm */
    ld_molecule receptor;
    ld_molecule ligand;
    const double *potential; /* DFIRE: LD_DFIRE_TABLE_LEN values of data/DCparams (src/dfire.rs:236-257) */
} ld_scorer_desc;

typedef struct ld_scorer ld_scorer;

ld_scorer *ld_scorer_create(const ld_scorer_desc *desc);

/* Same, but with the host-side model builder of this library doing the PDB walk, atom
 * typing and restraint lookup: DFIRE::new / DNA::new including DFIREDockingModel::new
 * (src/dfire.rs:115-190) and DNADockingModel::new (src/dna.rs:249-364).  Restraint ids are
 * "chain.resname.serial[icode]" strings (src/dfire.rs:139-142).  Passive lists are accepted
 * and ignored exactly like the reference (src/dfire.rs:164-175, never read by energy).
 * nmodes arrays are the flat contents of rec_nm.npy / lig_nm.npy (may be NULL, len 0). */
ld_scorer *ld_scorer_create_from_pdb(int method, const char *receptor_pdb, const char *ligand_pdb,
                                     const char *const *rec_active, size_t n_rec_active,
                                     const char *const *rec_passive, size_t n_rec_passive,
                                     const double *rec_nmodes, size_t rec_nmodes_len, size_t rec_num_anm,
                                     const char *const *lig_active, size_t n_lig_active,
                                     const char *const *lig_passive, size_t n_lig_passive,
                                     const double *lig_nmodes, size_t lig_nmodes_len, size_t lig_num_anm,
                                     int use_anm, const double *potential);

void ld_scorer_destroy(ld_scorer *s); /* Drop of the Box<dyn Score> */

/* ------------------------------------------------------------------------------------
 * Host-side model builder on its own (no GPU needed): DFIREDockingModel::new
 * (src/dfire.rs:115-190) / DNADockingModel::new (src/dna.rs:249-364).  The returned view
 * borrows the model's arrays and is what ld_scorer_create takes.
 * ---------------------------------------------------------------------------------- */
typedef struct ld_model ld_model;
ld_model *ld_model_from_pdb(int method, const char *pdb_path, const char *const *active, size_t n_active,
                            const char *const *passive, size_t n_passive, const double *nmodes, size_t nmodes_len,
                            size_t num_anm);
int ld_model_view(const ld_model *m, ld_molecule *out);
void ld_model_destroy(ld_model *m);

/* Host-side constants of the DFIRE kernel, exported for tests (DESIGN.md "bin LUT"):
 * the table bin DIST_TO_BINS[(sqrt(d2)*2-1) as usize]-1 (src/dfire.rs:49-53,336-337) of any
 * d2 in [0, 225] equals  b = lut[floor(4*d2)];  b += (d2 >= steps[b+1]);
 * lut: 901 cells of 0.25 A^2; steps[b], b = 0..20: first d2 the reference puts in bin >= b.
 * interface_d2: the largest d2 whose d = sqrt(d2)*2-1 is <= 3.9 (src/dfire.rs:339). */
int ld_dfire_bin_lut(uint8_t *lut_out /* 901 */, double *steps_out /* 21 */, double *interface_d2_out);
/* The cell LUT of the default DFIRE kernel, which tests pairs in f32 and recomputes in f64 only
 * where the f32 distance cannot decide the reference's result (host-side, no GPU; for tests).
 * With D' = cells_per_unit * 4 d2 + 1/2 evaluated in f32 on coordinates within `ubound` of the
 * frame centre (record units), word = words_out[min((unsigned)D', 1024 * cells_per_unit)]:
 *   word < 0x00800000            byte offset of the bin within a table patch: every f64 d2 that can
 *                                produce this cell has that bin, is inside the cutoff, sets no flag
 *   word == 0x00800000           every such d2 is beyond the cutoff (src/dfire.rs:334)
 *   word & 0x40000000            flagged: (word >> 24) & 15 == 1: one bin step exactly at the cell's
 *                                middle, bits 0..11 the offset below it, bits 12..23 its growth above;
 *                                other codes: decided by comparisons on the f64 distance.
 * eps_out: the bound on |D_f32 - 4 d2| (units of 4 d2) the LUT was built for.
 * words_out: 1028 * cells_per_unit entries; cells_per_unit is 1 or 2. */
int ld_dfire_packed_lut(int cells_per_unit, double ubound, uint32_t *words_out, double *eps_out);
/* The cell LUT of the block-major DFIRE kernels (kernels/dfire_bm.hpp; host-side, no GPU), for tests.  The kernel
 * computes E = 14583.5 - 64 d2 in f32 (error below eps_cells / 2, which depends on the frame `ubound` in record units
 * of 1/8 A and on the ligand's largest |local coordinate| `lig_extent` in A) and reads codes_out[floor(E)], E < 0 reading
 * cell 0:
 *   code 160                     every f64 d2 that can produce this cell is beyond the cutoff (src/dfire.rs:334)
 *   code 8 * bin, bin 0..19      every such d2 has that bin (src/dfire.rs:336-337) and is inside the cutoff: the byte offset of
 *                                the bin's slot in a table row of the kernel
 *   code 168                     flagged (the slot of the row's marker): the pair is recomputed in f64 -- a bin step or the
 *                                cutoff inside the cell's interval (hence r = 15.0 exactly, the reference's read past the row, :338)
 * codes_out: 14592 entries. */
int ld_dfire_bm_lut(double ubound, double lig_extent, uint8_t *codes_out, double *eps_cells_out);
/* The fixed-point scale of that kernel (host-side, no GPU): table values enter a pose's sum as rint(v * scale), scale =
 * 2^(44 - e - x), 2^e >= table_vmax, integer adds in any order (the sum src/dfire.rs:325-345 takes in f64).  x makes the sum of
 * one (pose, ligand tile) -- 64 ligand atoms x the receptor atoms within `reach` = cutoff + the tile's radius of its centre --
 * fit 63 bits: reach_count_out = an upper bound on the receptor atoms inside ANY ball of that radius (n_rec itself below 8192
 * atoms: no search), x = extra_bits_out = the bits that count takes beyond 2^13.  scale_out = 0.0: no scale fits (a table
 * value beyond 1024 or not finite, or a count beyond 2^23) -- such a scorer runs the pose-major kernels. */
int ld_dfire_bm_fix_scale(const double *rec_xyz /* n_rec x 3 */, size_t n_rec, double reach, double table_vmax,
                          uint64_t *reach_count_out, int *extra_bits_out, double *scale_out);
/* The atom order the tiled DFIRE kernel uses (host-side, no GPU): order_out[slot] = original atom
 * index, UINT32_MAX for padding; length = ceil(n/64)*64.  Consecutive 8 slots ("subtile") and 64
 * slots ("tile") are spatially compact; padding only at the tail.  The energy is a plain sum over
 * pairs (src/dfire.rs:325-345), so the order is free.  Returns the padded length. */
size_t ld_spatial_tile_order(const double *xyz /* n x 3 */, size_t n, uint32_t *order_out);
/* The complete layout the DFIRE scorer uses for one molecule (host-side, no GPU): the order above
 * refined so that atom types which share a 128-byte patch of the re-laid-out potential sit in the
 * same subtile, and the renumbering of the DFIRE types (type_perm_out[type] = number in the patch
 * layout; 2k and 2k+1 share patches).  order_out: ceil(n/64)*64 entries, type_perm_out: 169.
 * Returns the padded length. */
size_t ld_dfire_tile_layout(const double *xyz /* n x 3 */, const uint32_t *dfire_types /* n */, size_t n,
                            uint32_t *order_out, uint32_t *type_perm_out);

/* rand 0.7.3 StdRng::seed_from_u64 -> the 8 ChaCha20 key words the GSO kernel uses (src/lib.rs:38). */
void ld_stdrng_key(uint64_t seed, uint32_t key_out[8]);

/* DFIRE::load_potentials (src/dfire.rs:236-257): first 169*169*20 lines of a text file. */
int ld_load_dcparams(const char *path, double *out /* LD_DFIRE_TABLE_LEN */);

/* introspection (host copies of what the model builder produced) */
size_t ld_scorer_num_atoms(const ld_scorer *s, int side /* 0 receptor, 1 ligand */);
size_t ld_scorer_pose_len(const ld_scorer *s);  /* 7, or 7 + anm_rec + anm_lig when use_anm */
int ld_scorer_method(const ld_scorer *s);
int ld_scorer_model_arrays(const ld_scorer *s, int side, double *coordinates /* n*3 or NULL */,
                           uint32_t *dfire_types /* n or NULL */, double *ele_charges, double *vdw_charges,
                           double *vdw_radii);

/* Bind the handle to a HIP stream (hipStream_t passed as void*); NULL = the default
 * stream.  All *_device calls and kernels of this handle are enqueued there. */
int ld_scorer_set_stream(ld_scorer *s, void *hip_stream);

/* `Score::energy` (src/scoring.rs:11-19; impls src/dfire.rs:264-363, src/dna.rs:410-529):
 * one pose in, one f64 out; rec_nm / lig_nm may be NULL when the scorer has no ANM.
 * Synchronous (host pointers). */
int ld_scorer_energy(ld_scorer *s, const double translation[3], const double rotation_wxyz[4],
                     const double *rec_nmodes, const double *lig_nmodes, double *energy_out);

/* Batched form of the same call: what Swarm::update_luciferin (src/swarm.rs:66-70) does
 * one glowworm at a time.  poses: n rows of `stride` doubles (stride >= pose_len).
 * Host-pointer version copies in/out and synchronises. */
int ld_scorer_energy_batch(ld_scorer *s, size_t n, const double *poses, size_t stride, double *energies_out);

/* Device-pointer version: poses and energies already live in HBM (hipMalloc / a torch CUDA
 * tensor's data_ptr); asynchronous on the handle's stream.  `active` (device, n bytes, may
 * be NULL) skips poses whose byte is 0 and leaves their output untouched -- this is the
 * `if self.moved || self.step == 0` of Glowworm::compute_luciferin (src/glowworm.rs:62).
 * `pair_counts` (device, n x uint32, may be NULL) receives the number of atom pairs inside
 * the outer cutoff (DFIRE d2 <= 225, src/dfire.rs:334; DNA d2 <= 900, src/dna.rs:481): the
 * P_cut of the algorithmic-bytes model. */
int ld_scorer_energy_batch_device(ld_scorer *s, size_t n, const double *d_poses, size_t stride,
                                  const uint8_t *d_active, double *d_energies_out, uint32_t *d_pair_counts);

/* Diagnostics of the box-culled DFIRE kernel: after a ld_scorer_energy_batch_device call WITH
 * pair_counts, the number of 8x8 atom-pair blocks each of those n poses actually evaluated
 * (64 pair tests each; compare with n_rec*n_lig/64 for all pairs).  Synchronises.  Returns
 * LD_ERR_UNSUPPORTED for scorers that run the all-pairs kernel. */
int ld_scorer_last_block_counts(ld_scorer *s, size_t n, uint32_t *blocks_out_host);

/* Diagnostics of the block-major DFIRE path: the number of receptor subtiles (8 atoms of the tile order) whose atoms' rows of the
 * potential -- atoma * 169 * 20 + atomb * 20 + bin, src/dfire.rs:338, all 20 bins and the read past the row at r = 15.0 -- are 0.0
 * against every ligand type of the complex (membrane beads, if the DCparams at hand has zero rows for them).  Such a subtile adds
 * nothing to any sum: the culling lists its blocks within the interface distance only (r <= 2.45 A, src/dfire.rs:339; never, when
 * neither it nor the ligand holds a restraint atom or a bead).  0 for a table without such rows and for the other kernels. */
int ld_scorer_bm_quiet_subtiles(const ld_scorer *s, uint32_t *count_out);

/* Per-launch facts for the measurement harness. */
typedef struct ld_kernel_info {
    const char *pair_kernel_name; /* symbol of the dominant (pair loop) kernel */
    uint32_t block_threads;
    uint32_t receptor_chunks;     /* workgroups per pose */
    uint32_t lds_bytes;
    uint64_t pair_tests_per_pose; /* n_rec * n_lig */
    uint64_t stream_bytes_per_pose; /* algorithmic bytes excluding the 8*P_cut gather term */
} ld_kernel_info;
int ld_scorer_kernel_info(const ld_scorer *s, ld_kernel_info *out);

/* Measurement hook: when enabled, every ld_scorer_energy_batch_device call brackets its
 * pair kernel with HIP events on the handle's stream.  ld_scorer_pair_kernel_time
 * synchronises those events and returns the summed duration (ms) and launch count since
 * the last reset; reading also resets. */
int ld_scorer_enable_timing(ld_scorer *s, int enable);
int ld_scorer_pair_kernel_time(ld_scorer *s, double *total_ms_out, uint64_t *launches_out);

/* ------------------------------------------------------------------------------------
 * GSO: batched over independent swarms.  Replaces GSO::new / GSO::run
 * (src/lib.rs:27-58), Swarm (src/swarm.rs) and Glowworm (src/glowworm.rs).
 * ---------------------------------------------------------------------------------- */
typedef struct ld_gso ld_gso;

/* positions: n_swarms x n_glowworms rows of pose_len doubles (src/swarm.rs:26-64).
 * seeds: one u64 per swarm (GSO::new's `seed`, src/lib.rs:38), or NULL for DEFAULT_SEED
 * 324324 (src/constants.rs:2) everywhere. */
ld_gso *ld_gso_create(ld_scorer *scorer, size_t n_swarms, size_t n_glowworms, const double *positions,
                      const uint64_t *seeds);
void ld_gso_destroy(ld_gso *g);

/* One iteration of the loop body of GSO::run (src/lib.rs:47-50): update_luciferin
 * (pose-energy kernel over the glowworms that moved) then movement_phase.  Asynchronous. */
int ld_gso_step(ld_gso *g);
/* `steps` iterations back to back (hipGraph replay when available). */
int ld_gso_run(ld_gso *g, uint32_t steps);
uint32_t ld_gso_steps_done(const ld_gso *g);
uint64_t ld_gso_num_evals(ld_gso *g); /* energy evaluations so far, all swarms (synchronises) */

/* Snapshot of one swarm (synchronises).  Any pointer may be NULL.
 * poses: n_glowworms x pose_len; neighbors = neighbour count of the last movement phase. */
int ld_gso_read(ld_gso *g, size_t swarm, double *poses, double *luciferin, double *vision_range,
                double *scoring, int32_t *n_neighbors, int32_t *moved, int32_t *target);

/* Swarm::save (src/swarm.rs:128-167): writes "<dir>/gso_<step>.out" for one swarm. */
int ld_gso_save(ld_gso *g, size_t swarm, uint32_t step, const char *dir);
/* The same for many swarms of one ld_gso at once (what a multi-swarm launcher does after a save
 * step): dirs[k] receives swarm swarms[k].  The state is copied from the device once and the files
 * are written by a few host threads. */
int ld_gso_save_many(ld_gso *g, size_t n, const size_t *swarms, const char *const *dirs, uint32_t step);

/* ------------------------------------------------------------------------------------
 * The reference command line (src/bin/lightdock-rust.rs:77-333) as a function:
 *   argv = { prog, setup.json, initial_positions_N.dat, steps, dfire|dna|pydock }
 * Same stdout lines, same files, same "usage errors return 0" behaviour.
 * ---------------------------------------------------------------------------------- */
int ld_cli_main(int argc, char **argv);

#ifdef __cplusplus
}
#endif
#endif /* LIGHTDOCK_HIP_H */
