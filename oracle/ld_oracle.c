/*
 * ld_oracle.c -- CPU ORACLE (test infrastructure, see ld_oracle.h).
 *
 * Plain C restatement of lightdock-rust v0.3.2's GSO + DFIRE/DNA path.  The loops keep
 * the reference's order of floating-point operations (receptor outer / ligand inner,
 * sequential +=, no FMA contraction: build with -ffp-contract=off) so that the
 * reference's exact-equality known-answer tests are reproducible bit for bit.
 */
#define _GNU_SOURCE
#include "ld_oracle.h"
#include "dna_tables.h"

#include <errno.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread char g_err[512];

static void set_err(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
const char *orc_last_error(void) { return g_err; }
void orc_free(void *p) { free(p); }

/* ------------------------------------------------------------------------------------
 * constants, src/constants.rs:1-28
 * ---------------------------------------------------------------------------------- */
#define DEFAULT_SEED 324324ULL
#define DEFAULT_TRANSLATION_STEP 0.5
#define DEFAULT_ROTATION_STEP 0.5
#define DEFAULT_NMODES_STEP 0.5
#define LINEAR_THRESHOLD 0.9995
#define INTERFACE_CUTOFF 3.9
#define INTERFACE_CUTOFF2 (INTERFACE_CUTOFF * INTERFACE_CUTOFF)
#define MEMBRANE_PENALTY_SCORE 999.0

/* ------------------------------------------------------------------------------------
 * Quaternion, src/qt.rs.  q = {w, x, y, z}
 * ---------------------------------------------------------------------------------- */
void orc_q_conjugate(const double q[4], double out[4]) { /* qt.rs:24-26 */
    out[0] = q[0]; out[1] = -q[1]; out[2] = -q[2]; out[3] = -q[3];
}
double orc_q_dot(const double a[4], const double b[4]) { /* qt.rs:28-30 */
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
}
double orc_q_norm2(const double q[4]) { /* qt.rs:32-34 */
    return q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
}
double orc_q_norm(const double q[4]) { /* qt.rs:36-38 */
    return sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
}
void orc_q_normalize(double q[4]) { /* qt.rs:40-46 */
    double n = orc_q_norm(q);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
void orc_q_inverse(const double q[4], double out[4]) { /* qt.rs:48-50, Div qt.rs:187-197 */
    double c[4];
    orc_q_conjugate(q, c);
    double n2 = orc_q_norm2(q);
    out[0] = c[0] / n2; out[1] = c[1] / n2; out[2] = c[2] / n2; out[3] = c[3] / n2;
}
double orc_q_distance(const double a[4], const double b[4]) { /* qt.rs:52-55 */
    double d = orc_q_dot(a, b);
    return 1.0 - d * d;
}
void orc_q_mul(const double a[4], const double b[4], double out[4]) { /* qt.rs:174-185 */
    double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    out[0] = w; out[1] = x; out[2] = y; out[3] = z;
}
void orc_q_rotate(const double q[4], const double v[3], double out[3]) { /* qt.rs:57-61 */
    double vq[4] = {0.0, v[0], v[1], v[2]};
    double inv[4], qv[4], r[4];
    orc_q_inverse(q, inv);
    orc_q_mul(q, vq, qv); /* (*self * v) * self.inverse() */
    orc_q_mul(qv, inv, r);
    out[0] = r[1]; out[1] = r[2]; out[2] = r[3];
}
static void q_scale(const double q[4], double s, double out[4]) { /* qt.rs:161-172 */
    out[0] = s * q[0]; out[1] = s * q[1]; out[2] = s * q[2]; out[3] = s * q[3];
}
void orc_q_lerp(const double a[4], const double b[4], double t, double out[4]) { /* qt.rs:63-65 */
    double sa[4], sb[4];
    q_scale(a, 1.0 - t, sa);
    q_scale(b, t, sb);
    for (int i = 0; i < 4; i++) out[i] = sa[i] + sb[i];
}
void orc_q_slerp(const double a[4], const double b[4], double t, double out[4]) { /* qt.rs:67-91 */
    double q1[4] = {a[0], a[1], a[2], a[3]};
    double q2[4] = {b[0], b[1], b[2], b[3]};
    orc_q_normalize(q1);
    orc_q_normalize(q2);
    double q_dot = orc_q_dot(q1, q2);
    if (q_dot < 0.0) { /* short path */
        for (int i = 0; i < 4; i++) q1[i] = -q1[i];
        q_dot *= -1.0;
    }
    if (q_dot > LINEAR_THRESHOLD) {
        double r[4];
        for (int i = 0; i < 4; i++) r[i] = q1[i] + t * (q2[i] - q1[i]); /* q1 + (q2 - q1) * t; Mul<f64> is scalar*component */
        orc_q_normalize(r);
        for (int i = 0; i < 4; i++) out[i] = r[i];
    } else {
        q_dot = fmax(fmin(q_dot, 1.0), -1.0);
        double omega = acos(q_dot);
        double so = sin(omega);
        double s1 = sin((1.0 - t) * omega) / so;
        double s2 = sin(t * omega) / so;
        for (int i = 0; i < 4; i++) out[i] = s1 * q1[i] + s2 * q2[i];
    }
}

/* ------------------------------------------------------------------------------------
 * rand 0.7.3 StdRng = rand_chacha 0.2 ChaCha20Rng; seed_from_u64 = rand_core 0.5 PCG32
 * expander.  Not in /root/reference (Cargo.toml:12 dependency); restated from the
 * published algorithm, pinned by qt.rs:451-463 and the gso goldens.
 * ---------------------------------------------------------------------------------- */
struct orc_rng {
    uint32_t key[8];
    uint64_t counter;  /* next block number */
    uint32_t buf[64];  /* 4 blocks, as the crate buffers them */
    int index;         /* next unread word in buf; 64 = empty */
};

static inline uint32_t rotl32(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }
#define QR(a, b, c, d)                      \
    a += b; d ^= a; d = rotl32(d, 16);      \
    c += d; b ^= c; b = rotl32(b, 12);      \
    a += b; d ^= a; d = rotl32(d, 8);       \
    c += d; b ^= c; b = rotl32(b, 7);

static void chacha20_block(const uint32_t key[8], uint64_t counter, uint32_t out[16]) {
    uint32_t s[16], x[16];
    s[0] = 0x61707865u; s[1] = 0x3320646eu; s[2] = 0x79622d32u; s[3] = 0x6b206574u;
    for (int i = 0; i < 8; i++) s[4 + i] = key[i];
    s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32);
    s[14] = 0; s[15] = 0; /* stream id 0 */
    memcpy(x, s, sizeof x);
    for (int r = 0; r < 10; r++) {
        QR(x[0], x[4], x[8], x[12]) QR(x[1], x[5], x[9], x[13])
        QR(x[2], x[6], x[10], x[14]) QR(x[3], x[7], x[11], x[15])
        QR(x[0], x[5], x[10], x[15]) QR(x[1], x[6], x[11], x[12])
        QR(x[2], x[7], x[8], x[13]) QR(x[3], x[4], x[9], x[14])
    }
    for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}

orc_rng *orc_rng_new(uint64_t seed) {
    orc_rng *r = (orc_rng *)calloc(1, sizeof *r);
    uint64_t state = seed;
    for (int i = 0; i < 8; i++) { /* rand_core::SeedableRng::seed_from_u64 */
        state = state * 6364136223846793005ULL + 11634580027462260723ULL;
        uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
        uint32_t rot = (uint32_t)(state >> 59);
        r->key[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    }
    r->counter = 0;
    r->index = 64;
    return r;
}
void orc_rng_free(orc_rng *r) { free(r); }

static void rng_refill(orc_rng *r) {
    for (int b = 0; b < 4; b++) chacha20_block(r->key, r->counter + (uint64_t)b, r->buf + 16 * b);
    r->counter += 4;
}
uint64_t orc_rng_next_u64(orc_rng *r) { /* rand_core::block::BlockRng::next_u64 */
    if (r->index < 63) {
        uint64_t v = ((uint64_t)r->buf[r->index + 1] << 32) | r->buf[r->index];
        r->index += 2;
        return v;
    } else if (r->index >= 64) {
        rng_refill(r);
        r->index = 2;
        return ((uint64_t)r->buf[1] << 32) | r->buf[0];
    } else { /* index == 63: straddles two buffers (never hit with u64-only use) */
        uint64_t lo = r->buf[63];
        rng_refill(r);
        r->index = 1;
        return ((uint64_t)r->buf[0] << 32) | lo;
    }
}
double orc_rng_f64(orc_rng *r) { /* rand::distributions::Standard for f64: 53 bits */
    return (double)(orc_rng_next_u64(r) >> 11) * (1.0 / 9007199254740992.0);
}
void orc_q_random(orc_rng *r, double out[4]) { /* qt.rs:93-103 */
    const double PI = 3.14159265358979323846264338327950288;
    double u1 = orc_rng_f64(r), u2 = orc_rng_f64(r), u3 = orc_rng_f64(r);
    out[0] = sqrt(1.0 - u1) * sin(2.0 * PI * u2);
    out[1] = sqrt(1.0 - u1) * cos(2.0 * PI * u2);
    out[2] = sqrt(u1) * sin(2.0 * PI * u3);
    out[3] = sqrt(u1) * cos(2.0 * PI * u3);
}

/* ------------------------------------------------------------------------------------
 * PDB reader.  The reference uses pdbtbx 0.11.0 (Cargo.toml:15; not vendored) and walks
 * chains -> residues -> atoms (dfire.rs:133-144).  pdbtbx files every ATOM/HETATM record
 * under its chain id, then its (serial, insertion code) residue, then its (name, altloc)
 * conformer, each kept in first-appearance order; we rebuild that order here.
 * ---------------------------------------------------------------------------------- */
typedef struct {
    char name[8];
    char resname[8];
    char chain[4];
    long resseq;
    char icode; /* ' ' if none */
    char altloc;
    double x, y, z;
    /* grouping ranks */
    int chain_rank, res_rank, conf_rank, file_rank;
} pdb_atom;

typedef struct {
    pdb_atom *atoms;
    size_t n;
} pdb_t;

static void trim_copy(char *dst, size_t cap, const char *src, size_t len) {
    size_t b = 0, e = len;
    while (b < e && (src[b] == ' ' || src[b] == '\t')) b++;
    while (e > b && (src[e - 1] == ' ' || src[e - 1] == '\t' || src[e - 1] == '\r' || src[e - 1] == '\n')) e--;
    size_t n = e - b;
    if (n >= cap) n = cap - 1;
    memcpy(dst, src + b, n);
    dst[n] = 0;
}
static double field_f64(const char *line, size_t len, size_t a, size_t b) {
    char tmp[32];
    if (a >= len) return 0.0;
    if (b > len) b = len;
    trim_copy(tmp, sizeof tmp, line + a, b - a);
    return strtod(tmp, NULL);
}
static int cmp_pdb_atom(const void *pa, const void *pb) {
    const pdb_atom *a = (const pdb_atom *)pa, *b = (const pdb_atom *)pb;
    if (a->chain_rank != b->chain_rank) return a->chain_rank < b->chain_rank ? -1 : 1;
    if (a->res_rank != b->res_rank) return a->res_rank < b->res_rank ? -1 : 1;
    if (a->conf_rank != b->conf_rank) return a->conf_rank < b->conf_rank ? -1 : 1;
    return a->file_rank < b->file_rank ? -1 : (a->file_rank > b->file_rank);
}

static int pdb_open(const char *path, pdb_t *out) {
    FILE *f = fopen(path, "r");
    if (!f) { set_err("cannot open PDB file %s: %s", path, strerror(errno)); return -1; }
    size_t cap = 1024, n = 0;
    pdb_atom *atoms = (pdb_atom *)malloc(cap * sizeof *atoms);
    char *line = NULL;
    size_t lcap = 0;
    ssize_t got;
    while ((got = getline(&line, &lcap, f)) > 0) {
        size_t len = (size_t)got;
        if (len < 54) continue;
        if (strncmp(line, "ATOM  ", 6) != 0 && strncmp(line, "HETATM", 6) != 0) continue;
        if (n == cap) { cap *= 2; atoms = (pdb_atom *)realloc(atoms, cap * sizeof *atoms); }
        pdb_atom *a = &atoms[n];
        memset(a, 0, sizeof *a);
        trim_copy(a->name, sizeof a->name, line + 12, 4);
        a->altloc = line[16];
        trim_copy(a->resname, sizeof a->resname, line + 17, 3);
        trim_copy(a->chain, sizeof a->chain, line + 21, 1);
        char tmp[16];
        trim_copy(tmp, sizeof tmp, line + 22, 4);
        a->resseq = strtol(tmp, NULL, 10);
        a->icode = line[26];
        a->x = field_f64(line, len, 30, 38);
        a->y = field_f64(line, len, 38, 46);
        a->z = field_f64(line, len, 46, 54);
        a->file_rank = (int)n;
        n++;
    }
    free(line);
    fclose(f);
    /* first-appearance ranks */
    int n_chain = 0;
    for (size_t i = 0; i < n; i++) {
        pdb_atom *a = &atoms[i];
        a->chain_rank = a->res_rank = a->conf_rank = -1;
        for (size_t j = 0; j < i; j++) {
            pdb_atom *b = &atoms[j];
            if (strcmp(a->chain, b->chain) != 0) continue;
            a->chain_rank = b->chain_rank;
            if (a->resseq == b->resseq && a->icode == b->icode) {
                a->res_rank = b->res_rank;
                if (strcmp(a->resname, b->resname) == 0 && a->altloc == b->altloc) {
                    a->conf_rank = b->conf_rank;
                    break;
                }
            }
        }
        if (a->chain_rank < 0) a->chain_rank = n_chain++;
        if (a->res_rank < 0) a->res_rank = (int)i;  /* unique, increasing with first appearance */
        if (a->conf_rank < 0) a->conf_rank = (int)i;
    }
    qsort(atoms, n, sizeof *atoms, cmp_pdb_atom);
    out->atoms = atoms;
    out->n = n;
    return 0;
}

/* ------------------------------------------------------------------------------------
 * Docking model shared by DFIRE and DNA (dfire.rs:104-190, dna.rs:235-364)
 * ---------------------------------------------------------------------------------- */
typedef struct {
    size_t n;
    uint32_t *atoms;        /* DFIRE types, dfire.rs:105 */
    double *coordinates;    /* n*3, dfire.rs:106 */
    uint32_t *membrane; size_t n_membrane;
    /* active restraints: res_id -> atom indices (HashMap in the reference; order irrelevant) */
    char (*restraint_ids)[32]; size_t n_groups;
    uint32_t *group_offsets; /* n_groups+1 */
    uint32_t *group_atoms;
    size_t n_passive_groups; /* parsed, unused by energy (dfire.rs:164-175) */
    int num_anm;
    double *nmodes; size_t nmodes_len;
    double *vdw_radii, *vdw_charges, *ele_charges; /* DNA, dna.rs:243-245 */
} model_t;

struct orc_scorer {
    int method;
    int use_anm;
    model_t receptor, ligand;
    double *potential; /* DFIRE */
};

static void model_free(model_t *m) {
    free(m->atoms); free(m->coordinates); free(m->membrane); free(m->restraint_ids);
    free(m->group_offsets); free(m->group_atoms); free(m->nmodes);
    free(m->vdw_radii); free(m->vdw_charges); free(m->ele_charges);
    memset(m, 0, sizeof *m);
}

/* --- DFIRE atom typing, dfire.rs:18-101 ------------------------------------------- */
static int dfire_r3_to_numerical(const char *res) { /* dfire.rs:18-46 */
    static const char *names[] = {"ALA", "CYS", "ASP", "GLU", "PHE", "GLY", "HIS", "ILE", "LYS", "LEU", "MET",
                                  "ASN", "PRO", "GLN", "ARG", "SER", "THR", "VAL", "TRP", "TYR", "MMB"};
    for (int i = 0; i < 21; i++)
        if (strcmp(res, names[i]) == 0) return i;
    if (strcmp(res, "MMY") == 0) return 0;
    return -1;
}
/* ATOMNUMBER (dfire.rs:56-77): "<RES><ATOM>" -> column; listed per residue in column order. */
static const char *const DFIRE_ATOMNUMBER[][15] = {
    {"ALA", "N", "CA", "C", "O", "CB", 0},
    {"CYS", "N", "CA", "C", "O", "CB", "SG", 0},
    {"ASP", "N", "CA", "C", "O", "CB", "CG", "OD1", "OD2", 0},
    {"GLU", "N", "CA", "C", "O", "CB", "CG", "CD", "OE1", "OE2", 0},
    {"PHE", "N", "CA", "C", "O", "CB", "CG", "CD1", "CD2", "CE1", "CE2", "CZ", 0},
    {"GLY", "N", "CA", "C", "O", 0},
    {"HIS", "N", "CA", "C", "O", "CB", "CG", "ND1", "CD2", "CE1", "NE2", 0},
    {"ILE", "N", "CA", "C", "O", "CB", "CG1", "CG2", "CD1", 0},
    {"LYS", "N", "CA", "C", "O", "CB", "CG", "CD", "CE", "NZ", 0},
    {"LEU", "N", "CA", "C", "O", "CB", "CG", "CD1", "CD2", 0},
    {"MET", "N", "CA", "C", "O", "CB", "CG", "SD", "CE", 0},
    {"ASN", "N", "CA", "C", "O", "CB", "CG", "OD1", "ND2", 0},
    {"PRO", "N", "CA", "C", "O", "CB", "CG", "CD", 0},
    {"GLN", "N", "CA", "C", "O", "CB", "CG", "CD", "OE1", "NE2", 0},
    {"ARG", "N", "CA", "C", "O", "CB", "CG", "CD", "NE", "CZ", "NH1", "NH2", 0},
    {"SER", "N", "CA", "C", "O", "CB", "OG", 0},
    {"THR", "N", "CA", "C", "O", "CB", "OG1", "CG2", 0},
    {"VAL", "N", "CA", "C", "O", "CB", "CG1", "CG2", 0},
    {"TRP", "N", "CA", "C", "O", "CB", "CG", "CD1", "CD2", "CE2", "NE1", "CE3", "CZ3", "CH2", "CZ2"},
    {"TYR", "N", "CA", "C", "O", "CB", "CG", "CD1", "CD2", "CE1", "CE2", "CZ", "OH", 0},
    {"MMB", "BJ", 0},
    {"MMY", "DU", 0},
};
static int dfire_atomnumber(const char *key) { /* key = resname+atomname concatenated */
    for (size_t r = 0; r < sizeof DFIRE_ATOMNUMBER / sizeof DFIRE_ATOMNUMBER[0]; r++) {
        const char *res = DFIRE_ATOMNUMBER[r][0];
        size_t rl = strlen(res);
        if (strncmp(key, res, rl) != 0) continue;
        for (int c = 1; c < 15 && DFIRE_ATOMNUMBER[r][c]; c++)
            if (strcmp(key + rl, DFIRE_ATOMNUMBER[r][c]) == 0) return c - 1;
    }
    return -1;
}
/* ATOMRES (dfire.rs:80-101): rows by r3_to_numerical, columns by ATOMNUMBER */
static const uint8_t DFIRE_ATOMRES[22][14] = {
    {74, 75, 76, 77, 78, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {0, 1, 2, 3, 4, 5, 0, 0, 0, 0, 0, 0, 0, 0},
    {122, 123, 124, 125, 126, 127, 128, 129, 0, 0, 0, 0, 0, 0},
    {113, 114, 115, 116, 117, 118, 119, 120, 121, 0, 0, 0, 0, 0},
    {14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 0, 0, 0},
    {79, 80, 81, 82, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {130, 131, 132, 133, 134, 135, 136, 137, 138, 139, 0, 0, 0, 0},
    {25, 26, 27, 28, 29, 30, 31, 32, 0, 0, 0, 0, 0, 0},
    {151, 152, 153, 154, 155, 156, 157, 158, 159, 0, 0, 0, 0, 0},
    {33, 34, 35, 36, 37, 38, 39, 40, 0, 0, 0, 0, 0, 0},
    {6, 7, 8, 9, 10, 11, 12, 13, 0, 0, 0, 0, 0, 0},
    {105, 106, 107, 108, 109, 110, 111, 112, 0, 0, 0, 0, 0, 0},
    {160, 161, 162, 163, 164, 165, 166, 0, 0, 0, 0, 0, 0, 0},
    {96, 97, 98, 99, 100, 101, 102, 103, 104, 0, 0, 0, 0, 0},
    {140, 141, 142, 143, 144, 145, 146, 147, 148, 149, 150, 0, 0, 0},
    {90, 91, 92, 93, 94, 95, 0, 0, 0, 0, 0, 0, 0, 0},
    {83, 84, 85, 86, 87, 88, 89, 0, 0, 0, 0, 0, 0, 0},
    {41, 42, 43, 44, 45, 46, 47, 0, 0, 0, 0, 0, 0, 0},
    {48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61},
    {62, 63, 64, 65, 66, 67, 68, 69, 70, 71, 72, 73, 0, 0},
    {167, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {74, 75, 76, 77, 78, 0, 0, 0, 0, 0, 0, 0, 0, 0},
};

/* DIST_TO_BINS, dfire.rs:49-53 */
static const int DIST_TO_BINS[51] = {1,  1,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 14,
                                     15, 15, 16, 16, 17, 17, 18, 18, 19, 19, 20, 20, 21, 21, 22, 22, 23,
                                     23, 24, 24, 25, 25, 26, 26, 27, 27, 28, 28, 29, 29, 30, 30, 31, 32};

static size_t rust_f64_as_usize(double d) { /* `d as usize`: saturating, NaN -> 0 */
    if (!(d > 0.0)) return 0;
    if (d >= 18446744073709551615.0) return (size_t)-1;
    return (size_t)d;
}
int orc_dfire_bin(double dist2) { /* dfire.rs:336-337 */
    double d = sqrt(dist2) * 2.0 - 1.0;
    return DIST_TO_BINS[rust_f64_as_usize(d)] - 1;
}

/* --- DNA parameter lookup, dna.rs:65-232 ----------------------------------------- */
static int kv_num_find(const orc_kv_num *t, int n, const char *key, double *out) {
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        int mid = (lo + hi) / 2;
        int c = strcmp(key, t[mid].key);
        if (c == 0) { *out = t[mid].val; return 1; }
        if (c < 0) hi = mid - 1; else lo = mid + 1;
    }
    return 0;
}
static const char *kv_str_find(const orc_kv_str *t, int n, const char *key) {
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        int mid = (lo + hi) / 2;
        int c = strcmp(key, t[mid].key);
        if (c == 0) return t[mid].val;
        if (c < 0) hi = mid - 1; else lo = mid + 1;
    }
    return NULL;
}

/* the six generic entries PYDOCK adds to the AMBER type / charge tables (src/pydock.rs AMBER_TYPES,
 * ELE_CHARGES: "*-C", "*-F", "*-H", "*-N", "*-O", "*-S") */
static const struct { char element; const char *amber; double charge; } PYDOCK_GENERIC[] = {
    {'C', "C", 0.5973}, {'F', "F", -0.342}, {'H', "H", 0.2719}, {'N', "N", -0.4157}, {'O', "O", -0.5679}, {'S', "S", -0.2737},
};

static int str_in_list(const char *s, const char *const *list, int n) {
    for (int i = 0; i < n; i++)
        if (list[i] && strcmp(s, list[i]) == 0) return 1;
    return 0;
}

/* DFIREDockingModel::new (dfire.rs:115-190) / DNADockingModel::new (dna.rs:249-364) */
static int model_build(model_t *m, int method, const char *pdb_path, const char *const *active, int n_active,
                       const char *const *passive, int n_passive, const double *nmodes, size_t nmodes_len,
                       int num_anm) {
    pdb_t pdb;
    memset(m, 0, sizeof *m);
    if (pdb_open(pdb_path, &pdb) != 0) return -1;
    size_t n = pdb.n;
    m->n = n;
    m->atoms = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t));
    m->coordinates = (double *)calloc(n ? 3 * n : 1, sizeof(double));
    m->membrane = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t));
    m->restraint_ids = (char(*)[32])calloc(n ? n : 1, 32);
    uint32_t *atom_group = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t)); /* group id or ~0 */
    char(*passive_ids)[32] = (char(*)[32])calloc(n ? n : 1, 32);
    if (method != ORC_METHOD_DFIRE) {
        m->vdw_radii = (double *)calloc(n ? n : 1, sizeof(double));
        m->vdw_charges = (double *)calloc(n ? n : 1, sizeof(double));
        m->ele_charges = (double *)calloc(n ? n : 1, sizeof(double));
    }
    m->num_anm = num_anm;
    m->nmodes_len = nmodes_len;
    m->nmodes = (double *)malloc((nmodes_len ? nmodes_len : 1) * sizeof(double));
    if (nmodes_len) memcpy(m->nmodes, nmodes, nmodes_len * sizeof(double));

    int rc = 0;
    for (size_t i = 0; i < n && rc == 0; i++) {
        const pdb_atom *a = &pdb.atoms[i];
        char res_id[32]; /* "{chain}.{res}.{serial}" + insertion code, dfire.rs:139-142 */
        int k = snprintf(res_id, sizeof res_id, "%s.%s.%ld", a->chain, a->resname, a->resseq);
        if (a->icode != ' ' && a->icode != 0 && k < (int)sizeof res_id - 1) { res_id[k] = a->icode; res_id[k + 1] = 0; }
        char rec_atom_type[20];
        snprintf(rec_atom_type, sizeof rec_atom_type, "%s%s", a->resname, a->name);
        if (strcmp(rec_atom_type, "MMBBJ") == 0) m->membrane[m->n_membrane++] = (uint32_t)i; /* dfire.rs:146-149 */

        atom_group[i] = (uint32_t)-1;
        if (str_in_list(res_id, active, n_active)) { /* dfire.rs:151-162 */
            size_t g;
            for (g = 0; g < m->n_groups; g++)
                if (strcmp(m->restraint_ids[g], res_id) == 0) break;
            if (g == m->n_groups) { strcpy(m->restraint_ids[g], res_id); m->n_groups++; }
            atom_group[i] = (uint32_t)g;
        }
        if (str_in_list(res_id, passive, n_passive)) { /* dfire.rs:164-175; stored, never read */
            size_t g;
            for (g = 0; g < m->n_passive_groups; g++)
                if (strcmp(passive_ids[g], res_id) == 0) break;
            if (g == m->n_passive_groups) { strcpy(passive_ids[g], res_id); m->n_passive_groups++; }
        }

        if (method == ORC_METHOD_DFIRE) { /* dfire.rs:177-183 */
            int rnuma = dfire_r3_to_numerical(a->resname);
            if (rnuma < 0) { set_err("Residue name not supported in DFIRE scoring function"); rc = -1; break; }
            int anuma = dfire_atomnumber(rec_atom_type);
            if (anuma < 0) { set_err("Not supported atom type \"%s\"", rec_atom_type); rc = -1; break; }
            m->atoms[i] = DFIRE_ATOMRES[rnuma][anuma];
        } else { /* dna.rs:314-358 */
            char atom_id[24];
            snprintf(atom_id, sizeof atom_id, "%s-%s", a->resname, a->name);
            const char *who = method == ORC_METHOD_PYDOCK ? "PYDOCK" : "DNA";
            const char *amber = kv_str_find(ORC_AMBER_TYPES, ORC_AMBER_TYPES_LEN, atom_id);
            double charge, eps, radius;
            int have_charge = 0;
            if (!amber) {
                if (!strcmp(a->name, "H1") || !strcmp(a->name, "H2") || !strcmp(a->name, "H3")) {
                    snprintf(atom_id, sizeof atom_id, "%s-H", a->resname);
                    amber = kv_str_find(ORC_AMBER_TYPES, ORC_AMBER_TYPES_LEN, atom_id);
                } else if (method == ORC_METHOD_PYDOCK) { /* pydock.rs:332-345: "*-<first letter>" */
                    if (a->name[0] == 0) { set_err("PYDOCK Error: Atom element could not be guessed from [\"%s\"]", a->name); rc = -1; break; }
                    snprintf(atom_id, sizeof atom_id, "*-%c", a->name[0]);
                    for (size_t g = 0; g < sizeof PYDOCK_GENERIC / sizeof PYDOCK_GENERIC[0]; g++)
                        if (PYDOCK_GENERIC[g].element == a->name[0]) { amber = PYDOCK_GENERIC[g].amber; charge = PYDOCK_GENERIC[g].charge; have_charge = 1; }
                }
                if (!amber) { set_err("%s Error: Atom [\"%s\"] not supported", who, atom_id); rc = -1; break; }
            }
            if (!have_charge && !kv_num_find(ORC_ELE_CHARGES, ORC_ELE_CHARGES_LEN, atom_id, &charge) &&
                !kv_num_find(ORC_NT_ELE_CHARGES, ORC_NT_ELE_CHARGES_LEN, atom_id, &charge)) {
                set_err("%s Error: Atom [\"%s\"] electrostatics charge not found", who, atom_id); rc = -1; break;
            }
            if (!kv_num_find(ORC_VDW_CHARGES, ORC_VDW_CHARGES_LEN, amber, &eps)) {
                set_err("%s Error: Atom [\"%s\"] VDW charge not found", who, atom_id); rc = -1; break;
            }
            if (!kv_num_find(ORC_VDW_RADII, ORC_VDW_RADII_LEN, amber, &radius)) {
                set_err("%s Error: Atom [\"%s\"] VDW radius not found", who, atom_id); rc = -1; break;
            }
            m->ele_charges[i] = charge;
            m->vdw_charges[i] = eps;
            m->vdw_radii[i] = radius;
        }
        m->coordinates[3 * i] = a->x;
        m->coordinates[3 * i + 1] = a->y;
        m->coordinates[3 * i + 2] = a->z;
    }
    if (rc == 0) { /* CSR of the active groups */
        m->group_offsets = (uint32_t *)calloc(m->n_groups + 1, sizeof(uint32_t));
        m->group_atoms = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t));
        for (size_t i = 0; i < n; i++)
            if (atom_group[i] != (uint32_t)-1) m->group_offsets[atom_group[i] + 1]++;
        for (size_t g = 0; g < m->n_groups; g++) m->group_offsets[g + 1] += m->group_offsets[g];
        uint32_t *fill = (uint32_t *)calloc(m->n_groups + 1, sizeof(uint32_t));
        for (size_t i = 0; i < n; i++)
            if (atom_group[i] != (uint32_t)-1) {
                uint32_t g = atom_group[i];
                m->group_atoms[m->group_offsets[g] + fill[g]++] = (uint32_t)i;
            }
        free(fill);
    }
    free(atom_group);
    free(passive_ids);
    free(pdb.atoms);
    if (rc != 0) model_free(m);
    return rc;
}

int orc_load_dcparams(const char *path, double *out) { /* dfire.rs:236-257 */
    FILE *f = fopen(path, "r");
    if (!f) { set_err("Unable to open DFIRE parameters: %s", path); return -1; }
    char *line = NULL;
    size_t cap = 0;
    size_t n = 0;
    while (n < ORC_DFIRE_TABLE_LEN && getline(&line, &cap, f) > 0) {
        char *end;
        double v = strtod(line, &end); /* trim().parse::<f64>() */
        if (end == line) { set_err("bad DFIRE parameter at line %zu", n + 1); free(line); fclose(f); return -1; }
        out[n++] = v;
    }
    free(line);
    fclose(f);
    /* fewer lines than 169*169*20: the reference then panics on the first out-of-range
     * lookup (dfire.rs:338); we refuse up front. */
    if (n < ORC_DFIRE_TABLE_LEN) { set_err("DFIRE parameters: only %zu values", n); return -1; }
    return 0;
}

orc_scorer *orc_scorer_new(int method, const char *receptor_pdb, const char *ligand_pdb,
                           const char *const *rec_active, int n_rec_active, const char *const *rec_passive,
                           int n_rec_passive, const double *rec_nmodes, size_t rec_nmodes_len, int rec_num_anm,
                           const char *const *lig_active, int n_lig_active, const char *const *lig_passive,
                           int n_lig_passive, const double *lig_nmodes, size_t lig_nmodes_len, int lig_num_anm,
                           int use_anm, const double *potential) {
    if (method != ORC_METHOD_DFIRE && method != ORC_METHOD_DNA && method != ORC_METHOD_PYDOCK) { set_err("Error: method not supported"); return NULL; }
    orc_scorer *s = (orc_scorer *)calloc(1, sizeof *s);
    s->method = method;
    s->use_anm = use_anm;
    if (model_build(&s->receptor, method, receptor_pdb, rec_active, n_rec_active, rec_passive, n_rec_passive,
                    rec_nmodes, rec_nmodes_len, rec_num_anm) != 0) { free(s); return NULL; }
    if (model_build(&s->ligand, method, ligand_pdb, lig_active, n_lig_active, lig_passive, n_lig_passive,
                    lig_nmodes, lig_nmodes_len, lig_num_anm) != 0) { model_free(&s->receptor); free(s); return NULL; }
    if (method == ORC_METHOD_DFIRE) {
        if (!potential) { set_err("Unable to open DFIRE parameters"); orc_scorer_free(s); return NULL; }
        s->potential = (double *)malloc(ORC_DFIRE_TABLE_LEN * sizeof(double));
        memcpy(s->potential, potential, ORC_DFIRE_TABLE_LEN * sizeof(double));
    }
    if (use_anm) { /* bin:233,250 panics */
        if (rec_num_anm > 0 && rec_nmodes_len != s->receptor.n * 3 * (size_t)rec_num_anm) {
            set_err("Number of read ANM in receptor does not correspond to the number of atoms");
            orc_scorer_free(s); return NULL;
        }
        if (lig_num_anm > 0 && lig_nmodes_len != s->ligand.n * 3 * (size_t)lig_num_anm) {
            set_err("Number of read ANM in ligand does not correspond to the number of atoms");
            orc_scorer_free(s); return NULL;
        }
    }
    return s;
}
void orc_scorer_free(orc_scorer *s) {
    if (!s) return;
    model_free(&s->receptor);
    model_free(&s->ligand);
    free(s->potential);
    free(s);
}

static const model_t *side_of(const orc_scorer *s, int side) { return side ? &s->ligand : &s->receptor; }
size_t orc_scorer_num_atoms(const orc_scorer *s, int side) { return side_of(s, side)->n; }
const double *orc_scorer_coordinates(const orc_scorer *s, int side) { return side_of(s, side)->coordinates; }
const uint32_t *orc_scorer_dfire_types(const orc_scorer *s, int side) { return side_of(s, side)->atoms; }
const double *orc_scorer_ele_charges(const orc_scorer *s, int side) { return side_of(s, side)->ele_charges; }
const double *orc_scorer_vdw_charges(const orc_scorer *s, int side) { return side_of(s, side)->vdw_charges; }
const double *orc_scorer_vdw_radii(const orc_scorer *s, int side) { return side_of(s, side)->vdw_radii; }
size_t orc_scorer_num_membrane(const orc_scorer *s, int side) { return side_of(s, side)->n_membrane; }
const uint32_t *orc_scorer_membrane(const orc_scorer *s, int side) { return side_of(s, side)->membrane; }
size_t orc_scorer_num_restraint_groups(const orc_scorer *s, int side) { return side_of(s, side)->n_groups; }
const uint32_t *orc_scorer_restraint_offsets(const orc_scorer *s, int side) { return side_of(s, side)->group_offsets; }
const uint32_t *orc_scorer_restraint_atoms(const orc_scorer *s, int side) { return side_of(s, side)->group_atoms; }

/* scoring.rs:21-36 */
static double satisfied_restraints(const uint8_t *interface, const model_t *m) {
    if (m->n_groups == 0) return 0.0;
    size_t num_residues = 0;
    for (size_t g = 0; g < m->n_groups; g++)
        for (uint32_t k = m->group_offsets[g]; k < m->group_offsets[g + 1]; k++)
            if (interface[m->group_atoms[k]] == 1) { num_residues++; break; }
    return (double)num_residues / (double)m->n_groups;
}
/* scoring.rs:38-47 */
static double membrane_intersection(const uint8_t *interface, const model_t *m) {
    if (m->n_membrane == 0) return 0.0;
    size_t num_beads = 0;
    for (size_t k = 0; k < m->n_membrane; k++) num_beads += interface[m->membrane[k]];
    return (double)num_beads / (double)m->n_membrane;
}

/* pose transform shared by both scorers: dfire.rs:275-323 == dna.rs:419-467 */
static void pose_coordinates(const orc_scorer *s, const double t[3], const double q[4], const double *rec_nm,
                             const double *lig_nm, double *rc, double *lc) {
    const model_t *R = &s->receptor, *L = &s->ligand;
    size_t rn = R->n, ln = L->n;
    memcpy(rc, R->coordinates, 3 * rn * sizeof(double));
    memcpy(lc, L->coordinates, 3 * ln * sizeof(double));
    for (size_t i = 0; i < ln; i++) {
        double *c = &lc[3 * i];
        double rot[3];
        orc_q_rotate(q, c, rot);
        c[0] = rot[0] + t[0];
        c[1] = rot[1] + t[1];
        c[2] = rot[2] + t[2];
        if (s->use_anm && L->num_anm > 0)
            for (size_t k = 0; k < (size_t)L->num_anm; k++) {
                c[0] += L->nmodes[k * ln * 3 + i * 3] * lig_nm[k];
                c[1] += L->nmodes[k * ln * 3 + i * 3 + 1] * lig_nm[k];
                c[2] += L->nmodes[k * ln * 3 + i * 3 + 2] * lig_nm[k];
            }
    }
    for (size_t i = 0; i < rn; i++) {
        double *c = &rc[3 * i];
        if (s->use_anm && R->num_anm > 0)
            for (size_t k = 0; k < (size_t)R->num_anm; k++) {
                c[0] += R->nmodes[k * rn * 3 + i * 3] * rec_nm[k];
                c[1] += R->nmodes[k * rn * 3 + i * 3 + 1] * rec_nm[k];
                c[2] += R->nmodes[k * rn * 3 + i * 3 + 2] * rec_nm[k];
            }
    }
}

/* DNA constants, dna.rs:15-25 */
#define DNA_EPSILON 4.0
#define DNA_FACTOR 332.0
#define DNA_VDW_CUTOFF 1.0
#define DNA_ELEC_DIST_CUTOFF2 (30.0 * 30.0)
#define DNA_VDW_DIST_CUTOFF2 (10.0 * 10.0)
#define DNA_ELEC_MAX_CUTOFF (1.0 * DNA_EPSILON / DNA_FACTOR)
#define DNA_ELEC_MIN_CUTOFF (-1.0 * DNA_EPSILON / DNA_FACTOR)

static inline double powi3(double x) { return x * x * x; }            /* f64::powi(3) */
static inline double powi6(double x) { double x2 = x * x; return x2 * (x2 * x2); } /* f64::powi(6): x^2 * x^4 */

double orc_scorer_energy_ex(const orc_scorer *s, const double t[3], const double q[4], const double *rec_nm,
                            const double *lig_nm, double stats[8]) {
    const model_t *R = &s->receptor, *L = &s->ligand;
    size_t rn = R->n, ln = L->n;
    double *rc = (double *)malloc((3 * rn + 1) * sizeof(double));
    double *lc = (double *)malloc((3 * ln + 1) * sizeof(double));
    uint8_t *iface_r = (uint8_t *)calloc(rn + 1, 1);
    uint8_t *iface_l = (uint8_t *)calloc(ln + 1, 1);
    pose_coordinates(s, t, q, rec_nm, lig_nm, rc, lc);
    double score;
    uint64_t in_cut = 0;
    double raw0 = 0.0, raw1 = 0.0;

    if (s->method == ORC_METHOD_DFIRE) { /* dfire.rs:325-347 */
        score = 0.0;
        for (size_t i = 0; i < rn; i++) {
            double x1 = rc[3 * i], y1 = rc[3 * i + 1], z1 = rc[3 * i + 2];
            size_t atoma = R->atoms[i];
            for (size_t j = 0; j < ln; j++) {
                const double *la = &lc[3 * j];
                double dist = (x1 - la[0]) * (x1 - la[0]) + (y1 - la[1]) * (y1 - la[1]) + (z1 - la[2]) * (z1 - la[2]);
                if (dist <= 225.) {
                    size_t atomb = L->atoms[j];
                    double d = sqrt(dist) * 2.0 - 1.0;
                    size_t dfire_bin = (size_t)DIST_TO_BINS[rust_f64_as_usize(d)] - 1;
                    score += s->potential[atoma * 169 * 20 + atomb * 20 + dfire_bin];
                    in_cut++;
                    if (d <= INTERFACE_CUTOFF) { iface_r[i] = 1; iface_l[j] = 1; }
                }
            }
        }
        raw0 = score;
        score = (score * 0.0157 - 4.7) * -1.0;
    } else { /* dna.rs:469-514 */
        double total_elec = 0.0, total_vdw = 0.0;
        for (size_t i = 0; i < rn; i++) {
            double x1 = rc[3 * i], y1 = rc[3 * i + 1], z1 = rc[3 * i + 2];
            for (size_t j = 0; j < ln; j++) {
                const double *la = &lc[3 * j];
                double distance2 =
                    (x1 - la[0]) * (x1 - la[0]) + (y1 - la[1]) * (y1 - la[1]) + (z1 - la[2]) * (z1 - la[2]);
                if (distance2 <= DNA_ELEC_DIST_CUTOFF2) {
                    double atom_elec = R->ele_charges[i] * L->ele_charges[j] / distance2;
                    if (atom_elec > DNA_ELEC_MAX_CUTOFF) atom_elec = DNA_ELEC_MAX_CUTOFF;
                    if (atom_elec < DNA_ELEC_MIN_CUTOFF) atom_elec = DNA_ELEC_MIN_CUTOFF;
                    total_elec += atom_elec;
                    in_cut++;
                }
                if (distance2 <= DNA_VDW_DIST_CUTOFF2) {
                    double vdw_energy = sqrt(R->vdw_charges[i] * L->vdw_charges[j]);
                    double vdw_radius = R->vdw_radii[i] + L->vdw_radii[j];
                    double p6 = powi6(vdw_radius) / powi3(distance2);
                    double k = vdw_energy * (p6 * p6 - 2.0 * p6);
                    if (k > DNA_VDW_CUTOFF) k = DNA_VDW_CUTOFF;
                    total_vdw += k;
                }
                if (distance2 <= INTERFACE_CUTOFF2) { iface_r[i] = 1; iface_l[j] = 1; }
            }
        }
        raw0 = total_elec;
        raw1 = total_vdw;
        total_elec = total_elec * DNA_FACTOR / DNA_EPSILON;
        score = (total_elec + total_vdw) * -1.0;
    }

    /* dfire.rs:349-361 == dna.rs:516-528 */
    double perc_receptor_restraints = satisfied_restraints(iface_r, R);
    double perc_ligand_restraints = satisfied_restraints(iface_l, L);
    double membrane_penalty = 0.0;
    double intersection = membrane_intersection(iface_r, R);
    if (intersection > 0.0) membrane_penalty = MEMBRANE_PENALTY_SCORE * intersection;
    double energy = score + perc_receptor_restraints * score + perc_ligand_restraints * score - membrane_penalty;

    if (stats) {
        size_t nr = 0, nl = 0;
        for (size_t i = 0; i < rn; i++) nr += iface_r[i];
        for (size_t j = 0; j < ln; j++) nl += iface_l[j];
        stats[0] = raw0; stats[1] = raw1;
        stats[2] = perc_receptor_restraints; stats[3] = perc_ligand_restraints;
        stats[4] = intersection; stats[5] = (double)in_cut;
        stats[6] = (double)nr; stats[7] = (double)nl;
    }
    free(rc); free(lc); free(iface_r); free(iface_l);
    return energy;
}

double orc_scorer_energy(const orc_scorer *s, const double t[3], const double q[4], const double *rec_nm,
                         const double *lig_nm) {
    return orc_scorer_energy_ex(s, t, q, rec_nm, lig_nm, NULL);
}

/* ------------------------------------------------------------------------------------
 * Glowworm / Swarm / GSO: src/glowworm.rs, src/swarm.rs, src/lib.rs
 * ---------------------------------------------------------------------------------- */
typedef struct {
    uint32_t id;
    double translation[3];
    double rotation[4];
    double *rec_nmodes; int n_rec_nm;
    double *lig_nmodes; int n_lig_nm;
    double rho, gamma, beta, luciferin, vision_range, max_vision_range;
    uint32_t max_neighbors;
    uint32_t *neighbors; int n_neighbors;
    double *probabilities;
    double scoring;
    int moved;
    uint32_t step;
    int use_anm;
    uint32_t last_target;
} glowworm_t;

struct orc_gso {
    glowworm_t *g;
    int n, row_len;
    const orc_scorer *scorer;
    orc_rng *rng;
    uint64_t n_evals;
};

orc_gso *orc_gso_new(const double *positions, int n, int row_len, uint64_t seed, const orc_scorer *scorer,
                     int use_anm, int rec_num_anm, int lig_num_anm) {
    if (row_len < 7) { set_err("pose rows need at least 7 columns"); return NULL; }
    orc_gso *G = (orc_gso *)calloc(1, sizeof *G);
    G->n = n; G->row_len = row_len; G->scorer = scorer;
    G->rng = orc_rng_new(seed); /* lib.rs:38 */
    G->g = (glowworm_t *)calloc(n ? n : 1, sizeof(glowworm_t));
    for (int i = 0; i < n; i++) { /* swarm.rs:34-63, glowworm.rs:29-58 */
        const double *p = positions + (size_t)i * row_len;
        glowworm_t *w = &G->g[i];
        w->id = (uint32_t)i;
        w->translation[0] = p[0]; w->translation[1] = p[1]; w->translation[2] = p[2];
        w->rotation[0] = p[3]; w->rotation[1] = p[4]; w->rotation[2] = p[5]; w->rotation[3] = p[6];
        w->rec_nmodes = (double *)calloc(row_len, sizeof(double));
        w->lig_nmodes = (double *)calloc(row_len, sizeof(double));
        if (use_anm && rec_num_anm > 0)
            for (int j = 7; j < 7 + rec_num_anm && j < row_len; j++) w->rec_nmodes[w->n_rec_nm++] = p[j];
        if (use_anm && lig_num_anm > 0)
            for (int j = 7 + rec_num_anm; j < row_len; j++) w->lig_nmodes[w->n_lig_nm++] = p[j];
        w->rho = 0.5; w->gamma = 0.4; w->beta = 0.08; w->luciferin = 5.0; w->vision_range = 0.2;
        w->max_vision_range = 5.0; w->max_neighbors = 5;
        w->neighbors = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t));
        w->probabilities = (double *)calloc(n ? n : 1, sizeof(double));
        w->scoring = 0.0; w->moved = 0; w->step = 0; w->use_anm = use_anm;
        w->last_target = w->id;
    }
    return G;
}
void orc_gso_free(orc_gso *G) {
    if (!G) return;
    for (int i = 0; i < G->n; i++) {
        free(G->g[i].rec_nmodes); free(G->g[i].lig_nmodes); free(G->g[i].neighbors); free(G->g[i].probabilities);
    }
    free(G->g);
    orc_rng_free(G->rng);
    free(G);
}

static double glowworm_distance(const glowworm_t *one, const glowworm_t *two) { /* glowworm.rs:193-202 */
    double x1 = one->translation[0], x2 = two->translation[0];
    double y1 = one->translation[1], y2 = two->translation[1];
    double z1 = one->translation[2], z2 = two->translation[2];
    return sqrt((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2));
}

static void move_towards(glowworm_t *w, uint32_t other_id, const double *other_position,
                         const double *other_rotation, const double *other_anm_rec, const double *other_anm_lig) {
    /* glowworm.rs:128-190 */
    w->moved = w->id != other_id;
    if (w->id == other_id) return;
    double delta_x[3] = {other_position[0] - w->translation[0], other_position[1] - w->translation[1],
                         other_position[2] - w->translation[2]};
    double norm = sqrt(delta_x[0] * delta_x[0] + delta_x[1] * delta_x[1] + delta_x[2] * delta_x[2]);
    double coef = DEFAULT_TRANSLATION_STEP / norm;
    delta_x[0] *= coef; delta_x[1] *= coef; delta_x[2] *= coef;
    w->translation[0] += delta_x[0]; w->translation[1] += delta_x[1]; w->translation[2] += delta_x[2];
    double r[4];
    orc_q_slerp(w->rotation, other_rotation, DEFAULT_ROTATION_STEP, r);
    memcpy(w->rotation, r, sizeof r);
    if (w->use_anm && w->n_rec_nm > 0) {
        double cum_norm = 0.0;
        double delta[64];
        for (int i = 0; i < w->n_rec_nm; i++) {
            double diff = other_anm_rec[i] - w->rec_nmodes[i];
            delta[i] = diff;
            cum_norm += diff * diff;
        }
        double c = DEFAULT_NMODES_STEP / sqrt(cum_norm);
        for (int i = 0; i < w->n_rec_nm; i++) { delta[i] *= c; w->rec_nmodes[i] += delta[i]; }
    }
    if (w->use_anm && w->n_lig_nm > 0) {
        double cum_norm = 0.0;
        double delta[64];
        for (int i = 0; i < w->n_lig_nm; i++) {
            double diff = other_anm_lig[i] - w->lig_nmodes[i];
            delta[i] = diff;
            cum_norm += diff * diff;
        }
        double c = DEFAULT_NMODES_STEP / sqrt(cum_norm);
        for (int i = 0; i < w->n_lig_nm; i++) { delta[i] *= c; w->lig_nmodes[i] += delta[i]; }
    }
}

void orc_gso_step(orc_gso *G) {
    int n = G->n;
    /* Swarm::update_luciferin, swarm.rs:66-70 -> Glowworm::compute_luciferin, glowworm.rs:61-72 */
    for (int i = 0; i < n; i++) {
        glowworm_t *w = &G->g[i];
        if (w->moved || w->step == 0) {
            w->scoring = orc_scorer_energy(G->scorer, w->translation, w->rotation, w->rec_nmodes, w->lig_nmodes);
            G->n_evals++;
        }
        w->luciferin = (1.0 - w->rho) * w->luciferin + w->gamma * w->scoring;
        w->step += 1;
    }
    /* Swarm::movement_phase, swarm.rs:72-126 */
    int rl = G->row_len;
    double *positions = (double *)malloc((size_t)(n ? n : 1) * 3 * sizeof(double));
    double *rotations = (double *)malloc((size_t)(n ? n : 1) * 4 * sizeof(double));
    double *anm_recs = (double *)malloc((size_t)(n ? n : 1) * rl * sizeof(double));
    double *anm_ligs = (double *)malloc((size_t)(n ? n : 1) * rl * sizeof(double));
    double *luciferins = (double *)malloc((size_t)(n ? n : 1) * sizeof(double));
    for (int i = 0; i < n; i++) {
        const glowworm_t *w = &G->g[i];
        memcpy(positions + 3 * i, w->translation, 3 * sizeof(double));
        memcpy(rotations + 4 * i, w->rotation, 4 * sizeof(double));
        memcpy(anm_recs + (size_t)i * rl, w->rec_nmodes, (size_t)w->n_rec_nm * sizeof(double));
        memcpy(anm_ligs + (size_t)i * rl, w->lig_nmodes, (size_t)w->n_lig_nm * sizeof(double));
    }
    for (int i = 0; i < n; i++) { /* neighbour search, swarm.rs:85-102 */
        glowworm_t *g1 = &G->g[i];
        int cnt = 0;
        for (int j = 0; j < n; j++) {
            if (i == j) continue;
            const glowworm_t *g2 = &G->g[j];
            if (g1->luciferin < g2->luciferin) {
                double distance = glowworm_distance(g1, g2);
                if (distance < g1->vision_range) g1->neighbors[cnt++] = g2->id;
            }
        }
        g1->n_neighbors = cnt;
    }
    for (int i = 0; i < n; i++) luciferins[i] = G->g[i].luciferin; /* swarm.rs:105-108 */
    for (int i = 0; i < n; i++) { /* glowworm.rs:98-112 */
        glowworm_t *w = &G->g[i];
        double total_sum = 0.0;
        for (int k = 0; k < w->n_neighbors; k++) {
            double difference = luciferins[w->neighbors[k]] - w->luciferin;
            w->probabilities[k] = difference;
            total_sum += difference;
        }
        for (int k = 0; k < w->n_neighbors; k++) w->probabilities[k] /= total_sum;
    }
    for (int i = 0; i < n; i++) { /* swarm.rs:116-125 */
        glowworm_t *w = &G->g[i];
        double random_number = orc_rng_f64(G->rng);
        uint32_t neighbor_id;
        if (w->n_neighbors == 0) { /* glowworm.rs:114-126 */
            neighbor_id = w->id;
        } else {
            double sum_probabilities = 0.0;
            int k = 0;
            while (sum_probabilities < random_number) {
                if (k >= w->n_neighbors) break; /* the reference would panic (index out of bounds) */
                sum_probabilities += w->probabilities[k];
                k += 1;
            }
            if (k == 0) k = 1; /* random_number == 0.0: the reference would panic on neighbors[-1] */
            neighbor_id = w->neighbors[k - 1];
        }
        w->last_target = neighbor_id;
        move_towards(w, neighbor_id, positions + 3 * neighbor_id, rotations + 4 * neighbor_id,
                     anm_recs + (size_t)neighbor_id * rl, anm_ligs + (size_t)neighbor_id * rl);
        /* update_vision_range, glowworm.rs:91-96 */
        double v = w->vision_range + w->beta * (double)((int32_t)w->max_neighbors - (int32_t)w->n_neighbors);
        w->vision_range = fmin(w->max_vision_range, fmax(0.0, v));
    }
    free(positions); free(rotations); free(anm_recs); free(anm_ligs); free(luciferins);
}

/* Rust `{:.N}` and C "%.Nf" both print the exactly rounded (ties-to-even) decimal; only
 * the spelling of non-finite values differs. */
static void fmt_fixed(char *buf, size_t cap, double v, int prec) {
    if (isnan(v)) snprintf(buf, cap, "NaN");
    else if (isinf(v)) snprintf(buf, cap, v < 0 ? "-inf" : "inf");
    else snprintf(buf, cap, "%.*f", prec, v);
}

int orc_gso_save(const orc_gso *G, int step, const char *dir) { /* swarm.rs:128-167 */
    char path[4096];
    snprintf(path, sizeof path, "%s/gso_%d.out", dir, step);
    FILE *f = fopen(path, "w");
    if (!f) { set_err("Error saving GSO output: %s: %s", path, strerror(errno)); return -1; }
    fprintf(f, "#Coordinates  RecID  LigID  Luciferin  Neighbor's number  Vision Range  Scoring\n");
    char b[64];
    for (int i = 0; i < G->n; i++) {
        const glowworm_t *w = &G->g[i];
        double head[7] = {w->translation[0], w->translation[1], w->translation[2], w->rotation[0],
                          w->rotation[1],    w->rotation[2],    w->rotation[3]};
        fputc('(', f);
        for (int k = 0; k < 7; k++) { fmt_fixed(b, sizeof b, head[k], 7); fprintf(f, k ? ", %s" : "%s", b); }
        if (w->use_anm && w->n_rec_nm > 0)
            for (int k = 0; k < w->n_rec_nm; k++) { fmt_fixed(b, sizeof b, w->rec_nmodes[k], 7); fprintf(f, ", %s", b); }
        if (w->use_anm && w->n_lig_nm > 0)
            for (int k = 0; k < w->n_lig_nm; k++) { fmt_fixed(b, sizeof b, w->lig_nmodes[k], 7); fprintf(f, ", %s", b); }
        char l[64], v[64], s[64];
        fmt_fixed(l, sizeof l, w->luciferin, 8);
        fmt_fixed(v, sizeof v, w->vision_range, 3);
        fmt_fixed(s, sizeof s, w->scoring, 8);
        fprintf(f, ")    0    0   %s  %d %s %s\n", l, w->n_neighbors, v, s);
    }
    fclose(f);
    return 0;
}

int orc_gso_run(orc_gso *G, int steps, const char *dir) { /* lib.rs:46-58 */
    for (int step = 1; step < steps + 1; step++) {
        orc_gso_step(G);
        if (step % 10 == 0 || step == 1)
            if (orc_gso_save(G, step, dir) != 0) return -1;
    }
    return 0;
}
int orc_gso_num_glowworms(const orc_gso *G) { return G->n; }
int orc_gso_row_len(const orc_gso *G) { return G->row_len; }
uint64_t orc_gso_num_evals(const orc_gso *G) { return G->n_evals; }

void orc_gso_state(const orc_gso *G, double *poses, double *luciferin, double *vision, double *scoring,
                   int32_t *n_neighbors, int32_t *moved, int32_t *target) {
    for (int i = 0; i < G->n; i++) {
        const glowworm_t *w = &G->g[i];
        if (poses) {
            double *p = poses + (size_t)i * G->row_len;
            memset(p, 0, (size_t)G->row_len * sizeof(double));
            memcpy(p, w->translation, 3 * sizeof(double));
            memcpy(p + 3, w->rotation, 4 * sizeof(double));
            memcpy(p + 7, w->rec_nmodes, (size_t)w->n_rec_nm * sizeof(double));
            memcpy(p + 7 + w->n_rec_nm, w->lig_nmodes, (size_t)w->n_lig_nm * sizeof(double));
        }
        if (luciferin) luciferin[i] = w->luciferin;
        if (vision) vision[i] = w->vision_range;
        if (scoring) scoring[i] = w->scoring;
        if (n_neighbors) n_neighbors[i] = w->n_neighbors;
        if (moved) moved[i] = w->moved;
        if (target) target[i] = (int32_t)w->last_target;
    }
}
int orc_gso_neighbors(const orc_gso *G, int i, int32_t *out, int cap) {
    const glowworm_t *w = &G->g[i];
    for (int k = 0; k < w->n_neighbors && k < cap; k++) out[k] = (int32_t)w->neighbors[k];
    return w->n_neighbors;
}

/* ------------------------------------------------------------------------------------
 * file helpers
 * ---------------------------------------------------------------------------------- */
double *orc_parse_positions(const char *path, int *n_rows, int *row_len) { /* bin:60-75 */
    FILE *f = fopen(path, "r");
    if (!f) { set_err("Error reading the input file %s", path); return NULL; }
    size_t cap = 4096, n = 0;
    double *vals = (double *)malloc(cap * sizeof(double));
    int rows = 0, cols = -1;
    char *line = NULL;
    size_t lcap = 0;
    ssize_t got;
    while ((got = getline(&line, &lcap, f)) >= 0) {
        /* str::lines(): strip the trailing \n / \r\n; an empty final piece is not a line */
        while (got > 0 && (line[got - 1] == '\n' || line[got - 1] == '\r')) line[--got] = 0;
        int c = 0;
        char *p = line;
        for (;;) { /* split(' '): every piece must parse (unwrap) */
            char *sp = strchr(p, ' ');
            if (sp) *sp = 0;
            char *end;
            double v = strtod(p, &end);
            while (*end == '\t') end++;
            if (end == p || *end != 0) {
                set_err("invalid float literal in %s line %d", path, rows + 1);
                free(vals); free(line); fclose(f); return NULL;
            }
            if (n == cap) { cap *= 2; vals = (double *)realloc(vals, cap * sizeof(double)); }
            vals[n++] = v;
            c++;
            if (!sp) break;
            p = sp + 1;
        }
        if (cols < 0) cols = c;
        else if (c != cols) {
            set_err("ragged pose rows in %s (line %d has %d columns, expected %d)", path, rows + 1, c, cols);
            free(vals); free(line); fclose(f); return NULL;
        }
        rows++;
    }
    free(line);
    fclose(f);
    *n_rows = rows;
    *row_len = cols < 0 ? 0 : cols;
    return vals;
}

double *orc_read_npy_f64(const char *path, size_t *len) { /* bin:221-252 (npyz 0.8.3) */
    FILE *f = fopen(path, "rb");
    if (!f) { set_err("Error reading ANM file [\"%s\"]: %s", path, strerror(errno)); return NULL; }
    unsigned char hdr[12];
    if (fread(hdr, 1, 10, f) != 10 || memcmp(hdr, "\x93NUMPY", 6) != 0) { set_err("%s: not an npy file", path); fclose(f); return NULL; }
    size_t hlen;
    if (hdr[6] == 1) hlen = hdr[8] | (hdr[9] << 8);
    else {
        if (fread(hdr + 10, 1, 2, f) != 2) { set_err("%s: truncated npy header", path); fclose(f); return NULL; }
        hlen = hdr[8] | (hdr[9] << 8) | (hdr[10] << 16) | ((size_t)hdr[11] << 24);
    }
    char *h = (char *)malloc(hlen + 1);
    if (fread(h, 1, hlen, f) != hlen) { set_err("%s: truncated npy header", path); free(h); fclose(f); return NULL; }
    h[hlen] = 0;
    if (!strstr(h, "'<f8'") || strstr(h, "'fortran_order': True")) { set_err("%s: need C-order <f8 data", path); free(h); fclose(f); return NULL; }
    char *sh = strstr(h, "'shape':");
    size_t count = 1;
    if (sh) {
        char *p = strchr(sh, '(');
        char *e = p ? strchr(p, ')') : NULL;
        if (!p || !e) { set_err("%s: bad npy shape", path); free(h); fclose(f); return NULL; }
        p++;
        while (p < e) {
            while (p < e && (*p == ' ' || *p == ',')) p++;
            if (p >= e) break;
            count *= (size_t)strtoull(p, &p, 10);
        }
    }
    free(h);
    double *data = (double *)malloc((count ? count : 1) * sizeof(double));
    if (fread(data, sizeof(double), count, f) != count) { set_err("%s: truncated npy data", path); free(data); fclose(f); return NULL; }
    fclose(f);
    *len = count;
    return data;
}

/* --- bench baseline: many poses on several host threads --------------------------------------
 * What `ant_thony.py --cores N` does for the reference (example/1czy/execution.sh:24): N
 * independent single-threaded evaluations side by side.  Thread k takes rows k, k+threads, ...
 * Row layout = the .dat pose row [t(3) q(4) rec_nm lig_nm]; each evaluation is orc_scorer_energy. */
#include <pthread.h>
typedef struct {
    const orc_scorer *s;
    const double *rows;
    size_t n, stride;
    int first, step, anm_rec, anm_lig;
    double *out;
} rows_job_t;
static void *rows_worker(void *arg) {
    rows_job_t *j = (rows_job_t *)arg;
    for (size_t i = (size_t)j->first; i < j->n; i += (size_t)j->step) {
        const double *r = j->rows + i * j->stride;
        j->out[i] = orc_scorer_energy(j->s, r, r + 3, j->anm_rec ? r + 7 : NULL, j->anm_lig ? r + 7 + j->anm_rec : NULL);
    }
    return NULL;
}
int orc_scorer_energy_rows_mt(const orc_scorer *s, const double *rows, size_t n, size_t stride, int threads, double *out) {
    if (!s || !rows || !out || threads < 1 || threads > 1024) return -1;
    const int anm_rec = s->use_anm ? s->receptor.num_anm : 0, anm_lig = s->use_anm ? s->ligand.num_anm : 0;
    if (stride < (size_t)(7 + anm_rec + anm_lig)) return -1;
    pthread_t *th = (pthread_t *)malloc((size_t)threads * sizeof *th);
    rows_job_t *jobs = (rows_job_t *)malloc((size_t)threads * sizeof *jobs);
    if (!th || !jobs) { free(th); free(jobs); return -1; }
    int started = 0;
    for (int k = 0; k < threads; k++) {
        jobs[k] = (rows_job_t){s, rows, n, stride, k, threads, anm_rec, anm_lig, out};
        if (pthread_create(&th[k], NULL, rows_worker, &jobs[k]) != 0) break;
        started++;
    }
    for (int k = started; k < threads; k++) rows_worker(&jobs[k]); /* could not start: do its share here */
    for (int k = 0; k < started; k++) pthread_join(th[k], NULL);
    free(th);
    free(jobs);
    return 0;
}
