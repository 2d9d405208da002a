/*
 * orc_cli.c -- the reference command line on top of the CPU oracle (test infrastructure).
 *
 * Follows /root/reference/src/bin/lightdock-rust.rs:27-333:
 *     <prog> <setup.json> <initial_positions_N.dat> <steps> <dfire|dna>
 * Paths: PDBs relative to dirname(setup.json) with the "lightdock_" prefix, swarm_N/,
 * rec_nm.npy, lig_nm.npy and $LIGHTDOCK_DATA|data/DCparams relative to the CWD.
 * Usage errors print to stderr and return 0 like the reference (bin:101,112,121,142);
 * what is a panic there (exit 101) returns 101 here.
 */
#define _GNU_SOURCE
#include "ld_oracle.h"

#include <ctype.h>
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

/* ---- a small JSON reader: enough for setup.json (bin:27-58, serde_json) --------- */
typedef enum { J_NULL, J_BOOL, J_NUM, J_STR, J_ARR, J_OBJ } jtype;
typedef struct jval {
    jtype t;
    int b;
    double num;
    char *numtext;
    char *str;
    struct jval **items; /* J_ARR: values; J_OBJ: values */
    char **keys;         /* J_OBJ */
    size_t n;
} jval;

static void jskip(const char **p) { while (**p && isspace((unsigned char)**p)) (*p)++; }
static jval *jparse(const char **p);
static void jfree(jval *v) {
    if (!v) return;
    for (size_t i = 0; i < v->n; i++) { jfree(v->items[i]); if (v->keys) free(v->keys[i]); }
    free(v->items); free(v->keys); free(v->str); free(v->numtext); free(v);
}
static char *jstring(const char **p) {
    if (**p != '"') return NULL;
    (*p)++;
    size_t cap = 64, n = 0;
    char *s = (char *)malloc(cap);
    while (**p && **p != '"') {
        char c = *(*p)++;
        if (c == '\\') {
            char e = *(*p)++;
            switch (e) {
                case 'n': c = '\n'; break; case 't': c = '\t'; break; case 'r': c = '\r'; break;
                case 'b': c = '\b'; break; case 'f': c = '\f'; break;
                case 'u': { /* keep BMP code points below 0x80 only; setup.json is ASCII */
                    unsigned code = 0;
                    for (int i = 0; i < 4 && **p; i++) { code = code * 16 + (unsigned)(isdigit((unsigned char)**p) ? **p - '0' : (tolower((unsigned char)**p) - 'a' + 10)); (*p)++; }
                    c = code < 0x80 ? (char)code : '?';
                    break;
                }
                default: c = e;
            }
        }
        if (n + 2 > cap) { cap *= 2; s = (char *)realloc(s, cap); }
        s[n++] = c;
    }
    if (**p != '"') { free(s); return NULL; }
    (*p)++;
    s[n] = 0;
    return s;
}
static jval *jparse(const char **p) {
    jskip(p);
    jval *v = (jval *)calloc(1, sizeof *v);
    if (**p == '{' || **p == '[') {
        int obj = **p == '{';
        char close = obj ? '}' : ']';
        v->t = obj ? J_OBJ : J_ARR;
        (*p)++;
        jskip(p);
        if (**p == close) { (*p)++; return v; }
        for (;;) {
            jskip(p);
            char *key = NULL;
            if (obj) {
                key = jstring(p);
                if (!key) { jfree(v); return NULL; }
                jskip(p);
                if (**p != ':') { free(key); jfree(v); return NULL; }
                (*p)++;
            }
            jval *item = jparse(p);
            if (!item) { free(key); jfree(v); return NULL; }
            v->items = (jval **)realloc(v->items, (v->n + 1) * sizeof *v->items);
            if (obj) v->keys = (char **)realloc(v->keys, (v->n + 1) * sizeof *v->keys);
            v->items[v->n] = item;
            if (obj) v->keys[v->n] = key;
            v->n++;
            jskip(p);
            if (**p == ',') { (*p)++; continue; }
            if (**p == close) { (*p)++; return v; }
            jfree(v);
            return NULL;
        }
    }
    if (**p == '"') { v->t = J_STR; v->str = jstring(p); if (!v->str) { jfree(v); return NULL; } return v; }
    if (!strncmp(*p, "true", 4)) { v->t = J_BOOL; v->b = 1; *p += 4; return v; }
    if (!strncmp(*p, "false", 5)) { v->t = J_BOOL; v->b = 0; *p += 5; return v; }
    if (!strncmp(*p, "null", 4)) { v->t = J_NULL; *p += 4; return v; }
    char *end;
    v->num = strtod(*p, &end);
    if (end == *p) { jfree(v); return NULL; }
    v->t = J_NUM;
    v->numtext = strndup(*p, (size_t)(end - *p));
    *p = end;
    return v;
}
static jval *jget(const jval *o, const char *key) { /* serde: the last duplicate wins; none here */
    if (!o || o->t != J_OBJ) return NULL;
    for (size_t i = 0; i < o->n; i++)
        if (!strcmp(o->keys[i], key)) return o->items[i];
    return NULL;
}

typedef struct {
    int has_seed; uint64_t seed;
    int use_anm; size_t anm_rec, anm_lig;
    char receptor_pdb[1024], ligand_pdb[1024];
    int has_rec_restraints, has_lig_restraints;
    char **rec_active, **rec_passive, **lig_active, **lig_passive;
    int n_rec_active, n_rec_passive, n_lig_active, n_lig_passive;
} setup_t;

static int want_uint(const jval *o, const char *key, uint64_t *out, char *err, size_t cap) {
    jval *v = jget(o, key);
    if (!v) { snprintf(err, cap, "missing field `%s`", key); return -1; }
    if (v->t != J_NUM || strpbrk(v->numtext, ".eE-")) { snprintf(err, cap, "invalid type for `%s`: expected unsigned integer", key); return -1; }
    if (out) *out = strtoull(v->numtext, NULL, 10);
    return 0;
}
static int want_bool(const jval *o, const char *key, int *out, char *err, size_t cap) {
    jval *v = jget(o, key);
    if (!v) { snprintf(err, cap, "missing field `%s`", key); return -1; }
    if (v->t != J_BOOL) { snprintf(err, cap, "invalid type for `%s`: expected a boolean", key); return -1; }
    if (out) *out = v->b;
    return 0;
}
static int want_str(const jval *o, const char *key, char *out, size_t ocap, char *err, size_t cap) {
    jval *v = jget(o, key);
    if (!v) { snprintf(err, cap, "missing field `%s`", key); return -1; }
    if (v->t != J_STR) { snprintf(err, cap, "invalid type for `%s`: expected a string", key); return -1; }
    if (out) snprintf(out, ocap, "%s", v->str);
    return 0;
}
static int opt_str(const jval *o, const char *key, char *err, size_t cap) {
    jval *v = jget(o, key);
    if (v && v->t != J_NULL && v->t != J_STR) { snprintf(err, cap, "invalid type for `%s`: expected a string", key); return -1; }
    return 0;
}
/* Option<HashMap<String, Vec<String>>> */
static int opt_restraints(const jval *o, const char *key, int *present, char ***active, int *n_active,
                          char ***passive, int *n_passive, char *err, size_t cap, int *missing_key) {
    jval *v = jget(o, key);
    *present = 0;
    if (!v || v->t == J_NULL) return 0;
    if (v->t != J_OBJ) { snprintf(err, cap, "invalid type for `%s`: expected a map", key); return -1; }
    for (size_t i = 0; i < v->n; i++) {
        jval *l = v->items[i];
        if (l->t != J_ARR) { snprintf(err, cap, "invalid type in `%s`: expected a sequence", key); return -1; }
        for (size_t k = 0; k < l->n; k++)
            if (l->items[k]->t != J_STR) { snprintf(err, cap, "invalid type in `%s`: expected a string", key); return -1; }
    }
    *present = 1;
    const char *names[2] = {"active", "passive"};
    char ***outs[2] = {active, passive};
    int *ns[2] = {n_active, n_passive};
    for (int w = 0; w < 2; w++) {
        jval *l = jget(v, names[w]);
        if (!l) { *missing_key = 1; continue; } /* restraints["active"] panics, bin:257-272 */
        *outs[w] = (char **)calloc(l->n ? l->n : 1, sizeof(char *));
        for (size_t k = 0; k < l->n; k++) (*outs[w])[k] = strdup(l->items[k]->str);
        *ns[w] = (int)l->n;
    }
    return 0;
}

static void setup_free(setup_t *s) {
    char **lists[4] = {s->rec_active, s->rec_passive, s->lig_active, s->lig_passive};
    int ns[4] = {s->n_rec_active, s->n_rec_passive, s->n_lig_active, s->n_lig_passive};
    for (int w = 0; w < 4; w++) {
        for (int k = 0; lists[w] && k < ns[w]; k++) free(lists[w][k]);
        free(lists[w]);
    }
    memset(s, 0, sizeof *s);
}

static int read_setup(const char *path, setup_t *s, char *err, size_t cap, int *missing_key) {
    memset(s, 0, sizeof *s);
    FILE *f = fopen(path, "rb");
    if (!f) { snprintf(err, cap, "%s (os error %d)", strerror(errno), errno); return -1; }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *text = (char *)malloc((size_t)sz + 1);
    if (fread(text, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(text); snprintf(err, cap, "read error"); return -1; }
    text[sz] = 0;
    fclose(f);
    const char *p = text;
    jval *root = jparse(&p);
    if (root) { jskip(&p); if (*p) { jfree(root); root = NULL; } }
    if (!root || root->t != J_OBJ) { jfree(root); free(text); snprintf(err, cap, "JSON syntax error"); return -1; }
    int rc = 0;
    uint64_t u = 0;
    jval *seed = jget(root, "seed");
    if (seed && seed->t != J_NULL) {
        if (want_uint(root, "seed", &s->seed, err, cap)) rc = -1; else s->has_seed = 1;
    }
    /* required fields of SetupFile, bin:27-48 */
    if (!rc && want_uint(root, "anm_seed", NULL, err, cap)) rc = -1;
    if (!rc && opt_str(root, "ftdock_file", err, cap)) rc = -1;
    if (!rc && want_bool(root, "noh", NULL, err, cap)) rc = -1;
    if (!rc && want_uint(root, "anm_rec", &u, err, cap)) rc = -1; else s->anm_rec = (size_t)u;
    if (!rc && want_uint(root, "anm_lig", &u, err, cap)) rc = -1; else s->anm_lig = (size_t)u;
    if (!rc && want_uint(root, "swarms", NULL, err, cap)) rc = -1;
    if (!rc && want_uint(root, "starting_points_seed", NULL, err, cap)) rc = -1;
    if (!rc && want_bool(root, "verbose_parser", NULL, err, cap)) rc = -1;
    if (!rc && want_bool(root, "noxt", NULL, err, cap)) rc = -1;
    if (!rc && want_bool(root, "now", NULL, err, cap)) rc = -1;
    if (!rc && opt_str(root, "restraints", err, cap)) rc = -1;
    if (!rc && want_bool(root, "use_anm", &s->use_anm, err, cap)) rc = -1;
    if (!rc && want_uint(root, "glowworms", NULL, err, cap)) rc = -1;
    if (!rc && want_bool(root, "membrane", NULL, err, cap)) rc = -1;
    if (!rc && want_str(root, "receptor_pdb", s->receptor_pdb, sizeof s->receptor_pdb, err, cap)) rc = -1;
    if (!rc && want_str(root, "ligand_pdb", s->ligand_pdb, sizeof s->ligand_pdb, err, cap)) rc = -1;
    if (!rc && opt_restraints(root, "receptor_restraints", &s->has_rec_restraints, &s->rec_active, &s->n_rec_active,
                              &s->rec_passive, &s->n_rec_passive, err, cap, missing_key)) rc = -1;
    if (!rc && opt_restraints(root, "ligand_restraints", &s->has_lig_restraints, &s->lig_active, &s->n_lig_active,
                              &s->lig_passive, &s->n_lig_passive, err, cap, missing_key)) rc = -1;
    jfree(root);
    free(text);
    return rc;
}

static int parse_swarm_id(const char *path, int *id) { /* bin:150-156 */
    const char *base = strrchr(path, '/');
    base = base ? base + 1 : path;
    const char *pre = "initial_positions_";
    size_t pl = strlen(pre), bl = strlen(base);
    if (bl < pl + 4 || strncmp(base, pre, pl) != 0 || strcmp(base + bl - 4, ".dat") != 0) return -1;
    char num[64];
    size_t nl = bl - pl - 4;
    if (nl == 0 || nl >= sizeof num) return -1;
    memcpy(num, base + pl, nl);
    num[nl] = 0;
    char *end;
    errno = 0;
    long v = strtol(num, &end, 10); /* i32::from_str accepts an optional sign and digits */
    if (*end || errno || v > 2147483647L || v < -2147483648L) return -1;
    if (!isdigit((unsigned char)num[0]) && !((num[0] == '-' || num[0] == '+') && isdigit((unsigned char)num[1]))) return -1;
    *id = (int)v;
    return 0;
}

static void rust_debug_str(const char *s, char *out, size_t cap) { /* {:?} of a &str, ASCII subset */
    size_t n = 0;
    if (n + 1 < cap) out[n++] = '"';
    for (; *s && n + 3 < cap; s++) {
        if (*s == '"' || *s == '\\') out[n++] = '\\';
        out[n++] = *s;
    }
    out[n++] = '"';
    out[n] = 0;
}

int orc_cli_main(int argc, char **argv) {
    if (argc != 5) { /* bin:141-146 */
        fprintf(stderr, "Wrong command line. Usage: %s setup_filename swarm_filename steps method\n", argc > 0 ? argv[0] : "lightdock-rust");
        return 0;
    }
    const char *setup_filename = argv[1], *swarm_filename = argv[2];
    char *end;
    errno = 0;
    const char *st = argv[3];
    if (*st == '+') st++; /* u32::from_str accepts a leading '+' */
    unsigned long long steps_ull = strtoull(st, &end, 10);
    if (!isdigit((unsigned char)*st) || *end || errno || steps_ull > 4294967295ULL) {
        fprintf(stderr, "Error: steps argument must be a number\n"); /* bin:98-103 */
        return 0;
    }
    int steps = (int)steps_ull;
    char method_type[32];
    snprintf(method_type, sizeof method_type, "%s", argv[4]);
    for (char *c = method_type; *c; c++) *c = (char)tolower((unsigned char)*c);
    int method;
    const char *method_dbg;
    if (!strcmp(method_type, "dfire")) { method = ORC_METHOD_DFIRE; method_dbg = "DFIRE"; }
    else if (!strcmp(method_type, "dna")) { method = ORC_METHOD_DNA; method_dbg = "DNA"; }
    else if (!strcmp(method_type, "pydock")) { method = ORC_METHOD_PYDOCK; method_dbg = "PYDOCK"; }
    else { fprintf(stderr, "Error: method not supported\n"); return 0; } /* bin:105-115 */

    setup_t setup;
    char err[512], dbg[1200];
    int missing_key = 0;
    if (read_setup(setup_filename, &setup, err, sizeof err, &missing_key) != 0) { /* bin:118-129 */
        rust_debug_str(setup_filename, dbg, sizeof dbg);
        fprintf(stderr, "Error reading setup file [%s]: \"%s\"\n", dbg, err);
        setup_free(&setup);
        return 0;
    }
    int code = 0;
    double *potential = NULL, *positions = NULL, *rec_nm = NULL, *lig_nm = NULL;
    orc_scorer *scorer = NULL;
    orc_gso *gso = NULL;
    /* simulation path = parent of the setup file, bin:131 */
    char simulation_path[2048];
    snprintf(simulation_path, sizeof simulation_path, "%s", setup_filename);
    char *slash = strrchr(simulation_path, '/');
    if (slash) { if (slash == simulation_path) slash[1] = 0; else *slash = 0; } else simulation_path[0] = 0;

    uint64_t seed = setup.has_seed ? setup.seed : 324324ULL; /* bin:165-168 */
    rust_debug_str(swarm_filename, dbg, sizeof dbg);
    printf("Reading starting positions from %s\n", dbg);
    int swarm_id;
    if (parse_swarm_id(swarm_filename, &swarm_id) != 0) {
        fprintf(stderr, "Could not parse swarm from swarm filename\n");
        { code = 101; goto done; }
    }
    printf("Swarm ID %d\n", swarm_id);
    char swarm_directory[64];
    snprintf(swarm_directory, sizeof swarm_directory, "swarm_%d", swarm_id);
    struct stat sb;
    if (stat(swarm_directory, &sb) != 0 || !S_ISDIR(sb.st_mode)) { /* bin:176-185 */
        fprintf(stderr, "Output directory does not exist for swarm %d, creating it\n", swarm_id);
        if (mkdir(swarm_directory, 0777) != 0) { fprintf(stderr, "Error creating directory\n"); { code = 101; goto done; } }
    }
    rust_debug_str(swarm_directory, dbg, sizeof dbg);
    printf("Writing to swarm dir %s\n", dbg);
    int n_rows = 0, row_len = 0;
    positions = orc_parse_positions(swarm_filename, &n_rows, &row_len);
    if (!positions) { fprintf(stderr, "%s\n", orc_last_error()); { code = 101; goto done; } }

    char receptor_filename[4096], ligand_filename[4096];
    if (simulation_path[0] == 0) {
        snprintf(receptor_filename, sizeof receptor_filename, "lightdock_%s", setup.receptor_pdb);
        snprintf(ligand_filename, sizeof ligand_filename, "lightdock_%s", setup.ligand_pdb);
    } else {
        snprintf(receptor_filename, sizeof receptor_filename, "%s/lightdock_%s", simulation_path, setup.receptor_pdb);
        snprintf(ligand_filename, sizeof ligand_filename, "%s/lightdock_%s", simulation_path, setup.ligand_pdb);
    }
    /* pdbtbx::open(..).unwrap() happens here, before the ANM files are read (bin:190-214) */
    printf("Reading receptor input structure: %s\n", receptor_filename);
    { FILE *t = fopen(receptor_filename, "r"); if (!t) { fflush(stdout); fprintf(stderr, "cannot open PDB file %s\n", receptor_filename); { code = 101; goto done; } } fclose(t); }
    printf("Reading ligand input structure: %s\n", ligand_filename);
    { FILE *t = fopen(ligand_filename, "r"); if (!t) { fflush(stdout); fprintf(stderr, "cannot open PDB file %s\n", ligand_filename); { code = 101; goto done; } } fclose(t); }

    (void)0;
    size_t rec_nm_len = 0, lig_nm_len = 0;
    if (setup.use_anm) { /* bin:216-254 */
        if (setup.anm_rec > 0 && !(rec_nm = orc_read_npy_f64("rec_nm.npy", &rec_nm_len))) { fprintf(stderr, "%s\n", orc_last_error()); { code = 101; goto done; } }
        if (setup.anm_lig > 0 && !(lig_nm = orc_read_npy_f64("lig_nm.npy", &lig_nm_len))) { fprintf(stderr, "%s\n", orc_last_error()); { code = 101; goto done; } }
    }
    if (missing_key) { fprintf(stderr, "restraints map lacks \"active\"/\"passive\"\n"); { code = 101; goto done; } }

    printf("Loading %s scoring function\n", method_dbg);
    (void)0;
    if (method == ORC_METHOD_DFIRE) { /* dfire.rs:236-257 */
        const char *data = getenv("LIGHTDOCK_DATA");
        char ppath[4096];
        snprintf(ppath, sizeof ppath, "%s/DCparams", data ? data : "data");
        potential = (double *)malloc(ORC_DFIRE_TABLE_LEN * sizeof(double));
        if (orc_load_dcparams(ppath, potential) != 0) { fprintf(stderr, "%s\n", orc_last_error()); { code = 101; goto done; } }
    }
    scorer = orc_scorer_new(method, receptor_filename, ligand_filename,
                                        (const char *const *)setup.rec_active, setup.n_rec_active,
                                        (const char *const *)setup.rec_passive, setup.n_rec_passive, rec_nm, rec_nm_len,
                                        (int)setup.anm_rec, (const char *const *)setup.lig_active, setup.n_lig_active,
                                        (const char *const *)setup.lig_passive, setup.n_lig_passive, lig_nm, lig_nm_len,
                                        (int)setup.anm_lig, setup.use_anm, potential);
    if (!scorer) { fprintf(stderr, "%s\n", orc_last_error()); { code = 101; goto done; } }

    printf("Creating GSO with %d glowworms\n", n_rows);
    gso = orc_gso_new(positions, n_rows, row_len, seed, scorer, setup.use_anm, (int)setup.anm_rec, (int)setup.anm_lig);
    if (!gso) { fprintf(stderr, "%s\n", orc_last_error()); { code = 101; goto done; } }
    printf("Starting optimization (%d steps)\n", steps);
    fflush(stdout);
    int rc = orc_gso_run(gso, steps, swarm_directory);
    if (rc != 0) { fprintf(stderr, "%s\n", orc_last_error()); { code = 101; goto done; } }
done:
    orc_gso_free(gso);
    orc_scorer_free(scorer);
    free(potential); free(positions); free(rec_nm); free(lig_nm);
    setup_free(&setup);
    return code;
}
