/*
 * grample_oracle.c -- CPU restatement of grample's Gibbs hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: a plain-C,
 * double-precision restatement of the reference's CPU algorithm.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * it.  Nothing under grample_amd/ (the product) links, imports or calls it.
 *
 * Pinning: the reference is Go and cannot be built here (no go toolchain, RNG
 * dependency github.com/seehuhn/mt19937 v1.0.0 not vendored).  The oracle is
 * therefore pinned by the reference's OWN golden vectors and known-answer
 * tests, restated in tests/test_oracle_*.py:
 *   rand/rand_test.go:17-39         MT19937-64 init_by_array64 vector
 *   model/error_test.go:11-119      ErrorSuite goldens
 *   model/uai_test.go:141-170       one.uai.MAR -> MAE .25, Hellinger .18459
 *   model/function_test.go:81-157   index order / log space
 *   model/variable_iter_test.go     odometer order
 *   sampler/sampler_test.go:69-220  selector edge cases, 2:1 weight ratio
 *   sampler/chain_test.go:11-80     MergeChains sums + collapsed rule
 *   sampler/gibbs-collapsed_test.go:14-111  collapse KATs
 *   sampler/gibbs-simple_test.go:13-38      one.uai both values occur
 * plus res/<model>.uai.MAR exact marginals as statistical pins.
 * "Parity unpinned" item: the single-seed path mt19937.Seed(int64) has no
 * vector in the reference; canonical init_genrand64 is assumed.
 *
 * All file:line citations are relative to /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ errors */
static __thread char orc_err[512];
static int orc_fail(const char *msg) {
    snprintf(orc_err, sizeof orc_err, "%s", msg);
    return -1;
}
ORC_API const char *orc_last_error(void) { return orc_err; }

/* --------------------------------------------------------------- MT19937-64
 * Third-party dependency github.com/seehuhn/mt19937 v1.0.0 (go.mod:7) is the
 * published MT19937-64 of Matsumoto & Nishimura (2004): init_genrand64,
 * init_by_array64, genrand64_int64.  Call sites: rand/rand.go:26-35. */
#define MT_NN 312
#define MT_MM 156
#define MT_MATRIX_A 0xB5026F5AA96619E9ULL
#define MT_UM 0xFFFFFFFF80000000ULL
#define MT_LM 0x7FFFFFFFULL

typedef struct {
    uint64_t mt[MT_NN];
    int mti;
    uint64_t draws; /* number of Int63 values handed out (diagnostic) */
} orc_gen;

static void mt_seed(orc_gen *g, uint64_t seed) { /* init_genrand64 */
    g->mt[0] = seed;
    for (g->mti = 1; g->mti < MT_NN; g->mti++)
        g->mt[g->mti] = 6364136223846793005ULL * (g->mt[g->mti - 1] ^ (g->mt[g->mti - 1] >> 62)) + (uint64_t)g->mti;
}

static void mt_seed_slice(orc_gen *g, const uint64_t *key, int klen) { /* init_by_array64 */
    mt_seed(g, 19650218ULL);
    uint64_t i = 1, j = 0;
    uint64_t k = (MT_NN > klen) ? MT_NN : (uint64_t)klen;
    for (; k; k--) {
        g->mt[i] = (g->mt[i] ^ ((g->mt[i - 1] ^ (g->mt[i - 1] >> 62)) * 3935559000370003845ULL)) + key[j] + j;
        i++; j++;
        if (i >= MT_NN) { g->mt[0] = g->mt[MT_NN - 1]; i = 1; }
        if (j >= (uint64_t)klen) j = 0;
    }
    for (k = MT_NN - 1; k; k--) {
        g->mt[i] = (g->mt[i] ^ ((g->mt[i - 1] ^ (g->mt[i - 1] >> 62)) * 2862933555777941757ULL)) - i;
        i++;
        if (i >= MT_NN) { g->mt[0] = g->mt[MT_NN - 1]; i = 1; }
    }
    g->mt[0] = 1ULL << 63;
}

static uint64_t mt_u64(orc_gen *g) { /* genrand64_int64 */
    static const uint64_t mag01[2] = {0ULL, MT_MATRIX_A};
    uint64_t x;
    if (g->mti >= MT_NN) {
        int i;
        for (i = 0; i < MT_NN - MT_MM; i++) {
            x = (g->mt[i] & MT_UM) | (g->mt[i + 1] & MT_LM);
            g->mt[i] = g->mt[i + MT_MM] ^ (x >> 1) ^ mag01[(int)(x & 1ULL)];
        }
        for (; i < MT_NN - 1; i++) {
            x = (g->mt[i] & MT_UM) | (g->mt[i + 1] & MT_LM);
            g->mt[i] = g->mt[i + (MT_MM - MT_NN)] ^ (x >> 1) ^ mag01[(int)(x & 1ULL)];
        }
        x = (g->mt[MT_NN - 1] & MT_UM) | (g->mt[0] & MT_LM);
        g->mt[MT_NN - 1] = g->mt[MT_MM - 1] ^ (x >> 1) ^ mag01[(int)(x & 1ULL)];
        g->mti = 0;
    }
    x = g->mt[g->mti++];
    x ^= (x >> 29) & 0x5555555555555555ULL;
    x ^= (x << 17) & 0x71D67FFFEDA60000ULL;
    x ^= (x << 37) & 0xFFF7EEE000000000ULL;
    x ^= (x >> 43);
    return x;
}

/* rand/rand.go:19-49: NewGeneratorSlice / NewGenerator.  One entry => Seed(),
 * else SeedFromSlice().  (The goroutine + channel only buffers the stream.) */
ORC_API orc_gen *orc_gen_new_slice(const uint64_t *seed, int n) {
    if (n < 1) { orc_fail("Invalid generator seed array"); return NULL; }
    orc_gen *g = (orc_gen *)calloc(1, sizeof *g);
    if (n == 1) mt_seed(g, seed[0]); else mt_seed_slice(g, seed, n);
    return g;
}
ORC_API orc_gen *orc_gen_new(int64_t seed) { uint64_t s = (uint64_t)seed; return orc_gen_new_slice(&s, 1); }
ORC_API void orc_gen_free(orc_gen *g) { free(g); }
ORC_API uint64_t orc_gen_draws(const orc_gen *g) { return g->draws; }

/* rand/rand.go:52-54 Int63; the 63-bit value is the raw output with the top
 * bit cleared -- pinned by rand/rand_test.go:32-38 (v & 0x7fff...). */
ORC_API int64_t orc_int63(orc_gen *g) { g->draws++; return (int64_t)(mt_u64(g) & 0x7FFFFFFFFFFFFFFFULL); }

/* rand/rand.go:57-73 */
ORC_API int64_t orc_int63n(orc_gen *g, int64_t n) {
    if ((n & (n - 1)) == 0) return orc_int63(g) & (n - 1);
    int64_t max = (int64_t)((1ULL << 63) - 1 - (1ULL << 63) % (uint64_t)n);
    int64_t v = orc_int63(g);
    while (v > max) v = orc_int63(g);
    return v % n;
}
/* rand/rand.go:76-78 */
ORC_API int32_t orc_int31(orc_gen *g) { return (int32_t)(orc_int63(g) >> 32); }
/* rand/rand.go:81-98 */
ORC_API int32_t orc_int31n(orc_gen *g, int32_t n) {
    if ((n & (n - 1)) == 0) return orc_int31(g) & (n - 1);
    int32_t max = (int32_t)((1U << 31) - 1 - (1U << 31) % (uint32_t)n);
    int32_t v = orc_int31(g);
    while (v > max) v = orc_int31(g);
    return v % n;
}
/* rand/rand.go:102-105 */
ORC_API double orc_float64(orc_gen *g) { return (double)orc_int63n(g, 1LL << 53) / (double)(1LL << 53); }

/* ------------------------------------------------------------------- model */
typedef struct {
    int id, card, fixed_val, collapsed;
    double *marginal;  /* len card; NewVariable => uniform (variable.go:22-51) */
    double selections; /* State["Selections"], gibbs-simple.go:165 */
} orc_var;

typedef struct {
    int uid;      /* stands in for Function.Name (unique; used by Collapse's delete-by-name) */
    int nvars;
    int *vars;    /* variable ids, most significant first (function.go:13-36) */
    int tabsize;
    double *table;
    int is_log;
} orc_func;

typedef struct {
    int nvars;
    orc_var *vars;
    int nfuncs;
    orc_func **funcs;
    int next_uid;
} orc_model;

/* model/variable.go:106-147 NormMarginal */
static int var_norm_marginal(orc_var *v) {
    if (v->card < 1) return 0;
    if (v->card == 1) v->marginal[0] = 1.0;
    double sum = 0;
    for (int i = 0; i < v->card; i++) sum += v->marginal[i];
    const double EPS = 1e-8;
    if (fabs(sum - 1.0) < EPS) return 0;
    if (fabs(sum) < EPS) {
        double p = 1.0 / (double)v->card;
        for (int i = 0; i < v->card; i++) v->marginal[i] = p;
        return 0;
    }
    for (int i = 0; i < v->card; i++) v->marginal[i] /= sum;
    return 0;
}
ORC_API void orc_norm_marginal(double *m, int card) { orc_var v = {0, card, -1, 0, m, 0}; var_norm_marginal(&v); }

/* model/function.go:47-57 calcTabSize */
static int calc_tab_size(const orc_model *m, const int *vars, int n) {
    int ts = 0;
    for (int i = 0; i < n; i++) { if (i == 0) ts = 1; ts *= m->vars[vars[i]].card; }
    return ts;
}
#define ORC_MAX_TAB (1 << 23) /* function.go:59 */

/* model/function.go:62-93 NewFunction (zero table, linear space) */
static orc_func *func_new(orc_model *m, const int *vars, int n) {
    if (n < 1) { orc_fail("Empty variable list for function is invalid"); return NULL; }
    int ts = calc_tab_size(m, vars, n);
    if (ts < 1) { orc_fail("Function is invalid - could not calculate table size"); return NULL; }
    if (ts > ORC_MAX_TAB) { orc_fail("Function table size exceeds maxTabSize"); return NULL; }
    orc_func *f = (orc_func *)calloc(1, sizeof *f);
    f->uid = m->next_uid++;
    f->nvars = n;
    f->vars = (int *)malloc(sizeof(int) * n);
    memcpy(f->vars, vars, sizeof(int) * n);
    f->tabsize = ts;
    f->table = (double *)calloc(ts, sizeof(double));
    return f;
}
static void func_free(orc_func *f) { if (f) { free(f->vars); free(f->table); free(f); } }

/* model/function.go:180-202 calcIndex: last scope var fastest; -1 on error */
static int func_calc_index(const orc_model *m, const orc_func *f, const int *values, int nvalues) {
    if (nvalues != f->nvars) { orc_fail("Value vector does not match variables"); return -1; }
    int digit = 1, location = 0;
    for (int i = nvalues - 1; i >= 0; i--) {
        int val = values[i], card = m->vars[f->vars[i]].card;
        if (val < 0 || val >= card) { orc_fail("Value invalid for cardinality"); return -1; }
        location += digit * val;
        digit *= card;
    }
    return location;
}
/* model/function.go:146-157 Eval: NaN + error on bad input */
static int func_eval(const orc_model *m, const orc_func *f, const int *values, int n, double *out) {
    int i = func_calc_index(m, f, values, n);
    if (i < 0 || i >= f->tabsize) { *out = NAN; if (i >= 0) orc_fail("Could not find table entry"); return -1; }
    *out = f->table[i];
    return 0;
}
/* model/function.go:126-142 UseLogSpace */
static int func_use_log_space(orc_func *f) {
    if (f->is_log) return orc_fail("IsLog already set - double-call detected");
    const double eps = 1e-6;
    for (int i = 0; i < f->tabsize; i++) {
        double v = f->table[i];
        if (v < eps) v += eps;
        f->table[i] = log(v);
    }
    f->is_log = 1;
    return 0;
}
/* model/function.go:161-177 AddValue */
static int func_add_value(const orc_model *m, orc_func *f, const int *values, int n, double inc) {
    if (f->is_log) return orc_fail("Can not AddValue if function is already in log space");
    int i = func_calc_index(m, f, values, n);
    if (i < 0) return -1;
    if (i >= f->tabsize) return orc_fail("Could not find table entry");
    f->table[i] += inc;
    return 0;
}

/* Flat-array constructor (what model/uai.go:53-179 ReadModel produces).
 * table[] holds linear-space values exactly as parsed.  fixed[] may be NULL. */
ORC_API orc_model *orc_model_create(int n_vars, const int *card, const int *fixed, int n_funcs,
                                    const int *scope_off, const int *scope_var, const int *tab_off,
                                    const double *table) {
    if (n_vars < 1) { orc_fail("Invalid variable count"); return NULL; }
    orc_model *m = (orc_model *)calloc(1, sizeof *m);
    m->nvars = n_vars;
    m->vars = (orc_var *)calloc(n_vars, sizeof(orc_var));
    for (int i = 0; i < n_vars; i++) {
        orc_var *v = &m->vars[i];
        v->id = i; v->card = card[i]; v->fixed_val = fixed ? fixed[i] : -1;
        v->marginal = (double *)calloc(card[i] > 0 ? card[i] : 1, sizeof(double));
        var_norm_marginal(v); /* zero sum => uniform */
    }
    m->nfuncs = n_funcs;
    m->funcs = (orc_func **)calloc(n_funcs > 0 ? n_funcs : 1, sizeof(orc_func *));
    for (int i = 0; i < n_funcs; i++) {
        int n = scope_off[i + 1] - scope_off[i];
        orc_func *f = func_new(m, scope_var + scope_off[i], n);
        if (!f) return NULL;
        if (f->tabsize != tab_off[i + 1] - tab_off[i]) { orc_fail("Read table size != previous Clique size"); return NULL; }
        memcpy(f->table, table + tab_off[i], sizeof(double) * f->tabsize);
        m->funcs[i] = f;
    }
    return m;
}
/* model/model.go:32-49 Clone (deep) */
ORC_API orc_model *orc_model_clone(const orc_model *s) {
    orc_model *m = (orc_model *)calloc(1, sizeof *m);
    m->nvars = s->nvars; m->nfuncs = s->nfuncs; m->next_uid = s->next_uid;
    m->vars = (orc_var *)calloc(s->nvars, sizeof(orc_var));
    for (int i = 0; i < s->nvars; i++) {
        m->vars[i] = s->vars[i];
        m->vars[i].marginal = (double *)malloc(sizeof(double) * (s->vars[i].card > 0 ? s->vars[i].card : 1));
        memcpy(m->vars[i].marginal, s->vars[i].marginal, sizeof(double) * s->vars[i].card);
    }
    m->funcs = (orc_func **)calloc(s->nfuncs > 0 ? s->nfuncs : 1, sizeof(orc_func *));
    for (int i = 0; i < s->nfuncs; i++) {
        const orc_func *sf = s->funcs[i];
        orc_func *f = (orc_func *)calloc(1, sizeof *f);
        *f = *sf;
        f->vars = (int *)malloc(sizeof(int) * sf->nvars); memcpy(f->vars, sf->vars, sizeof(int) * sf->nvars);
        f->table = (double *)malloc(sizeof(double) * sf->tabsize); memcpy(f->table, sf->table, sizeof(double) * sf->tabsize);
        m->funcs[i] = f;
    }
    return m;
}
ORC_API void orc_model_free(orc_model *m) {
    if (!m) return;
    for (int i = 0; i < m->nvars; i++) free(m->vars[i].marginal);
    for (int i = 0; i < m->nfuncs; i++) func_free(m->funcs[i]);
    free(m->vars); free(m->funcs); free(m);
}
/* model/model.go:115-157 Check (the parts that can fail for flat input) */
ORC_API int orc_model_check(const orc_model *m) {
    int fix = 0;
    for (int i = 0; i < m->nvars; i++) {
        const orc_var *v = &m->vars[i];
        if (v->fixed_val != -1 && (v->fixed_val < 0 || v->fixed_val >= v->card)) return orc_fail("Variable has bad fixed val");
        double sum = 0; for (int c = 0; c < v->card; c++) sum += v->marginal[c];
        if (v->card > 0 && fabs(sum - 1.0) >= 1e-8) return orc_fail("Variable has marginal dist with sum != 1");
        if (v->fixed_val > -1) fix++;
    }
    if (fix >= m->nvars) return orc_fail("Fixed variable count - all vars are fixed!");
    for (int i = 0; i < m->nfuncs; i++) {
        const orc_func *f = m->funcs[i];
        int ts = calc_tab_size(m, f->vars, f->nvars);
        if (ts < 1 || ts != f->tabsize) return orc_fail("Function has invalid table size");
    }
    return 0;
}

/* accessors for tests */
ORC_API int orc_model_nvars(const orc_model *m) { return m->nvars; }
ORC_API int orc_model_nfuncs(const orc_model *m) { return m->nfuncs; }
ORC_API int orc_var_card(const orc_model *m, int v) { return m->vars[v].card; }
ORC_API int orc_var_fixed(const orc_model *m, int v) { return m->vars[v].fixed_val; }
ORC_API void orc_var_set_fixed(orc_model *m, int v, int val) { m->vars[v].fixed_val = val; }
ORC_API int orc_var_collapsed(const orc_model *m, int v) { return m->vars[v].collapsed; }
ORC_API void orc_var_set_collapsed(orc_model *m, int v, int c) { m->vars[v].collapsed = c; }
ORC_API void orc_var_get_marginal(const orc_model *m, int v, double *out) { memcpy(out, m->vars[v].marginal, sizeof(double) * m->vars[v].card); }
ORC_API void orc_var_set_marginal(orc_model *m, int v, const double *in) { memcpy(m->vars[v].marginal, in, sizeof(double) * m->vars[v].card); }
ORC_API double orc_var_selections(const orc_model *m, int v) { return m->vars[v].selections; }
ORC_API int orc_func_nvars(const orc_model *m, int f) { return m->funcs[f]->nvars; }
ORC_API int orc_func_tabsize(const orc_model *m, int f) { return m->funcs[f]->tabsize; }
ORC_API int orc_func_is_log(const orc_model *m, int f) { return m->funcs[f]->is_log; }
ORC_API void orc_func_get_vars(const orc_model *m, int f, int *out) { memcpy(out, m->funcs[f]->vars, sizeof(int) * m->funcs[f]->nvars); }
ORC_API void orc_func_get_table(const orc_model *m, int f, double *out) { memcpy(out, m->funcs[f]->table, sizeof(double) * m->funcs[f]->tabsize); }
ORC_API int orc_func_eval(const orc_model *m, int f, const int *values, int n, double *out) { return func_eval(m, m->funcs[f], values, n, out); }
ORC_API int orc_func_use_log_space(orc_model *m, int f) { return func_use_log_space(m->funcs[f]); }
ORC_API int orc_func_add_value(orc_model *m, int f, const int *values, int n, double inc) { return func_add_value(m, m->funcs[f], values, n, inc); }
/* appends a zero function over vars (NewFunction) and returns its index, or -1 */
ORC_API int orc_model_new_function(orc_model *m, const int *vars, int n) {
    orc_func *f = func_new(m, vars, n);
    if (!f) return -1;
    m->funcs = (orc_func **)realloc(m->funcs, sizeof(orc_func *) * (m->nfuncs + 1));
    m->funcs[m->nfuncs] = f;
    return m->nfuncs++;
}

/* ------------------------------------------------------------ VariableIter
 * model/variable_iter.go:15-74: odometer, last var fastest, fixed pinned. */
typedef struct { const orc_model *m; int n; int *vars; int *last; int honor_fixed; } orc_viter;

ORC_API orc_viter *orc_viter_new(const orc_model *m, const int *vars, int n, int honor_fixed) {
    if (n < 1) { orc_fail("At least one variable required for iteration"); return NULL; }
    orc_viter *vi = (orc_viter *)calloc(1, sizeof *vi);
    vi->m = m; vi->n = n; vi->honor_fixed = honor_fixed;
    vi->vars = (int *)malloc(sizeof(int) * n); memcpy(vi->vars, vars, sizeof(int) * n);
    vi->last = (int *)calloc(n, sizeof(int));
    if (honor_fixed)
        for (int i = 0; i < n; i++) if (m->vars[vars[i]].fixed_val >= 0) vi->last[i] = m->vars[vars[i]].fixed_val;
    return vi;
}
ORC_API void orc_viter_val(const orc_viter *vi, int *curr) { memcpy(curr, vi->last, sizeof(int) * vi->n); }
ORC_API int orc_viter_next(orc_viter *vi) {
    for (int i = vi->n - 1; i >= 0; i--) {
        const orc_var *v = &vi->m->vars[vi->vars[i]];
        if (vi->honor_fixed && v->fixed_val >= 0) { vi->last[i] = v->fixed_val; continue; }
        int prop = vi->last[i] + 1;
        if (prop < v->card) { vi->last[i] = prop; return 1; }
        vi->last[i] = 0;
    }
    return 0;
}
ORC_API void orc_viter_free(orc_viter *vi) { if (vi) { free(vi->vars); free(vi->last); free(vi); } }

/* ---------------------------------------------------------- UniformSampler
 * sampler/sampler.go:67-174 */
#define ORC_MAX_CARD (1 << 30)

/* sampler.go:72-86 UniSample */
ORC_API int orc_uni_sample(orc_gen *g, int card, int *out) {
    if (card < 1) return orc_fail("Can not sample if Cardinality < 1");
    if (card > ORC_MAX_CARD) return orc_fail("Cardinality above maxCard not supported");
    if (card == 1) { *out = 0; return 0; }
    *out = (int)orc_int31n(g, (int32_t)card);
    return 0;
}
/* sampler.go:90-130 WeightedSample */
ORC_API int orc_weighted_sample(orc_gen *g, int card, const double *weights, int nweights, int *out) {
    if (card < 1) return orc_fail("Can not sample if Cardinality < 1");
    if (card > ORC_MAX_CARD) return orc_fail("Cardinality above maxCard not supported");
    if (nweights != card) return orc_fail("Weight array size must match cardinality");
    if (card == 1) { *out = 0; return 0; }
    double tot = 0.0;
    for (int i = 0; i < card; i++) {
        if (weights[i] <= 0.0) return orc_fail("Weights must be > 0.0");
        tot += weights[i];
    }
    double r = orc_float64(g) * tot;
    int sel = -1;
    for (int i = 0; i < card; i++) {
        if (r <= weights[i]) { sel = i; break; }
        r -= weights[i];
    }
    if (sel < 0) return orc_fail("Failed to sample");
    *out = sel;
    return 0;
}
/* sampler.go:135-174 VarSample over a model's variables */
static int var_sample(orc_gen *g, const orc_model *m, int exclude_collapsed, int *out) {
    if (m->nvars < 1) return orc_fail("Can not sample from an empty variable list");
    int count = 0, first = -1;
    /* two passes instead of the pooled index slice: same result, same draws */
    for (int i = 0; i < m->nvars; i++) {
        if (exclude_collapsed && m->vars[i].collapsed) continue;
        if (m->vars[i].fixed_val >= 0) continue;
        if (count == 0) first = i;
        count++;
    }
    if (count < 1) return orc_fail("No Variables to select");
    if (count == 1) { *out = first; return 0; } /* no draw (sampler.go:162-165) */
    int k;
    if (orc_uni_sample(g, count, &k)) return -1;
    for (int i = 0; i < m->nvars; i++) {
        if (exclude_collapsed && m->vars[i].collapsed) continue;
        if (m->vars[i].fixed_val >= 0) continue;
        if (k-- == 0) { *out = i; return 0; }
    }
    return orc_fail("unreachable");
}
ORC_API int orc_var_sample(orc_gen *g, const orc_model *m, int exclude_collapsed, int *out) { return var_sample(g, m, exclude_collapsed, out); }

/* ------------------------------------------------------------- GibbsSimple
 * sampler/gibbs-simple.go */
typedef struct {
    orc_gen *gen;
    orc_model *pgm;   /* not owned */
    int *vf_off;      /* varFuncs as CSR: for var v, funcs vf_fun[vf_off[v]..vf_off[v+1]) in pgm.Funcs order */
    int *vf_fun;
    int *last;
    double *weights;  /* valuePool buffer, len maxCard */
    int *callvals;    /* varPool buffer, len nvars */
    int max_card;
} orc_simple;

static void simple_build_varfuncs(orc_simple *s) {
    const orc_model *m = s->pgm;
    free(s->vf_off); free(s->vf_fun);
    s->vf_off = (int *)calloc(m->nvars + 1, sizeof(int));
    int tot = 0;
    for (int f = 0; f < m->nfuncs; f++) tot += m->funcs[f]->nvars;
    s->vf_fun = (int *)malloc(sizeof(int) * (tot > 0 ? tot : 1));
    for (int f = 0; f < m->nfuncs; f++) for (int j = 0; j < m->funcs[f]->nvars; j++) s->vf_off[m->funcs[f]->vars[j] + 1]++;
    for (int v = 0; v < m->nvars; v++) s->vf_off[v + 1] += s->vf_off[v];
    int *fill = (int *)calloc(m->nvars, sizeof(int));
    for (int f = 0; f < m->nfuncs; f++) /* append order == m.Funcs order (gibbs-simple.go:72-81) */
        for (int j = 0; j < m->funcs[f]->nvars; j++) { int v = m->funcs[f]->vars[j]; s->vf_fun[s->vf_off[v] + fill[v]++] = f; }
    free(fill);
}

/* gibbs-simple.go:25-115 NewGibbsSimple */
ORC_API orc_simple *orc_simple_new(orc_gen *gen, orc_model *m) {
    if (!m) { orc_fail("No model supplied"); return NULL; }
    if (m->nvars < 1) { orc_fail("Invalid max var count"); return NULL; }
    orc_simple *s = (orc_simple *)calloc(1, sizeof *s);
    s->gen = gen; s->pgm = m;
    for (int i = 0; i < m->nvars; i++) if (m->vars[i].card > s->max_card) s->max_card = m->vars[i].card;
    s->last = (int *)calloc(m->nvars, sizeof(int));
    s->weights = (double *)calloc(s->max_card > 0 ? s->max_card : 1, sizeof(double));
    s->callvals = (int *)calloc(m->nvars, sizeof(int));
    for (int f = 0; f < m->nfuncs; f++)
        if (func_use_log_space(m->funcs[f])) return NULL; /* 72-76 */
    simple_build_varfuncs(s);
    for (int i = 0; i < m->nvars; i++) { /* 84-112 */
        orc_var *v = &m->vars[i];
        v->selections = 0.0;
        if (s->vf_off[i + 1] - s->vf_off[i] < 1) { orc_fail("There are no functions for var"); return NULL; }
        if (v->fixed_val >= 0) s->last[i] = v->fixed_val;
        else if (orc_uni_sample(gen, v->card, &s->last[i])) return NULL;
    }
    return s;
}
ORC_API void orc_simple_free(orc_simple *s) {
    if (!s) return;
    free(s->vf_off); free(s->vf_fun); free(s->last); free(s->weights); free(s->callvals); free(s);
}
/* gibbs-simple.go:119-145 FunctionsChanged */
static int simple_functions_changed(orc_simple *s) {
    orc_model *m = s->pgm;
    for (int f = 0; f < m->nfuncs; f++)
        if (!m->funcs[f]->is_log) return orc_fail("Function is not in log space on FunctionsChanged");
    simple_build_varfuncs(s);
    for (int i = 0; i < m->nvars; i++) {
        orc_var *v = &m->vars[i];
        if (v->fixed_val >= 0) s->last[i] = v->fixed_val;
        else if (orc_weighted_sample(s->gen, v->card, v->marginal, v->card, &s->last[i])) return -1;
    }
    return 0;
}

/* gibbs-simple.go:163-271 SampleVar.  Returns updated index (>=0), -1 on the
 * "(-1, nil)" draw-failure quirk (line 262-265), -2 on error.
 * If probs_out != NULL the post-floor weights and their re-summed total (what
 * WeightedSample sees) are written there: probs_out[0..card) = w_i / tot. */
static int simple_sample_var(orc_simple *g, int var_idx, int *s, double *probs_out) {
    orc_model *m = g->pgm;
    orc_var *sv = &m->vars[var_idx];
    sv->selections += 1.0;
    if (sv->fixed_val >= 0) { orc_fail("Selected sample variable which has FixedVal"); return -2; }
    double *w = g->weights;
    for (int i = 0; i < sv->card; i++) w[i] = 0.0;
    for (int k = g->vf_off[var_idx]; k < g->vf_off[var_idx + 1]; k++) { /* 186-220 */
        const orc_func *fun = m->funcs[g->vf_fun[k]];
        int *cv = g->callvals, call_idx = -1;
        for (int i = 0; i < fun->nvars; i++) {
            cv[i] = g->last[fun->vars[i]];
            if (fun->vars[i] == sv->id) call_idx = i;
        }
        if (call_idx < 0) { orc_fail("Var not in function var list?!"); return -2; }
        for (int v = 0; v < sv->card; v++) {
            cv[call_idx] = v;
            double r;
            if (func_eval(m, fun, cv, fun->nvars, &r)) return -2;
            w[v] += r;
        }
    }
    double minw = w[0]; /* 227-237 */
    for (int i = 1; i < sv->card; i++) if (w[i] < minw) minw = w[i];
    if (minw < -8.0) for (int i = 0; i < sv->card; i++) w[i] = w[i] - (minw - 1.5);
    double tot = 0.0; /* 239-244 */
    for (int i = 0; i < sv->card; i++) { double v = exp(w[i]); tot += v; w[i] = v; }
    for (int i = 0; i < sv->card; i++) { /* 248-258: sequential floor, mutating tot */
        if (w[i] / tot < 1e-6) {
            double delta = tot * 1e-6;
            if (delta <= 1e-12) { orc_fail("Logic error: delta"); return -2; }
            tot += delta;
            w[i] += delta;
        }
    }
    if (probs_out) {
        double t2 = 0.0;
        for (int i = 0; i < sv->card; i++) t2 += w[i];
        for (int i = 0; i < sv->card; i++) probs_out[i] = w[i] / t2;
    }
    int next;
    if (orc_weighted_sample(g->gen, sv->card, w, sv->card, &next)) return -1; /* 262-265 */
    g->last[var_idx] = next;
    memcpy(s, g->last, sizeof(int) * m->nvars); /* 267-268 */
    return var_idx;
}
/* gibbs-simple.go:148-160 Sample */
ORC_API int orc_simple_sample(orc_simple *g, int *s, int slen) {
    if (slen != g->pgm->nvars) { orc_fail("Sample size != Var size"); return -2; }
    int v;
    if (var_sample(g->gen, g->pgm, 0, &v)) return -2;
    return simple_sample_var(g, v, s, NULL);
}
ORC_API int orc_simple_sample_var(orc_simple *g, int var_idx, int *s, double *probs_out) { return simple_sample_var(g, var_idx, s, probs_out); }
ORC_API void orc_simple_get_last(const orc_simple *g, int *out) { memcpy(out, g->last, sizeof(int) * g->pgm->nvars); }
ORC_API void orc_simple_set_last(orc_simple *g, const int *in) { memcpy(g->last, in, sizeof(int) * g->pgm->nvars); }

/* Conditional law of var v given a full state, without consuming a draw: the
 * normalised post-floor weights of gibbs-simple.go:175-258.  Used to check the
 * device's conditional tables entry by entry. */
ORC_API int orc_simple_conditional(orc_simple *g, int var_idx, const int *state, double *probs_out) {
    int *save = (int *)malloc(sizeof(int) * g->pgm->nvars);
    memcpy(save, g->last, sizeof(int) * g->pgm->nvars);
    memcpy(g->last, state, sizeof(int) * g->pgm->nvars);
    orc_gen tmp = *g->gen; /* run the draw on a scratch copy of the generator */
    orc_gen *keep = g->gen; g->gen = &tmp;
    double sel = g->pgm->vars[var_idx].selections;
    int *scratch = (int *)malloc(sizeof(int) * g->pgm->nvars);
    int rc = simple_sample_var(g, var_idx, scratch, probs_out);
    g->pgm->vars[var_idx].selections = sel;
    g->gen = keep;
    memcpy(g->last, save, sizeof(int) * g->pgm->nvars);
    free(save); free(scratch);
    return rc == -2 ? -1 : 0;
}

/* ---------------------------------------------------------- GibbsCollapsed
 * sampler/gibbs-collapsed.go */
#define ORC_NEIGHBOR_VAR_MAX 12 /* gibbs-collapsed.go:93 */

typedef struct {
    orc_simple *base;
    unsigned char *nbr; /* varNeighbors as an nvars x nvars bit matrix (row v = set of v's neighbours incl. v) */
    int *nbr_count;
} orc_collapsed;

/* gibbs-collapsed.go:44-78 */
static int collapsed_functions_changed(orc_collapsed *g) {
    orc_simple *b = g->base; const orc_model *m = b->pgm; int n = m->nvars;
    memset(g->nbr, 0, (size_t)n * n);
    for (int v = 0; v < n; v++)
        for (int k = b->vf_off[v]; k < b->vf_off[v + 1]; k++) {
            const orc_func *f = m->funcs[b->vf_fun[k]];
            for (int j = 0; j < f->nvars; j++) g->nbr[(size_t)v * n + f->vars[j]] = 1;
        }
    for (int v = 0; v < n; v++) {
        int c = 0; for (int u = 0; u < n; u++) c += g->nbr[(size_t)v * n + u];
        g->nbr_count[v] = c;
        if (m->vars[v].collapsed && c > 0) return orc_fail("Var is collapsed but has a blanket");
    }
    return 0;
}
/* gibbs-collapsed.go:23-40 */
ORC_API orc_collapsed *orc_collapsed_new(orc_gen *gen, orc_model *m) {
    orc_simple *b = orc_simple_new(gen, m);
    if (!b) return NULL;
    orc_collapsed *g = (orc_collapsed *)calloc(1, sizeof *g);
    g->base = b;
    g->nbr = (unsigned char *)calloc((size_t)m->nvars * m->nvars, 1);
    g->nbr_count = (int *)calloc(m->nvars, sizeof(int));
    if (collapsed_functions_changed(g)) return NULL;
    return g;
}
ORC_API void orc_collapsed_free(orc_collapsed *g) { if (g) { orc_simple_free(g->base); free(g->nbr); free(g->nbr_count); free(g); } }
ORC_API orc_simple *orc_collapsed_base(orc_collapsed *g) { return g->base; }
ORC_API int orc_collapsed_blanket_size(const orc_collapsed *g, int v) { return g->nbr_count[v]; }                 /* 81-83 */
ORC_API int orc_collapsed_function_count(const orc_collapsed *g, int v) { return g->base->vf_off[v + 1] - g->base->vf_off[v]; } /* 86-88 */

/* gibbs-collapsed.go:98-314 Collapse.  Returns the collapsed var index or -1.
 * The reference iterates a Go map to build the blanket (line 154), so the new
 * factor's variable order is unspecified there; the oracle uses ascending
 * variable id, one of the orders the reference can produce. */
ORC_API int orc_collapsed_collapse(orc_collapsed *g, int var_idx) {
    orc_simple *base = g->base; orc_model *pgm = base->pgm; int n = pgm->nvars;
    if (var_idx < 0) { /* 102-120 */
        for (int i = 0; i < n; i++) {
            if (var_sample(base->gen, pgm, 1, &var_idx)) return -1;
            if (g->nbr_count[var_idx] <= ORC_NEIGHBOR_VAR_MAX) break;
            var_idx = -1;
        }
    }
    if (var_idx < 0) { orc_fail("Failed to randomly select a variable to collapse"); return -1; }
    if (var_idx >= n) { orc_fail("Invalid variable index"); return -1; }
    orc_var *src = &pgm->vars[var_idx];
    if (src->fixed_val >= 0) { orc_fail("Can not collapse Fixed Val variable"); return -1; }
    if (src->collapsed) { orc_fail("Already collapsed variable"); return -1; }
    int card = src->card;
    double *marg = (double *)malloc(sizeof(double) * card);
    for (int i = 0; i < card; i++) marg[i] = 1e-12; /* 138-140 */

    int *blanket = (int *)malloc(sizeof(int) * n), nb = 0;
    int *xref = (int *)malloc(sizeof(int) * n);
    int *newvars = (int *)malloc(sizeof(int) * n), nnew = 0, coll_idx = -1;
    for (int vi = 0; vi < n; vi++) { /* 150-167 */
        xref[vi] = -1;
        if (!g->nbr[(size_t)var_idx * n + vi]) continue;
        blanket[nb] = vi; xref[vi] = nb;
        if (vi == var_idx) coll_idx = nb; else newvars[nnew++] = vi;
        nb++;
    }
    int rc = -1; orc_func *post = NULL; orc_viter *it = NULL;
    int *state = (int *)calloc(n, sizeof(int)), *cv = (int *)calloc(n, sizeof(int));
    if (coll_idx < 0) { orc_fail("Collapsing variable not in its own blanket"); goto done; }
    if (nnew != nb - 1) { orc_fail("New function size mismatch"); goto done; }
    if (nnew < 1) { orc_fail("New function would have 0 variables"); goto done; } /* 175-177 */
    for (int k = base->vf_off[var_idx]; k < base->vf_off[var_idx + 1]; k++)
        if (!pgm->funcs[base->vf_fun[k]]->is_log) { orc_fail("Function is not set up for Log Space"); goto done; }
    post = func_new(pgm, newvars, nnew); /* 192-196 (fails above maxTabSize) */
    if (!post) goto done;
    it = orc_viter_new(pgm, blanket, nb, 1);
    for (;;) { /* 206-260 */
        orc_viter_val(it, state);
        int mval = state[coll_idx];
        double fr = 0.0;
        for (int k = base->vf_off[var_idx]; k < base->vf_off[var_idx + 1]; k++) {
            const orc_func *fun = pgm->funcs[base->vf_fun[k]];
            for (int i = 0; i < fun->nvars; i++) cv[i] = state[xref[fun->vars[i]]];
            double r;
            if (func_eval(pgm, fun, cv, fun->nvars, &r)) goto done;
            fr += r;
        }
        fr = exp(fr);
        marg[mval] += fr;
        for (int i = 0; i < nnew; i++) cv[i] = state[xref[newvars[i]]];
        if (func_add_value(pgm, post, cv, nnew, fr)) goto done;
        if (!orc_viter_next(it)) break;
    }
    { orc_var tmp = {var_idx, card, -1, 0, marg, 0}; var_norm_marginal(&tmp); } /* 263 */
    if (func_use_log_space(post)) goto done;                                    /* 269 */
    { /* 275-291: append post, drop var's factors */
        int nf = pgm->nfuncs;
        unsigned char *del = (unsigned char *)calloc(nf + 1, 1);
        for (int k = base->vf_off[var_idx]; k < base->vf_off[var_idx + 1]; k++) del[base->vf_fun[k]] = 1;
        pgm->funcs = (orc_func **)realloc(pgm->funcs, sizeof(orc_func *) * (nf + 1));
        pgm->funcs[nf] = post; post = NULL;
        int ins = 0;
        for (int i = 0; i <= nf; i++) {
            if (del[i]) { func_free(pgm->funcs[i]); continue; }
            pgm->funcs[ins++] = pgm->funcs[i];
        }
        free(del);
        if (ins < 1) { orc_fail("No functions left after collapse!"); goto done; }
        pgm->nfuncs = ins;
    }
    if (simple_functions_changed(base)) goto done;  /* 296 (re-draws start state from marginals) */
    if (collapsed_functions_changed(g)) goto done;  /* 300 */
    if (orc_model_check(pgm)) goto done;            /* 304 */
    src->collapsed = 1;                              /* 311-313 */
    memcpy(src->marginal, marg, sizeof(double) * card);
    rc = var_idx;
done:
    if (post) func_free(post);
    orc_viter_free(it);
    free(marg); free(blanket); free(xref); free(newvars); free(state); free(cv);
    return rc;
}
/* gibbs-collapsed.go:317-334 Sample */
ORC_API int orc_collapsed_sample(orc_collapsed *g, int *s, int slen) {
    if (slen != g->base->pgm->nvars) { orc_fail("Samples size is wrong"); return -2; }
    int v;
    if (var_sample(g->base->gen, g->base->pgm, 1, &v)) return -2;
    return simple_sample_var(g->base, v, s, NULL);
}

/* ------------------------------------------------------------ CircularInt
 * buffer/circular.go:1-102: ring of the last BufSize ints, iterable as an older
 * and a newer half. */
typedef struct { int *buffer; int pos, buf_size, count; int64_t total_seen; } orc_ring;

ORC_API orc_ring *orc_ring_new(int total_size) { /* circular.go:16-28: odd sizes are rounded down */
    orc_ring *r = (orc_ring *)calloc(1, sizeof *r);
    int half = total_size / 2;
    r->buf_size = half + half;
    r->buffer = (int *)calloc(r->buf_size > 0 ? r->buf_size : 1, sizeof(int));
    return r;
}
ORC_API void orc_ring_free(orc_ring *r) { if (r) { free(r->buffer); free(r); } }
ORC_API void orc_ring_add(orc_ring *r, int v) { /* circular.go:36-49 (a zero-sized ring would panic in Go; here it only counts) */
    r->total_seen++;
    if (r->buf_size < 1) return;
    r->buffer[r->pos] = v;
    r->pos = (r->pos + 1) % r->buf_size;
    r->count++;
    if (r->count > r->buf_size) r->count = r->buf_size;
}
/* circular.go:54-84: copies the older (which=0) or newer (which=1) half into out[buf_size/2];
 * returns the number of values or -1 while the ring is not full (Go returns a nil iterator) */
ORC_API int orc_ring_half(const orc_ring *r, int which, int *out) {
    if (r->count < r->buf_size || r->buf_size < 1) return -1;
    int half = r->buf_size / 2;
    int cur = which ? (r->pos + half) % r->buf_size : r->pos;
    for (int i = 0; i < half; i++) { out[i] = r->buffer[cur]; cur = (cur + 1) % r->buf_size; }
    return half;
}
ORC_API int orc_ring_buf_size(const orc_ring *r) { return r->buf_size; }
ORC_API int orc_ring_count(const orc_ring *r) { return r->count; }

/* ------------------------------------------------------------------- Chain
 * sampler/chain.go */
typedef struct {
    orc_model *target;       /* not owned */
    orc_simple *simple;      /* exactly one of simple / collapsed is set */
    orc_collapsed *coll;
    int cw;
    orc_ring **history;      /* ChainHistory: one ring of cw values per variable (chain.go:156-164) */
    int64_t *total_seen;
    int64_t total_sample_count;
    int *last_sample;
} orc_chain;

/* chain.go:221-246 oneSample */
static int chain_one_sample(orc_chain *c, int update_vars) {
    int n = c->target->nvars;
    int idx = c->coll ? orc_collapsed_sample(c->coll, c->last_sample, n) : orc_simple_sample(c->simple, c->last_sample, n);
    if (idx == -2) return -1;
    if (idx < 0 || c->target->vars[idx].fixed_val >= 0) return orc_fail("Invalid sample");
    if (update_vars) {
        int value = c->last_sample[idx];
        orc_var *v = &c->target->vars[idx];
        if (!v->collapsed) v->marginal[value] += 1.0;
        orc_ring_add(c->history[idx], value); /* chain.go:238 */
        c->total_seen[idx]++;
        c->total_sample_count++;
    }
    return 0;
}
/* chain.go:151-175 NewChain (burn-in included) */
static orc_chain *chain_new(orc_model *m, orc_simple *s, orc_collapsed *g, int cw, int64_t burn_in) {
    orc_chain *c = (orc_chain *)calloc(1, sizeof *c);
    c->target = m; c->simple = s; c->coll = g; c->cw = cw;
    c->total_seen = (int64_t *)calloc(m->nvars, sizeof(int64_t));
    c->last_sample = (int *)calloc(m->nvars, sizeof(int));
    c->history = (orc_ring **)calloc(m->nvars, sizeof(orc_ring *));
    for (int i = 0; i < m->nvars; i++) c->history[i] = orc_ring_new(cw);
    for (int64_t i = 0; i < burn_in; i++)
        if (chain_one_sample(c, 0)) { orc_fail("Failure during chain burn in"); return NULL; }
    return c;
}
ORC_API orc_chain *orc_chain_new_simple(orc_model *m, orc_simple *s, int cw, int64_t burn_in) { return chain_new(m, s, NULL, cw, burn_in); }
ORC_API orc_chain *orc_chain_new_collapsed(orc_model *m, orc_collapsed *g, int cw, int64_t burn_in) { return chain_new(m, NULL, g, cw, burn_in); }
ORC_API orc_chain *orc_chain_new_nosampler(orc_model *m) { return chain_new(m, NULL, NULL, 0, 0); } /* chain_test.go:32 */
/* Test hooks (tests/test_gpu_adaptive.py): a chain without a sampler whose ChainHistory rings hold ring_size values while its
 * ConvergenceWindow is cw -- one such chain stands for a 64-chain group of the GPU path, its rings holding the values of all
 * the group's chains (the distance measures of ChainDist normalise their histograms, chain.go:253-290) -- and the ring
 * write of oneSample alone (chain.go:238-241). */
ORC_API orc_chain *orc_chain_new_history(orc_model *m, int cw, int ring_size) {
    orc_chain *c = chain_new(m, NULL, NULL, 0, 0);
    if (!c) return NULL;
    c->cw = cw;
    for (int i = 0; i < m->nvars; i++) { orc_ring_free(c->history[i]); c->history[i] = orc_ring_new(ring_size); }
    return c;
}
ORC_API int orc_chain_history_add(orc_chain *c, int var, int value) {
    if (var < 0 || var >= c->target->nvars || value < 0 || value >= c->target->vars[var].card) return orc_fail("history add out of range");
    orc_ring_add(c->history[var], value);
    c->total_seen[var]++;
    return 0;
}
ORC_API void orc_chain_free(orc_chain *c) {
    if (!c) return;
    if (c->history) for (int i = 0; i < c->target->nvars; i++) orc_ring_free(c->history[i]);
    free(c->history); free(c->total_seen); free(c->last_sample); free(c);
}
ORC_API int64_t orc_chain_total_samples(const orc_chain *c) { return c->total_sample_count; }
ORC_API void orc_chain_get_last(const orc_chain *c, int *out) { memcpy(out, c->last_sample, sizeof(int) * c->target->nvars); }

/* chain.go:180-218 AdvanceChain (run synchronously; the goroutine only adds concurrency) */
ORC_API int orc_chain_advance(orc_chain *c) {
    int n = c->target->nvars;
    int64_t *thresh = (int64_t *)malloc(sizeof(int64_t) * n);
    for (int i = 0; i < n; i++) thresh[i] = c->total_seen[i] + (int64_t)c->cw + 1;
    int batch = n * 2, rc = 0;
    for (;;) {
        int keep = 0;
        for (int i = 0; i < n; i++) {
            const orc_var *v = &c->target->vars[i];
            if (!v->collapsed && v->fixed_val < 0 && c->total_seen[i] < thresh[i]) { keep = 1; break; }
        }
        if (!keep) break;
        for (int i = 0; i < batch; i++) if (chain_one_sample(c, 1)) { rc = -1; goto out; }
    }
out:
    free(thresh);
    return rc;
}
/* fixed-length advance: `updates` calls of oneSample(true) (the inner loop of chain.go:207-214) */
ORC_API int orc_chain_run(orc_chain *c, int64_t updates) {
    for (int64_t i = 0; i < updates; i++) if (chain_one_sample(c, 1)) return -1;
    return 0;
}

/* chain.go:96-148 MergeChains.  out_marginal: concatenated per-variable
 * marginals (sum of cards); out_collapsed[v]: merged var's Collapsed flag. */
ORC_API int orc_merge_chains(orc_chain **chains, int n, double *out_marginal, int *out_collapsed) {
    if (n < 1) return orc_fail("Can not merge 0 chains");
    int nv = chains[0]->target->nvars;
    int *off = (int *)malloc(sizeof(int) * (nv + 1)); off[0] = 0;
    for (int v = 0; v < nv; v++) off[v + 1] = off[v] + chains[0]->target->vars[v].card;
    int *is_coll = (int *)calloc(nv, sizeof(int));
    for (int v = 0; v < nv; v++) {
        const orc_var *found = NULL;
        if (n > 1) for (int c = 0; c < n; c++) if (chains[c]->target->vars[v].collapsed) { found = &chains[c]->target->vars[v]; break; }
        const orc_var *srcv = found ? found : &chains[0]->target->vars[v];
        is_coll[v] = found != NULL;
        memcpy(out_marginal + off[v], srcv->marginal, sizeof(double) * srcv->card);
        if (out_collapsed) out_collapsed[v] = srcv->collapsed;
    }
    int rc = 0;
    for (int c = 1; c < n; c++) { /* n == 1 returns chain 0's vars untouched (line 101-103) */
        if (chains[c]->target->nvars != nv) { rc = orc_fail("Cannot merge chains with different var counts"); break; }
        for (int v = 0; v < nv; v++) {
            if (is_coll[v]) continue;
            const orc_var *sv = &chains[c]->target->vars[v];
            for (int k = 0; k < sv->card; k++) out_marginal[off[v] + k] += sv->marginal[k];
        }
    }
    free(off); free(is_coll);
    return rc;
}

/* ----------------------------------------------------------------- metrics
 * model/error.go.  Marginals are passed as concatenated arrays + cards. */
static void norm_tot(const double *a, const double *b, int card, double *ta, double *tb) {
    const double eps = 1e-12; double t1 = 0, t2 = 0;
    for (int c = 0; c < card; c++) { t1 += a[c]; t2 += b[c]; }
    if (t1 < eps) t1 = eps;
    if (t2 < eps) t2 = eps;
    *ta = t1; *tb = t2;
}
/* error.go:81-113 */
ORC_API double orc_max_abs_diff(const double *a, const double *b, int card) {
    double t1, t2, mx = 0; norm_tot(a, b, card, &t1, &t2);
    for (int c = 0; c < card; c++) { double e = fabs(a[c] / t1 - b[c] / t2); if (c == 0 || e > mx) mx = e; }
    return mx;
}
/* error.go:116-151 */
ORC_API double orc_mean_abs_diff(const double *a, const double *b, int card) {
    if (card < 1) return 0;
    double t1, t2, s = 0; norm_tot(a, b, card, &t1, &t2);
    for (int c = 0; c < card; c++) s += fabs(a[c] / t1 - b[c] / t2);
    return s / (double)card;
}
/* error.go:158-190 */
ORC_API double orc_hellinger(const double *a, const double *b, int card) {
    double t1, t2, s = 0; norm_tot(a, b, card, &t1, &t2);
    for (int c = 0; c < card; c++) { double d = sqrt(a[c] / t1) - sqrt(b[c] / t2); s += pow(d, 2); }
    return sqrt(s) / M_SQRT2;
}
/* error.go:197-213 */
static double kl_div(const double *p, const double *q, int n) {
    const double eps = 1e-12; double d = 0;
    for (int i = 0; i < n; i++) {
        double p1 = p[i], p2 = q[i];
        if (p1 < eps) p1 = eps;
        if (p2 < eps) p2 = eps;
        d += p1 * log2(p1 / p2);
    }
    return d;
}
/* error.go:217-249 */
ORC_API double orc_js_divergence(const double *a, const double *b, int card) {
    double t1, t2; norm_tot(a, b, card, &t1, &t2);
    double *p1 = (double *)malloc(sizeof(double) * card * 3), *p2 = p1 + card, *mid = p2 + card;
    for (int i = 0; i < card; i++) { p1[i] = a[i] / t1; p2[i] = b[i] / t2; mid[i] = (p1[i] + p2[i]) * 0.5; }
    double r = 0.5 * (kl_div(p1, mid, card) + kl_div(p2, mid, card));
    free(p1);
    return r;
}
/* error.go:28-78 NewErrorSuite.  out[8] = MeanMeanAbs, MeanMaxAbs, MeanHellinger,
 * MeanJS, MaxMeanAbs, MaxMaxAbs, MaxHellinger, MaxJS.  fixed1/fixed2 may be NULL. */
ORC_API int orc_error_suite(int nvars, const int *card1, const int *card2, const int *fixed1, const int *fixed2,
                            const double *m1, const double *m2, double *out) {
    int count = 0;
    for (int i = 0; i < nvars; i++) {
        if (card1[i] != card2[i]) return orc_fail("Variable card mismatch");
        int f1 = fixed1 ? fixed1[i] : -1, f2 = fixed2 ? fixed2[i] : -1;
        if (f1 < 0 && f2 < 0) count++;
    }
    if (count < 1) return orc_fail("No un-fixed vars to score");
    for (int k = 0; k < 8; k++) out[k] = 0;
    int off = 0;
    for (int i = 0; i < nvars; i++) {
        int c = card1[i];
        int fixed = (fixed1 && fixed1[i] >= 0) || (fixed2 && fixed2[i] >= 0);
        double d;
        d = fixed ? 0.0 : orc_mean_abs_diff(m1 + off, m2 + off, c); out[0] += d; out[4] = fmax(d, out[4]);
        d = fixed ? 0.0 : orc_max_abs_diff(m1 + off, m2 + off, c);  out[1] += d; out[5] = fmax(d, out[5]);
        d = fixed ? 0.0 : orc_hellinger(m1 + off, m2 + off, c);     out[2] += d; out[6] = fmax(d, out[6]);
        d = fixed ? 0.0 : orc_js_divergence(m1 + off, m2 + off, c); out[3] += d; out[7] = fmax(d, out[7]);
        off += c;
    }
    for (int k = 0; k < 4; k++) out[k] /= (double)count;
    return 0;
}

/* --------------------------------------------------------- ChainConvergence
 * sampler/chain.go:32-92, 253-290.  measure: 0 Hellinger, 1 JS, 2 MaxAbs, 3 MeanAbs. */
static double measure_fn(int measure, const double *a, const double *b, int card) {
    switch (measure) {
    case 1: return orc_js_divergence(a, b, card);
    case 2: return orc_max_abs_diff(a, b, card);
    case 3: return orc_mean_abs_diff(a, b, card);
    default: return orc_hellinger(a, b, card);
    }
}
/* chain.go:253-290 ChainDist: within = d(first half, second half), between = d(merged, first + second) */
static int chain_dist(const orc_chain *c, int measure, int v, const double *merged, int merged_card, double *within, double *between) {
    const orc_ring *hist = c->history[v];
    if (hist->total_seen < (int64_t)c->cw) return orc_fail("Total seen < Convergence Window");
    int card = c->target->vars[v].card;
    if (card != merged_card) return orc_fail("Variable mismatch");
    double *v1 = (double *)malloc(sizeof(double) * card * 2), *v2 = v1 + card;
    for (int i = 0; i < card; i++) v1[i] = v2[i] = 1e-8;
    int half = hist->buf_size / 2;
    int *tmp = (int *)malloc(sizeof(int) * (half > 0 ? half : 1));
    int n1 = orc_ring_half(hist, 0, tmp);
    for (int i = 0; i < n1; i++) v1[tmp[i]] += 1.0;
    int n2 = orc_ring_half(hist, 1, tmp);
    for (int i = 0; i < n2; i++) v2[tmp[i]] += 1.0;
    free(tmp);
    if (n1 < 0 || n2 < 0) { free(v1); return orc_fail("history ring not full (Go: nil iterator)"); }
    *within = measure_fn(measure, v1, v2, card);
    for (int i = 0; i < card; i++) v1[i] += v2[i];
    *between = measure_fn(measure, merged, v1, card);
    free(v1);
    return 0;
}
/* chain.go:32-92.  merged / merged_collapsed may be NULL (then MergeChains is called). out[nvars]. */
ORC_API int orc_chain_convergence(orc_chain **chains, int n, int measure, const double *merged, const int *merged_collapsed, double *out) {
    if (n < 2) return orc_fail("Convergence requires at least 2 chains");
    int nv = chains[0]->target->nvars, tot = 0;
    for (int v = 0; v < nv; v++) tot += chains[0]->target->vars[v].card;
    double *mm = NULL; int *mc = NULL;
    if (!merged) {
        mm = (double *)malloc(sizeof(double) * tot); mc = (int *)malloc(sizeof(int) * nv);
        if (orc_merge_chains(chains, n, mm, mc)) { free(mm); free(mc); return -1; }
        merged = mm; merged_collapsed = mc;
    }
    double sample_count = (double)chains[0]->cw, chain_count = (double)n;
    double b_norm = sample_count / (chain_count - 1);
    double w_factor = (sample_count - 1) / sample_count;
    double b_factor = (chain_count + 1) / (chain_count * sample_count);
    int off = 0, rc = 0;
    for (int i = 0; i < nv && !rc; i++) {
        int card = chains[0]->target->vars[i].card;
        if ((merged_collapsed && merged_collapsed[i]) || chains[0]->target->vars[i].fixed_val >= 0) { out[i] = 1.0; off += card; continue; }
        double W = 1e-8, B = 1e-8;
        for (int c = 0; c < n; c++) {
            double w1, b1;
            if (chain_dist(chains[c], measure, i, merged + off, card, &w1, &b1)) { rc = -1; break; }
            W += w1; B += b1;
        }
        W /= chain_count;
        B *= b_norm;
        double vhat = (w_factor * W) + (b_factor * B);
        out[i] = sqrt((4.0 * vhat) / (2.0 * W));
        off += card;
    }
    free(mm); free(mc);
    return rc;
}

/* ---------------------------------------------------------------- ensemble
 * What cmd/root.go:373-430 builds for `--sampler adaptive` plus ConvergenceSampler.Adapt
 * (sampler/adaptive.go:57-157).  Owns the generator, the base model and every chain. */
#define ORC_MAX_CHAINS 128 /* adaptive.go:49 */
typedef struct {
    orc_gen *gen; orc_model *base;
    int n, cap;
    orc_chain **ch; orc_model **mods; orc_collapsed **cs;
} orc_ensemble;

static int ensemble_push(orc_ensemble *e, orc_model *m, orc_collapsed *g, orc_chain *c) {
    if (e->n == e->cap) {
        e->cap = e->cap ? e->cap * 2 : 8;
        e->ch = (orc_chain **)realloc(e->ch, sizeof(*e->ch) * e->cap);
        e->mods = (orc_model **)realloc(e->mods, sizeof(*e->mods) * e->cap);
        e->cs = (orc_collapsed **)realloc(e->cs, sizeof(*e->cs) * e->cap);
    }
    e->ch[e->n] = c; e->mods[e->n] = m; e->cs[e->n] = g; e->n++;
    return 0;
}
/* root.go:408-415 + 422: NewGibbsCollapsed without a collapse, NewChain(cw, burn-in) for every base chain */
ORC_API orc_ensemble *orc_ensemble_new(const orc_model *base, int64_t seed, int n_base, int cw, int64_t burn_in) {
    orc_ensemble *e = (orc_ensemble *)calloc(1, sizeof *e);
    e->gen = orc_gen_new(seed);
    e->base = orc_model_clone(base);
    for (int c = 0; c < n_base; c++) {
        orc_model *m = orc_model_clone(base);
        orc_collapsed *g = orc_collapsed_new(e->gen, m);
        orc_chain *ch = g ? orc_chain_new_collapsed(m, g, cw, burn_in) : NULL;
        if (!ch) return NULL;
        ensemble_push(e, m, g, ch);
    }
    return e;
}
ORC_API void orc_ensemble_free(orc_ensemble *e) {
    if (!e) return;
    for (int c = 0; c < e->n; c++) { orc_chain_free(e->ch[c]); orc_collapsed_free(e->cs[c]); orc_model_free(e->mods[c]); }
    free(e->ch); free(e->mods); free(e->cs); orc_model_free(e->base); orc_gen_free(e->gen); free(e);
}
ORC_API int orc_ensemble_n(const orc_ensemble *e) { return e->n; }
ORC_API orc_chain *orc_ensemble_chain(orc_ensemble *e, int i) { return e->ch[i]; }
ORC_API int orc_ensemble_advance(orc_ensemble *e) { /* root.go:476-479 */
    for (int c = 0; c < e->n; c++) if (orc_chain_advance(e->ch[c])) return -1;
    return 0;
}
ORC_API int orc_ensemble_run(orc_ensemble *e, int64_t updates) {
    for (int c = 0; c < e->n; c++) if (orc_chain_run(e->ch[c], updates)) return -1;
    return 0;
}
ORC_API int orc_ensemble_merge(orc_ensemble *e, double *out_marginal, int *out_collapsed) { return orc_merge_chains(e->ch, e->n, out_marginal, out_collapsed); }
ORC_API int orc_ensemble_convergence(orc_ensemble *e, int measure, double *out) { return orc_chain_convergence(e->ch, e->n, measure, NULL, NULL, out); }

/* adaptive.go:57-157 Adapt.  out_vars[<= new_chain_count] receives the collapsed variable of every chain added;
 * out_converge[nvars] (may be NULL) the scores used for ranking (NaN when ranking was not needed).
 * Ranking: the reference sorts descending with sort.Slice (not stable) and takes from the TAIL, i.e. the
 * lowest scores; ties are broken here by descending variable id at the tail (= ascending id picked last). */
ORC_API int orc_ensemble_adapt(orc_ensemble *e, int new_chain_count, int *out_vars, int *out_n, double *out_converge) {
    *out_n = 0;
    int nv = e->base->nvars;
    if (out_converge) for (int v = 0; v < nv; v++) out_converge[v] = NAN;
    if (e->n < 2) return orc_fail("At least 2 chains required for adaptation");
    if (e->n >= ORC_MAX_CHAINS) return 0;
    orc_model *mod_clone = orc_model_clone(e->base);
    orc_collapsed *samp = orc_collapsed_new(e->gen, mod_clone);
    if (!samp) return -1;
    int tot = 0; for (int v = 0; v < nv; v++) tot += e->base->vars[v].card;
    double *merged = (double *)malloc(sizeof(double) * tot);
    int *mcoll = (int *)malloc(sizeof(int) * nv);
    int rc = orc_merge_chains(e->ch, e->n, merged, mcoll);
    int *vars = (int *)malloc(sizeof(int) * nv), nvars = 0;
    for (int v = 0; v < nv && !rc; v++) {
        int sz = orc_collapsed_blanket_size(samp, v);
        if (e->base->vars[v].fixed_val < 0 && !mcoll[v] && sz > 1 && sz <= ORC_NEIGHBOR_VAR_MAX) vars[nvars++] = v;
    }
    int *targets = (int *)malloc(sizeof(int) * (new_chain_count > 0 ? new_chain_count : 1)), nt = 0;
    if (!rc && nvars >= 1) {
        if (nvars <= new_chain_count) {
            for (int i = 0; i < nvars; i++) targets[nt++] = vars[i];
        } else {
            double *conv = (double *)malloc(sizeof(double) * nv);
            rc = orc_chain_convergence(e->ch, e->n, 0, merged, mcoll, conv);
            if (!rc) {
                if (out_converge) memcpy(out_converge, conv, sizeof(double) * nv);
                /* descending by score (insertion sort, stable); then take from the tail */
                for (int i = 1; i < nvars; i++) {
                    int x = vars[i], j = i - 1;
                    while (j >= 0 && conv[vars[j]] < conv[x]) { vars[j + 1] = vars[j]; j--; }
                    vars[j + 1] = x;
                }
                for (int cc = 0, pos = nvars - 1; cc < new_chain_count; cc++, pos--) targets[nt++] = vars[pos];
            }
            free(conv);
        }
    }
    const orc_chain *last = e->ch[e->n - 1];
    for (int i = 0; i < nt && !rc; i++) { /* adaptive.go:131-153 */
        if (!samp) {
            mod_clone = orc_model_clone(e->base);
            samp = orc_collapsed_new(e->gen, mod_clone);
            if (!samp) { rc = -1; break; }
        }
        if (orc_collapsed_collapse(samp, targets[i]) < 0) { rc = -1; break; }
        orc_chain *nc = orc_chain_new_collapsed(mod_clone, samp, last->cw, 2);
        if (!nc) { rc = -1; break; }
        ensemble_push(e, mod_clone, samp, nc);
        out_vars[(*out_n)++] = targets[i];
        samp = NULL; mod_clone = NULL;
    }
    if (samp) { orc_collapsed_free(samp); orc_model_free(mod_clone); }
    free(merged); free(mcoll); free(vars); free(targets);
    return rc;
}

/* ------------------------------------------------- multi-chain convenience
 * What cmd/root.go:373-430 + 475-479 + 507 do for `--sampler simple|collapsed`,
 * with one generator shared by all chains, chains built and burnt-in
 * sequentially, then each advanced by a fixed number of updates (sequentially:
 * a legal interleaving of the reference's racing goroutines).
 * collapse_mode: 0 simple; 1 collapsed with Collapse(-1); 2 collapsed with
 * Collapse(collapse_vars[c]) for chain c.
 * out_marginal: MergeChains result (incl. the 1/Card prior per chain). */
ORC_API int orc_run_chains(const orc_model *base, int64_t seed, int n_chains, int64_t burn_in, int64_t updates,
                           int collapse_mode, const int *collapse_vars, double *out_marginal, int *out_collapsed,
                           int *out_collapsed_var, int64_t *out_total) {
    orc_gen *gen = orc_gen_new(seed);
    orc_chain **ch = (orc_chain **)calloc(n_chains, sizeof(*ch));
    orc_model **mods = (orc_model **)calloc(n_chains, sizeof(*mods));
    orc_simple **ss = (orc_simple **)calloc(n_chains, sizeof(*ss));
    orc_collapsed **cs = (orc_collapsed **)calloc(n_chains, sizeof(*cs));
    int rc = 0; int64_t total = 0;
    for (int c = 0; c < n_chains && !rc; c++) {
        mods[c] = orc_model_clone(base);
        if (collapse_mode == 0) {
            ss[c] = orc_simple_new(gen, mods[c]);
            if (!ss[c]) { rc = -1; break; }
            ch[c] = orc_chain_new_simple(mods[c], ss[c], (int)burn_in, burn_in);
        } else {
            cs[c] = orc_collapsed_new(gen, mods[c]);
            if (!cs[c]) { rc = -1; break; }
            int cv = orc_collapsed_collapse(cs[c], collapse_mode == 1 ? -1 : collapse_vars[c]);
            if (cv < 0) { rc = -1; break; }
            if (out_collapsed_var) out_collapsed_var[c] = cv;
            ch[c] = orc_chain_new_collapsed(mods[c], cs[c], (int)burn_in, burn_in);
        }
        if (!ch[c]) rc = -1;
    }
    for (int c = 0; c < n_chains && !rc; c++) { rc = orc_chain_run(ch[c], updates); total += ch[c]->total_sample_count; }
    if (!rc) rc = orc_merge_chains(ch, n_chains, out_marginal, out_collapsed);
    if (out_total) *out_total = total;
    for (int c = 0; c < n_chains; c++) {
        orc_chain_free(ch[c]);
        if (cs[c]) orc_collapsed_free(cs[c]); else orc_simple_free(ss[c]);
        orc_model_free(mods[c]);
    }
    free(ch); free(mods); free(ss); free(cs); orc_gen_free(gen);
    return rc;
}

/* Throughput leg for bench.py's cpu_baseline ("port"): n_threads OS threads,
 * each running its own chains with its own generator (seed + thread index) --
 * the reference shares one generator through a channel; per-thread generators
 * remove that contention, so this is an upper bound on the port's speed.
 * Returns total single-variable updates executed in the counted phase. */
#ifdef _OPENMP
#include <omp.h>
#endif
ORC_API int64_t orc_bench_simple(const orc_model *base, int64_t seed, int n_threads, int chains_per_thread,
                                 int64_t burn_in, int64_t updates, double *out_marginal) {
    int nv = base->nvars, tot_card = 0;
    for (int v = 0; v < nv; v++) tot_card += base->vars[v].card;
    double *acc = (double *)calloc((size_t)tot_card * n_threads, sizeof(double));
    int64_t total = 0; int bad = 0;
#pragma omp parallel for num_threads(n_threads) reduction(+ : total) reduction(| : bad) schedule(static, 1)
    for (int t = 0; t < n_threads; t++) {
        orc_gen *gen = orc_gen_new(seed + t);
        for (int c = 0; c < chains_per_thread; c++) {
            orc_model *m = orc_model_clone(base);
            orc_simple *s = orc_simple_new(gen, m);
            orc_chain *ch = s ? orc_chain_new_simple(m, s, 0, burn_in) : NULL;
            if (!ch || orc_chain_run(ch, updates)) bad |= 1;
            else {
                total += ch->total_sample_count;
                int off = 0;
                for (int v = 0; v < nv; v++) { for (int k = 0; k < m->vars[v].card; k++) acc[(size_t)t * tot_card + off + k] += m->vars[v].marginal[k]; off += m->vars[v].card; }
            }
            orc_chain_free(ch); orc_simple_free(s); orc_model_free(m);
        }
        orc_gen_free(gen);
    }
    if (out_marginal) {
        for (int k = 0; k < tot_card; k++) { double s = 0; for (int t = 0; t < n_threads; t++) s += acc[(size_t)t * tot_card + k]; out_marginal[k] = s; }
    }
    free(acc);
    return bad ? -1 : total;
}
