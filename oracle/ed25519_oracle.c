/* ORACLE (test infrastructure, NOT the product): plain-C restatement of the reference's
 * group work on the AC20 hot path, for sizes where the big-int Python oracle is too slow
 * and as the `cpu_baseline` ("port") leg of bench.py.
 *
 * It follows the REFERENCE ALGORITHM, not a fast one:
 *   vector_commitment  = one right-to-left double-and-add ladder per term, then the
 *                        pairwise product tree, then * h**gamma
 *                        (verifiable_mpc/ac20/pivot.py:139-145, list_mul :26-28)
 *   fold               = (g_l[i] ** c) * g_r[i]   (verifiable_mpc/ac20/compressed_pivot.py:64)
 *   fixed base         = h ** r_i                 (verifiable_mpc/ac20/circuit_sat_r1cs.py:64-70)
 * with MPyC's projective formulas as recalled in oracle/ed25519_ref.py (add-2008-bbjlp,
 * dbl-2008-bbjlp) [mpyc-recall; parity with real MPyC unpinned].  Pinned against
 * oracle/ed25519_ref.py (exact X:Y:Z representatives) and RFC 8032 by
 * tests/test_oracle_c.py.
 *
 * Field: GF(2^255-19), 5 x 51-bit limbs, unsigned __int128 products.  Single thread by default,
 * as the reference is (that is what `cpu_baseline` times); oracle_set_threads() spreads the independent
 * pieces of the SAME algorithm over cores for the parity tests at N = 2^20 / 2^24 (see below).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t v[5]; } fe;
typedef struct { fe X, Y, Z; } pt;

#define MASK51 ((1ULL << 51) - 1)

static void fe_frombytes(fe *h, const uint8_t *s) {
    uint64_t w[4];
    memcpy(w, s, 32);
    h->v[0] = w[0] & MASK51;
    h->v[1] = ((w[0] >> 51) | (w[1] << 13)) & MASK51;
    h->v[2] = ((w[1] >> 38) | (w[2] << 26)) & MASK51;
    h->v[3] = ((w[2] >> 25) | (w[3] << 39)) & MASK51;
    h->v[4] = (w[3] >> 12) & MASK51;   /* inputs are canonical (< p), bit 255 clear */
}

static void fe_carry(fe *h) {
    uint64_t c;
    for (int k = 0; k < 2; k++) {
        c = h->v[0] >> 51; h->v[0] &= MASK51; h->v[1] += c;
        c = h->v[1] >> 51; h->v[1] &= MASK51; h->v[2] += c;
        c = h->v[2] >> 51; h->v[2] &= MASK51; h->v[3] += c;
        c = h->v[3] >> 51; h->v[3] &= MASK51; h->v[4] += c;
        c = h->v[4] >> 51; h->v[4] &= MASK51; h->v[0] += 19 * c;
    }
}

static void fe_tobytes(uint8_t *s, const fe *a) {
    fe t = *a;
    fe_carry(&t);
    /* canonical: subtract p if t >= p */
    uint64_t q = (t.v[0] + 19) >> 51;
    q = (t.v[1] + q) >> 51;
    q = (t.v[2] + q) >> 51;
    q = (t.v[3] + q) >> 51;
    q = (t.v[4] + q) >> 51;
    t.v[0] += 19 * q;
    uint64_t c;
    c = t.v[0] >> 51; t.v[0] &= MASK51; t.v[1] += c;
    c = t.v[1] >> 51; t.v[1] &= MASK51; t.v[2] += c;
    c = t.v[2] >> 51; t.v[2] &= MASK51; t.v[3] += c;
    c = t.v[3] >> 51; t.v[3] &= MASK51; t.v[4] += c;
    t.v[4] &= MASK51;
    uint64_t w[4];
    w[0] = t.v[0] | (t.v[1] << 51);
    w[1] = (t.v[1] >> 13) | (t.v[2] << 38);
    w[2] = (t.v[2] >> 26) | (t.v[3] << 25);
    w[3] = (t.v[3] >> 39) | (t.v[4] << 12);
    memcpy(s, w, 32);
}

static void fe_add(fe *h, const fe *a, const fe *b) {
    for (int i = 0; i < 5; i++) h->v[i] = a->v[i] + b->v[i];
}

/* h = a - b, with a bias of 4p so limbs stay non-negative (inputs carried: limbs < 2^52) */
static void fe_sub(fe *h, const fe *a, const fe *b) {
    h->v[0] = a->v[0] + 0x1FFFFFFFFFFFB4ULL - b->v[0];   /* 4*(2^51-19) */
    h->v[1] = a->v[1] + 0x1FFFFFFFFFFFFCULL - b->v[1];   /* 4*(2^51-1)  */
    h->v[2] = a->v[2] + 0x1FFFFFFFFFFFFCULL - b->v[2];
    h->v[3] = a->v[3] + 0x1FFFFFFFFFFFFCULL - b->v[3];
    h->v[4] = a->v[4] + 0x1FFFFFFFFFFFFCULL - b->v[4];
    fe_carry(h);
}

static void fe_mul(fe *h, const fe *f, const fe *g) {
    const uint64_t *a = f->v, *b = g->v;
    uint64_t b1_19 = 19 * b[1], b2_19 = 19 * b[2], b3_19 = 19 * b[3], b4_19 = 19 * b[4];
    u128 t0 = (u128)a[0] * b[0] + (u128)a[1] * b4_19 + (u128)a[2] * b3_19 + (u128)a[3] * b2_19 + (u128)a[4] * b1_19;
    u128 t1 = (u128)a[0] * b[1] + (u128)a[1] * b[0] + (u128)a[2] * b4_19 + (u128)a[3] * b3_19 + (u128)a[4] * b2_19;
    u128 t2 = (u128)a[0] * b[2] + (u128)a[1] * b[1] + (u128)a[2] * b[0] + (u128)a[3] * b4_19 + (u128)a[4] * b3_19;
    u128 t3 = (u128)a[0] * b[3] + (u128)a[1] * b[2] + (u128)a[2] * b[1] + (u128)a[3] * b[0] + (u128)a[4] * b4_19;
    u128 t4 = (u128)a[0] * b[4] + (u128)a[1] * b[3] + (u128)a[2] * b[2] + (u128)a[3] * b[1] + (u128)a[4] * b[0];
    uint64_t c;
    t1 += (uint64_t)(t0 >> 51); h->v[0] = (uint64_t)t0 & MASK51;
    t2 += (uint64_t)(t1 >> 51); h->v[1] = (uint64_t)t1 & MASK51;
    t3 += (uint64_t)(t2 >> 51); h->v[2] = (uint64_t)t2 & MASK51;
    t4 += (uint64_t)(t3 >> 51); h->v[3] = (uint64_t)t3 & MASK51;
    c = (uint64_t)(t4 >> 51);   h->v[4] = (uint64_t)t4 & MASK51;
    h->v[0] += 19 * c;
    c = h->v[0] >> 51; h->v[0] &= MASK51; h->v[1] += c;
}

static void fe_sqr(fe *h, const fe *f) { fe_mul(h, f, f); }

static void fe_inv(fe *out, const fe *z) {
    /* z^(p-2), plain square-and-multiply over the bits of p-2 = 2^255-21 */
    fe r, base = *z;
    memset(&r, 0, sizeof r);
    r.v[0] = 1;
    /* p-2 in binary: bits 0..254; p-2 = 2^255 - 21 = ...11101011 */
    for (int i = 0; i < 255; i++) {
        int bit;
        if (i >= 5) bit = 1;
        else bit = ((0x0B >> i) & 1);   /* low 5 bits of (2^255-21) = 01011b */
        if (bit) fe_mul(&r, &r, &base);
        fe_sqr(&base, &base);
    }
    *out = r;
}

static const fe FE_D = {{0x34dca135978a3ULL, 0x1a8283b156ebdULL, 0x5e7a26001c029ULL, 0x739c663a03cbbULL, 0x52036cee2b6ffULL}};

/* add-2008-bbjlp, a = -1 (oracle/ed25519_ref.py pt_add) */
static void pt_add(pt *r, const pt *p, const pt *q) {
    fe A, B, C, D, E, F, G, s, t, u;
    fe_mul(&A, &p->Z, &q->Z);
    fe_sqr(&B, &A);
    fe_mul(&C, &p->X, &q->X);
    fe_mul(&D, &p->Y, &q->Y);
    fe_mul(&E, &FE_D, &C);
    fe_mul(&E, &E, &D);
    fe_sub(&F, &B, &E);
    fe_add(&G, &B, &E); fe_carry(&G);
    fe_add(&s, &p->X, &p->Y); fe_carry(&s);
    fe_add(&t, &q->X, &q->Y); fe_carry(&t);
    fe_mul(&u, &s, &t);
    fe_sub(&u, &u, &C);
    fe_sub(&u, &u, &D);
    fe_mul(&s, &A, &F);
    fe_mul(&r->X, &s, &u);
    fe_add(&t, &D, &C); fe_carry(&t);
    fe_mul(&s, &A, &G);
    fe_mul(&r->Y, &s, &t);
    fe_mul(&r->Z, &F, &G);
}

/* dbl-2008-bbjlp, a = -1 (oracle/ed25519_ref.py pt_dbl) */
static void pt_dbl(pt *r, const pt *p) {
    fe B, C, D, E, F, H, J, s, zero;
    memset(&zero, 0, sizeof zero);
    fe_add(&s, &p->X, &p->Y); fe_carry(&s);
    fe_sqr(&B, &s);
    fe_sqr(&C, &p->X);
    fe_sqr(&D, &p->Y);
    fe_sub(&E, &zero, &C);
    fe_add(&F, &E, &D); fe_carry(&F);
    fe_sqr(&H, &p->Z);
    fe_add(&s, &H, &H); fe_carry(&s);
    fe_sub(&J, &F, &s);
    fe_sub(&s, &B, &C);
    fe_sub(&s, &s, &D);
    fe_mul(&r->X, &s, &J);
    fe_sub(&s, &E, &D);
    fe_mul(&r->Y, &F, &s);
    fe_mul(&r->Z, &F, &J);
}

static void pt_identity(pt *r) {
    memset(r, 0, sizeof *r);
    r->Y.v[0] = 1;
    r->Z.v[0] = 1;
}

static int bit_length(const uint8_t n[32]) {
    for (int i = 31; i >= 0; i--)
        if (n[i]) {
            int b = 0;
            uint8_t v = n[i];
            while (v) { b++; v >>= 1; }
            return 8 * i + b;
        }
    return 0;
}

/* a ** n, n >= 0: right-to-left binary double-and-add (oracle pt_repeat) */
static void pt_repeat(pt *r, const pt *a, const uint8_t n[32]) {
    int bl = bit_length(n);
    if (bl == 0) { pt_identity(r); return; }
    pt d = *a, c, t;
    pt_identity(&c);
    for (int i = 0; i < bl - 1; i++) {
        if ((n[i >> 3] >> (i & 7)) & 1) { pt_add(&t, &c, &d); c = t; }
        pt_dbl(&t, &d); d = t;
    }
    pt_add(r, &c, &d);
}

static void pt_load_proj(pt *p, const uint8_t *s) {
    fe_frombytes(&p->X, s); fe_frombytes(&p->Y, s + 32); fe_frombytes(&p->Z, s + 64);
}
static void pt_load_affine(pt *p, const uint8_t *s) {
    fe_frombytes(&p->X, s); fe_frombytes(&p->Y, s + 32);
    memset(&p->Z, 0, sizeof p->Z); p->Z.v[0] = 1;
}
static void pt_store_proj(uint8_t *s, const pt *p) {
    fe_tobytes(s, &p->X); fe_tobytes(s + 32, &p->Y); fe_tobytes(s + 64, &p->Z);
}
static void pt_store_affine(uint8_t *s, const pt *p) {
    fe zi, x, y;
    fe_inv(&zi, &p->Z);
    fe_mul(&x, &p->X, &zi); fe_mul(&y, &p->Y, &zi);
    fe_tobytes(s, &x); fe_tobytes(s + 32, &y);
}

/* l and signed-exponent handling (pivot._int on a signed field element) */
static const uint8_t ELL[32] = {0xed,0xd3,0xf5,0x5c,0x1a,0x63,0x12,0x58,0xd6,0x9c,0xf7,0xa2,0xde,0xf9,0xde,0x14,
                                0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0x10};
static int gt_half_l(const uint8_t s[32]) {  /* 2s >= l */
    uint8_t d[33]; unsigned c = 0;
    for (int i = 0; i < 32; i++) { unsigned v = ((unsigned)s[i] << 1) | c; d[i] = (uint8_t)v; c = v >> 8; }
    d[32] = (uint8_t)c;
    if (d[32]) return 1;
    for (int i = 31; i >= 0; i--) { if (d[i] > ELL[i]) return 1; if (d[i] < ELL[i]) return 0; }
    return 1;
}
static void l_minus(uint8_t out[32], const uint8_t s[32]) {
    int borrow = 0;
    for (int i = 0; i < 32; i++) { int v = (int)ELL[i] - (int)s[i] - borrow; borrow = v < 0; out[i] = (uint8_t)(v & 0xff); }
}

/* ------------------------------------------------------------------------------------------ */
/* Threads (round 6).  The reference is single-threaded Python and `cpu_baseline` times this file with ONE
 * thread (the default).  The parity tests at BASELINE's own sizes (N = 2^20 proofs, the 2^24-term
 * commitment) spread the INDEPENDENT pieces of the same algorithm over the box's cores: the per-term ladders
 * of pivot.py:143, the element-wise fold of compressed_pivot.py:64, and the pair-products of one level of
 * the mpctools.reduce tree (pivot.py:26-28).  Operation order inside every piece, the tree's shape and hence
 * every (X:Y:Z) representative are those of the single-threaded run (tests/test_oracle_c.py compares the two
 * and oracle/ed25519_ref.py). */
#include <pthread.h>
static int g_threads = 1;
void oracle_set_threads(int n) { g_threads = n < 1 ? 1 : (n > 1024 ? 1024 : n); }
int oracle_get_threads(void) { return g_threads; }

typedef void (*range_fn)(size_t lo, size_t hi, void *arg);
typedef struct { range_fn fn; void *arg; size_t n, chunk, next; pthread_mutex_t mu; } pf_job;
static void *pf_worker(void *p) {
    pf_job *j = (pf_job *)p;
    for (;;) {
        pthread_mutex_lock(&j->mu);
        size_t lo = j->next;
        j->next = lo + j->chunk;
        pthread_mutex_unlock(&j->mu);
        if (lo >= j->n) return NULL;
        j->fn(lo, lo + j->chunk < j->n ? lo + j->chunk : j->n, j->arg);
    }
}
/* fn over [0, n) in chunks handed out first come first served; the calling thread works too */
static void parallel_for(size_t n, size_t chunk, range_fn fn, void *arg) {
    size_t chunks = (n + chunk - 1) / chunk;
    size_t t = (size_t)g_threads < chunks ? (size_t)g_threads : chunks;
    if (t <= 1) { if (n) fn(0, n, arg); return; }
    pf_job job = { fn, arg, n, chunk, 0, PTHREAD_MUTEX_INITIALIZER };
    pthread_t *th = (pthread_t *)malloc(t * sizeof(pthread_t));
    size_t started = 0;
    if (th)
        for (size_t i = 0; i + 1 < t; i++)
            if (pthread_create(&th[started], NULL, pf_worker, &job) == 0) started++;
    pf_worker(&job);
    for (size_t i = 0; i < started; i++) pthread_join(th[i], NULL);
    free(th);
}

/* g_i ** x_i for i in [lo, hi): the ladders of pivot.py:143 */
typedef struct { const uint8_t *x, *g; int proj_in, signed_exp; pt *terms; } vc_job;
static void vc_ladders(size_t lo, size_t hi, void *arg) {
    vc_job *j = (vc_job *)arg;
    for (size_t i = lo; i < hi; i++) {
        pt b;
        if (j->proj_in) pt_load_proj(&b, j->g + 96 * i); else pt_load_affine(&b, j->g + 64 * i);
        const uint8_t *s = j->x + 32 * i;
        uint8_t mag[32];
        if (j->signed_exp && gt_half_l(s)) {
            l_minus(mag, s);
            fe zero; memset(&zero, 0, sizeof zero);
            fe_sub(&b.X, &zero, &b.X);
            s = mag;
        }
        pt_repeat(&j->terms[i], &b, s);
    }
}
/* one level of the tree, out of place: dst[odd + p] = src[odd + 2p] * src[odd + 2p + 1] */
typedef struct { const pt *src; pt *dst; size_t odd; } lvl_job;
static void vc_level(size_t lo, size_t hi, void *arg) {
    lvl_job *j = (lvl_job *)arg;
    for (size_t p = lo; p < hi; p++)
        pt_add(&j->dst[j->odd + p], &j->src[j->odd + 2 * p], &j->src[j->odd + 2 * p + 1]);
}

/* out_proj (96 B) = h**gamma * reduce_tree([g_i ** x_i] + [identity]); points are projective
 * 96-byte inputs when proj_in != 0, else affine 64-byte.  signed_exp: treat residues > l/2 as
 * negative exponents (field-element inputs), as the reference does.  gamma_neg: `gamma` holds |gamma| of a
 * negative exponent (`a ** -n` inverts the base first, oracle/ed25519_ref.py pt_repeat). */
int oracle_vector_commitment_ex(const uint8_t *x, const uint8_t *gamma, int gamma_neg, const uint8_t *g,
                                const uint8_t *h, size_t n, int proj_in, int signed_exp, uint8_t *out_proj,
                                uint8_t *out_affine) {
    size_t len = n + 1;
    pt *terms = (pt *)malloc(len * sizeof(pt));
    pt *spare = (pt *)malloc(len * sizeof(pt));
    if (!terms || !spare) { free(terms); free(spare); return -12; }
    vc_job vj = { x, g, proj_in, signed_exp, terms };
    parallel_for(n, 64, vc_ladders, &vj);
    pt_identity(&terms[n]);
    while (len > 1) {           /* mpctools.reduce tree, initial appended at the end */
        size_t odd = len & 1, pairs = (len - odd) / 2;
        if (odd) spare[0] = terms[0];   /* an odd level keeps its first element in front */
        lvl_job lj = { terms, spare, odd };
        parallel_for(pairs, 256, vc_level, &lj);
        pt *sw = terms; terms = spare; spare = sw;
        len = odd + pairs;
    }
    pt hb, hg, res;
    if (proj_in) pt_load_proj(&hb, h); else pt_load_affine(&hb, h);
    if (gamma_neg) { fe zero; memset(&zero, 0, sizeof zero); fe_sub(&hb.X, &zero, &hb.X); }
    pt_repeat(&hg, &hb, gamma);
    pt_add(&res, &hg, &terms[0]);
    free(terms); free(spare);
    if (out_proj) pt_store_proj(out_proj, &res);
    if (out_affine) pt_store_affine(out_affine, &res);
    return 0;
}
int oracle_vector_commitment(const uint8_t *x, const uint8_t *gamma, const uint8_t *g, const uint8_t *h,
                             size_t n, int proj_in, int signed_exp, uint8_t *out_proj, uint8_t *out_affine) {
    return oracle_vector_commitment_ex(x, gamma, 0, g, h, n, proj_in, signed_exp, out_proj, out_affine);
}

/* g'_i = (g_l[i] ** c) * g_r[i]   (compressed_pivot.py:64,178) */
typedef struct { const uint8_t *gl, *gr, *c; int proj_in; uint8_t *out_proj, *out_affine; } fold_job;
static void fold_range(size_t lo, size_t hi, void *arg) {
    fold_job *j = (fold_job *)arg;
    for (size_t i = lo; i < hi; i++) {
        pt a, b, t, r;
        if (j->proj_in) { pt_load_proj(&a, j->gl + 96 * i); pt_load_proj(&b, j->gr + 96 * i); }
        else { pt_load_affine(&a, j->gl + 64 * i); pt_load_affine(&b, j->gr + 64 * i); }
        pt_repeat(&t, &a, j->c);
        pt_add(&r, &t, &b);
        if (j->out_proj) pt_store_proj(j->out_proj + 96 * i, &r);
        if (j->out_affine) pt_store_affine(j->out_affine + 64 * i, &r);
    }
}
int oracle_fold(const uint8_t *gl, const uint8_t *gr, const uint8_t *c, size_t half, int proj_in,
                uint8_t *out_proj, uint8_t *out_affine) {
    fold_job j = { gl, gr, c, proj_in, out_proj, out_affine };
    parallel_for(half, 64, fold_range, &j);
    return 0;
}

/* out_i = base ** r_i   (circuit_sat_r1cs.py:64-70) */
typedef struct { pt b; const uint8_t *exps; uint8_t *out_proj, *out_affine; } fb_job;
static void fb_range(size_t lo, size_t hi, void *arg) {
    fb_job *j = (fb_job *)arg;
    for (size_t i = lo; i < hi; i++) {
        pt r;
        pt_repeat(&r, &j->b, j->exps + 32 * i);
        if (j->out_proj) pt_store_proj(j->out_proj + 96 * i, &r);
        if (j->out_affine) pt_store_affine(j->out_affine + 64 * i, &r);
    }
}
int oracle_fixed_base(const uint8_t *base_proj, const uint8_t *exps, size_t n, uint8_t *out_proj,
                      uint8_t *out_affine) {
    fb_job j;
    pt_load_proj(&j.b, base_proj);
    j.exps = exps; j.out_proj = out_proj; j.out_affine = out_affine;
    parallel_for(n, 64, fb_range, &j);
    return 0;
}
