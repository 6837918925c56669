/* ORACLE-SIDE BASELINE (test / bench infrastructure, NOT the product): a "strong CPU" Pedersen commitment.
 *
 * The reference computes  h^gamma * prod g_i^{x_i}  (verifiable_mpc/ac20/pivot.py:139-145) as n independent
 * double-and-add ladders in Python; oracle/ed25519_oracle.c restates exactly that.  This file answers a
 * different question - how fast can the host's cores produce the SAME GROUP ELEMENT with the algorithm class
 * the GPU uses - so that the GPU/CPU ratio in bench.py's cpu_baseline has an honest denominator:
 *
 *   field      GF(2^255-19), 5 limbs of 51 bits, unsigned __int128 products
 *   points     extended twisted-Edwards (a = -1), mixed addition against (y-x, y+x, 2dxy): 7M
 *   algorithm  Pippenger with signed c-bit windows per thread over a contiguous slice of the terms
 *              (pthreads, one slice per core), partial sums added at the end
 *
 * Only bench.py's cpu_baseline leg and tests/ may use it (tests/test_oracle_cpu_pippenger.py checks it
 * against oracle/ed25519_ref.py).  Plain C, gcc -O2 -pthread.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t v[5]; } fe;
typedef struct { fe X, Y, Z, T; } ge;
typedef struct { fe ymx, ypx, t2d; } niels;

#define M51 0x7ffffffffffffULL

static const fe FE_D2 = {{0x69b9426b2f159ULL, 0x35050762add7aULL, 0x3cf44c0038052ULL, 0x6738cc7407977ULL,
                          0x2406d9dc56dffULL}};

static void fe_0(fe *r) { memset(r, 0, sizeof *r); }
static void fe_1(fe *r) { fe_0(r); r->v[0] = 1; }

static void fe_frombytes(fe *r, const uint8_t *s) {
    uint64_t w[4];
    memcpy(w, s, 32);
    r->v[0] = w[0] & M51;
    r->v[1] = ((w[0] >> 51) | (w[1] << 13)) & M51;
    r->v[2] = ((w[1] >> 38) | (w[2] << 26)) & M51;
    r->v[3] = ((w[2] >> 25) | (w[3] << 39)) & M51;
    r->v[4] = (w[3] >> 12) & M51;
}

static void fe_carry(fe *r) {
    uint64_t c;
    for (int k = 0; k < 2; k++) {
        c = r->v[0] >> 51; r->v[0] &= M51; r->v[1] += c;
        c = r->v[1] >> 51; r->v[1] &= M51; r->v[2] += c;
        c = r->v[2] >> 51; r->v[2] &= M51; r->v[3] += c;
        c = r->v[3] >> 51; r->v[3] &= M51; r->v[4] += c;
        c = r->v[4] >> 51; r->v[4] &= M51; r->v[0] += 19 * c;
    }
}

static void fe_tobytes(uint8_t *s, const fe *a) {
    fe t = *a;
    fe_carry(&t);
    /* canonical: add 19, see if it overflows 2^255 */
    uint64_t q = (t.v[0] + 19) >> 51;
    q = (t.v[1] + q) >> 51; q = (t.v[2] + q) >> 51; q = (t.v[3] + q) >> 51; q = (t.v[4] + q) >> 51;
    t.v[0] += 19 * q;
    uint64_t c;
    c = t.v[0] >> 51; t.v[0] &= M51; t.v[1] += c;
    c = t.v[1] >> 51; t.v[1] &= M51; t.v[2] += c;
    c = t.v[2] >> 51; t.v[2] &= M51; t.v[3] += c;
    c = t.v[3] >> 51; t.v[3] &= M51; t.v[4] += c;
    t.v[4] &= M51;
    uint64_t w[4];
    w[0] = t.v[0] | (t.v[1] << 51);
    w[1] = (t.v[1] >> 13) | (t.v[2] << 38);
    w[2] = (t.v[2] >> 26) | (t.v[3] << 25);
    w[3] = (t.v[3] >> 39) | (t.v[4] << 12);
    memcpy(s, w, 32);
}

static inline void fe_add(fe *r, const fe *a, const fe *b) {
    for (int i = 0; i < 5; i++) r->v[i] = a->v[i] + b->v[i];
}
/* a - b + 2p (inputs below 2^52) */
static inline void fe_sub(fe *r, const fe *a, const fe *b) {
    r->v[0] = a->v[0] + 0xfffffffffffdaULL - b->v[0];
    for (int i = 1; i < 5; i++) r->v[i] = a->v[i] + 0xffffffffffffeULL - b->v[i];
}

static inline void fe_mul(fe *r, const fe *a, const fe *b) {
    const uint64_t a0 = a->v[0], a1 = a->v[1], a2 = a->v[2], a3 = a->v[3], a4 = a->v[4];
    const uint64_t b0 = b->v[0], b1 = b->v[1], b2 = b->v[2], b3 = b->v[3], b4 = b->v[4];
    const uint64_t b1_19 = 19 * b1, b2_19 = 19 * b2, b3_19 = 19 * b3, b4_19 = 19 * b4;
    u128 t0 = (u128)a0 * b0 + (u128)a1 * b4_19 + (u128)a2 * b3_19 + (u128)a3 * b2_19 + (u128)a4 * b1_19;
    u128 t1 = (u128)a0 * b1 + (u128)a1 * b0 + (u128)a2 * b4_19 + (u128)a3 * b3_19 + (u128)a4 * b2_19;
    u128 t2 = (u128)a0 * b2 + (u128)a1 * b1 + (u128)a2 * b0 + (u128)a3 * b4_19 + (u128)a4 * b3_19;
    u128 t3 = (u128)a0 * b3 + (u128)a1 * b2 + (u128)a2 * b1 + (u128)a3 * b0 + (u128)a4 * b4_19;
    u128 t4 = (u128)a0 * b4 + (u128)a1 * b3 + (u128)a2 * b2 + (u128)a3 * b1 + (u128)a4 * b0;
    uint64_t c;
    t1 += (uint64_t)(t0 >> 51); r->v[0] = (uint64_t)t0 & M51;
    t2 += (uint64_t)(t1 >> 51); r->v[1] = (uint64_t)t1 & M51;
    t3 += (uint64_t)(t2 >> 51); r->v[2] = (uint64_t)t2 & M51;
    t4 += (uint64_t)(t3 >> 51); r->v[3] = (uint64_t)t3 & M51;
    c = (uint64_t)(t4 >> 51); r->v[4] = (uint64_t)t4 & M51;
    r->v[0] += 19 * c;
    c = r->v[0] >> 51; r->v[0] &= M51; r->v[1] += c;
}
static inline void fe_sq(fe *r, const fe *a) { fe_mul(r, a, a); }

static void fe_invert(fe *r, const fe *z) {
    /* z^(p-2), plain square-and-multiply over the fixed exponent 2^255 - 21 */
    fe acc, base = *z;
    fe_1(&acc);
    /* exponent bits: 2^255 - 21 = ...: low bits of (p - 2) = 0b...11101011 */
    uint8_t e[32];
    memset(e, 0xff, 32);
    e[0] = 0xeb;
    e[31] = 0x7f;
    for (int i = 254; i >= 0; i--) {
        fe_sq(&acc, &acc);
        if ((e[i >> 3] >> (i & 7)) & 1) fe_mul(&acc, &acc, &base);
    }
    *r = acc;
}

static void ge_identity(ge *p) { fe_0(&p->X); fe_1(&p->Y); fe_1(&p->Z); fe_0(&p->T); }

/* p += q (mixed, q = (y-x, y+x, 2dxy)), optionally negated */
static inline void ge_madd(ge *p, const niels *q, int neg) {
    fe a, b, c, d, e, f, g, h, t;
    const fe *qm = neg ? &q->ypx : &q->ymx, *qp = neg ? &q->ymx : &q->ypx;
    fe_sub(&t, &p->Y, &p->X); fe_carry(&t); fe_mul(&a, &t, qm);
    fe_add(&t, &p->Y, &p->X); fe_mul(&b, &t, qp);
    fe_mul(&c, &p->T, &q->t2d);
    fe_add(&d, &p->Z, &p->Z);
    if (neg) { fe_add(&f, &d, &c); fe_sub(&g, &d, &c); }      /* -q: C -> -C */
    else     { fe_sub(&f, &d, &c); fe_add(&g, &d, &c); }
    fe_sub(&e, &b, &a);
    fe_add(&h, &b, &a);
    fe_carry(&e); fe_carry(&f); fe_carry(&g); fe_carry(&h);
    fe_mul(&p->X, &e, &f);
    fe_mul(&p->Y, &g, &h);
    fe_mul(&p->T, &e, &h);
    fe_mul(&p->Z, &g, &f);
}

static inline void ge_add(ge *p, const ge *q) {
    fe a, b, c, d, e, f, g, h, t, u;
    fe_sub(&t, &p->Y, &p->X); fe_sub(&u, &q->Y, &q->X); fe_carry(&t); fe_carry(&u); fe_mul(&a, &t, &u);
    fe_add(&t, &p->Y, &p->X); fe_add(&u, &q->Y, &q->X); fe_mul(&b, &t, &u);
    fe_mul(&c, &p->T, &q->T); fe_mul(&c, &c, &FE_D2);
    fe_mul(&d, &p->Z, &q->Z); fe_add(&d, &d, &d);
    fe_sub(&e, &b, &a); fe_sub(&f, &d, &c); fe_add(&g, &d, &c); fe_add(&h, &b, &a);
    fe_carry(&e); fe_carry(&f); fe_carry(&g); fe_carry(&h);
    fe_mul(&p->X, &e, &f);
    fe_mul(&p->Y, &g, &h);
    fe_mul(&p->T, &e, &h);
    fe_mul(&p->Z, &g, &f);
}

static inline void ge_dbl(ge *p) {
    fe a, b, c, e, f, g, h, t;
    fe_sq(&a, &p->X);
    fe_sq(&b, &p->Y);
    fe_sq(&c, &p->Z); fe_add(&c, &c, &c);
    fe_add(&h, &a, &b);
    fe_add(&t, &p->X, &p->Y); fe_sq(&t, &t);
    fe_sub(&e, &h, &t);                 /* E = H - (X+Y)^2 = -2XY */
    fe_sub(&g, &a, &b);                 /* G = A - B */
    fe_carry(&g);
    fe_add(&f, &c, &g);                 /* F = C + G */
    fe_carry(&e); fe_carry(&f); fe_carry(&h);
    fe_mul(&p->X, &e, &f);
    fe_mul(&p->Y, &g, &h);
    fe_mul(&p->T, &e, &h);
    fe_mul(&p->Z, &g, &f);
}

struct job {
    const uint8_t *scalars, *points;   /* n x 32 LE scalars < 2^253, n x 64 affine x||y */
    size_t n;
    int c;
    ge out;
};

static void *run_slice(void *arg) {
    struct job *j = (struct job *)arg;
    const size_t n = j->n;
    const int c = j->c, W = (253 + 1 + c - 1) / c, nb = 1 << (c - 1);
    niels *pts = (niels *)malloc(n * sizeof(niels));
    int32_t *digits = (int32_t *)malloc(n * (size_t)W * sizeof(int32_t));
    ge *buckets = (ge *)malloc((size_t)nb * sizeof(ge));
    uint8_t *used = (uint8_t *)malloc(nb);
    for (size_t i = 0; i < n; i++) {
        fe x, y, t;
        fe_frombytes(&x, j->points + 64 * i);
        fe_frombytes(&y, j->points + 64 * i + 32);
        fe_sub(&pts[i].ymx, &y, &x); fe_carry(&pts[i].ymx);
        fe_add(&pts[i].ypx, &y, &x); fe_carry(&pts[i].ypx);
        fe_mul(&t, &x, &y); fe_mul(&pts[i].t2d, &t, &FE_D2);
        /* signed c-bit digits */
        uint64_t w[5] = {0, 0, 0, 0, 0};
        memcpy(w, j->scalars + 32 * i, 32);
        uint32_t carry = 0;
        for (int k = 0; k < W; k++) {
            const int bit = k * c;
            uint64_t raw = (w[bit >> 6] >> (bit & 63));
            if ((bit & 63) + c > 64) raw |= w[(bit >> 6) + 1] << (64 - (bit & 63));
            raw = (raw & ((1u << c) - 1)) + carry;
            int32_t d = (int32_t)raw;
            carry = 0;
            if (raw >= (uint32_t)nb + (k == W - 1 ? nb : 0)) { d -= 1 << c; carry = 1; }
            digits[i * W + k] = d;
        }
    }
    ge acc;
    ge_identity(&acc);
    for (int k = W - 1; k >= 0; k--) {
        for (int s = 0; s < c; s++) ge_dbl(&acc);
        memset(used, 0, nb);
        for (size_t i = 0; i < n; i++) {
            const int32_t d = digits[i * W + k];
            if (!d) continue;
            const int b = (d < 0 ? -d : d) - 1;
            if (!used[b]) { ge_identity(&buckets[b]); used[b] = 1; }
            ge_madd(&buckets[b], &pts[i], d < 0);
        }
        ge run, sum;
        ge_identity(&run);
        ge_identity(&sum);
        int any = 0;
        for (int b = nb - 1; b >= 0; b--) {
            if (used[b]) { ge_add(&run, &buckets[b]); any = 1; }
            if (any) ge_add(&sum, &run);
        }
        ge_add(&acc, &sum);
    }
    j->out = acc;
    free(pts); free(digits); free(buckets); free(used);
    return NULL;
}

/* out[64] = affine x||y of sum_i scalars[i] * points[i];  threads >= 1, window c in 4..16 (0: pick) */
int cpu_pippenger_msm(const uint8_t *scalars, const uint8_t *points, size_t n, int threads, int c, uint8_t *out) {
    if (threads < 1) threads = 1;
    if ((size_t)threads > n) threads = n ? (int)n : 1;
    struct job *jobs = (struct job *)calloc(threads, sizeof(struct job));
    pthread_t *tid = (pthread_t *)calloc(threads, sizeof(pthread_t));
    const size_t per = (n + threads - 1) / threads;
    for (int t = 0; t < threads; t++) {
        size_t lo = (size_t)t * per, hi = lo + per > n ? n : lo + per;
        if (lo > n) lo = hi = n;
        jobs[t].scalars = scalars + 32 * lo;
        jobs[t].points = points + 64 * lo;
        jobs[t].n = hi - lo;
        int cc = c;
        if (cc <= 0) {          /* ~ log2(slice) - 3, the usual optimum of W * (n + 2^c) */
            cc = 4;
            while (cc < 13 && ((size_t)1 << (cc + 3)) < jobs[t].n) cc++;   /* beyond 13 the buckets leave the L2 */
        }
        jobs[t].c = cc;
        if (threads == 1) run_slice(&jobs[t]);
        else pthread_create(&tid[t], NULL, run_slice, &jobs[t]);
    }
    ge acc;
    ge_identity(&acc);
    for (int t = 0; t < threads; t++) {
        if (threads > 1) pthread_join(tid[t], NULL);
        ge_add(&acc, &jobs[t].out);
    }
    fe zi, x, y;
    fe_invert(&zi, &acc.Z);
    fe_mul(&x, &acc.X, &zi);
    fe_mul(&y, &acc.Y, &zi);
    fe_tobytes(out, &x);
    fe_tobytes(out + 32, &y);
    free(jobs); free(tid);
    return 0;
}
