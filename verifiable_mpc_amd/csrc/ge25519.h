// Ed25519 group arithmetic for gfx950.
//
// Two coordinate systems live here, on purpose:
//
//  (1) "ext"  - extended twisted-Edwards coordinates (X:Y:Z:T), a = -1, with the
//      Hisil-Wong-Carter-Dawson formulas.  This is the throughput path used by the
//      Pippenger MSM (vmpc msm.hip) that replaces the per-term double-and-add of
//      verifiable_mpc/ac20/pivot.py:143-144.
//
//  (2) "proj" - projective (X:Y:Z) with EFD add-2008-bbjlp / dbl-2008-bbjlp and the
//      right-to-left binary `repeat`.  This REPLAYS, operation for operation, what the
//      reference computes through MPyC for `g ** n` and `a * b`
//      (verifiable_mpc/ac20/compressed_pivot.py:64 fold, circuit_sat_r1cs.py:64-70,81
//      generator setup), because the reference's Fiat-Shamir pre-image contains the
//      UN-normalised coordinates of those results (compressed_pivot.py:52).  Field
//      arithmetic is exact, so replaying the same formula sequence yields the same
//      residues.  [mpyc-recall: formulas as restated in oracle/ed25519_ref.py]
#pragma once
#include "fe25519.h"

struct ge_ext {   // extended: x = X/Z, y = Y/Z, T = XY/Z
    fe X, Y, Z, T;
};
struct ge_niels {  // affine point cached for mixed addition: (y-x, y+x, 2d*x*y)
    fe ymx, ypx, t2d;
};
struct ge_proj {  // projective, representative preserved
    fe X, Y, Z;
};
struct ge_aff {
    fe x, y;
};

VMPC_HD ge_ext ge_ext_identity() {
    ge_ext r;
    r.X = fe_zero();
    r.Y = fe_one();
    r.Z = fe_one();
    r.T = fe_zero();
    return r;
}

VMPC_HD ge_proj ge_proj_identity() {
    ge_proj r;
    r.X = fe_zero();
    r.Y = fe_one();
    r.Z = fe_one();
    return r;
}

VMPC_HD ge_niels ge_niels_from_affine(const ge_aff &a) {
    ge_niels r;
    r.ymx = fe_sub(a.y, a.x);
    r.ypx = fe_add(a.y, a.x);
    r.t2d = fe_mul(fe_mul(a.x, a.y), fe_const_d2());
    return r;
}

VMPC_HD ge_niels ge_niels_neg(const ge_niels &a) {
    ge_niels r;
    r.ymx = a.ypx;
    r.ypx = a.ymx;
    r.t2d = fe_neg(a.t2d);
    return r;
}

// t2d of the result may be a lazy negation (limbs < 2^27): it is only ever a product operand
VMPC_HD ge_niels ge_niels_select_neg(const ge_niels &a, bool neg) {
    ge_niels r;
    r.ymx = fe_select(a.ymx, a.ypx, neg);
    r.ypx = fe_select(a.ypx, a.ymx, neg);
    r.t2d = fe_select(a.t2d, fe_neg_lazy(a.t2d), neg);
    return r;
}

// The extended formulas below keep their sums and differences lazy (no carry pass) wherever the
// fe_mul operand contract allows; only F is carried.  Inputs: coordinates reduced (they are
// products, constants or unpacked values).  Limb bounds in comments are for even limbs.

// second half shared by madd / add: A, B, C reduced, D < 2^27.1
VMPC_HD ge_ext ge_hwcd_tail(const fe &A, const fe &B, const fe &C, const fe &D) {
    fe E = fe_sub_lazy(B, A);      // < 2^27.6
    fe H = fe_add_lazy(B, A);      // < 2^27.1
    fe F = fe_sub(D, C);           // carried: reduced
    fe G = fe_add_lazy(D, C);      // < 2^27.6
    ge_ext r;
    r.X = fe_mul(E, F);            // 2^27.6 * 2^26
    r.Y = fe_mul(G, H);            // 2^27.6 * 2^27.1 = 2^54.7
    r.T = fe_mul(E, H);            // 2^54.7
    r.Z = fe_mul(G, F);
    return r;
}

// mixed addition ext + niels (madd-2008-hwcd-3 shape): 7M.  q.t2d may be a lazy negation.
VMPC_HD ge_ext ge_madd(const ge_ext &p, const ge_niels &q) {
    fe A = fe_mul(fe_sub_lazy(p.Y, p.X), q.ymx);     // 2^27.6 * 2^26
    fe B = fe_mul(fe_add_lazy(p.Y, p.X), q.ypx);     // 2^27.1 * 2^26
    fe C = fe_mul(p.T, q.t2d);                       // 2^26 * 2^27
    fe D = fe_add_lazy(p.Z, p.Z);
    return ge_hwcd_tail(A, B, C, D);
}

// full addition ext + ext (add-2008-hwcd-3): 9M.  Complete on Ed25519 (a = -1 square... d non-square).
VMPC_HD ge_ext ge_add(const ge_ext &p, const ge_ext &q) {
    fe A = fe_mul(fe_sub_lazy(p.Y, p.X), fe_sub_lazy(q.Y, q.X));   // 2^27.6 * 2^27.6 = 2^55.2
    fe B = fe_mul(fe_add_lazy(p.Y, p.X), fe_add_lazy(q.Y, q.X));   // 2^27.1 * 2^27.1
    fe C = fe_mul(fe_mul(p.T, q.T), fe_const_d2());
    fe zz = fe_mul(p.Z, q.Z);
    fe D = fe_add_lazy(zz, zz);
    return ge_hwcd_tail(A, B, C, D);
}

// doubling (dbl-2008-hwcd, a = -1): 4M + 4S
VMPC_HD ge_ext ge_dbl(const ge_ext &p) {
    fe A = fe_sqr(p.X);
    fe B = fe_sqr(p.Y);
    fe zz = fe_sqr(p.Z);
    fe S = fe_sqr(fe_add_lazy(p.X, p.Y));            // 2^27.1
    fe H = fe_add_lazy(A, B);                        // < 2^27.1
    fe E = fe_sub_lazy(H, S);                        // < 2^28
    fe G = fe_sub_lazy(A, B);                        // < 2^27.6
    fe F = fe_add(fe_add_lazy(zz, zz), G);           // carried: reduced  (2^27.1 + 2^27.6 < 2^31)
    ge_ext r;
    r.X = fe_mul(E, F);                              // 2^28 * 2^26
    r.Y = fe_mul(G, H);                              // 2^54.7
    r.T = fe_mul(E, H);                              // 2^28 * 2^27.1 = 2^55.1
    r.Z = fe_mul(G, F);
    return r;
}

VMPC_HD ge_ext ge_ext_neg(const ge_ext &p) {
    ge_ext r = p;
    r.X = fe_neg(p.X);
    r.T = fe_neg(p.T);
    return r;
}

VMPC_HD ge_ext ge_ext_select(const ge_ext &a, const ge_ext &b, bool pick_b) {
    ge_ext r;
    r.X = fe_select(a.X, b.X, pick_b);
    r.Y = fe_select(a.Y, b.Y, pick_b);
    r.Z = fe_select(a.Z, b.Z, pick_b);
    r.T = fe_select(a.T, b.T, pick_b);
    return r;
}

VMPC_HD ge_ext ge_ext_from_affine(const ge_aff &a) {
    ge_ext r;
    r.X = a.x;
    r.Y = a.y;
    r.Z = fe_one();
    r.T = fe_mul(a.x, a.y);
    return r;
}

VMPC_HD ge_aff ge_ext_to_affine(const ge_ext &p) {
    fe zi = fe_inv(p.Z);
    ge_aff r;
    r.x = fe_canon(fe_mul(p.X, zi));
    r.y = fe_canon(fe_mul(p.Y, zi));
    return r;
}

VMPC_HD bool ge_aff_on_curve(const ge_aff &a) {
    // -x^2 + y^2 = 1 + d x^2 y^2
    fe x2 = fe_sqr(a.x), y2 = fe_sqr(a.y);
    fe lhs = fe_sub(y2, x2);
    fe rhs = fe_add(fe_one(), fe_mul(fe_const_d(), fe_mul(x2, y2)));
    return fe_eq(lhs, rhs);
}

// ---------------- projective replay of the reference's formulas ----------------------

// add-2008-bbjlp, a = -1 (oracle/ed25519_ref.py pt_add).  Sums stay lazy where the fe_mul operand
// contract allows - the residues, which is all the replay has to reproduce, are unchanged.
VMPC_HD ge_proj ge_proj_add(const ge_proj &p, const ge_proj &q) {
    fe A = fe_mul(p.Z, q.Z);
    fe B = fe_sqr(A);
    fe C = fe_mul(p.X, q.X);
    fe D = fe_mul(p.Y, q.Y);
    fe E = fe_mul(fe_mul(fe_const_d(), C), D);
    fe F = fe_sub_lazy(B, E);                         // < 2^27.6
    fe G = fe_add_lazy(B, E);                         // < 2^27.1
    fe DC = fe_add_lazy(D, C);                        // < 2^27.1
    fe s = fe_sub(fe_mul(fe_add_lazy(p.X, p.Y), fe_add_lazy(q.X, q.Y)), DC);   // carried
    ge_proj r;
    r.X = fe_mul(fe_mul(A, F), s);
    r.Y = fe_mul(fe_mul(A, G), DC);
    r.Z = fe_mul(F, G);                               // 2^27.6 * 2^27.1
    return r;
}

// dbl-2008-bbjlp, a = -1 (oracle/ed25519_ref.py pt_dbl)
VMPC_HD ge_proj ge_proj_dbl(const ge_proj &p) {
    fe B = fe_sqr(fe_add_lazy(p.X, p.Y));
    fe C = fe_sqr(p.X);
    fe D = fe_sqr(p.Y);
    fe F = fe_sub_lazy(D, C);                         // E + D with E = -C;  < 2^27.6
    fe H = fe_sqr(p.Z);
    fe J = fe_sub(F, fe_add_lazy(H, H));              // carried
    fe nCD = fe_neg(fe_add_lazy(C, D));               // E - D, carried
    ge_proj r;
    r.X = fe_mul(fe_sub_lazy(fe_sub_lazy(B, C), D), J);   // B - C - D < 2^28.4 against a reduced J
    r.Y = fe_mul(F, nCD);
    r.Z = fe_mul(F, J);
    return r;
}

VMPC_HD ge_proj ge_proj_neg(const ge_proj &p) {
    ge_proj r = p;
    r.X = fe_neg(p.X);
    return r;
}

VMPC_HD ge_proj ge_proj_select(const ge_proj &a, const ge_proj &b, bool pick_b) {
    ge_proj r;
    r.X = fe_select(a.X, b.X, pick_b);
    r.Y = fe_select(a.Y, b.Y, pick_b);
    r.Z = fe_select(a.Z, b.Z, pick_b);
    return r;
}

VMPC_HD ge_proj ge_proj_canon(const ge_proj &p) {
    ge_proj r;
    r.X = fe_canon(p.X);
    r.Y = fe_canon(p.Y);
    r.Z = fe_canon(p.Z);
    return r;
}

VMPC_HD ge_aff ge_proj_to_affine(const ge_proj &p) {
    fe zi = fe_inv(p.Z);
    ge_aff r;
    r.x = fe_canon(fe_mul(p.X, zi));
    r.y = fe_canon(fe_mul(p.Y, zi));
    return r;
}

VMPC_HD ge_ext ge_ext_from_proj(const ge_proj &p) {
    // (X:Y:Z) -> (XZ : YZ : Z^2 : XY)
    ge_ext r;
    r.X = fe_mul(p.X, p.Z);
    r.Y = fe_mul(p.Y, p.Z);
    r.Z = fe_sqr(p.Z);
    r.T = fe_mul(p.X, p.Y);
    return r;
}

VMPC_HD int u256_bit_length(const uint32_t s[8]) {
    int bl = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (s[i]) {
            uint32_t v = s[i];
            int b = 0;
            while (v) {
                b++;
                v >>= 1;
            }
            bl = 32 * i + b;
        }
    }
    return bl;
}

// `a ** n` for 0 <= n < 2^256 given as 8 LE limbs: right-to-left binary double-and-add,
// exactly the operation sequence of oracle/ed25519_ref.py pt_repeat (n >= 0 branch).
// Lane-divergent n is handled by predication so that every lane replays ITS OWN
// sequence (the selects do not change values).
VMPC_HD ge_proj ge_proj_repeat(const ge_proj &a, const uint32_t n[8]) {
    int bl = u256_bit_length(n);
    if (bl == 0) return ge_proj_identity();
    ge_proj d = a;
    ge_proj c = ge_proj_identity();
    // word loop unrolled so that n[] is only indexed statically (stays in registers)
#pragma unroll
    for (int w = 0; w < 8; w++) {
        uint32_t word = n[w];
        int base = 32 * w;
        if (base >= bl - 1) break;
        int cnt = bl - 1 - base;
        if (cnt > 32) cnt = 32;
        for (int b = 0; b < cnt; b++) {
            if ((word >> b) & 1u) c = ge_proj_add(c, d);
            d = ge_proj_dbl(d);
        }
    }
    return ge_proj_add(c, d);
}
