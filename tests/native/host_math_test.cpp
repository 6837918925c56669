// Host-side harness for the VMPC_HD math headers (fe25519 / ge25519 / fr / fmt).
// Built with g++ by tests/test_native_host_math.py; reads one command per line on
// stdin (hex operands, little-endian 32-byte values written as big-endian hex ints)
// and prints the result, so the Python oracle can check the same device source on CPU.
#include <cstdio>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#define VMPC_HD inline
#include "../../verifiable_mpc_amd/csrc/fe25519.h"
#include "../../verifiable_mpc_amd/csrc/ge25519.h"
#include "../../verifiable_mpc_amd/csrc/fr.h"
#include "../../verifiable_mpc_amd/csrc/fmt.h"
#include "../../verifiable_mpc_amd/csrc/sw256.h"
#include "../../verifiable_mpc_amd/csrc/fp29.h"
#include "../../verifiable_mpc_amd/csrc/fe51_host.h"

static void parse_hex(const std::string &h, uint32_t *out, int limbs) {
    for (int i = 0; i < limbs; i++) out[i] = 0;
    int nib = 0;
    for (int i = (int)h.size() - 1; i >= 0; i--, nib++) {
        char c = h[i];
        uint32_t v = (c >= '0' && c <= '9') ? c - '0' : (c >= 'a' && c <= 'f') ? c - 'a' + 10 : c - 'A' + 10;
        if (nib / 8 < limbs) out[nib / 8] |= v << (4 * (nib % 8));
    }
}
static std::string to_hex(const uint32_t *v, int limbs) {
    char buf[16];
    std::string s;
    bool lead = true;
    for (int i = limbs - 1; i >= 0; i--) {
        if (lead && v[i] == 0 && i > 0) continue;
        snprintf(buf, sizeof buf, lead ? "%x" : "%08x", v[i]);
        s += buf;
        lead = false;
    }
    return s;
}
static fe rd_fe(std::istringstream &is) {
    std::string h;
    is >> h;
    uint32_t w[8];
    parse_hex(h, w, 8);
    return fe_unpack(w);
}
static void rd_raw8(std::istringstream &is, uint32_t w[8]) {
    std::string h;
    is >> h;
    parse_hex(h, w, 8);
}
static fr rd_fr(std::istringstream &is) {
    std::string h;
    is >> h;
    fr r;
    parse_hex(h, r.v, 8);
    return r;
}
static std::string fehex(const fe &a) {
    fe8 c = fe_pack(a);
    return to_hex(c.w, 8);
}
static std::string frhex(const fr &a) { return to_hex(a.v, 8); }

// ---- BN-256 helpers: operands are canonical integers, converted to Montgomery form inside
static fp rd_fp(std::istringstream &is) {
    std::string h;
    is >> h;
    uint32_t w[8];
    parse_hex(h, w, 8);
    return Fp1Ops::load(w);
}
static std::string fphex(const fp &a) {
    uint32_t w[8];
    Fp1Ops::store(w, a);
    return to_hex(w, 8);
}
static fp2 rd_fp2(std::istringstream &is) {
    fp2 r;
    r.a = rd_fp(is);
    r.b = rd_fp(is);
    return r;
}
static std::string fp2hex(const fp2 &a) { return fphex(a.a) + " " + fphex(a.b); }
template <class F> static typename F::elem rd_el(std::istringstream &is);
template <> fp rd_el<Fp1Ops>(std::istringstream &is) { return rd_fp(is); }
template <> fp2 rd_el<Fp2Ops>(std::istringstream &is) { return rd_fp2(is); }
static std::string elhex(const fp &a) { return fphex(a); }
static std::string elhex(const fp2 &a) { return fp2hex(a); }
// the latency field (fp29.h): canonical integers in and out through its own load / store
static fp29 rd_fp29(std::istringstream &is) {
    std::string h;
    is >> h;
    uint32_t w[8];
    parse_hex(h, w, 8);
    return Fp29Ops::load(w);
}
static std::string fp29hex(const fp29 &a) {
    uint32_t w[8];
    Fp29Ops::store(w, a);
    return to_hex(w, 8);
}
template <> fp29 rd_el<Fp29Ops>(std::istringstream &is) { return rd_fp29(is); }
template <> fp29x2 rd_el<Fp29x2Ops>(std::istringstream &is) {
    fp29x2 r;
    r.a = rd_fp29(is);
    r.b = rd_fp29(is);
    return r;
}
static std::string elhex(const fp29 &a) { return fp29hex(a); }
static std::string elhex(const fp29x2 &a) { return fp29hex(a.a) + " " + fp29hex(a.b); }

template <class F> static aff<F> rd_aff(std::istringstream &is) {
    aff<F> r;
    r.x = rd_el<F>(is);
    r.y = rd_el<F>(is);
    r.inf = F::is_zero(r.x) && F::is_zero(r.y);
    return r;
}
template <class F> static std::string affhex(const aff<F> &a) {
    if (a.inf) return "inf";
    return elhex(a.x) + " " + elhex(a.y);
}
template <class F> static jac<F> lift(const aff<F> &a) {
    jac<F> r = jac_identity<F>();
    return jac_madd<F>(r, a);
}
// curve commands: add (general, via de-normalised operands), madd, dbl, mul k
template <class F> static void curve_cmd(const std::string &op, std::istringstream &is) {
    aff<F> a = rd_aff<F>(is);
    if (op == "dbl") {
        std::cout << affhex<F>(jac_to_affine<F>(jac_dbl<F>(lift<F>(a)))) << "\n";
        return;
    }
    if (op == "mul") {
        std::string h;
        is >> h;
        uint32_t k[8];
        parse_hex(h, k, 8);
        jac<F> acc = jac_identity<F>(), d = lift<F>(a);
        for (int i = 0; i < 256; i++) {
            if ((k[i >> 5] >> (i & 31)) & 1u) acc = jac_add<F>(acc, d);
            d = jac_dbl<F>(d);
        }
        std::cout << affhex<F>(jac_to_affine<F>(acc)) << "\n";
        return;
    }
    aff<F> b = rd_aff<F>(is);
    jac<F> pa = jac_dbl<F>(lift<F>(a));           // 2a, Z != 1
    if (op == "madd") {
        std::cout << affhex<F>(jac_to_affine<F>(jac_madd<F>(pa, b))) << "\n";
    } else {
        jac<F> pb = jac_add<F>(jac_dbl<F>(lift<F>(b)), jac_identity<F>());   // 2b
        std::cout << affhex<F>(jac_to_affine<F>(jac_add<F>(pa, pb))) << "\n";
    }
}

int main() {
    std::string line;
    while (std::getline(std::cin, line)) {
        std::istringstream is(line);
        std::string cmd;
        is >> cmd;
        if (cmd == "h51lin3") {
            // A + c Q + c^2 B on the host curve code of fe51_host.h: six coordinates and c in, affine x y out
            uint32_t w[7][8];
            for (int i = 0; i < 7; i++) rd_raw8(is, w[i]);
            uint8_t pts[3][64];
            for (int i = 0; i < 3; i++) {
                memcpy(pts[i], w[2 * i], 32);
                memcpy(pts[i] + 32, w[2 * i + 1], 32);
            }
            const fr cf = fr_load(w[6]);
            uint8_t c2[32];
            fr_store((uint32_t *)c2, fr_mul(cf, cf));
            const fe51::el dd = fe51::d2();
            const fe51::pt a = fe51::pt_from_affine(pts[0]), q = fe51::pt_from_affine(pts[1]), b = fe51::pt_from_affine(pts[2]);
            const fe51::pt r = fe51::pt_add(fe51::pt_add(a, fe51::pt_mul(q, (const uint8_t *)w[6], dd), dd),
                                            fe51::pt_mul(b, c2, dd), dd);
            uint8_t out[64];
            fe51::pt_to_affine(out, r);
            uint32_t ox[8], oy[8];
            memcpy(ox, out, 32);
            memcpy(oy, out + 32, 32);
            std::cout << to_hex(ox, 8) << " " << to_hex(oy, 8) << "\n";
        } else if (cmd == "h51mul" || cmd == "h51inv") {
            // the host-only 51-bit-limb field (fe51_host.h): bytes in, canonical bytes out
            uint32_t wa[8], wb[8] = {1, 0, 0, 0, 0, 0, 0, 0};
            rd_raw8(is, wa);
            if (cmd == "h51mul") rd_raw8(is, wb);
            const fe51::el a = fe51::from_bytes((const uint8_t *)wa), b = fe51::from_bytes((const uint8_t *)wb);
            uint8_t out[32];
            fe51::to_bytes(out, cmd == "h51mul" ? fe51::mul(a, b) : fe51::inv(a));
            uint32_t wo[8];
            memcpy(wo, out, 32);
            std::cout << to_hex(wo, 8) << "\n";
        } else if (cmd == "femul") {
            fe a = rd_fe(is), b = rd_fe(is);
            std::cout << fehex(fe_mul(a, b)) << "\n";
        } else if (cmd == "fesqr") {
            fe a = rd_fe(is);
            std::cout << fehex(fe_sqr(a)) << "\n";
        } else if (cmd == "feadd") {
            fe a = rd_fe(is), b = rd_fe(is);
            std::cout << fehex(fe_add(a, b)) << "\n";
        } else if (cmd == "fesub") {
            fe a = rd_fe(is), b = rd_fe(is);
            std::cout << fehex(fe_sub(a, b)) << "\n";
        } else if (cmd == "femulu32") {
            fe a = rd_fe(is);
            uint32_t sw[8];
            rd_raw8(is, sw);
            std::cout << fehex(fe_mul_u32(a, sw[0])) << "\n";
        } else if (cmd == "feinv") {
            fe a = rd_fe(is);
            std::cout << fehex(fe_inv(a)) << "\n";
        } else if (cmd == "fecanon") {
            uint32_t w[8];
            rd_raw8(is, w);
            std::cout << fehex(fe_unpack(w)) << " " << (fe8_is_canonical(w) ? 1 : 0) << "\n";
        } else if (cmd == "felazy") {
            // (a + b) * (c + d) with lazy sums, and (a+b)^2: the operand bound of fe_mul / fe_sqr
            fe a = rd_fe(is), b = rd_fe(is), c = rd_fe(is), d = rd_fe(is);
            std::cout << fehex(fe_mul(fe_add_lazy(a, b), fe_add_lazy(c, d))) << " "
                      << fehex(fe_sqr(fe_add_lazy(a, b))) << " " << fehex(fe_sub(fe_add_lazy(a, b), fe_add_lazy(c, d)))
                      << "\n";
        } else if (cmd == "rawmul") {
            // fe_mul / fe_sqr on RAW limbs (decimal), at the operand-contract bounds
            fe a, b;
            for (int i = 0; i < FE_LIMBS; i++) is >> a.v[i];
            for (int i = 0; i < FE_LIMBS; i++) is >> b.v[i];
            std::cout << fehex(fe_mul(a, b)) << " " << fehex(fe_sqr(b)) << "\n";
        } else if (cmd == "consts") {
            std::cout << fehex(fe_const_d()) << " " << fehex(fe_const_d2()) << "\n";
        } else if (cmd == "fradd") {
            fr a = rd_fr(is), b = rd_fr(is);
            std::cout << frhex(fr_add(a, b)) << "\n";
        } else if (cmd == "frsub") {
            fr a = rd_fr(is), b = rd_fr(is);
            std::cout << frhex(fr_sub(a, b)) << "\n";
        } else if (cmd == "frmul") {
            fr a = rd_fr(is), b = rd_fr(is);
            std::cout << frhex(fr_mul(a, b)) << "\n";
        } else if (cmd == "frred") {
            std::string h;
            is >> h;
            uint32_t x[16];
            parse_hex(h, x, 16);
            std::cout << frhex(fr_reduce512(x)) << "\n";
        } else if (cmd == "frrepr") {
            fr a = rd_fr(is);
            int sg;
            is >> sg;
            char buf[100];
            int n = fr_repr_write(a, sg != 0, buf);
            buf[n] = 0;
            std::cout << buf << " " << fr_repr_len(a, sg != 0) << "\n";
        } else if (cmd == "dec") {
            uint32_t w[8];
            rd_raw8(is, w);
            char buf[100];
            int n = u256_write_decimal(w, buf);
            buf[n] = 0;
            std::cout << buf << " " << u256_decimal_len(w) << "\n";
        } else if (cmd == "padd" || cmd == "pdbl" || cmd == "prepeat" || cmd == "prepr" || cmd == "preprs" ||
                   cmd == "preprp") {
            ge_proj p;
            p.X = rd_fe(is);
            p.Y = rd_fe(is);
            p.Z = rd_fe(is);
            ge_proj r;
            if (cmd == "padd") {
                ge_proj q;
                q.X = rd_fe(is);
                q.Y = rd_fe(is);
                q.Z = rd_fe(is);
                r = ge_proj_add(p, q);
            } else if (cmd == "pdbl") {
                r = ge_proj_dbl(p);
            } else if (cmd == "prepeat") {
                uint32_t nw[8];
                rd_raw8(is, nw);
                r = ge_proj_repeat(p, nw);
            } else {
                fe8 cx = fe_pack(p.X), cy = fe_pack(p.Y), cz = fe_pack(p.Z);
                char buf[300];
                // prepr: "[X, Y, Z]" unsigned; preprs: signed coordinates; preprp: "(X, Y, Z)" signed
                const fmt_point_style style = {cmd == "preprp" ? '(' : '[', cmd == "preprp" ? ')' : ']',
                                               cmd == "prepr" ? 0 : 1};
                int n = proj_repr_write(cx.w, cy.w, cz.w, style, buf);
                buf[n] = 0;
                std::cout << buf << "|" << proj_repr_len(cx.w, cy.w, cz.w, style) << "\n";
                continue;
            }
            std::cout << fehex(r.X) << " " << fehex(r.Y) << " " << fehex(r.Z) << "\n";
        } else if (cmd == "eadd" || cmd == "emadd" || cmd == "edbl" || cmd == "emaddneg") {
            // affine inputs -> affine output through the extended-coordinate formulas
            ge_aff a;
            a.x = rd_fe(is);
            a.y = rd_fe(is);
            ge_ext p = ge_ext_from_affine(a);
            // de-normalise so Z != 1 is exercised: p = 2p - p is costly; scale by dbl+add instead
            ge_ext r;
            if (cmd == "edbl") {
                r = ge_dbl(ge_dbl(p));  // 4P
            } else {
                ge_aff b;
                b.x = rd_fe(is);
                b.y = rd_fe(is);
                ge_ext p3 = ge_add(ge_dbl(p), p);  // 3P with Z != 1
                if (cmd == "eadd")
                    r = ge_add(p3, ge_ext_from_affine(b));
                else if (cmd == "emadd")
                    r = ge_madd(p3, ge_niels_from_affine(b));
                else
                    r = ge_madd(p3, ge_niels_select_neg(ge_niels_from_affine(b), true));
            }
            ge_aff o = ge_ext_to_affine(r);
            std::cout << fehex(o.x) << " " << fehex(o.y) << " " << (ge_aff_on_curve(o) ? 1 : 0) << "\n";
        } else if (cmd == "bnmul" || cmd == "bnadd" || cmd == "bnsub") {
            fp a = rd_fp(is), b = rd_fp(is);
            std::cout << fphex(cmd == "bnmul" ? fp_mul(a, b) : cmd == "bnadd" ? fp_add(a, b) : fp_sub(a, b)) << "\n";
        } else if (cmd == "bninv") {
            fp a = rd_fp(is);
            std::cout << fphex(fp_inv(a)) << "\n";
        } else if (cmd == "bn2mul") {
            fp2 a = rd_fp2(is), b = rd_fp2(is);
            std::cout << fp2hex(fp2_mul(a, b)) << "\n";
        } else if (cmd == "bn2sqr" || cmd == "bn2inv") {
            fp2 a = rd_fp2(is);
            std::cout << fp2hex(cmd == "bn2sqr" ? fp2_sqr(a) : fp2_inv(a)) << "\n";
        } else if (cmd == "l29mul" || cmd == "l29add" || cmd == "l29sub") {
            fp29 a = rd_fp29(is), b = rd_fp29(is);
            std::cout << fp29hex(cmd == "l29mul" ? fp29_mul(a, b) : cmd == "l29add" ? fp29_add(a, b) : fp29_sub(a, b)) << "\n";
        } else if (cmd == "l29chain") {
            // sums and differences fed back into products (every value stays in [0, 2p)): ((a+b)(a-b) - a a + b b) (a + a)
            fp29 a = rd_fp29(is), b = rd_fp29(is);
            fp29 t = fp29_sub(fp29_mul(fp29_add(a, b), fp29_sub(a, b)), fp29_sqr(a));
            t = fp29_mul(fp29_add(t, fp29_sqr(b)), fp29_dbl(a));
            std::cout << fp29hex(t) << " " << (fp29_is_zero(t) ? 1 : 0) << " " << (fp29_is_zero(fp29_sub(a, b)) ? 1 : 0) << "\n";
        } else if (cmd == "l29sqr") {
            fp29 a = rd_fp29(is);
            std::cout << fp29hex(fp29_sqr(a)) << " " << fp29hex(fp29_sqr(fp29_add(a, a))) << "\n";
        } else if (cmd == "l29inv") {
            fp29 a = rd_fp29(is);
            std::cout << fp29hex(fp29_inv(a)) << "\n";
        } else if (cmd == "l29raw") {
            // the workspace / table format: canonical residue of x 2^261 in eight words, and back
            fp29 a = rd_fp29(is);
            uint32_t w[8];
            Fp29Ops::store_raw(w, fp29_add(a, fp29_zero()));
            const fp29 l = Fp29Ops::load_raw(w);
            std::cout << fp29hex(l) << " " << (Fp29Ops::raw_canonical(w) ? 1 : 0) << "\n";
        } else if (cmd == "l29x2mul") {
            fp29x2 a = rd_el<Fp29x2Ops>(is), b = rd_el<Fp29x2Ops>(is);
            std::cout << elhex(fp29x2_mul(a, b)) << " " << elhex(fp29x2_sqr(a)) << " " << elhex(Fp29x2Ops::inv(a)) << "\n";
        } else if (cmd.rfind("h1", 0) == 0) {
            curve_cmd<Fp29Ops>(cmd.substr(2), is);
        } else if (cmd.rfind("h2", 0) == 0) {
            curve_cmd<Fp29x2Ops>(cmd.substr(2), is);
        } else if (cmd.rfind("g1", 0) == 0) {
            curve_cmd<Fp1Ops>(cmd.substr(2), is);
        } else if (cmd.rfind("g2", 0) == 0) {
            curve_cmd<Fp2Ops>(cmd.substr(2), is);
        } else if (cmd == "quit") {
            break;
        } else {
            std::cout << "?\n";
        }
    }
    return 0;
}
