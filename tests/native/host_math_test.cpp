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
#include "../../verifiable_mpc_amd/csrc/fe25519.cuh"
#include "../../verifiable_mpc_amd/csrc/ge25519.cuh"
#include "../../verifiable_mpc_amd/csrc/fr.cuh"
#include "../../verifiable_mpc_amd/csrc/fmt.cuh"

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
    fe r;
    parse_hex(h, r.v, 8);
    return r;
}
static fr rd_fr(std::istringstream &is) {
    std::string h;
    is >> h;
    fr r;
    parse_hex(h, r.v, 8);
    return r;
}
static std::string fehex(const fe &a) {
    fe c = fe_canon(a);
    return to_hex(c.v, 8);
}
static std::string frhex(const fr &a) { return to_hex(a.v, 8); }

int main() {
    std::string line;
    while (std::getline(std::cin, line)) {
        std::istringstream is(line);
        std::string cmd;
        is >> cmd;
        if (cmd == "femul") {
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
            fe s = rd_fe(is);
            std::cout << fehex(fe_mul_u32(a, s.v[0])) << "\n";
        } else if (cmd == "feinv") {
            fe a = rd_fe(is);
            std::cout << fehex(fe_inv(a)) << "\n";
        } else if (cmd == "fecanon") {
            fe a = rd_fe(is);
            std::cout << fehex(a) << " " << (fe_is_canonical(a) ? 1 : 0) << "\n";
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
            fe a = rd_fe(is);
            char buf[100];
            int n = u256_write_decimal(a.v, buf);
            buf[n] = 0;
            std::cout << buf << " " << u256_decimal_len(a.v) << "\n";
        } else if (cmd == "padd" || cmd == "pdbl" || cmd == "prepeat" || cmd == "prepr") {
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
                fe n = rd_fe(is);
                r = ge_proj_repeat(p, n.v);
            } else {
                ge_proj c = ge_proj_canon(p);
                char buf[300];
                int n = proj_repr_write(c.X.v, c.Y.v, c.Z.v, buf);
                buf[n] = 0;
                std::cout << buf << "|" << proj_repr_len(c.X.v, c.Y.v, c.Z.v) << "\n";
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
        } else if (cmd == "quit") {
            break;
        } else {
            std::cout << "?\n";
        }
    }
    return 0;
}
