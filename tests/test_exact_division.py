"""The sweeps evaluate the push term (1.0-ALPHA)*x/(outdeg+1) of gpu/ExpandRev.cuh:72 without a
division per edge: rcp = RN(1/den) once per row, then q0 = a*rcp, rem = fma(-q0, den, a),
q = fma(rem, rcp, q0) (dppr::push_term in dppr_multi.hpp). The result must be the correctly rounded
quotient -- the same double the reference's expression gives -- for every operand the path can
see. Checked here on 2e7 random operands with gcc (-ffp-contract=off like the device build). CPU only."""
import subprocess
import textwrap

SRC = textwrap.dedent(r"""
    #include <math.h>
    #include <stdio.h>
    #include <stdint.h>
    static uint64_t s = 88172645463325252ull;
    static uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
    int main(void) {
        long bad = 0;
        for (long it = 0; it < 20000000L; ++it) {
            uint64_t a = rnd(), b = rnd();
            double m = 1.0 + (double)(a >> 12) / 4503599627370496.0;
            double x = ldexp(m, 1 - (int)(b % 62));              /* residuals from 2 down to 1e-18 */
            if (b & (1ull << 40)) x = -x;                          /* phase 1 pushes negative amounts */
            uint64_t sel = (b >> 8) % 5;
            uint64_t span = sel == 0 ? 16 : sel == 1 ? 1000 : sel == 2 ? 100000 : sel == 3 ? 16777214 : 2147483646ull;
            double den = 2.0 + (double)((b >> 16) % span);       /* outdeg + 1 */
            const double A = (1.0 - 0.15) * x;
            const double want = A / den;
            const double rcp = 1.0 / den;
            const double q0 = A * rcp;
            const double rem = fma(-q0, den, A);
            const double q = fma(rem, rcp, q0);
            if (q != want) ++bad;
        }
        printf("%ld\n", bad);
        return 0;
    }
""")


def test_reciprocal_fma_division_is_correctly_rounded(tmp_path):
    src = tmp_path / "div.c"
    src.write_text(SRC)
    exe = tmp_path / "div"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe), str(src), "-lm"])
    assert subprocess.check_output([str(exe)], text=True).strip() == "0"
