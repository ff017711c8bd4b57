"""gfw_math.h (the fixed-op-sequence atanf/tanf every kernel uses) vs this box's libm, exhaustively.

The header is compiled for the host with gcc -ffp-contract=off and compared with glibc's atanf/tanf on ALL 2^32
float bit patterns (OpenMP; ~25 s on 8 cores).  The reference (Rust std on linux-gnu) calls exactly these libm
routines (opencv_fisheye.rs:56,79; gopro.rs:52,67), so equality here is what makes the oracle's `libm` arithmetic and
the device's arithmetic the same function.
"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include "%s/gyroflow_amd/csrc/gfw_math.h"
int main(void) {
    long bad_atan = 0, bad_tan = 0;
    #pragma omp parallel for reduction(+:bad_atan,bad_tan) schedule(static)
    for (long i = 0; i < (1L << 32); ++i) {
        const uint32_t u = (uint32_t)i; const float x = gfw_u2f(u);
        const float a = atanf(x), b = gfw_atanf(x);
        if (gfw_f2u(a) != gfw_f2u(b) && !(a != a && b != b)) bad_atan++;
        const float c = tanf(x), d = gfw_tanf(x);
        if (gfw_f2u(c) != gfw_f2u(d) && !(c != c && d != d)) bad_tan++;
    }
    printf("%%ld %%ld\n", bad_atan, bad_tan);
    return 0;
}
"""


def test_atanf_tanf_bit_identical_to_libm_on_all_inputs(tmp_path):
    c = tmp_path / "m.c"
    c.write_text(SRC % ROOT)
    exe = tmp_path / "m"
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-fno-fast-math", str(c), "-o", str(exe), "-lm"])
    out = subprocess.check_output([str(exe)], timeout=900).decode().split()
    assert out == ["0", "0"], "mismatches vs libm: atanf %s, tanf %s" % tuple(out)


ROUND_SRC = r"""
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
/* Rust `v as i32`: truncate, saturate, NaN -> 0 (what v_cvt_i32_f32 does on the device) */
static inline int32_t as_i32(float v) { if (v != v) return 0; if (v >= 2147483648.0f) return INT32_MAX; if (v <= -2147483648.0f) return INT32_MIN; return (int32_t)v; }
int main(void) {
    long bad = 0;
    #pragma omp parallel for reduction(+:bad) schedule(static)
    for (long i = 0; i < (1L << 32); ++i) {
        const uint32_t u = (uint32_t)i; float x; memcpy(&x, &u, 4);
        const int32_t want = as_i32(roundf(x));                               /* f32::round() as i32 */
        const int32_t got = as_i32(x + copysignf(0x1.fffffep-2f, x));         /* gfw_frame.hip round_i32 */
        if (want != got) bad++;
    }
    printf("%ld\n", bad);
    return 0;
}
"""


def test_branch_free_round_half_away_matches_roundf_on_all_inputs(tmp_path):
    """The fused kernel's `round_i32` (x + copysign(pred(0.5), x), truncated by the cast) vs `roundf` + cast, all 2^32 floats."""
    c = tmp_path / "r.c"
    c.write_text(ROUND_SRC)
    exe = tmp_path / "r"
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-fno-fast-math", str(c), "-o", str(exe), "-lm"])
    assert subprocess.check_output([str(exe)], timeout=900).decode().split() == ["0"]


SINCOS_SRC = r"""
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include "%s/gyroflow_amd/csrc/gfw_math.h"
int main(void) {
    long bad_s = 0, bad_c = 0;
    #pragma omp parallel for reduction(+:bad_s,bad_c) schedule(static)
    for (long i = 0; i < (1L << 32); ++i) {
        const uint32_t u = (uint32_t)i; const float x = gfw_u2f(u);
        const float a = sinf(x), b = gfw_sinf(x);
        if (gfw_f2u(a) != gfw_f2u(b) && !(a != a && b != b)) bad_s++;
        const float c = cosf(x), d = gfw_cosf(x);
        if (gfw_f2u(c) != gfw_f2u(d) && !(c != c && d != d)) bad_c++;
    }
    printf("%%ld %%ld\n", bad_s, bad_c);
    return 0;
}
"""


def test_sinf_cosf_bit_identical_to_libm_on_all_inputs(tmp_path):
    """gfw_sinf / gfw_cosf (IBIS/OIS roll terms for device-resident matrices) vs this box's libm, all 2^32 floats.  glibc picks
    its FMA build of sinf/cosf on FMA-capable x86-64 CPUs; on a CPU without FMA this test would tell."""
    c = tmp_path / "sc.c"
    c.write_text(SINCOS_SRC % ROOT)
    exe = tmp_path / "sc"
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-fno-fast-math", str(c), "-o", str(exe), "-lm"])
    out = subprocess.check_output([str(exe)], timeout=900).decode().split()
    assert out == ["0", "0"], "mismatches vs libm: sinf %s, cosf %s" % tuple(out)


ATAN_TAB_SRC = r"""
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include "%s/gyroflow_amd/csrc/gfw_math.h"
static const float TAB[48] = GFW_ATAN_TAB_INIT;
int main(void) {
    long bad = 0;
    #pragma omp parallel for reduction(+:bad) schedule(static)
    for (long i = 0; i < 0x7f800000L; ++i) {                      /* every finite float >= +0 */
        const float x = gfw_u2f((uint32_t)i);
        const float *rec = TAB + 8 * gfw_atanf_tab_id(x);
        float num, den;
        gfw_atanf_tab_reduce(x, rec, &num, &den);
        const float got = gfw_atanf_tab_finish(num / den, rec[4], rec[5]);
        if (gfw_f2u(got) != gfw_f2u(gfw_atanf(x))) bad++;
    }
    printf("%%ld\n", bad);
    return 0;
}
"""


def test_table_driven_atan_reduction_equals_gfw_atanf_on_every_finite_non_negative_float(tmp_path):
    """gfw_atanf_tab_{id,reduce,finish} (the select-free reduction prepared for the fused kernel, GFW_ATAN_TABLE) composed with an
    IEEE division, against gfw_atanf — which the first test of this file pins to libm — on all 2^31 - 2^23 finite floats >= 0."""
    c = tmp_path / "t.c"
    c.write_text(ATAN_TAB_SRC % ROOT)
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-fno-fast-math", str(c), "-o", str(exe), "-lm"])
    assert subprocess.check_output([str(exe)], timeout=900).decode().split() == ["0"]
