/* Host replica of the division-free requantisation fast path (tengine_amd/csrc/epilogue.h: round_div_sat and
 * rq_value/rq_round_fast with the +-127.49*s clamp folded in) checked against the reference expression
 * sat127((int)round(f / s)) on random and adversarial (boundary-hugging) inputs.  IEEE binary32 mul / fma /
 * div behave identically on the host and on gfx950, so this pins the exactness argument.
 * build: gcc -O2 -ffp-contract=off fast_requant_check.c -lm ; prints the number of mismatches. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define EPS 0x1p-14f
static int sat127(int v) { return v > 127 ? 127 : (v < -127 ? -127 : v); }
static int ref(float f, float s)
{
    float d = f / s;
    double r = round((double)d);
    if (r > 1e9) r = 1e9;
    if (r < -1e9) r = -1e9;
    return sat127((int)r);
}
static long risky_count = 0;
static float fractf_(float a) { return a - floorf(a); }
static int cvt_i32(float y) { return y > 2e9f ? 2147483647 : (y < -2e9f ? -2147483647 - 1 : (int)y); }

/* general form */
static int fast_general(float f, float s, float inv)
{
    float y = fmaf(f, inv, copysignf(0.5f + EPS, f));
    int q = sat127(cvt_i32(y));
    float ay = fabsf(y);
    if (fractf_(ay) < 2.f * EPS && ay < 129.f) { risky_count++; q = ref(f, s); }
    return q;
}
/* conv form: f is first clamped to +-127.49*s, no integer clamp afterwards */
static int fast_clamped(float f, float s, float inv)
{
    float lim = 127.49f * s;
    float fc = f > lim ? lim : (f < -lim ? -lim : f);
    float y = fmaf(fc, inv, copysignf(0.5f + EPS, fc));
    int q = cvt_i32(y);
    if (fractf_(fabsf(y)) < 2.f * EPS) { risky_count++; q = ref(fc, s); }
    return q;
}
static uint64_t st = 0x9E3779B97F4A7C15ull;
static uint64_t rnd(void) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; }
static float urand(float lo, float hi) { return lo + (hi - lo) * (float)((rnd() >> 11) * (1.0 / 9007199254740992.0)); }

int main(int argc, char** argv)
{
    long n = argc > 1 ? atol(argv[1]) : 20000000L, bad = 0, total = 0;
    for (long i = 0; i < n; i++) {
        float s = (rnd() & 1) ? urand(1e-3f, 0.5f) : urand(1e-6f, 40.f);
        volatile float inv = 1.0f / s;
        float f;
        int kind = (int)(rnd() % 4);
        if (kind == 0) f = urand(-200.f, 200.f) * s;                 /* generic, incl. saturating */
        else if (kind == 1) f = urand(0.f, 6.f);                      /* relu6 range */
        else {                                                        /* hug a rounding boundary k+0.5 */
            int k = (int)(rnd() % 131) - 1;
            float b = ((float)k + 0.5f) * s;
            int32_t bits; memcpy(&bits, &b, 4);
            bits += (int)(rnd() % 33) - 16;                           /* +-16 ulps around it */
            memcpy(&f, &bits, 4);
            if (rnd() & 1) f = -f;
        }
        total++;
        int r = ref(f, s);
        if (fast_general(f, s, inv) != r || fast_clamped(f, s, inv) != r) {
            if (bad < 5) printf("MISMATCH f=%a s=%a ref %d general %d clamped %d\n", f, s, r, fast_general(f, s, inv), fast_clamped(f, s, inv));
            bad++;
        }
    }
    printf("checked %ld mismatches %ld slow_path %ld\n", total, bad, risky_count);
    return bad != 0;
}
