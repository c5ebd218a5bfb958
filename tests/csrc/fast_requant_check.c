/* Host replica of round_div_sat() (tengine_amd/csrc/epilogue.h) checked against the reference expression
 * sat127((int)round(f / s)) on random and adversarial (boundary-hugging) inputs.  IEEE binary32 mul / add /
 * floor / div behave identically on the host and on gfx950, so this pins the exactness argument.
 * build: gcc -O2 -ffp-contract=off fast_requant_check.c -lm ; prints the number of mismatches. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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
static int fast(float f, float s, float inv)
{
    volatile float t = f * inv;
    volatile float y = fabsf(t) + 0.5f;
    float fr = y - floorf(y);
    double yy = y > 1e9f ? 1e9 : y;
    int q = (int)yy;
    q = q > 127 ? 127 : q;
    q = t < 0.f ? -q : q;
    int risky = (fabsf(fr - 0.5f) > 0.5f - 0x1p-14f) && y < 129.f;
    if (risky) { risky_count++; q = ref(f, s); }
    return q;
}
static uint64_t st = 0x9E3779B97F4A7C15ull;
static uint64_t rnd(void) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; }
static float urand(float lo, float hi) { return lo + (hi - lo) * (float)((rnd() >> 11) * (1.0 / 9007199254740992.0)); }

int main(int argc, char** argv)
{
    long n = argc > 1 ? atol(argv[1]) : 20000000L, bad = 0, total = 0;
    for (long i = 0; i < n; i++) {
        float s = urand(1e-3f, 0.5f);
        volatile float inv = 1.0f / s;
        float f;
        int kind = (int)(rnd() % 4);
        if (kind == 0) f = urand(-200.f, 200.f) * s;                 /* generic */
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
        if (fast(f, s, inv) != ref(f, s)) { if (bad < 5) printf("MISMATCH f=%a s=%a\n", f, s); bad++; }
    }
    printf("checked %ld mismatches %ld slow_path %ld\n", total, bad, risky_count);
    return bad != 0;
}
