/* Host replica of quant_round_sat_u8 (tengine_amd/csrc/u8_kernels.hip): the uint8 requantisation
 *     q = sat_u8((int)clamp(roundf(s / scale), +-65536) + zp)          conv_kernel_x86.c:1783-1788, conv_kernel_ref_uint8.c:177-182, fc_ref.c:196-202
 * without the IEEE division on the common path: y = fma(s, fl(1/scale), copysign(0.5 + e, s)), e = 2^-13; trunc(y) is
 * round_half_away(fl(s / scale)) unless fract(|y|) < 2e (then the reference expression decides).  The argument needs |d| < 300:
 * |s*inv - d| <= 2 |d| 2^-24, y rounds within 2^-16 (|y| < 512): 5.1e-5 < e.  Beyond 300 the result is saturated whatever the
 * rounding did (zp is a byte), so those values are never handed over.  Same IEEE operations on host and device.
 * build: gcc -O2 -ffp-contract=off u8_round_check.c -lm */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define E 0x1p-13f
static int sat_u8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
static int ref(float s, float scale, int zp)
{
    volatile float d = s / scale;
    float r = roundf(d);
    r = fminf(fmaxf(r, -65536.f), 65536.f);
    return sat_u8((int)r + zp);
}
static long handed = 0;
static int fast(float s, float scale, float inv, int zp)
{
    const float y = fmaf(s, inv, copysignf(0.5f + E, s));
    const float ay = fabsf(y);
    float r = truncf(fminf(fmaxf(y, -65536.f), 65536.f));
    if ((ay - floorf(ay)) < 2.f * E && ay < 300.5f) { handed++; return ref(s, scale, zp); }
    return sat_u8((int)r + zp);
}
/* quant_round_in: sat_u8((int)clamp(roundf(fl(f / scale) + zp))) -- relu_kernel_ref_uint8.c:83-89, upsample_ref.c:118-125 */
static int ref_in(float f, float scale, int zp)
{
    volatile float d = f / scale;
    volatile float x = d + (float)zp;
    float r = roundf(x);
    r = fminf(fmaxf(r, -65536.f), 65536.f);
    return sat_u8((int)r);
}
static int fast_in(float f, float scale, float inv, int zp)
{
    const float y = fmaf(f, inv, (float)zp);
    volatile float y2 = y + copysignf(0.5f + E, y);
    const float ay = fabsf(y2);
    if ((ay - floorf(ay)) < 2.f * E && ay < 300.5f) { handed++; return ref_in(f, scale, zp); }
    return sat_u8((int)truncf(fminf(fmaxf(y2, -65536.f), 65536.f)));
}
static uint64_t st = 0x9E3779B97F4A7C15ull;
static uint64_t rnd(void) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; }
static float urand(float lo, float hi) { return lo + (hi - lo) * (float)((rnd() >> 11) * (1.0 / 9007199254740992.0)); }

int main(int argc, char** argv)
{
    long n = argc > 1 ? atol(argv[1]) : 20000000L, bad = 0;
    for (long i = 0; i < n; i++) {
        const float scale = (rnd() & 1) ? urand(1e-3f, 0.5f) : (float)exp(urand(logf(1e-6f), logf(40.f)));
        volatile float inv = 1.0f / scale;
        const int zp = (int)(rnd() % 256);
        float s;
        const int kind = (int)(rnd() % 4);
        if (kind == 0) s = urand(-700.f, 700.f) * scale;
        else if (kind == 1) s = urand(-70000.f, 70000.f) * scale;
        else {                                                        /* hug a rounding boundary k + 0.5 */
            const int k = (int)(rnd() % 300);
            float b = ((float)k + 0.5f) * scale;
            int32_t bits; memcpy(&bits, &b, 4);
            bits += (int)(rnd() % 33) - 16;
            memcpy(&s, &bits, 4);
            if (rnd() & 1) s = -s;
        }
        const int r = ref(s, scale, zp), f = fast(s, scale, inv, zp);
        if (r != f) { if (bad < 5) printf("MISMATCH s=%a scale=%a zp=%d ref %d fast %d\n", s, scale, zp, r, f); bad++; }
        /* zero point inside the round: boundaries of x = d + zp are the same k + 0.5 grid shifted by an integer */
        const int ri = ref_in(s, scale, zp), fi = fast_in(s, scale, inv, zp);
        if (ri != fi) { if (bad < 5) printf("MISMATCH(in) s=%a scale=%a zp=%d ref %d fast %d\n", s, scale, zp, ri, fi); bad++; }
    }
    printf("checked %ld mismatches %ld handed_over %ld\n", n, bad, handed);
    return bad != 0;
}
