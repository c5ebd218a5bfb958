/* Host replica of the ONE-FMA requantisation fast path (tengine_amd/csrc/epilogue.h: requant4 / requant1) against the
 * reference chain it replaces
 *
 *     f = fl(fl((float)acc * m1) * m2[c]) ;  f = clamp(f, lo, hi) ;  q = sat127((int)round(fl(f / s)))
 *     (conv_kernel_x86.c:1826-1889, conv_kernel_ref_int8.c:137-167, fc_ref.c:252-257 folded into (m1, m2, lo, hi, s))
 *
 * Fast path (planner-folded constants):  M = RN32(double(m1) * double(m2[c]) / double(s)),  e = 2^-14,
 *     y  = fma((float)acc, M, 128.5 + e)                  -- one rounding
 *     yc = med3(y, 128 + q_lo + 0.25, 128 + q_hi + 0.75)  -- q_lo / q_hi = the reference expression applied to lo / hi (host)
 *     q  = trunc(yc) - 128,   taken unless fract(yc) < 2e (then the reference chain itself decides)
 *
 * Why it is exact: with d_ref = fl(f / s) (unclamped) and |d_ref| < 128.6,
 *     |y - (d_ref + 128.5 + e)| <= |d|*(4 + 2^-20)*2^-24 + 2^-16 < 4.6e-5 < e,   so  y - (d_ref + 128.5) in (0, 2e):
 * an integer can lie between d_ref + 128.5 and y only if fract(y) < 2e, and ties (d_ref + 128.5 integral) are flagged too, so
 * floor(y) == floor(d_ref + 128.5) == round_half_away(d_ref) + 128 for every unflagged, unclamped value (both signs).  The
 * reference's clamp commutes with its monotone round(fl(x / s)): clamp(R(f), R(lo), R(hi)), which the clamp of y implements.
 * IEEE binary32 cvt / fma / mul / div behave identically on the host and on gfx950, so this program pins the argument:
 * random layers (scales over six decades, every activation window), random and boundary-hugging accumulators.
 * build: gcc -O2 -ffp-contract=off fold_requant_check.c -lm ; exit status = mismatches != 0. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define E 0x1p-14f
static int sat127(int v) { return v > 127 ? 127 : (v < -127 ? -127 : v); }
static float clampf(float f, float lo, float hi) { return f < lo ? lo : (f > hi ? hi : f); }
static int R(float x, float s)       /* sat127((int)round(x / s)) */
{
    volatile float d = x / s;
    double r = round((double)d);
    if (r > 1e9) r = 1e9;
    if (r < -1e9) r = -1e9;
    return sat127((int)r);
}
static int ref_chain(int acc, float m1, float m2, float lo, float hi, float s)
{
    volatile float a = (float)acc;
    volatile float t = a * m1;
    volatile float f = t * m2;
    return R(clampf(f, lo, hi), s);
}
static long flagged = 0, binade_checked = 0, binade_bad = 0;
/* epilogue.h's form for windows that start at 128.25 (a fused ReLU): every clamped value lies in [128, 256), so the kernels read the
 * result byte and the hand-over flag off the bit pattern (byte 2 / low half below thr * 2^16) instead of converting.  Both must be
 * the conversions' results for EVERY value -- checked here on every value the program draws (also by check_elt with thr = 2^-12). */
static void check_one_binade(float yc, float thr)
{
    uint32_t b;
    memcpy(&b, &yc, 4);
    const int byte2 = (int)((b >> 16) & 0xffu), low_flag = (b & 0xffffu) < (uint32_t)(thr * 65536.f);
    binade_checked++;
    if (byte2 != (int)yc - 128 || low_flag != ((yc - floorf(yc)) < thr)) {
        if (binade_bad < 5) printf("ONE-BINADE MISMATCH yc=%a: byte2 %d trunc-128 %d, low-half flag %d fract flag %d\n", yc, byte2, (int)yc - 128, low_flag, (yc - floorf(yc)) < thr);
        binade_bad++;
    }
}
static int fast(int acc, float M, float ylo, float yhi, int* risky)
{
    float y = fmaf((float)acc, M, 128.5f + E);
    float yc = y < ylo ? ylo : (y > yhi ? yhi : y);     /* med3: ylo < yhi */
    *risky = (yc - floorf(yc)) < 2.f * E;
    if (ylo >= 128.f) check_one_binade(yc, 2.f * E);
    return (int)yc - 128;
}
static uint64_t st = 0x243F6A8885A308D3ull;
static uint64_t rnd(void) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; }
static double urand(double lo, double hi) { return lo + (hi - lo) * ((rnd() >> 11) * (1.0 / 9007199254740992.0)); }
static float logu(double lo, double hi) { return (float)exp(urand(log(lo), log(hi))); }

/* ---- the residual tail (epilogue.h: elt_sum16_fold): f = fl(fl(qc*sc) + fl(qr*sr)), y = sat127(round(fl(f / s))), [max(y, 0)]
 * against  t = fma(qc + 128, Mc, K0), yb = fma(qr + 128, Mr, t), window, truncate;  folded only when (sc + sr) / s <= 2 */
static long check_elt(long layers, long* total_out, long* flagged_out)
{
    long bad = 0, total = 0, fl = 0;
    for (long L = 0; L < layers; L++) {
        const float s = logu(1e-3, 1.0);
        float sc, sr;
        do { sc = s * (float)urand(0.02, 1.9); sr = s * (float)urand(0.02, 1.9); } while (((double)sc + (double)sr) / (double)s > 2.0);
        const int relu = (int)(rnd() & 1);
        const float e = 0x1p-13f;
        const float mc = (float)((double)sc / (double)s), mr = (float)((double)sr / (double)s);
        const float k0 = (float)(128.5 + (double)e - 128.0 * ((double)mc + (double)mr));
        const float ylo = relu ? 128.25f : 1.25f, yhi = 255.75f, thr = 2.f * e;
        for (int qc = -127; qc <= 127; qc++)
            for (int qr = -128; qr <= 127; qr++) {
                volatile float p1 = (float)qc * sc, p2 = (float)qr * sr;
                volatile float f = p1 + p2;
                int r = R(f, s);
                if (relu && r < 0) r = 0;
                const float t = fmaf((float)(qc + 128), mc, k0);
                float y = fmaf((float)(qr + 128), mr, t);
                y = y < ylo ? ylo : (y > yhi ? yhi : y);
                if (relu) check_one_binade(y, thr);
                total++;
                if (y - floorf(y) < thr) { fl++; continue; }
                if ((int)y - 128 != r) {
                    if (bad < 5) printf("ELT MISMATCH qc=%d qr=%d sc=%a sr=%a s=%a relu=%d: fast %d ref %d\n", qc, qr, sc, sr, s, relu, (int)y - 128, r);
                    bad++;
                }
            }
    }
    *total_out = total; *flagged_out = fl;
    return bad;
}

int main(int argc, char** argv)
{
    long layers = argc > 1 ? atol(argv[1]) : 40000, per = 600, bad = 0, total = 0;
    {
        long et, ef;
        const long eb = check_elt(layers / 100 + 1, &et, &ef);        /* every (qc, qr) pair of each layer */
        printf("eltwise tail: checked %ld mismatches %ld flagged %ld (%.2e)\n", et, eb, ef, (double)ef / (double)et);
        bad += eb;
    }
    for (long L = 0; L < layers; L++) {
        float m1 = (rnd() & 3) ? logu(1e-4, 1.0) : 1.0f;
        float m2 = logu(1e-6, 1.0);
        float s = (rnd() & 7) ? logu(1e-3, 2.0) : 1.0f;
        float lo = -3.4028235e38f, hi = 3.4028235e38f;
        switch (rnd() % 5) {
        case 0: lo = 0.f; break;
        case 1: lo = 0.f; hi = 6.f; break;
        case 2: lo = -1.f; hi = 1.f; break;
        default: break;
        }
        /* the planner's folding (graph.hip: host_rq): saturation folded into lo / hi, then the reference expression on them */
        volatile float lim = 127.49f * s;
        float loc = lo > -lim ? lo : -lim, hic = hi < lim ? hi : lim;
        const float ylo = 128.f + (float)R(loc, s) + 0.25f, yhi = 128.f + (float)R(hic, s) + 0.75f;
        const float M = (float)((double)m1 * (double)m2 / (double)s);
        const double dper = (double)m1 * (double)m2 / (double)s;       /* output units per accumulator unit */
        for (long i = 0; i < per; i++) {
            int acc;
            const int kind = (int)(rnd() % 4);
            if (kind == 0) acc = (int)(rnd() % 2000001) - 1000000;
            else if (kind == 1) acc = (int)urand(-140.0 / dper, 140.0 / dper > 2e9 ? 2e9 : 140.0 / dper);
            else {                                                    /* hug a rounding boundary k + 0.5 (both signs) */
                const int k = (int)(rnd() % 130);
                double a0 = ((double)k + 0.5) / dper;
                if (a0 > 2e9) a0 = 2e9;
                acc = (int)a0 + (int)(rnd() % 5) - 2;
                if (rnd() & 1) acc = -acc;
            }
            if (fabs((double)acc * dper) > 1e30) continue;
            int risky;
            const int q = fast(acc, M, ylo, yhi, &risky), r = ref_chain(acc, m1, m2, loc, hic, s);
            const int r_raw = ref_chain(acc, m1, m2, lo, hi, s);      /* the unfolded reference: the lim clamp changes nothing */
            total++;
            if (r != r_raw) { if (bad < 5) printf("FOLD MISMATCH acc=%d m1=%a m2=%a s=%a lo=%g hi=%g\n", acc, m1, m2, s, lo, hi); bad++; }
            if (risky) { flagged++; continue; }
            if (q != r) {
                if (bad < 5) printf("MISMATCH acc=%d m1=%a m2=%a s=%a lo=%g hi=%g: fast %d ref %d\n", acc, m1, m2, s, lo, hi, q, r);
                bad++;
            }
        }
    }
    printf("requant: checked %ld mismatches %ld flagged %ld (%.2e)\n", total, bad, flagged, (double)flagged / (double)total);
    printf("one-binade form: checked %ld mismatches %ld\n", binade_checked, binade_bad);
    bad += binade_bad;
    if (bad == 0) printf("mismatches 0\n");
    return bad != 0;
}
