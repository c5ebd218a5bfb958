/* EVERY accumulator value, not a sample: for each (layer, channel) record {m1, m2, lo, hi, s} of a real model (written by
 * tools/exhaustive_requant.py from the synthetic BASELINE models' quantisation parameters, folded as graph.hip folds them) the
 * one-FMA requantisation of epilogue.h is compared with the reference chain over the whole accumulator range in which the result
 * is not yet saturated for good (|acc * M| <= 140, both signs).  Values the fast path hands over are counted, not compared (they
 * run the chain on the device).  Same IEEE operations as the device.
 * build: gcc -O2 -ffp-contract=off exhaustive_requant.c -lm ; usage: a.out records.bin */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define E 0x1p-14f
static int sat127(int v) { return v > 127 ? 127 : (v < -127 ? -127 : v); }
static int R(float x, float s)
{
    volatile float d = x / s;
    float r = roundf(d);
    return sat127(r > 1e9f ? 1000000000 : (r < -1e9f ? -1000000000 : (int)r));
}
static float clampf(float f, float lo, float hi) { return f < lo ? lo : (f > hi ? hi : f); }

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    float rec[5];
    long long total = 0, flagged = 0, bad = 0, records = 0;
    while (fread(rec, sizeof(float), 5, f) == 5) {
        const float m1 = rec[0], m2 = rec[1], lo0 = rec[2], hi0 = rec[3], s = rec[4];
        volatile float lim = 127.49f * s;
        const float lo = lo0 > -lim ? lo0 : -lim, hi = hi0 < lim ? hi0 : lim;
        const float ylo = 128.f + (float)R(lo, s) + 0.25f, yhi = 128.f + (float)R(hi, s) + 0.75f;
        const double dper = (double)m1 * (double)m2 / (double)s;
        const float M = (float)dper;
        if (!(dper > 0)) continue;
        double amaxd = 140.0 / dper + 8.0;
        if (amaxd > 67108864.0) amaxd = 67108864.0;
        const int amax = (int)amaxd;
        records++;
        for (int acc = -amax; acc <= amax; acc++) {
            const float a = (float)acc;
            const float y = fmaf(a, M, 128.5f + E);
            const float yc = y < ylo ? ylo : (y > yhi ? yhi : y);
            total++;
            if ((yc - floorf(yc)) < 2.f * E) { flagged++; continue; }
            volatile float t = a * m1;
            volatile float fv = t * m2;
            const int r = R(clampf(fv, lo, hi), s);
            if ((int)yc - 128 != r) {
                if (bad < 5) printf("MISMATCH acc=%d m1=%a m2=%a s=%a lo=%g hi=%g fast %d ref %d\n", acc, m1, m2, s, lo, hi, (int)yc - 128, r);
                bad++;
            }
        }
    }
    fclose(f);
    printf("records %lld accumulators %lld mismatches %lld handed_over %lld (%.2e)\n", records, total, bad, flagged, total ? (double)flagged / (double)total : 0.0);
    return bad != 0;
}
