/* Exhaustive check (test infrastructure): for EVERY finite float x,
 *     q = x * c ; r = fmaf(-d, q, x) ; q' = fmaf(r, c, q)      with c = 1.0f / d
 * equals the correctly rounded IEEE quotient x / d, for the constant divisors the hot path
 * uses (9: the 3x3 window mean, reference layers.py:266-270; 3: the channel mean,
 * train.py:977,982).  The gfx950 kernels use this 3-instruction form instead of the
 * ~10-instruction generic IEEE divide; this program is the proof that it changes no bit.
 * usage: check_constdiv d [d ...]   -> prints mismatches, exit code = (any mismatch)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline float constdiv(float x, float d, float c)
{
    float q = x * c;
    float r = fmaf(-d, q, x);
    return fmaf(r, c, q);
}

int main(int argc, char **argv)
{
    int bad_total = 0;
    for (int a = 1; a < argc; ++a) {
        float d = (float)atof(argv[a]);
        float c = 1.0f / d;
        long bad = 0;
        uint32_t first = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
        for (int64_t i = 0; i < (1LL << 32); ++i) {
            uint32_t u = (uint32_t)i;
            float x;
            memcpy(&x, &u, 4);
            if (!isfinite(x)) continue;
            float want = x / d, got = constdiv(x, d, c);
            if (memcmp(&want, &got, 4) != 0 && !(want == 0.0f && got == 0.0f)) {
                ++bad;
#pragma omp critical
                if (!first) first = u;
            }
        }
        printf("d=%g c=%a mismatches=%ld first=0x%08x\n", d, c, bad, first);
        bad_total += bad != 0;
    }
    return bad_total;
}
