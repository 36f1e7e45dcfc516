/* Host check of the cached-reciprocal division sequence the search kernel uses
 * at the root (product + two fused residual corrections): counts inputs where
 * it differs from the correctly rounded t / b.  Same operations as the device
 * code: IEEE fp64 multiply and fma, round to nearest even.
 * usage: div_check <n_random> <max_b>  -> prints "<tested> <mismatches>" */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t rng = 0x9e3779b97f4a7c15ull;
static uint64_t next(void)
{
    uint64_t z = (rng += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

static double seq(double t, double b, double rinv)
{
    double q = t * rinv;
    q = fma(fma(-q, b, t), rinv, q);
    q = fma(fma(-q, b, t), rinv, q);
    return q;
}

int main(int argc, char **argv)
{
    long n = argc > 1 ? atol(argv[1]) : 1000000;
    int max_b = argc > 2 ? atoi(argv[2]) : 4096;
    long tested = 0, bad = 0;
    /* random significands over the magnitudes cpuct * P * sqrt(nodeN + 1e-8) can take */
    for (long i = 0; i < n; i++) {
        uint64_t m = next();
        int e = (int)(next() % 160) - 120; /* 2^-120 .. 2^39 */
        double t = ldexp(1.0 + (double)(m >> 12) / 4503599627370496.0, e);
        double b = (double)(1 + next() % (uint64_t)max_b);
        double r = 1.0 / b;
        tested++;
        if (seq(t, b, r) != t / b) bad++;
        /* value sums divided by visit counts: either sign, |W| <= N */
        double w = ((double)(next() >> 11) / 9007199254740992.0 * 2.0 - 1.0) * b;
        tested++;
        if (seq(w, b, r) != w / b) bad++;
    }
    /* adversarial: numerators next to exact multiples and to rounding midpoints */
    for (int bi = 1; bi <= max_b; bi++) {
        double b = (double)bi, r = 1.0 / b;
        for (int j = 0; j < 64; j++) {
            uint64_t m = next();
            double q = 1.0 + (double)(m >> 12) / 4503599627370496.0;
            double t0 = q * b; /* rounded product */
            for (int d = -3; d <= 3; d++) {
                double t = t0;
                for (int s = 0; s < (d < 0 ? -d : d); s++) t = nextafter(t, d < 0 ? 0.0 : INFINITY);
                tested++;
                if (seq(t, b, r) != t / b) bad++;
            }
        }
        tested++;
        if (seq(0.0, b, r) != 0.0) bad++;
    }
    printf("%ld %ld\n", tested, bad);
    return 0;
}
