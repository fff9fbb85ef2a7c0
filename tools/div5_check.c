#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
static uint64_t s = 88172645463325252ull;
static inline uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static inline double div5(double n, double b, double rb) {
    double q = n * rb; double r = fma(-q, b, n); q = fma(r, rb, q); r = fma(-q, b, n); return fma(r, rb, q);
}
int main(void) {
    long bad = 0, bad3 = 0, N = 400000000L;
    for (long i = 0; i < N; i++) {
        /* b = (double) of a random float scale in [2^-24, 2); n = (double)(float x) * 0.5 with |x| <= b roughly; every 4th case: arbitrary doubles */
        double b, n;
        if (i & 3) {
            uint32_t ub = (uint32_t)(rnd() >> 41) | ((uint32_t)(103 + rnd() % 24) << 23);
            uint32_t un = (uint32_t)(rnd() >> 41) | ((uint32_t)(90 + rnd() % 40) << 23);
            float fb, fn; *(uint32_t*)&fb = ub; *(uint32_t*)&fn = un;
            b = (double)fb; n = (double)fn * 0.5; if (rnd() & 1) n = -n;
        } else {
            uint64_t ub = (rnd() >> 12) | ((uint64_t)(1000 + rnd() % 40) << 52), un = (rnd() >> 12) | ((uint64_t)(1000 + rnd() % 40) << 52);
            *(uint64_t*)&b = ub; *(uint64_t*)&n = un;
        }
        const double rb = 1.0 / b, want = n / b;
        const double got = div5(n, b, rb);
        double q = n * rb; double r = fma(-q, b, n); const double got3 = fma(r, rb, q);
        if (got != want) bad++;
        if (got3 != want) bad3++;
    }
    printf("5-op: %ld wrong of %ld; 3-op: %ld wrong\n", bad, N, bad3);
    return 0;
}
