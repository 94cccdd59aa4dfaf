/* rng.c -- the reference's per-atom random streams (random.c:21-75), needed bit-for-bit so that
 * every implementation starts from identical momenta and displacements.
 * lcg61: x <- (a*x mod 2^64) mod (2^61 - 1); mkSeed: two Knuth hashes + 10 warm-up draws;
 * gasdev: polar Box-Muller that never caches its second deviate. */
#include "comd_host.h"
#include <math.h>

double lcg61(uint64_t* seed)
{
   static const uint64_t kPrime = UINT64_C(2305843009213693951);
   static const double kInv = 1.0 / UINT64_C(2305843009213693951);
   uint64_t s = *seed;
   s *= UINT64_C(437799614237992725);
   s %= kPrime;
   *seed = s;
   return s * kInv;
}

uint64_t mkSeed(uint32_t id, uint32_t callSite)
{
   uint32_t h1 = id * UINT32_C(2654435761);
   uint32_t h2 = (id + callSite) * UINT32_C(2654435761);
   uint64_t seed = (UINT64_C(0x100000000) * h1) + h2;
   for (unsigned i = 0; i < 10; ++i) lcg61(&seed);
   return seed;
}

real_t gasdev(uint64_t* seed)
{
   real_t u, v, q;
   do {
      u = 2.0 * lcg61(seed) - 1.0;
      v = 2.0 * lcg61(seed) - 1.0;
      q = u * u + v * v;
   } while (q >= 1.0 || q == 0.0);
   return v * sqrt(-2.0 * log(q) / q);
}
