/*
 * iou_div_proof.c — checks that the division-free quotient of the IoU kernels (csrc/hd_common.h:
 * hd_div_exact2 / hd_quot_fast) is BIT-IDENTICAL to the IEEE-754 binary32 quotient n / u that the
 * oracle (oracle/hd_oracle.c: iou_xyxy) computes.  Test infrastructure only; plain C, no GPU.
 *
 * Sequence under test — every step is one binary32 operation, FMAs as written:
 *     y0 = RCP(u)                 ANY estimate within 1 ulp of the correctly rounded 1/u (v_rcp_f32's bound)
 *     e  = fma(-u, y0, 1);  t = fma(e, e, e);  y1 = fma(t, y0, y0)        second-order Newton step
 *     q0 = n * y1;  r = fma(-u, q0, n);  q = fma(r, y1, q0)                quotient + residual correction
 * except that a denominator whose significand is all ones (u = (2 - 2^-23) * 2^k) takes the IEEE division.
 *
 * Proof structure.
 *  (0) Scale invariance: every step commutes exactly with u -> u*2^a, n -> n*2^b while no intermediate
 *      leaves the normal range; the kernels only run the sequence when n, u lie in [2^-86, 2^84]
 *      (hd_common.h: hd_box_div_safe — coordinates 0 or 2^-20 <= |c| <= 2^40 — else IEEE division), where
 *      the smallest intermediate, r ~ n * 2^-24, is still normal.  So only the two 24-bit significands
 *      matter, plus whether n sits in u's binade or the one below.
 *  (1) Exhaustive over all 2^23 denominators and the three estimates RN(1/u) + {-1, 0, +1} ulp:
 *      y1 == RN(1/u) except for the few denominators printed as "exception set X".
 *  (2) For u outside X, y1 is the correctly rounded reciprocal, q0 = RN(n*y1) is within one ulp of n/u,
 *      and r is exact, so q = RN(q0 + r*y1) = RN(n/u) by Markstein's theorem (P. Markstein, "Computation
 *      of elementary functions on the IBM RISC System/6000 processor", IBM J. R&D 34(1), 1990, Thm 4.4;
 *      Cornea, Harrison, Tang, "Scientific Computing on Itanium-based Systems", Thm 8.5).
 *  (3) For every u in X: exhaustive over all 2^23 numerators x both binades x the three estimates.
 *      The only mismatches allowed are u = all-ones significand (the IEEE-division case above).
 *  (4) Belt and braces: all 2^23 values of one operand x 101 structured values of the other (boundary
 *      patterns, the four neighbours of the first operand, pseudo-random fill), in BOTH roles, both
 *      binades, the three estimates — must be mismatch-free outside the all-ones family.
 *
 * Exit code 0 iff (1)-(4) hold.   Build: gcc -O2 -mfma -fopenmp iou_div_proof.c -lm
 * (without -mfma fmaf() is libm's exact software fma: correct, much slower).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static inline float f_from_bits(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
static inline uint32_t bits_of(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }
static inline float sig(uint32_t m) { return f_from_bits(0x3F800000u | (m & 0x7FFFFFu)); } /* [1, 2) */
static inline int all_ones(float u) { return (bits_of(u) & 0x7FFFFFu) == 0x7FFFFFu; }

static inline float recip_step(float u, float y0) {
  const float e = fmaf(-u, y0, 1.0f);
  const float t = fmaf(e, e, e);
  return fmaf(t, y0, y0);
}
static inline float quot(float n, float u, float y0) {
  const float y1 = recip_step(u, y0);
  const float q0 = n * y1;
  const float r = fmaf(-u, q0, n);
  return fmaf(r, y1, q0);
}

#define NS 96
static uint32_t structured[NS];
static void build_structured(void) {
  int k = 0;
  const uint32_t fixed[] = {0x000000, 0x000001, 0x000002, 0x7FFFFF, 0x7FFFFE, 0x7FFFFD, 0x400000, 0x3FFFFF,
                            0x400001, 0x200000, 0x600000, 0x555555, 0x2AAAAA, 0x100000, 0x0FFFFF, 0x7F0000};
  for (size_t i = 0; i < sizeof(fixed) / sizeof(fixed[0]); ++i) structured[k++] = fixed[i];
  uint32_t s = 0x9E3779B9u;
  while (k < NS) {
    s ^= s << 13; s ^= s >> 17; s ^= s << 5;
    structured[k++] = s & 0x7FFFFFu;
  }
}

/* mismatches of quot() against n/u for one (n, u) over the three reciprocal estimates;
 * *only_all_ones is cleared when a mismatch happens for a denominator that is not all-ones */
static inline unsigned check_pair(float n, float u, int* only_all_ones) {
  const float ref = n / u, yr = 1.0f / u;
  unsigned bad = 0;
  for (int d = -1; d <= 1; ++d)
    if (bits_of(quot(n, u, f_from_bits(bits_of(yr) + (uint32_t)d))) != bits_of(ref)) ++bad;
  if (bad && !all_ones(u)) *only_all_ones = 0;
  return bad;
}

int main(void) {
  build_structured();
  int ok = 1;
  /* (1) exception set of the reciprocal */
  static uint32_t X[1024];
  int nx = 0;
  for (uint32_t m = 0; m < (1u << 23); ++m) {
    const float u = sig(m), yr = 1.0f / u;
    int ex = 0;
    for (int d = -1; d <= 1; ++d)
      if (bits_of(recip_step(u, f_from_bits(bits_of(yr) + (uint32_t)d))) != bits_of(yr)) ex = 1;
    if (ex && nx < 1024) X[nx++] = m;
  }
  printf("(1) y1 != RN(1/u) for %d of 2^23 denominators (exception set X):", nx);
  for (int i = 0; i < nx; ++i) printf(" %06x", X[i]);
  printf("\n");
  if (nx > 64) ok = 0;
  /* (3) exhaustive numerators for every u in X */
  for (int i = 0; i < nx; ++i) {
    const float u = sig(X[i]);
    unsigned long long bad = 0;
    int only = 1;
#pragma omp parallel for reduction(+ : bad) reduction(& : only) schedule(static)
    for (uint32_t k = 0; k < (1u << 23); ++k) {
      int o = 1;
      bad += check_pair(sig(k), u, &o);
      bad += check_pair(sig(k) * 0.5f, u, &o);
      only &= o;
    }
    printf("(3) u significand %06x: %llu mismatching (n, estimate) cases over all numerators%s\n", X[i], bad,
           bad ? (all_ones(u) ? "  [all-ones denominator: IEEE-division case]" : "  [UNEXPECTED]") : "");
    if (!only) ok = 0;
  }
  /* (4) structured sweeps in both roles */
  for (int role_u = 1; role_u >= 0; --role_u) {
    unsigned long long bad = 0;
    int only = 1;
#pragma omp parallel for reduction(+ : bad) reduction(& : only) schedule(static)
    for (uint32_t m = 0; m < (1u << 23); ++m) {
      int o = 1;
      for (int j = 0; j < NS + 5; ++j) {
        const uint32_t other = j < NS ? structured[j] : ((m + (uint32_t)(j - NS) - 2u) & 0x7FFFFFu);
        const float u = role_u ? sig(m) : sig(other);
        const float n = role_u ? sig(other) : sig(m);
        bad += check_pair(n, u, &o);
        bad += check_pair(n * 0.5f, u, &o);
      }
      only &= o;
    }
    printf("(4) exhaustive %s x %d structured: %llu mismatching cases%s\n", role_u ? "denominators" : "numerators",
           NS + 5, bad, only ? (bad ? " (all with an all-ones denominator)" : "") : "  [UNEXPECTED: other denominators]");
    if (!only) ok = 0;
  }
  printf(ok ? "PROOF OK\n" : "PROOF FAILED\n");
  return ok ? 0 : 1;
}
