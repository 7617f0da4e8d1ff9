/* Exhaustive check of csrc/stem_bf16.hip::div_const on the CPU: for the four constant divisors the stem pre-pass uses, the
 * three-operation quotient  q0 = x * RN(1/d);  r = fma(-d, q0, x);  q = fma(r, RN(1/d), q0)  equals the IEEE quotient x / d
 * for EVERY float x with 2^-40 <= x <= 512 (negative x follow by symmetry of round-to-nearest; 0 gives 0).
 * Build: gcc -O2 -mfma -ffp-contract=off  (hardware fma; prints one line per divisor, exit code = number of divisors that failed). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

int main(void) {
  const float ds[4] = {255.0f, 0.229f, 0.224f, 0.225f};
  int failed = 0;
  for (int k = 0; k < 4; ++k) {
    const float d = ds[k];
    const float y = 1.0f / d;
    uint32_t lo, hi;
    const float flo = ldexpf(1.0f, -40), fhi = 512.0f;
    memcpy(&lo, &flo, 4);
    memcpy(&hi, &fhi, 4);
    long long bad = 0, n = 0;
    for (uint32_t u = lo; u <= hi; ++u, ++n) {
      float x;
      memcpy(&x, &u, 4);
      volatile float ref = x / d;
      const float q0 = x * y;
      const float r = fmaf(-d, q0, x);
      const float q = fmaf(r, y, q0);
      if (q != ref) ++bad;
    }
    printf("d=%g values=%lld mismatches=%lld\n", d, n, bad);
    if (bad) ++failed;
  }
  return failed;
}
