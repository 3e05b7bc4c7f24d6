"""The strict march divides by 6 with a 2-operation sequence, q = RN(x c_hi + RN(x c_lo)) (csrc/march.hip: div6).
It must equal IEEE x / 6.0f for every f32 significand; checked exhaustively on the CPU with the
same fmaf sequence (one binade covers all significands, a few exponents cover the rest of the
range incl. negatives)."""
import os
import subprocess

SRC = r'''
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static float div6(float x){ const float c_hi = 0x1.555556p-3f, c_lo = -0x1.555556p-28f; return fmaf(x, c_hi, x*c_lo); }
int main(void){
  long bad = 0, n = 0;
  int exps[] = {-100, -60, -10, -1, 0, 1, 2, 3, 7, 40, 100};
  for (unsigned e = 0; e < sizeof(exps)/sizeof(exps[0]); ++e)
    for (uint32_t m = 0; m < (1u<<23); ++m) {
      uint32_t bits = ((uint32_t)(127 + exps[e]) << 23) | m; float x; memcpy(&x,&bits,4);
      volatile float want = x / 6.0f; float got = div6(x);
      if (got != want) { if (bad < 5) printf("mismatch x=%a got=%a want=%a\n", x, got, (float)want); ++bad; }
      volatile float wantn = (-x) / 6.0f; if (div6(-x) != wantn) ++bad;
      n += 2;
    }
  printf("checked %ld bad %ld\n", n, bad);
  return bad != 0;
}
'''


def test_div6_is_correctly_rounded(tmp_path):
    c = tmp_path / "div6.c"
    c.write_text(SRC)
    exe = tmp_path / "div6"
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-ffp-contract=off", str(c), "-o", str(exe), "-lm"])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "bad 0" in out.stdout
