/* orc_gomath.c - Go's own math.Log and math.Lgamma, restated (TEST INFRASTRUCTURE, part of the oracle).
 *
 * The reference computes its BQSR estimates with Go's math package (filters/bqsr.go:561-613), which on amd64 is pure Go for Log
 * (math/log.go, a port of FreeBSD's /usr/src/lib/msun/src/e_log.c) and Lgamma (math/lgamma.go, a port of e_lgamma_r.c): their results
 * are NOT glibc's (glibc's log is a different, table-driven algorithm), so an oracle that leans on libm can differ from the reference
 * in the last bit of a log-likelihood - and that bit decides an argmax over 61 bins.  The algorithms below follow the Go sources
 * statement by statement: same constants (checked against the bit patterns the sources print next to them, orc_gomath_selfcheck),
 * same operation order, no fused multiply-add (-ffp-contract=off; Go does not fuse on amd64).
 * math.Exp (round 6, VERDICT r5 next #3c): restated from the PURE-GO function (math/exp.go: exp / expmulti, a port of FreeBSD's e_exp.c) - what
 * Go runs on every architecture without an assembly kernel and the reading of the reference that is defined to the last bit.  CAVEAT: on
 * amd64 Go dispatches math.Exp to an assembly kernel (math/exp_amd64.s: another argument reduction and a Remez polynomial evaluated with
 * FMA where the CPU has it), so an elPrep binary built for amd64 can differ from this function in the last bit of exp - it feeds math.Pow's
 * fractional part (qualities that are not integers: none on the paths the tests reach - phred / -10 of an integer quality has a fractional
 * part, but every consumer rounds the result) and the library-size estimate's printed text.  arm64 / ppc64 / s390x builds fuse the
 * multiply-adds of the pure-Go function; this restatement is the unfused reading (-ffp-contract=off).  Only tests may call this file. */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "orc.h"

static const double Ln2Hi = 6.93147180369123816490e-01, /* 3fe62e42 fee00000 */
    Ln2Lo = 1.90821492927058770002e-10,                  /* 3dea39ef 35793c76 */
    L1 = 6.666666666666735130e-01,                       /* 3FE55555 55555593 */
    L2 = 3.999999999940941908e-01,                       /* 3FD99999 9997FA04 */
    L3 = 2.857142874366239149e-01,                       /* 3FD24924 94229359 */
    L4 = 2.222219843214978396e-01,                       /* 3FCC71C5 1D8E78AF */
    L5 = 1.818357216161805012e-01,                       /* 3FC74664 96CB03DE */
    L6 = 1.531383769920937332e-01,                       /* 3FC39A09 D078C69F */
    L7 = 1.479819860511658591e-01;                       /* 3FC2F112 DF3E5244 */
#define GO_SQRT2 1.41421356237309504880168872420969807856967187537694807317667974

/* math/log.go: func log(x float64) float64, for finite x > 0 (the callers never pass anything else; 0 -> -Inf, < 0 -> NaN as in Go) */
double orc_go_log(double x) {
  if (x != x || x == INFINITY) return x;
  if (x < 0) return NAN;
  if (x == 0) return -INFINITY;
  int ki;
  double f1 = frexp(x, &ki);
  if (f1 < GO_SQRT2 / 2) { f1 *= 2; ki--; }
  const double f = f1 - 1, k = (double)ki;
  const double s = f / (2 + f), s2 = s * s, s4 = s2 * s2;
  const double t1 = s2 * (L1 + s4 * (L3 + s4 * (L5 + s4 * L7)));
  const double t2 = s4 * (L2 + s4 * (L4 + s4 * L6));
  const double R = t1 + t2, hfsq = 0.5 * f * f;
  return k * Ln2Hi - ((hfsq - (s * (hfsq + R) + k * Ln2Lo)) - f);
}

/* math/exp.go: func exp(x float64) float64 (the pure-Go function) */
static double go_expmulti(double hi, double lo, int k) {
  static const double P1 = 1.66666666666666657415e-01, /* 0x3FC55555; 0x55555555 */
      P2 = -2.77777777770155933842e-03,                /* 0xBF66C16C; 0x16BEBD93 */
      P3 = 6.61375632143793436117e-05,                 /* 0x3F11566A; 0xAF25DE2C */
      P4 = -1.65339022054652515390e-06,                /* 0xBEBBBD41; 0xC5D26BF1 */
      P5 = 4.13813679705723846039e-08;                 /* 0x3E663769; 0x72BEA4D0 */
  const double r = hi - lo;
  const double t = r * r;
  const double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  const double y = 1 - ((lo - (r * c) / (2 - c)) - hi);
  return ldexp(y, k);
}
double orc_go_exp(double x) {
  static const double Log2e = 1.44269504088896338700e+00, Overflow = 7.09782712893383973096e+02, Underflow = -7.45133219101941108420e+02,
                      NearZero = 1.0 / (1 << 28);
  if (x != x || x == INFINITY) return x;
  if (x == -INFINITY) return 0;
  if (x > Overflow) return INFINITY;
  if (x < Underflow) return 0;
  if (-NearZero < x && x < NearZero) return 1 + x;
  int k = 0;
  if (x < 0) k = (int)(Log2e * x - 0.5);
  else if (x > 0) k = (int)(Log2e * x + 0.5);
  const double hi = x - (double)k * Ln2Hi;
  const double lo = (double)k * Ln2Lo;
  return go_expmulti(hi, lo, k);
}

static const double lgamA[12] = {7.72156649015328655494e-02, 3.22467033424113591611e-01, 6.73523010531292681824e-02, 2.05808084325167332806e-02,
                                 7.38555086081402883957e-03, 2.89051383673415629091e-03, 1.19270763183362067845e-03, 5.10069792153511336608e-04,
                                 2.20862790713908385557e-04, 1.08011567247583939954e-04, 2.52144565451257326939e-05, 4.48640949618915160150e-05};
static const double lgamR[7] = {1.0, 1.39200533467621045958e+00, 7.21935547567138069525e-01, 1.71933865632803078993e-01, 1.86459191715652901344e-02,
                                7.77942496381893596434e-04, 7.32668430744625636189e-06};
static const double lgamS[7] = {-7.72156649015328655494e-02, 2.14982415960608852501e-01, 3.25778796408930981787e-01, 1.46350472652464452805e-01,
                                2.66422703033638609560e-02, 1.84028451407337715652e-03, 3.19475326584100867617e-05};
static const double lgamT[15] = {4.83836122723810047042e-01,  -1.47587722994593911752e-01, 6.46249402391333854778e-02,  -3.27885410759859649565e-02,
                                 1.79706750811820387126e-02,  -1.03142241298341437450e-02, 6.10053870246291332635e-03,  -3.68452016781138256760e-03,
                                 2.25964780900612472250e-03,  -1.40346469989232843813e-03, 8.81081882437654011382e-04,  -5.38595305356740546715e-04,
                                 3.15632070903625950361e-04,  -3.12754168375120860518e-04, 3.35529192635519073543e-04};
static const double lgamU[6] = {-7.72156649015328655494e-02, 6.32827064025093366517e-01, 1.45492250137234768737e+00, 9.77717527963372745603e-01,
                                2.28963728064692451092e-01,  1.33810918536787660377e-02};
static const double lgamV[6] = {1.0, 2.45597793713041134822e+00, 2.12848976379893395361e+00, 7.69285150456672783825e-01, 1.04222645593369134254e-01,
                                3.21709242282423911810e-03};
static const double lgamW[7] = {4.18938533204672725052e-01,  8.33333333333329678849e-02, -2.77777777728775536470e-03, 7.93650558643019558500e-04,
                                -5.95187557450339963135e-04, 8.36339918996282139126e-04, -1.63092934096575273989e-03};

/* math/lgamma.go: func Lgamma(x float64) (lgamma float64, sign int), for x > 0.  The reference passes counts + 1 (bqsr.go:598-606):
 * integers >= 1, i.e. the branches x == 1 || x == 2, 2 < x < 8 with y == 0 and x >= 8 - those are what the parity rests on and what
 * tests/test_gomath.py checks against 40-digit values.  The branches below 2 are restated for completeness; their interval bounds are
 * written as the Go source writes them (Ymin +- 0.27), which puts [1.1916, 1.2316) and [0.1916, 0.2316) on the polynomial in x - Tc
 * where FreeBSD's e_lgamma_r.c (bounds 1.2316 / 0.2316) uses the rational in x - 1: there this function is ~5e-15 off the true value,
 * as the Go function is if the bounds are remembered right - no Go toolchain here to confirm, no caller that could notice. */
double orc_go_lgamma(double x) {
  const double Ymin = 1.461632144968362245, Two58 = 288230376151711744.0, Tiny = 1.0 / 1180591620717411303424.0 /* 1 / (1 << 70) */;
  const double Tc = 1.46163214496836224576e+00, Tf = -1.21486290535849611461e-01, Tt = -3.63867699703950536541e-18;
  if (x != x) return x;
  if (x == INFINITY) return x;
  if (x <= 0) return INFINITY; /* 0 -> +Inf as in Go; negative arguments (the reflection formula) are not restated: never used */
  if (x < Tiny) return -orc_go_log(x);
  double lg;
  if (x == 1 || x == 2) return 0;
  if (x < 2) {
    double y;
    int i;
    if (x <= 0.9) {
      lg = -orc_go_log(x);
      if (x >= (Ymin - 1 + 0.27)) { y = 1 - x; i = 0; }
      else if (x >= (Ymin - 1 - 0.27)) { y = x - (Tc - 1); i = 1; }
      else { y = x; i = 2; }
    } else {
      lg = 0;
      if (x >= (Ymin + 0.27)) { y = 2 - x; i = 0; }
      else if (x >= (Ymin - 0.27)) { y = x - Tc; i = 1; }
      else { y = x - 1; i = 2; }
    }
    if (i == 0) {
      const double z = y * y;
      const double p1 = lgamA[0] + z * (lgamA[2] + z * (lgamA[4] + z * (lgamA[6] + z * (lgamA[8] + z * lgamA[10]))));
      const double p2 = z * (lgamA[1] + z * (+lgamA[3] + z * (lgamA[5] + z * (lgamA[7] + z * (lgamA[9] + z * lgamA[11])))));
      const double p = y * p1 + p2;
      lg += (p - 0.5 * y);
    } else if (i == 1) {
      const double z = y * y, w = z * y;
      const double p1 = lgamT[0] + w * (lgamT[3] + w * (lgamT[6] + w * (lgamT[9] + w * lgamT[12])));
      const double p2 = lgamT[1] + w * (lgamT[4] + w * (lgamT[7] + w * (lgamT[10] + w * lgamT[13])));
      const double p3 = lgamT[2] + w * (lgamT[5] + w * (lgamT[8] + w * (lgamT[11] + w * lgamT[14])));
      const double p = z * p1 - (Tt - w * (p2 + y * p3));
      lg += (Tf + p);
    } else {
      const double p1 = y * (lgamU[0] + y * (lgamU[1] + y * (lgamU[2] + y * (lgamU[3] + y * (lgamU[4] + y * lgamU[5])))));
      const double p2 = 1 + y * (lgamV[1] + y * (lgamV[2] + y * (lgamV[3] + y * (lgamV[4] + y * lgamV[5]))));
      lg += (-0.5 * y + p1 / p2);
    }
    return lg;
  }
  if (x < 8) {
    const int i = (int)x;
    const double y = x - (double)i;
    const double p = y * (lgamS[0] + y * (lgamS[1] + y * (lgamS[2] + y * (lgamS[3] + y * (lgamS[4] + y * (lgamS[5] + y * lgamS[6]))))));
    const double q = 1 + y * (lgamR[1] + y * (lgamR[2] + y * (lgamR[3] + y * (lgamR[4] + y * (lgamR[5] + y * lgamR[6])))));
    lg = 0.5 * y + p / q;
    double z = 1.0;
    switch (i) {
      case 7: z *= (y + 6); /* fallthrough */
      case 6: z *= (y + 5); /* fallthrough */
      case 5: z *= (y + 4); /* fallthrough */
      case 4: z *= (y + 3); /* fallthrough */
      case 3: z *= (y + 2); lg += orc_go_log(z);
    }
    return lg;
  }
  if (x < Two58) {
    const double t = orc_go_log(x), z = 1 / x, y = z * z;
    const double w = lgamW[0] + z * (lgamW[1] + y * (lgamW[2] + y * (lgamW[3] + y * (lgamW[4] + y * (lgamW[5] + y * lgamW[6])))));
    return (x - 0.5) * (t - 1) + w;
  }
  return x * (orc_go_log(x) - 1);
}

/* the bit patterns the Go / FreeBSD sources print next to the constants of log: a typing error in a decimal literal shows up here */
int orc_gomath_selfcheck(void) {
  static const uint64_t want[9] = {0x3fe62e42fee00000ull, 0x3dea39ef35793c76ull, 0x3FE5555555555593ull, 0x3FD999999997FA04ull, 0x3FD2492494229359ull,
                                   0x3FCC71C51D8E78AFull, 0x3FC7466496CB03DEull, 0x3FC39A09D078C69Full, 0x3FC2F112DF3E5244ull};
  const double have[9] = {Ln2Hi, Ln2Lo, L1, L2, L3, L4, L5, L6, L7};
  int bad = 0;
  for (int k = 0; k < 9; k++) {
    uint64_t b;
    memcpy(&b, &have[k], 8);
    if (b != want[k]) bad |= 1 << k;
  }
  return bad;
}
