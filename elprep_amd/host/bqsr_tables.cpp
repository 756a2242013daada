// bqsr_tables.cpp — host-side (float64) BQSR table finalisation, LUT tabulation and report text.
//
// C++ mirror of the Go host code that stays on the CPU in the drop-in design:
//   FinalizeBQSRTables / calculateEmpiricalQuality / calculateBayesianEstimateOfEmpiricalQuality (filters/bqsr.go:553-694),
//   initializeCombinedBQSRTable (:655-674), quantisation (:746-899), estimateHierarchicalBayesianQuality (:901-919),
//   the quality mapping of ApplyBQSR (:959-999), PrintBQSRTables (filters/print-bqsr.go:49-298).
//
// Go's math.Log, math.Lgamma and math.Exp are restated from the pure-Go sources (math/log.go, lgamma.go, exp.go: ports of FreeBSD's msun),
// math.Log10 and math.Pow structurally on top of them (log2(x) * Ln2 / Ln10; frexp / ldexp exponentiation): libm only supplies frexp,
// ldexp, modf and sqrt, which are exact.  (math.Exp on amd64 builds of Go is an assembly kernel, not the pure-Go function: see go_exp.)
// Where the reference iterates a Go map (initializeCombinedBQSRTable) the (rg, qual) entries are visited in ascending qual.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/elprep_host.h"

namespace {

constexpr int NQ = 94, NX = 16;

// math.Log as Go computes it on amd64 (math/log.go, a port of FreeBSD's e_log.c; glibc's log is another algorithm and differs in
// the last bit for one argument in ~300): reduction x = 2^k (1 + f), sqrt(2)/2 < 1 + f < sqrt(2); log(1 + f) = f - s (f - R) with
// s = f / (2 + f) and R a degree-14 polynomial in s; same constants, same operation order, no contraction (-ffp-contract=off)
double go_log(double x) {
  constexpr double Ln2Hi = 6.93147180369123816490e-01, Ln2Lo = 1.90821492927058770002e-10, L1 = 6.666666666666735130e-01,
                   L2 = 3.999999999940941908e-01, L3 = 2.857142874366239149e-01, L4 = 2.222219843214978396e-01, L5 = 1.818357216161805012e-01,
                   L6 = 1.531383769920937332e-01, L7 = 1.479819860511658591e-01,
                   Sqrt2 = 1.41421356237309504880168872420969807856967187537694807317667974;
  if (x != x || x == HUGE_VAL) return x;
  if (x < 0) return std::nan("");
  if (x == 0) return -HUGE_VAL;
  int ki;
  double f1 = std::frexp(x, &ki);
  if (f1 < Sqrt2 / 2) { f1 *= 2; ki--; }
  const double f = f1 - 1, k = double(ki);
  const double s = f / (2 + f), s2 = s * s, s4 = s2 * s2;
  const double t1 = s2 * (L1 + s4 * (L3 + s4 * (L5 + s4 * L7)));
  const double t2 = s4 * (L2 + s4 * (L4 + s4 * L6));
  const double R = t1 + t2, hfsq = 0.5 * f * f;
  return k * Ln2Hi - ((hfsq - (s * (hfsq + R) + k * Ln2Lo)) - f);
}
inline double go_log2(double x) {
  int e;
  double frac = std::frexp(x, &e);
  if (frac == 0.5) return double(e - 1);
  return go_log(frac) * 1.44269504088896340735992468100189213742664595415298593413 + double(e);
}
inline double go_log10(double x) { return go_log2(x) * 0.30102999566398119521373889472449302676818988146210854131; }

// math.Lgamma for the arguments the reference passes (counts + 1: integers >= 1, bqsr.go:598-606), math/lgamma.go: 0 at 1 and 2; for
// 3 .. 7 the logarithm of (x - 1)! built as the product (y + 2) .. (y + 6) with y = 0; from 8 on Stirling's series in 1 / x
double go_lgamma_count(double x) {
  constexpr double W0 = 4.18938533204672725052e-01, W1 = 8.33333333333329678849e-02, W2 = -2.77777777728775536470e-03,
                   W3 = 7.93650558643019558500e-04, W4 = -5.95187557450339963135e-04, W5 = 8.36339918996282139126e-04,
                   W6 = -1.63092934096575273989e-03, Two58 = 288230376151711744.0;
  if (x == 1 || x == 2) return 0;
  if (x < 8) {
    const int i = int(x);
    const double y = x - double(i);  // 0 for a count; the polynomial part p / q of the source is then exactly 0
    double lg = 0.5 * y;
    double z = 1.0;
    switch (i) {
      case 7: z *= (y + 6); [[fallthrough]];
      case 6: z *= (y + 5); [[fallthrough]];
      case 5: z *= (y + 4); [[fallthrough]];
      case 4: z *= (y + 3); [[fallthrough]];
      case 3: z *= (y + 2); lg += go_log(z);
    }
    return lg;
  }
  if (x < Two58) {
    const double t = go_log(x), z = 1 / x, y = z * z;
    const double w = W0 + z * (W1 + y * (W2 + y * (W3 + y * (W4 + y * (W5 + y * W6)))));
    return (x - 0.5) * (t - 1) + w;
  }
  return x * (go_log(x) - 1);
}

// math.Exp as the pure-Go function computes it (math/exp.go: exp / expmulti, FreeBSD's e_exp.c): k = round(x / ln 2), r = x - k ln 2 in two
// pieces, a degree-5 polynomial in r^2, scaled by 2^k; no contraction.  An amd64 build of Go dispatches to math/exp_amd64.s instead
// (another reduction, FMA where the CPU has it) and can differ in the last bit: the reference itself is only bit-defined in this reading.
double go_exp(double x) {
  constexpr double Ln2Hi = 6.93147180369123816490e-01, Ln2Lo = 1.90821492927058770002e-10, Log2e = 1.44269504088896338700e+00,
                   Overflow = 7.09782712893383973096e+02, Underflow = -7.45133219101941108420e+02, NearZero = 1.0 / (1 << 28),
                   P1 = 1.66666666666666657415e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                   P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  if (x != x || x == HUGE_VAL) return x;
  if (x == -HUGE_VAL) return 0;
  if (x > Overflow) return HUGE_VAL;
  if (x < Underflow) return 0;
  if (-NearZero < x && x < NearZero) return 1 + x;
  int k = 0;
  if (x < 0) k = int(Log2e * x - 0.5);
  else if (x > 0) k = int(Log2e * x + 0.5);
  const double hi = x - double(k) * Ln2Hi, lo = double(k) * Ln2Lo;
  const double r = hi - lo, t = r * r;
  const double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  const double y = 1 - ((lo - (r * c) / (2 - c)) - hi);
  return std::ldexp(y, k);
}

double go_pow(double x, double y) {  // finite x > 0
  if (y == 0 || x == 1) return 1;
  if (y == 1) return x;
  if (y == 0.5) return std::sqrt(x);
  if (y == -0.5) return 1 / std::sqrt(x);
  double yi;
  double yf = std::modf(std::fabs(y), &yi);
  double a1 = 1.0;
  int ae = 0;
  if (yf != 0) {
    if (yf > 0.5) { yf--; yi++; }
    a1 = go_exp(yf * go_log(x));
  }
  int xe;
  double x1 = std::frexp(x, &xe);
  for (long long i = (long long)yi; i != 0; i >>= 1) {
    if (xe < -(1 << 12) || (1 << 12) < xe) { ae += xe; break; }
    if (i & 1) { a1 *= x1; ae += xe; }
    x1 *= x1;
    xe <<= 1;
    if (x1 < .5) { x1 += x1; xe--; }
  }
  if (y < 0) { a1 = 1 / a1; ae = -ae; }
  return std::ldexp(a1, ae);
}

inline double q_to_err(double phred) { return go_pow(10, phred / -10); }
inline double q_to_prob(double phred) { return 1 - go_pow(10, phred / -10); }

const double kPrior[21] = {-0.045757490560675115, -0.9143464543671788, -3.5201133457866898, -7.863058164819208, -13.943180911464733,
                           -21.760481585723266,   -31.314960187594806, -42.606616717079355, -55.63545117417691, -70.40146355888747,
                           -86.90465387121104,    -105.14502211114761, -125.1225682786972,  -146.83729237385978, -170.2891943966354,
                           -195.47827434702398,   -222.4045322250256,  -251.06796803064023, -281.46858176386786, -313.60637342472336,
                           -1.7976931348623157e308};

inline double log10_gamma(long long n) {
  return go_lgamma_count(double(n)) * 0.43429448190325182765112891891660508229439700580366656611445378316586464920887077;
}

// log10(1 - 10^(-Q/10)) for Q = 1..60: the only transcendental work of the likelihood that depends on the bin alone
struct Log10MinP {
  double v[61];
  Log10MinP() {
    v[0] = 0.0;
    for (int i = 1; i <= 60; i++) v[i] = go_log10(1.0 - go_pow(10, double(i) / -10.0));
  }
};
const Log10MinP kLog10MinP;

// calculateBayesianEstimateOfEmpiricalQuality, bqsr.go:623-642, in three pieces that share work between the entries of a table row
// and between the two priors every entry is evaluated with (its reported quality in FinalizeBQSRTables, the row's conditional
// estimate in ApplyBQSR's hierarchy).  The binomial coefficient term (three Lgamma calls) does not depend on the bin, log10(1-p)
// only on the bin, the prior only on (bin, prior): every floating-point operation and its order is the reference's
// (log10QualEmpiricalLikelihood :598-613), so the argmax is unchanged.
constexpr int NBIN = 61, NBIN_PAD = 64;
struct BinConst {
  alignas(64) double log10p[NBIN_PAD], log10minp[NBIN_PAD];
  BinConst() {
    for (int i = 0; i < NBIN_PAD; i++) { log10p[i] = 0.0; log10minp[i] = 0.0; }
    for (int i = 0; i < NBIN; i++) { log10p[i] = double(i) / -10.0; log10minp[i] = kLog10MinP.v[i]; }
  }
};
const BinConst kBin;

struct Curve { alignas(64) double v[NBIN_PAD]; };

inline void clamp_counts(long long &obs, long long &mism) {
  const long long kMax = 2147483647LL - 1;
  if (obs > kMax) {
    mism = (long long)std::round(double(mism) * (double(kMax) / double(obs)));
    obs = kMax;
  }
}
// the likelihood of every bin for one table entry (obs > 0 after the clamp)
__attribute__((target_clones("avx2", "default"))) void like_curve(long long obs, long long mism, Curve &out) {
  const double c = log10_gamma(obs + 1) - log10_gamma(mism + 1) - log10_gamma(obs - mism + 1);
  const double dm = double(mism), dn = double(obs - mism);
  for (int i = 0; i < NBIN_PAD; i++) out.v[i] = c + kBin.log10p[i] * dm + kBin.log10minp[i] * dn;
  out.v[0] = -DBL_MAX;  // log10p == 0
}
// the prior of every bin: the same for all entries of a (covariate, quality) row
inline void prior_curve(double prior, Curve &out) {
  for (int i = 0; i < NBIN; i++) {
    int d = int(double(i) - prior);
    if (d < 0) d = -d;
    if (d > 20) d = 20;
    out.v[i] = kPrior[d];
  }
  for (int i = NBIN; i < NBIN_PAD; i++) out.v[i] = 0.0;
}
// the first bin with the largest posterior (:630-641: `if best < post`), capped as calculateEmpiricalQuality caps it (:644-649)
// = per lane of four (bins 4 g + lane) the first largest posterior and its group, then the lanes' best with ties to the smaller bin;
// two priors at a time (independent dependency chains over the same likelihoods)
typedef double v4d __attribute__((vector_size(32)));
typedef long long v4i __attribute__((vector_size(32)));
inline uint8_t argmax_lanes(const v4d &m4, const v4i &g4, double last) {
  double m = -DBL_MAX;
  int arg = 0;  // no bin ever passes `best < post`: bin 0
  for (int k = 0; k < 4; k++) {
    const int bin = int(g4[k]) * 4 + k;
    const bool take = m4[k] > m || (m4[k] == m && m4[k] > -DBL_MAX && bin < arg);
    m = take ? m4[k] : m;
    arg = take ? bin : arg;
  }
  if (last > m) arg = NBIN - 1;  // bin 60, alone in its group
  return uint8_t(arg < 93 ? arg : 93);
}
__attribute__((target_clones("avx2", "default"))) void argmax_posterior2(const Curve &prior_a, const Curve &prior_b, const Curve &like,
                                                                         uint8_t &a, uint8_t &b) {
  const v4d *pa = reinterpret_cast<const v4d *>(prior_a.v), *pb = reinterpret_cast<const v4d *>(prior_b.v);
  const v4d *lv = reinterpret_cast<const v4d *>(like.v);
  v4d ma = {-DBL_MAX, -DBL_MAX, -DBL_MAX, -DBL_MAX}, mb = ma;
  v4i ga = {0, 0, 0, 0}, gb = ga;
  for (int g = 0; g < NBIN_PAD / 4 - 1; g++) {
    const v4d l = lv[g], qa = pa[g] + l, qb = pb[g] + l;
    const v4i ua = qa > ma, ub = qb > mb;
    const v4i gg = {g, g, g, g};
    ma = ua ? qa : ma; ga = ua ? gg : ga;
    mb = ub ? qb : mb; gb = ub ? gg : gb;
  }
  a = argmax_lanes(ma, ga, prior_a.v[NBIN - 1] + like.v[NBIN - 1]);
  b = argmax_lanes(mb, gb, prior_b.v[NBIN - 1] + like.v[NBIN - 1]);
}
inline uint8_t argmax_posterior(const Curve &prior, const Curve &like) {
  uint8_t a, b;
  argmax_posterior2(prior, prior, like, a, b);
  return a;
}
// one entry, one prior (the q table, the combined entries: a handful per call)
uint8_t empirical(long long obs, long long mism, double prior) {  // bqsr.go:644-649: obs + 2, mism + 1
  obs += 2; mism += 1;
  clamp_counts(obs, mism);
  Curve like, pr;
  like_curve(obs, mism, like);
  prior_curve(prior, pr);
  return argmax_posterior(pr, like);
}
// one entry, the two priors of its row
inline void empirical2(long long obs, long long mism, const Curve &prior_a, const Curve &prior_b, uint8_t &a, uint8_t &b) {
  obs += 2; mism += 1;
  clamp_counts(obs, mism);
  Curve like;
  like_curve(obs, mism, like);
  argmax_posterior2(prior_a, prior_b, like, a, b);
}

// rows are independent (the results do not depend on the split): a small pool of workers that lives as long as the library,
// so that a call costs two condition-variable round trips instead of thread creations; rows are handed out one at a time
class RowPool {
 public:
  static RowPool &get() { static RowPool p; return p; }
  void run(size_t n, const std::function<void(size_t)> &f) {
    if (workers_.empty() || n < 2) { for (size_t i = 0; i < n; i++) f(i); return; }
    std::lock_guard<std::mutex> call(call_mu_);  // one parallel region at a time
    {
      std::lock_guard<std::mutex> lk(mu_);
      f_ = &f; n_ = n; next_.store(0); running_ = workers_.size(); gen_++;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [&] { return running_ == 0; });
    f_ = nullptr;
  }

 private:
  RowPool() {
    // ELP_HOST_THREADS=<n>: threads per process for the table path (a host that runs one process per GPU on one node divides the
    // cores among them: eight processes that each wake fifteen workers at the same moment - right behind the all-reduce - on sixteen
    // cores make each other wait); default: the hardware's threads, at most 16
    unsigned hw = std::thread::hardware_concurrency();
    if (const char *e = std::getenv("ELP_HOST_THREADS")) { const long v = std::strtol(e, nullptr, 10); if (v >= 1) hw = (unsigned)std::min<long>(v, 64); }
    const unsigned nt = hw > 1 ? std::min(hw, 16u) - 1 : 0;  // the caller works too
    for (unsigned k = 0; k < nt; k++) workers_.emplace_back([this] { loop(); });
  }
  ~RowPool() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
    cv_.notify_all();
    for (auto &t : workers_) t.join();
  }
  void work() { for (size_t i; (i = next_.fetch_add(1)) < n_;) (*f_)(i); }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
        if (stop_) return;
        seen = gen_;
      }
      work();
      std::lock_guard<std::mutex> lk(mu_);
      if (--running_ == 0) done_.notify_one();
    }
  }
  std::vector<std::thread> workers_;
  std::mutex mu_, call_mu_;
  std::condition_variable cv_, done_;
  const std::function<void(size_t)> *f_ = nullptr;
  size_t n_ = 0, running_ = 0;
  std::atomic<size_t> next_{0};
  uint64_t gen_ = 0;
  bool stop_ = false;
};
// `small`: the whole job is well under a millisecond on one core (one or two read groups with a handful of qualities: what a binned
// sequencer run gives) - run it on the caller's thread: no wake-ups, and no waiting for a worker the scheduler took off its core (on a
// box with busy neighbours a 0.5 ms job spread over 16 threads was seen to take 6 ms)
template <class F>
void parallel_rows(size_t n, F f, bool small = false) {
  if (small) { for (size_t i = 0; i < n; i++) f(i); return; }
  RowPool::get().run(n, std::function<void(size_t)>(f));
}
inline bool small_job(const std::vector<uint8_t> &live, int n_cov) {
  size_t rows = 0;
  for (uint8_t l : live) rows += l;
  return n_cov <= 2 && rows <= 12;
}

struct Interval { int next; double rate; long long nobs, leaf, nerr; };
inline double err_rate(long long nobs, long long nerr) { return nobs == 0 ? 0.0 : double(nerr + 1) / double(nobs + 1); }

}  // namespace

// a dense array whose untouched pages stay untouched: calloc hands out zero pages that cost nothing until they are written.  Most
// (covariate, quality) rows of the cycle and context tables are empty (the qualities a sequencer never reports, the cycles beyond the
// read length): with 16 read groups the tables are 24 MB of which 1 MB is used
struct LazyZero {
  long long *p = nullptr;
  size_t n = 0;
  LazyZero() = default;
  LazyZero(const LazyZero &) = delete;
  LazyZero &operator=(const LazyZero &) = delete;
  ~LazyZero() { std::free(p); }
  bool alloc(size_t count) { std::free(p); n = count; p = static_cast<long long *>(std::calloc(count ? count : 1, sizeof(long long))); return p != nullptr; }
  long long &operator[](size_t i) { return p[i]; }
  const long long &operator[](size_t i) const { return p[i]; }
  size_t size() const { return n; }
};

struct elp_bqsr_tables {
  int n_cov, max_cycle, ncyc;
  std::vector<long long> q;                 // {obs, mism} pairs
  LazyZero c, x;
  std::vector<uint8_t> live;                // per (cov, quality) row: some cycle or context entry is not zero
  std::vector<uint8_t> qe, ce, xe;          // EmpiricalQuality, 255 = absent
  std::vector<uint8_t> ce_cond, xe_cond;    // the same entries under the prior of ApplyBQSR's hierarchy (cond[row]), for the LUT
  std::vector<double> cond;                 // per (cov, quality): deltaQReported + deltaQ + epsilon (:959-975)
  std::vector<double> rep;                  // combined reportedQuality per cov
  std::vector<uint8_t> cemp, present;
  std::vector<long long> cobs, cmism;
  bool finalized = false;

  size_t qi(int cov, int qual) const { return size_t(cov) * NQ + qual; }
};

extern "C" {

// adds the rows of ct / xt that hold anything into the tables (rows are independent: one task each)
static void add_rows(elp_bqsr_tables *t, const int64_t *ct, const int64_t *xt) {
  const size_t nq = size_t(t->n_cov) * NQ, cw = size_t(t->ncyc) * 2, xw = size_t(NX) * 2;
  parallel_rows(nq, [&](size_t row) {
    const long long *cs = ct ? reinterpret_cast<const long long *>(ct) + row * cw : nullptr;
    const long long *xs = xt ? reinterpret_cast<const long long *>(xt) + row * xw : nullptr;
    bool any = false;
    for (size_t k = 0; cs && k < cw && !any; k++) any = cs[k] != 0;
    for (size_t k = 0; xs && k < xw && !any; k++) any = xs[k] != 0;
    if (!any) return;
    if (cs) { long long *d = &t->c[row * cw]; for (size_t k = 0; k < cw; k++) d[k] += cs[k]; }
    if (xs) { long long *d = &t->x[row * xw]; for (size_t k = 0; k < xw; k++) d[k] += xs[k]; }
    t->live[row] = 1;
  }, t->n_cov <= 2);
}

elp_bqsr_tables *elp_bqsr_tables_new(int n_cov, int max_cycle, const int64_t *qt, const int64_t *ct, const int64_t *xt) {
  if (n_cov < 0 || max_cycle < 1) return nullptr;
  auto *t = new elp_bqsr_tables();
  t->n_cov = n_cov; t->max_cycle = max_cycle; t->ncyc = 2 * max_cycle + 1;
  const size_t nq = size_t(n_cov) * NQ;
  static_assert(sizeof(long long) == sizeof(int64_t), "tables are kept as long long");
  if (qt) t->q.assign(reinterpret_cast<const long long *>(qt), reinterpret_cast<const long long *>(qt) + nq * 2);
  else t->q.assign(nq * 2, 0);
  t->live.assign(nq, 0);
  if (!t->c.alloc(nq * t->ncyc * 2) || !t->x.alloc(nq * NX * 2)) { delete t; return nullptr; }
  add_rows(t, ct, xt);
  return t;
}
// The same object from the rows of the qualities that CAN hold anything (round 5): with many read groups the dense tables are tens of
// megabytes of zeros around a few hundred rows - n_cov x 94 rows of 16 KB, of which n_cov x (qualities in the read set) are used - and
// carrying them over PCIe and scanning them was most of the host's time per step.  quals: ascending quality values; q_rows
// [n_cov][n_quals][2], c_rows [n_cov][n_quals][2*max_cycle+1][2], x_rows [n_cov][n_quals][16][2]; every other row is empty.
elp_bqsr_tables *elp_bqsr_tables_new_rows(int n_cov, int max_cycle, const uint8_t *quals, int n_quals, const int64_t *q_rows, const int64_t *c_rows,
                                          const int64_t *x_rows) {
  if (n_cov < 0 || max_cycle < 1 || n_quals < 0 || n_quals > NQ || (n_quals && (!quals || !q_rows || !c_rows || !x_rows))) return nullptr;
  for (int k = 0; k < n_quals; k++)
    if (quals[k] >= NQ || (k && quals[k] <= quals[k - 1])) return nullptr;
  auto *t = new elp_bqsr_tables();
  t->n_cov = n_cov; t->max_cycle = max_cycle; t->ncyc = 2 * max_cycle + 1;
  const size_t nq = size_t(n_cov) * NQ, cw = size_t(t->ncyc) * 2, xw = size_t(NX) * 2;
  t->q.assign(nq * 2, 0);
  t->live.assign(nq, 0);
  if (!t->c.alloc(nq * t->ncyc * 2) || !t->x.alloc(nq * NX * 2)) { delete t; return nullptr; }
  parallel_rows(size_t(n_cov) * size_t(n_quals), [&](size_t k) {
    const size_t cv = k / size_t(n_quals), row = cv * NQ + quals[k % size_t(n_quals)];
    const long long *qs = reinterpret_cast<const long long *>(q_rows) + k * 2;
    const long long *cs = reinterpret_cast<const long long *>(c_rows) + k * cw;
    const long long *xs = reinterpret_cast<const long long *>(x_rows) + k * xw;
    t->q[2 * row] = qs[0]; t->q[2 * row + 1] = qs[1];
    bool any = false;
    for (size_t j = 0; j < cw && !any; j++) any = cs[j] != 0;
    for (size_t j = 0; j < xw && !any; j++) any = xs[j] != 0;
    if (!any) return;
    std::memcpy(&t->c[row * cw], cs, cw * sizeof(long long));
    std::memcpy(&t->x[row * xw], xs, xw * sizeof(long long));
    t->live[row] = 1;
  }, n_cov <= 2);
  return t;
}
void elp_bqsr_tables_free(elp_bqsr_tables *t) { delete t; }

int elp_bqsr_tables_merge(elp_bqsr_tables *t, const int64_t *qt, const int64_t *ct, const int64_t *xt) {
  if (!t || !qt || !ct || !xt) return -1;
  for (size_t i = 0; i < t->q.size(); i++) t->q[i] += qt[i];
  add_rows(t, ct, xt);
  t->finalized = false;
  return 0;
}

int elp_bqsr_tables_finalize(elp_bqsr_tables *t) {
  if (!t) return -1;
  const size_t nq = size_t(t->n_cov) * NQ;
  const int ncyc = t->ncyc;
  t->qe.assign(nq, 255); t->ce.assign(nq * ncyc, 255); t->xe.assign(nq * NX, 255);
  t->ce_cond.assign(nq * ncyc, 255); t->xe_cond.assign(nq * NX, 255); t->cond.assign(nq, 0.0);
  for (size_t i = 0; i < nq; i++)
    if (t->q[2 * i] > 0) t->qe[i] = empirical(t->q[2 * i], t->q[2 * i + 1], double(i % NQ));
  t->rep.assign(t->n_cov, 0.0); t->cemp.assign(t->n_cov, 0); t->present.assign(t->n_cov, 0);
  t->cobs.assign(t->n_cov, 0); t->cmism.assign(t->n_cov, 0);
  for (int cv = 0; cv < t->n_cov; cv++) {
    for (int ql = 0; ql < NQ; ql++) {
      const long long obs = t->q[2 * t->qi(cv, ql)], mism = t->q[2 * t->qi(cv, ql) + 1];
      if (obs <= 0) continue;
      if (t->present[cv]) {
        const double sum = double(t->cobs[cv]) * q_to_err(t->rep[cv]) + double(obs) * q_to_err(double(ql));
        t->cobs[cv] += obs; t->cmism[cv] += mism;
        t->rep[cv] = -10 * go_log10(sum / double(t->cobs[cv]));
      } else {
        t->present[cv] = 1; t->rep[cv] = double(ql); t->cobs[cv] = obs; t->cmism[cv] = mism;
      }
    }
    if (!t->present[cv]) continue;
    t->cemp[cv] = empirical(t->cobs[cv], t->cmism[cv], t->rep[cv]);
    // the prior under which ApplyBQSR's hierarchy reads the cycle and context entries of a row (:959-975; globalQualityScorePrior = -1)
    const double epsilon = t->rep[cv], d_global = double(t->cemp[cv]) - epsilon;
    for (int ql = 0; ql < NQ; ql++) {
      const size_t qi = t->qi(cv, ql);
      double d_reported = 0;
      if (t->q[2 * qi] > 0) d_reported = double(empirical(t->q[2 * qi], t->q[2 * qi + 1], d_global + epsilon)) - d_global - epsilon;
      t->cond[qi] = d_reported + d_global + epsilon;
    }
  }
  // the cycle and context entries: a task = one eighth of a (covariate, quality) row's cycles (+ the row's contexts with the first)
  constexpr int kParts = 8;
  const int part_len = (ncyc + kParts - 1) / kParts;
  parallel_rows(nq * kParts, [&](size_t task) {
    const size_t row = task / kParts;
    const int part = int(task % kParts);
    const int lo = part * part_len, hi = std::min(ncyc, lo + part_len);
    if (!t->live[row]) return;
    bool any = false;
    for (int cy = lo; cy < hi && !any; cy++) any = t->c[2 * (row * ncyc + cy)] > 0;
    if (part == 0) for (int cx = 0; cx < NX && !any; cx++) any = t->x[2 * (row * NX + cx)] > 0;
    if (!any) return;
    Curve pr_fin, pr_cond;
    prior_curve(double(row % NQ), pr_fin);
    prior_curve(t->cond[row], pr_cond);
    for (int cy = lo; cy < hi; cy++) {
      const size_t i = row * ncyc + cy;
      if (t->c[2 * i] > 0) empirical2(t->c[2 * i], t->c[2 * i + 1], pr_fin, pr_cond, t->ce[i], t->ce_cond[i]);
    }
    if (part == 0)
      for (int cx = 0; cx < NX; cx++) {
        const size_t i = row * NX + cx;
        if (t->x[2 * i] > 0) empirical2(t->x[2 * i], t->x[2 * i + 1], pr_fin, pr_cond, t->xe[i], t->xe_cond[i]);
      }
  }, small_job(t->live, t->n_cov));
  t->finalized = true;
  return 0;
}

int elp_bqsr_tables_empirical(const elp_bqsr_tables *t, uint8_t *qe, uint8_t *ce, uint8_t *xe) {
  if (!t || !t->finalized) return -1;
  std::memcpy(qe, t->qe.data(), t->qe.size()); std::memcpy(ce, t->ce.data(), t->ce.size()); std::memcpy(xe, t->xe.data(), t->xe.size());
  return 0;
}
int elp_bqsr_tables_combined(const elp_bqsr_tables *t, double *rep, uint8_t *emp, int64_t *obs, int64_t *mism, uint8_t *present) {
  if (!t || !t->finalized) return -1;
  for (int c = 0; c < t->n_cov; c++) { rep[c] = t->rep[c]; emp[c] = t->cemp[c]; obs[c] = t->cobs[c]; mism[c] = t->cmism[c]; present[c] = t->present[c]; }
  return 0;
}

int elp_bqsr_tables_quantize(const elp_bqsr_tables *t, int levels, int64_t *counts, uint8_t *scores) {
  if (!t || !t->finalized) return -1;
  for (int i = 0; i < 94; i++) { counts[i] = 0; scores[i] = 0; }
  if (levels == 0) { for (int i = 0; i < 94; i++) scores[i] = uint8_t(i); return 0; }
  for (size_t i = 0; i < t->qe.size(); i++)
    if (t->q[2 * i] > 0) counts[t->qe[i]] += t->q[2 * i];
  Interval iv[94];
  for (int i = 0; i < 94; i++) {
    const double er = q_to_err(double(i));
    iv[i] = Interval{i + 1 == 94 ? -1 : i + 1, er, counts[i], counts[i], (long long)(double(counts[i]) * er)};
  }
  auto leaf_penalty = [&](int k, double global) { return k <= 6 ? 0.0 : std::fabs(go_log10(iv[k].rate) - go_log10(global)) * double(iv[k].leaf); };
  auto merge_penalty = [&](int i, int j) {
    const double rate = err_rate(iv[i].nobs + iv[j].nobs, iv[i].nerr + iv[j].nerr);
    if (rate == 0) return 0.0;
    double si = 0, sj = 0;
    for (int k = i; k < j; k++) si += leaf_penalty(k, rate);
    const int kend = iv[j].next >= 0 ? iv[j].next : 94;
    for (int k = j; k < kend; k++) sj += leaf_penalty(k, rate);
    return si + sj;
  };
  int n = 94;
  while (n > levels) {
    int i = 0, j = iv[0].next;
    if (j < 0) break;
    int min_i = 0;
    double pen = merge_penalty(i, j);
    for (;;) {
      i = j; j = iv[i].next;
      if (j < 0) break;
      const double p = merge_penalty(i, j);
      if (p < pen) { min_i = i; pen = p; }
    }
    Interval &a = iv[min_i], &b = iv[a.next];
    const long long nobs = a.nobs + b.nobs, nerr = a.nerr + b.nerr;
    a.next = b.next; a.nobs = nobs; a.nerr = nerr;
    n--;
  }
  for (int i = 0; i >= 0;) {
    const bool leaf = iv[i].next < 0 ? (i == 93) : (iv[i].next == i + 1);
    uint8_t qs;
    if (leaf) qs = uint8_t(i);
    else {
      const double prob = err_rate(iv[i].nobs, iv[i].nerr);
      int qv = 93;
      if (prob != 0.0) { qv = int(std::round(-10 * go_log10(prob))); if (qv > 93) qv = 93; if (qv < 1) qv = 1; }
      qs = uint8_t(qv);
    }
    const int kend = iv[i].next >= 0 ? iv[i].next : 94;
    for (int k = i; k < kend; k++) scores[k] = qs;
    i = iv[i].next;
  }
  return 0;
}

static void static_quantized(const uint8_t *quals_in, int n, uint8_t *out) {  // bqsr.go:710-744
  std::vector<uint8_t> quals(quals_in, quals_in + n);
  std::memset(out, 0, 254);
  for (int i = 0; i < 6; i++) out[i] = uint8_t(i);
  if (n == 1) { for (int i = 6; i < 254; i++) out[i] = quals[0]; return; }
  std::sort(quals.begin(), quals.end());
  uint8_t prev_q = 6;
  double prev_p = q_to_prob(double(prev_q));
  for (uint8_t next_q : quals) {
    for (uint8_t i = prev_q; i < next_q; i++) {  // the reference advances prevProb/prevQual inside this loop (:727-737)
      const double next_p = q_to_prob(double(next_q)), ip = q_to_prob(double(i));
      out[i] = (ip - prev_p > next_p - ip) ? next_q : prev_q;
      prev_p = next_p;
      prev_q = next_q;
    }
  }
  for (int i = prev_q; i < 254; i++) out[i] = prev_q;
}

// The reference memoises applyKey{rg, qual, cycle, context} -> uint8 (bqsr.go:970-999).  estimateHierarchicalBayesianQuality
// (:901-919) factorises: deltaGlobal depends on rg; deltaReported and the conditional prior on (rg, qual); the cycle term on
// (rg, qual, cycle); the context term on (rg, qual, context).  Tabulating the three factors and combining them with the very
// same float64 operations in the same order (conditionalPrior + (cycleTerm + contextTerm)) reproduces every memo value.
// one (covariate, quality) row of the LUT, [ncyc][17] bytes at lq; returns the byte every cycle WITHOUT a table entry holds at context
// index 16 (a row that is not live is that byte everywhere)
static uint8_t lut_row(const elp_bqsr_tables *t, int cv, int ql, const uint8_t *quantized, const uint8_t *stat, int n_sqq, uint8_t *lq) {
  const int ncyc = t->ncyc;
  std::vector<double> dcyc(ncyc), dctx(17);
  const size_t qi = t->qi(cv, ql);
  const double cond = t->cond[qi];  // deltaQReported + deltaQ + epsilon, and the entries' estimates under it: elp_bqsr_tables_finalize
  const bool live = t->live[qi] != 0;  // else: no cycle or context entry in this row
  for (int cy = 0; live && cy < ncyc; cy++) {
    const size_t ci = qi * ncyc + cy;
    dcyc[cy] = t->c[2 * ci] > 0 ? double(t->ce_cond[ci]) - cond : 0.0;
  }
  for (int cx = 0; cx < 16; cx++) {
    const size_t xi = qi * NX + cx;
    dctx[cx] = live && t->x[2 * xi] > 0 ? double(t->xe_cond[xi]) - cond : 0.0;
  }
  auto entry = [&](bool has_c, int cy, int cx) {
    const bool has_x = live && cx < 16 && t->x[2 * (qi * NX + cx)] > 0;
    double d_cov = 0;
    if (has_c) d_cov = dcyc[cy];
    if (has_x) d_cov += dctx[cx];
    const double est = cond + d_cov;
    int r = int(std::round(est));
    if (r > 93) r = 93;
    if (r < 1) r = 1;
    uint8_t o = quantized[r];
    if (n_sqq > 0) o = stat[o];
    return o;
  };
  uint8_t no_cycle[17];  // the 17 values of a cycle without table entry (the same for every such cycle)
  for (int cx = 0; cx < 17; cx++) no_cycle[cx] = entry(false, 0, cx);
  if (!lq) return no_cycle[16];
  // the whole (cov, quality) row = that pattern repeated (filled by doubling), then the cycles that do have an entry
  const size_t row_bytes = size_t(ncyc) * 17;
  std::memcpy(lq, no_cycle, 17);
  for (size_t have = 17; have < row_bytes;) {
    const size_t n = std::min(have, row_bytes - have);
    std::memcpy(lq + have, lq, n);
    have += n;
  }
  for (int cy = 0; live && cy < ncyc; cy++) {
    if (t->c[2 * (qi * ncyc + cy)] > 0) {
      uint8_t *le = lq + size_t(cy) * 17;
      for (int cx = 0; cx < 17; cx++) le[cx] = entry(true, cy, cx);
    }
  }
  return no_cycle[16];
}

int elp_bqsr_tables_build_lut(const elp_bqsr_tables *t, int quantize_levels, const uint8_t *sqq, int n_sqq, uint8_t *lut, uint8_t *cov_present) {
  if (!t || !t->finalized || !lut || !cov_present) return -1;
  int64_t counts[94];
  uint8_t quantized[94], stat[254];
  elp_bqsr_tables_quantize(t, quantize_levels, counts, quantized);
  if (n_sqq > 0) static_quantized(sqq, n_sqq, stat);
  const int ncyc = t->ncyc;
  for (int cv = 0; cv < t->n_cov; cv++) {
    cov_present[cv] = t->present[cv];
    if (!t->present[cv]) std::memset(lut + size_t(cv) * NQ * ncyc * 17, 0, size_t(NQ) * ncyc * 17);
  }
  parallel_rows(size_t(t->n_cov) * NQ, [&](size_t row) {  // one (covariate, quality) row of the LUT per task
    const int cv = int(row / NQ), ql = int(row % NQ);
    if (!t->present[cv]) return;
    (void)lut_row(t, cv, ql, quantized, stat, n_sqq, lut + (size_t(cv) * NQ + size_t(ql)) * ncyc * 17);
  }, small_job(t->live, t->n_cov));
  return 0;
}

// The LUT in the rows form (round 5): rows [n_cov][n_quals][2*max_cycle+1][17] for the qualities `quals` (ascending), and for every other
// (covariate, quality) the ONE byte its whole row consists of (a row without cycle and context entries: the estimate is the row's prior
// whatever the cycle and the context), defaults [n_cov][94].  Expanding the two gives elp_bqsr_tables_build_lut's dense LUT byte for
// byte (elp_bqsr_lut_upload_rows does that on the device); returns -2 if a quality outside `quals` has table entries.
int elp_bqsr_tables_build_lut_rows(const elp_bqsr_tables *t, int quantize_levels, const uint8_t *sqq, int n_sqq, const uint8_t *quals, int n_quals, uint8_t *rows,
                                   uint8_t *defaults, uint8_t *cov_present) {
  if (!t || !t->finalized || !defaults || !cov_present || n_quals < 0 || n_quals > NQ || (n_quals && (!quals || !rows))) return -1;
  int slot[NQ];
  for (int q = 0; q < NQ; q++) slot[q] = -1;
  for (int k = 0; k < n_quals; k++) {
    if (quals[k] >= NQ || (k && quals[k] <= quals[k - 1])) return -1;
    slot[quals[k]] = k;
  }
  for (size_t row = 0; row < size_t(t->n_cov) * NQ; row++)
    if (t->live[row] && slot[row % NQ] < 0) return -2;
  int64_t counts[94];
  uint8_t quantized[94], stat[254];
  elp_bqsr_tables_quantize(t, quantize_levels, counts, quantized);
  if (n_sqq > 0) static_quantized(sqq, n_sqq, stat);
  const int ncyc = t->ncyc;
  for (int cv = 0; cv < t->n_cov; cv++) cov_present[cv] = t->present[cv];
  parallel_rows(size_t(t->n_cov) * NQ, [&](size_t row) {
    const int cv = int(row / NQ), ql = int(row % NQ), k = slot[ql];
    uint8_t *lq = k >= 0 ? rows + (size_t(cv) * size_t(n_quals) + size_t(k)) * ncyc * 17 : nullptr;
    if (!t->present[cv]) {
      defaults[row] = 0;
      if (lq) std::memset(lq, 0, size_t(ncyc) * 17);
      return;
    }
    defaults[row] = lut_row(t, cv, ql, quantized, stat, n_sqq, lq);
  }, small_job(t->live, t->n_cov));
  return 0;
}

// ---------------------------------------------------------------- report text (filters/print-bqsr.go)
namespace {
struct Sb {
  std::string s;
  void f(const char *fmt, ...) __attribute__((format(printf, 2, 3))) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    int w = vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (w >= (int)sizeof buf) {
      std::vector<char> big(w + 1);
      va_start(ap, fmt);
      vsnprintf(big.data(), big.size(), fmt, ap);
      va_end(ap);
      s.append(big.data(), w);
    } else {
      s.append(buf, w);
    }
  }
};
inline int ilen(long long v) { char b[32]; return snprintf(b, sizeof b, "%lld", v); }
struct Row2 { std::string rg; int qual; bool cycle; std::string text; long long obs, mism; int emp; };
}  // namespace

char *elp_bqsr_tables_report(const elp_bqsr_tables *t, const char *const *names, const char *prefix) {
  if (!t || !t->finalized) return nullptr;
  Sb o;
  o.f("#:%sReport.v1.1:5\n", prefix);
  o.f("#:%sTable:2:17:%%s:%%s:;\n", prefix);
  o.f("#:%sTable:Arguments:Recalibration argument collection values used in this run\n", prefix);
  static const char *kArgs[][2] = {{"Argument", "Value"}, {"binary_tag_name", "null"},
      {"covariate", "ReadGroupCovariate,QualityScoreCovariate,ContextCovariate,CycleCovariate"}, {"default_platform", "null"},
      {"deletions_default_quality", "45"}, {"force_platform", "null"}, {"indels_context_size", "3"}, {"insertions_default_quality", "45"},
      {"low_quality_tail", "2"}, {"maximum_cycle_value", "500"}, {"mismatches_context_size", "2"}, {"mismatches_default_quality", "-1"},
      {"no_standard_covs", "false"}, {"quantizing_levels", "16"}, {"recalibration_report", "null"}, {"run_without_dbsnp", "false"},
      {"solid_nocall_strategy", "THROW_EXCEPTION"}, {"solid_recal_mode", "SET_Q_ZERO"}};
  for (auto &a : kArgs) o.f("%-26s  %-72s\n", a[0], a[1]);  // fixed-width literal lines of print-bqsr.go:275-292
  o.f("\n");
  {  // printQuantizationTable :49-76
    int64_t counts[94]; uint8_t scores[94];
    elp_bqsr_tables_quantize(t, 16, counts, scores);
    o.f("#:%sTable:3:%d:%%d:%%d:%%d:;\n", prefix, 94);
    o.f("#:%sTable:Quantized:Quality quantization map\n", prefix);
    int w1 = 12, w2 = 5, w3 = 14;
    for (int i = 0; i < 94; i++) { w1 = std::max(w1, ilen(i)); w2 = std::max(w2, ilen(counts[i])); w3 = std::max(w3, ilen(scores[i])); }
    o.f("%-*s  %-*s  %-*s\n", w1, "QualityScore", w2, "Count", w3, "QuantizedScore");
    for (int i = 0; i < 94; i++) o.f("%*d  %*lld  %*d\n", w1, i, w2, (long long)counts[i], w3, scores[i]);
    o.f("\n");
  }
  std::vector<int> order;
  for (int c = 0; c < t->n_cov; c++) order.push_back(c);
  std::sort(order.begin(), order.end(), [&](int a, int b) { return std::strcmp(names[a], names[b]) < 0; });
  {  // printCombinedBQSRTable :78-124
    int n = 0;
    for (int c = 0; c < t->n_cov; c++) n += t->present[c];
    o.f("#:%sTable:6:%d:%%s:%%s:%%.4f:%%.4f:%%d:%%.2f:;\n", prefix, n);
    o.f("#:%sTable:RecalTable0:\n", prefix);
    int wrg = 9, wev = 9, wemp = 16, west = 18, wobs = 12, werr = 6;
    char b[64];
    for (int c = 0; c < t->n_cov; c++) {
      if (!t->present[c]) continue;
      wrg = std::max(wrg, (int)std::strlen(names[c])); wemp = std::max(wemp, ilen(t->cemp[c]) + 5);
      west = std::max(west, snprintf(b, sizeof b, "%.4f", t->rep[c])); wobs = std::max(wobs, ilen(t->cobs[c])); werr = std::max(werr, ilen(t->cmism[c]) + 3);
    }
    o.f("%-*s  %-*s  %-*s  %-*s  %-*s  %-*s\n", wrg, "ReadGroup", wev, "EventType", wemp, "EmpiricalQuality", west, "EstimatedQReported", wobs,
        "Observations", werr, "Errors");
    for (int c : order) {
      if (!t->present[c]) continue;
      o.f("%-*s  %-*s  %*d.0000  %*.4f  %*lld  %*lld.00\n", wrg, names[c], wev, "M", wemp - 5, t->cemp[c], west, t->rep[c], wobs, t->cobs[c], werr - 3,
          t->cmism[c]);
    }
    o.f("\n");
  }
  {  // printBQSRTable :126-178
    int n = 0;
    for (size_t i = 0; i < t->qe.size(); i++) n += t->q[2 * i] > 0;
    o.f("#:%sTable:6:%d:%%s:%%d:%%s:%%.4f:%%d:%%.2f:;\n", prefix, n);
    o.f("#:%sTable:RecalTable1:\n", prefix);
    int wrg = 9, wq = 12, wev = 9, wemp = 16, wobs = 12, werr = 6;
    for (int c = 0; c < t->n_cov; c++)
      for (int q = 0; q < NQ; q++) {
        const size_t i = t->qi(c, q);
        if (t->q[2 * i] <= 0) continue;
        wrg = std::max(wrg, (int)std::strlen(names[c])); wq = std::max(wq, ilen(q)); wemp = std::max(wemp, ilen(t->qe[i]) + 5);
        wobs = std::max(wobs, ilen(t->q[2 * i])); werr = std::max(werr, ilen(t->q[2 * i + 1]) + 3);
      }
    o.f("%-*s  %-*s  %-*s  %-*s  %-*s  %-*s\n", wrg, "ReadGroup", wq, "QualityScore", wev, "EventType", wemp, "EmpiricalQuality", wobs, "Observations",
        werr, "Errors");
    for (int c : order)
      for (int q = 0; q < NQ; q++) {
        const size_t i = t->qi(c, q);
        if (t->q[2 * i] <= 0) continue;
        o.f("%-*s  %*d  %-*s  %*d.0000  %*lld  %*lld.00\n", wrg, names[c], wq, q, wev, "M", wemp - 5, t->qe[i], wobs, t->q[2 * i], werr - 3, t->q[2 * i + 1]);
      }
    o.f("\n");
  }
  {  // printOtherCovariateTable :186-266
    std::vector<Row2> rows;
    static const char kB[] = "ACGT";
    for (int c = 0; c < t->n_cov; c++)
      for (int q = 0; q < NQ; q++) {
        const size_t qi = t->qi(c, q);
        for (int cy = 0; cy < t->ncyc; cy++) {
          const size_t i = qi * t->ncyc + cy;
          if (t->c[2 * i] > 0) rows.push_back(Row2{names[c], q, true, std::to_string(cy - t->max_cycle), t->c[2 * i], t->c[2 * i + 1], t->ce[i]});
        }
        for (int cx = 0; cx < NX; cx++) {
          const size_t i = qi * NX + cx;
          if (t->x[2 * i] > 0) rows.push_back(Row2{names[c], q, false, std::string{kB[cx & 3], kB[(cx >> 2) & 3]}, t->x[2 * i], t->x[2 * i + 1], t->xe[i]});
        }
      }
    int wrg = 9, wq = 12, wcv = 14, wcn = 13, wev = 9, wemp = 16, wobs = 12, werr = 6;
    for (auto &r : rows) {
      wrg = std::max(wrg, (int)r.rg.size()); wq = std::max(wq, ilen(r.qual)); wcv = std::max(wcv, (int)r.text.size());
      wemp = std::max(wemp, ilen(r.emp) + 5); wobs = std::max(wobs, ilen(r.obs)); werr = std::max(werr, ilen(r.mism) + 3);
    }
    o.f("#:%sTable:8:%zu:%%s:%%d:%%s:%%s:%%s:%%.4f:%%d:%%.2f:;\n", prefix, rows.size());
    o.f("#:%sTable:RecalTable2:\n", prefix);
    o.f("%-*s  %-*s  %-*s  %-*s  %-*s  %-*s  %-*s  %-*s\n", wrg, "ReadGroup", wq, "QualityScore", wcv, "CovariateValue", wcn, "CovariateName", wev,
        "EventType", wemp, "EmpiricalQuality", wobs, "Observations", werr, "Errors");
    std::sort(rows.begin(), rows.end(), [](const Row2 &a, const Row2 &b) {  // (ReadGroup, Qual, covariate AS TEXT) :229-243
      if (a.rg != b.rg) return a.rg < b.rg;
      if (a.qual != b.qual) return a.qual < b.qual;
      return a.text < b.text;
    });
    for (auto &r : rows)
      o.f("%-*s  %*d  %-*s  %-*s  %-*s  %*d.0000  %*lld  %*lld.00\n", wrg, r.rg.c_str(), wq, r.qual, wcv, r.text.c_str(), wcn, r.cycle ? "Cycle" : "Context",
          wev, "M", wemp - 5, r.emp, wobs, r.obs, werr - 3, r.mism);
    o.f("\n");
  }
  char *out = (char *)std::malloc(o.s.size() + 1);
  std::memcpy(out, o.s.c_str(), o.s.size() + 1);
  return out;
}

void elp_host_free(void *p) { std::free(p); }

// ---------------------------------------------------------------- duplication metrics (filters/mark-optical-duplicates.go:527-699)
static double f_lib(double x, double c, double n) { return c / x - 1 + go_exp(-n / x); }
static long long estimate_library_size(long long n_pairs, long long n_unique) {
  const double n = double(n_pairs), c = double(n_unique);
  if (n_pairs > 0 && n_pairs - n_unique > 0) {
    double m = 1.0, M = 100.0;
    double fd = f_lib(M * c, c, n);
    while (fd >= 0.0) { M *= 10.0; fd = f_lib(M * c, c, n); }
    for (int i = 0; i < 40; i++) {
      const double r = (m + M) / 2.0, u = f_lib(r * c, c, n);
      if (u == 0.0) break;
      if (u > 0.0) m = r;
      if (u < 0.0) M = r;
    }
    return (long long)(c * ((m + M) / 2.0));
  }
  return 0;
}

int elp_dup_derived(const int64_t *k, double *pct, int64_t *lib_size) {
  if (!k) return -1;
  // order: UnpairedReadsExamined, ReadPairsExamined, SecondaryOrSupplementary, UnmappedReads, UnpairedReadDuplicates, ReadPairDuplicates, ReadPairOpticalDuplicates
  if (lib_size) *lib_size = k[1] > 0 ? estimate_library_size(k[1] - k[6], k[1] - k[5]) : 0;
  if (pct) *pct = double(k[4] + k[5] * 2) / double(k[0] + k[1] * 2);
  return 0;
}

static std::string format_float(double v) {  // formatFloat :590-605
  char b[64];
  snprintf(b, sizeof b, "%.6f", v);
  std::string s(b);
  size_t dot = s.find('.');
  if (dot == std::string::npos) return s;
  size_t j = s.size() - 1;
  while (j > dot && s[j] == '0') j--;
  if (j == dot) return s;  // all zeros after the point: Go returns the untrimmed string
  return s.substr(0, j + 1);
}

char *elp_dup_metrics_report_hist(const int64_t *ctr, const int64_t *hist, int hist_len, int n_lib, const char *const *lib_names,
                                  const char *command_line) {
  Sb o;
  o.f("## htsjdk.samtools.metrics.StringHeader\n");
  o.f("# %s\n", command_line ? command_line : "");
  o.f("## htsjdk.samtools.metrics.StringHeader\n");
  o.f("# Started on: (timestamp omitted)\n\n");
  o.f("## METRICS CLASS\tpicard.sam.DuplicationMetrics\n");
  o.f("LIBRARY\tUNPAIRED_READS_EXAMINED\tREAD_PAIRS_EXAMINED\tSECONDARY_OR_SUPPLEMENTARY_RDS\tUNMAPPED_READS\tUNPAIRED_READ_DUPLICATES\tREAD_PAIR_DUPLICATES\t"
      "READ_PAIR_OPTICAL_DUPLICATES\tPERCENT_DUPLICATION\tESTIMATED_LIBRARY_SIZE\n");
  for (int l = 0; l <= n_lib; l++) {  // the reference iterates a Go map (:621): row order is not part of the contract
    const int64_t *k = ctr + size_t(l) * 7;
    const char *name = l < n_lib ? lib_names[l] : "Unknown Library";
    double pct; int64_t ls;
    elp_dup_derived(k, &pct, &ls);
    std::string ps = std::isnan(pct) ? "NaN" : format_float(pct);
    if (k[1] > 0)
      o.f("%s\t%lld\t%lld\t%lld\t%lld\t%lld\t%lld\t%lld\t%s\t%lld\n", name, (long long)k[0], (long long)k[1], (long long)k[2], (long long)k[3], (long long)k[4],
          (long long)k[5], (long long)k[6], ps.c_str(), (long long)ls);
    else
      o.f("%s\t%lld\t%lld\t%lld\t%lld\t%lld\t%lld\t%lld\t%s\n", name, (long long)k[0], (long long)k[1], (long long)k[2], (long long)k[3], (long long)k[4],
          (long long)k[5], (long long)k[6], ps.c_str());
  }
  o.f("\n");
  // the histogram block is only written if exactly one library has pairs (:628-646)
  int one = -1, many = 0;
  for (int l = 0; l <= n_lib; l++)
    if (ctr[size_t(l) * 7 + 1] > 0) { many += one >= 0; one = l; }
  if (!hist || one < 0 || many) {
    o.f("\n");
  } else {
    const int64_t *k = ctr + size_t(one) * 7;
    const int64_t *h0 = hist + size_t(one) * 3 * hist_len, *h1 = h0 + hist_len, *h2 = h1 + hist_len;  // all / non-optical / optical sets
    const auto at = [&](const int64_t *h, int b) { return b < hist_len ? (long long)h[b] : 0ll; };
    double pct; int64_t ls;
    elp_dup_derived(k, &pct, &ls);
    const int64_t n_pairs = k[1], n_unique = k[1] - k[5];
    o.f("## HISTOGRAM\tjava.lang.Double\n");
    o.f("BIN\tCoverageMult\tall_sets\toptical_sets\tnon_optical_sets\n");
    for (int x = 1; x <= 100; x++) {  // histogramRoi / estimateRoi :576-588
      const double roi = double(ls) * (1.0 - go_exp(-double(int64_t(x) * n_pairs) / double(ls))) / double(n_unique);
      o.f("%d.0\t%s\t%lld\t%lld\t%lld\n", x, format_float(roi).c_str(), at(h0, x), at(h2, x), at(h1, x));
    }
    for (int b = 101; b < hist_len; b++)  // set sizes beyond 100 that occur (the reference sorts its map keys, :659-697)
      if (h0[b] || h1[b] || h2[b]) o.f("%d.0\t0\t%lld\t%lld\t%lld\n", b, (long long)h0[b], (long long)h2[b], (long long)h1[b]);
    o.f("\n");
  }
  char *out = (char *)std::malloc(o.s.size() + 1);
  std::memcpy(out, o.s.c_str(), o.s.size() + 1);
  return out;
}

char *elp_dup_metrics_report(const int64_t *ctr, int n_lib, const char *const *lib_names, const char *command_line) {
  return elp_dup_metrics_report_hist(ctr, nullptr, 0, n_lib, lib_names, command_line);
}

}  // extern "C"
