// tests/host_harness.cpp — a compiled consumer of include/elprep_hip.h + include/elprep_host.h: the `elprep filter` sequence of
// cmd/filter.go:142-211 (MarkDuplicates while the records stream in, coordinate sort as the pipeline's Finalize, optical-duplicate
// metrics, Recalibrate, FinalizeBQSRTables, ApplyBQSR) driven end to end through the two C ABIs, the way a C/C++ host (or the cgo
// adapter of INTEGRATION.md) would.  Input: a binary batch file written by tests/test_gpu_harness.py (columns + header + reference
// + known sites); output: a binary file with every result, which the test compares with the oracle.
//
//   host_harness <in.bin> <out.bin> [n_stage_threads]
//
// Built by tests (g++, links libelprep_hip.so and libelprep_host.so); test infrastructure, not product.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../include/elprep_hip.h"
#include "../include/elprep_host.h"

static std::vector<uint8_t> slurp(const char *path) {
  FILE *f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> b((size_t)n);
  if (n && fread(b.data(), 1, (size_t)n, f) != (size_t)n) { perror("read"); exit(2); }
  fclose(f);
  return b;
}
struct Reader {
  const uint8_t *p;
  template <class T> T get() { T v; memcpy(&v, p, sizeof v); p += sizeof v; return v; }
  template <class T> const T *arr(size_t n) { const T *r = reinterpret_cast<const T *>(p); p += n * sizeof(T); p += (8 - (n * sizeof(T)) % 8) % 8; return r; }
};
static void put(FILE *f, const void *p, size_t n) { fwrite(p, 1, n, f); }
#define CHECK(ctx, call)                                                                  \
  do {                                                                                    \
    int rc__ = (call);                                                                    \
    if (rc__ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc__, elp_last_error(ctx)); return 3; } \
  } while (0)

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: host_harness in.bin out.bin [threads]\n"); return 2; }
  const int n_threads = argc > 3 ? atoi(argv[3]) : 3;
  std::vector<uint8_t> in = slurp(argv[1]);
  Reader r{in.data()};
  // ---- header: @SQ LN, @RG -> LB / PU dictionaries (elp_header)
  elp_header h;
  h.n_ref = r.get<int32_t>(); h.n_rg = r.get<int32_t>(); h.n_lib = r.get<int32_t>(); h.n_cov = r.get<int32_t>();
  const int32_t max_cycle = r.get<int32_t>(), pixel = r.get<int32_t>();
  const uint64_t n = r.get<uint64_t>();
  h.ref_len = r.arr<int32_t>((size_t)h.n_ref);
  h.rg_lib = r.arr<uint16_t>((size_t)h.n_rg);
  h.rg_cov = r.arr<uint16_t>((size_t)h.n_rg);
  // ---- one batch, column by column (elp_batch)
  elp_batch b;
  memset(&b, 0, sizeof b);
  b.n = n;
  b.refid = r.arr<int32_t>(n); b.pos = r.arr<int32_t>(n); b.next_refid = r.arr<int32_t>(n); b.pnext = r.arr<int32_t>(n); b.tlen = r.arr<int32_t>(n);
  b.flag = r.arr<uint16_t>(n); b.mapq = r.arr<uint8_t>(n); b.rgid = r.arr<uint16_t>(n); b.has_sr = r.arr<uint8_t>(n); b.l_seq = r.arr<uint32_t>(n);
  b.qname_off = r.arr<uint64_t>(n + 1); b.qname = r.arr<uint8_t>((size_t)b.qname_off[n]);
  b.cigar_off = r.arr<uint64_t>(n + 1); b.cigar = r.arr<uint32_t>((size_t)b.cigar_off[n]);
  b.seq_off = r.arr<uint64_t>(n + 1); b.seq4 = r.arr<uint8_t>((size_t)b.seq_off[n]);
  b.qual_off = r.arr<uint64_t>(n + 1); b.qual = r.arr<uint8_t>((size_t)b.qual_off[n]);

  elp_ctx *c = nullptr;
  if (elp_create(0, &c) != 0) { fprintf(stderr, "elp_create failed: no gfx950 device\n"); return 4; }
  // the flat entry points (every array an argument of its own): what the cgo binding of INTEGRATION.md section 2 calls
  CHECK(c, elp_set_header_columns(c, h.n_ref, h.ref_len, h.n_rg, h.rg_lib, h.rg_cov, h.n_lib, h.n_cov));
  // record batching: the batches arrive from several threads, as the reference's LimitedPar parse nodes deliver them
  // (sam/filter-pipeline.go:290-292); elp_stage / elp_stage_columns serialise internally, the staging order is the order of the calls, so the
  // threads take turns in batch order here (the reference's Slice node is ordered as well, :108-124)
  {
    const uint64_t per = (n + (uint64_t)n_threads - 1) / (uint64_t)(n_threads ? n_threads : 1);
    int rc_all = 0;
    for (int t = 0; t < n_threads; t++) {
      const uint64_t lo = std::min<uint64_t>(n, per * (uint64_t)t), hi = std::min<uint64_t>(n, lo + per);
      if (lo >= hi) continue;
      std::thread th([&, lo, hi] {
        elp_batch s = b;
        s.n = hi - lo;
        s.refid += lo; s.pos += lo; s.next_refid += lo; s.pnext += lo; s.tlen += lo; s.flag += lo; s.mapq += lo; s.rgid += lo; s.has_sr += lo; s.l_seq += lo;
        s.qname_off += lo; s.cigar_off += lo; s.seq_off += lo; s.qual_off += lo;  // offsets are rebased by elp_stage
        const int rc = elp_stage_columns(c, s.n, s.refid, s.pos, s.next_refid, s.pnext, s.tlen, s.flag, s.mapq, s.rgid, s.has_sr, s.l_seq, s.qname_off, s.qname,
                                         s.cigar_off, s.cigar, s.seq_off, s.seq4, s.qual_off, s.qual, nullptr);
        if (rc) rc_all = rc;
      });
      th.join();
    }
    if (rc_all) { fprintf(stderr, "elp_stage -> %d: %s\n", rc_all, elp_last_error(c)); return 3; }
  }
  // ---- NewBaseRecalibrator inputs
  for (int32_t k = 0; k < h.n_ref; k++) {
    const int64_t len = r.get<int64_t>();
    const uint8_t *bases = r.arr<uint8_t>((size_t)len);
    CHECK(c, elp_bqsr_set_reference(c, k, bases, len));
    const int64_t ns = r.get<int64_t>();
    const int32_t *sites = r.arr<int32_t>((size_t)(2 * ns));
    CHECK(c, elp_bqsr_set_known_sites(c, k, sites, ns));
  }

  // ---- the filter sequence
  CHECK(c, elp_mark_duplicates(c, 1));
  CHECK(c, elp_sort_coordinate(c));
  std::vector<uint32_t> perm(n);
  std::vector<uint16_t> flags(n);
  CHECK(c, elp_get_permutation(c, perm.data()));
  CHECK(c, elp_get_flags(c, flags.data()));
  std::vector<int64_t> ctr((size_t)(h.n_lib + 1) * ELP_NCTR);
  CHECK(c, elp_dup_metrics(c, pixel, ctr.data()));
  const size_t ncyc = 2 * (size_t)max_cycle + 1, nq = (size_t)h.n_cov * ELP_NQUAL * 2;
  std::vector<int64_t> qt(nq), ct(nq * ncyc), xt(nq * ELP_NCTX);
  CHECK(c, elp_bqsr_gather(c, max_cycle, qt.data(), ct.data(), xt.data()));
  // host side, float64: FinalizeBQSRTables + the dense tabulation of ApplyBQSR's memo
  elp_bqsr_tables *t = elp_bqsr_tables_new(h.n_cov, max_cycle, qt.data(), ct.data(), xt.data());
  if (!t || elp_bqsr_tables_finalize(t) != 0) { fprintf(stderr, "finalize failed\n"); return 3; }
  std::vector<uint8_t> lut((size_t)h.n_cov * ELP_NQUAL * ncyc * 17), present((size_t)h.n_cov);
  if (elp_bqsr_tables_build_lut(t, 0, nullptr, 0, lut.data(), present.data()) != 0) { fprintf(stderr, "build_lut failed\n"); return 3; }
  CHECK(c, elp_bqsr_apply(c, max_cycle, lut.data(), present.data()));
  std::vector<uint8_t> qual((size_t)elp_num_qual_bytes(c));
  CHECK(c, elp_get_qual(c, qual.data()));
  std::vector<std::string> names;
  std::vector<const char *> cn;
  for (int k = 0; k < h.n_cov; k++) names.push_back("cov" + std::to_string(k));
  for (auto &s : names) cn.push_back(s.c_str());
  char *report = elp_bqsr_tables_report(t, cn.data(), "GATK");

  FILE *f = fopen(argv[2], "wb");
  if (!f) { perror(argv[2]); return 2; }
  const uint64_t n_sorted = elp_num_sorted(c), qb = qual.size(), rl = strlen(report);
  put(f, &n_sorted, 8); put(f, &qb, 8); put(f, &rl, 8);
  put(f, perm.data(), n * 4); put(f, flags.data(), n * 2); put(f, ctr.data(), ctr.size() * 8);
  put(f, qt.data(), qt.size() * 8); put(f, ct.data(), ct.size() * 8); put(f, xt.data(), xt.size() * 8);
  put(f, qual.data(), qb); put(f, report, rl);
  fclose(f);
  elp_host_free(report);
  elp_bqsr_tables_free(t);
  elp_destroy(c);
  printf("host_harness ok: %llu records, %llu sorted\n", (unsigned long long)n, (unsigned long long)n_sorted);
  return 0;
}
