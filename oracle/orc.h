/*
 * oracle/orc.h — CPU ORACLE for the elPrep hot path (coordinate sort -> mark duplicates
 * -> optical-duplicate metrics -> BQSR gather -> finalize -> apply).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (elprep_amd/, libelprep_hip.so)
 * never includes, links or calls anything in oracle/.
 *
 * It is a sequential plain-C restatement of the reference's Go algorithms, each function
 * citing the reference file:line it follows (paths relative to ExaScience/elprep v5.1.3).
 *
 * PARITY UNPINNED: the reference has no tests, golden vectors or fixtures for this path
 * (its only test file is intervals/intervals_test.go) and cannot be built here (no Go
 * toolchain, un-vendored github.com/exascience/pargo v1.1.0).  The oracle is pinned only
 * on (a) the interval KATs of intervals/intervals_test.go and (b) hand-derived known-answer
 * vectors of SURVEY.md §8(c), re-derived in tests/test_oracle_kat.py.
 *
 * Where the (parallel, racy) reference admits several valid executions the oracle follows
 * the single-threaded execution: records are processed in input order (one batch in flight).
 */
#ifndef ORC_H
#define ORC_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NIL16 0xFFFFu

/* SAM FLAG bits, sam/sam-types.go:484-520 */
enum {
  ORC_MULTIPLE = 0x1, ORC_PROPER = 0x2, ORC_UNMAPPED = 0x4, ORC_NEXT_UNMAPPED = 0x8,
  ORC_REVERSED = 0x10, ORC_NEXT_REVERSED = 0x20, ORC_FIRST = 0x40, ORC_LAST = 0x80,
  ORC_SECONDARY = 0x100, ORC_QCFAILED = 0x200, ORC_DUPLICATE = 0x400, ORC_SUPPLEMENTARY = 0x800
};

/* One batch of alignment records, column-wise.  Same layout as elp_batch in
 * include/elprep_hip.h (tests hand the same buffers to both sides). */
typedef struct orc_batch {
  uint64_t n;
  const int32_t *refid, *pos, *next_refid, *pnext, *tlen; /* POS/PNEXT 1-based; refid -1 == RNAME '*' */
  const uint16_t *flag;
  const uint8_t *mapq;
  const uint16_t *rgid;      /* dense id of the RG:Z tag string, ORC_NIL16 = no RG tag */
  const uint8_t *has_sr;     /* 1 = record carries the sr:i tag (elprep split copy); may be NULL */
  const uint32_t *l_seq;     /* number of bases in SEQ */
  const uint64_t *qname_off; const uint8_t *qname;   /* n+1 offsets, raw bytes, no terminator */
  const uint64_t *cigar_off; const uint32_t *cigar;  /* n+1 offsets (in ops), BAM encoding len<<4|op, op index into "MIDNSHP=X" */
  const uint64_t *seq_off;   const uint8_t *seq4;    /* n+1 byte offsets, BAM nibbles high nibble first, "=ACMGRSVTWYHKDBN" */
  const uint64_t *qual_off;  const uint8_t *qual;    /* n+1 byte offsets, raw phred (no +33) */
  const uint16_t *split;     /* layout parity with elp_batch only: the oracle is run once per split file, as the reference is */
} orc_batch;

/* Header facts the path needs (sam.Header @SQ / @RG) */
typedef struct orc_header {
  int32_t n_ref; const int32_t *ref_len;   /* @SQ LN per refid */
  int32_t n_rg;  const uint16_t *rg_lib;   /* library id (dense id of the LB string) per rgid, ORC_NIL16 = RG has no LB / unknown RG */
                 const uint16_t *rg_cov;   /* BQSR read-group covariate id per rgid: dense id of (PU if present else ID), filters/bqsr.go:35-51 */
  int32_t n_lib; int32_t n_cov;
} orc_header;

/* ---- sort (sam/sam-types.go:408-473, 599-641) ---- */
uint16_t orc_mod_flag(uint16_t flag);
/* perm_out[k] = index of the record at sorted position k; equal records keep input order */
int orc_sort_coordinate(const orc_batch *b, uint32_t *perm_out);
uint64_t orc_num_sorted(const orc_batch *b); /* records without the sr tag (RemoveOptionalReads): the sorted prefix of perm_out */
int orc_coordinate_less(const orc_batch *b, uint64_t i, uint64_t j);

/* ---- mark duplicates (filters/mark-duplicates.go) ---- */
int32_t orc_phred_score(const uint8_t *qual, uint32_t n, int *invalid);
int32_t orc_unclipped_position(int32_t pos, uint16_t flag, const uint32_t *cigar, uint32_t n_cigar);
/* flag_out[i] = FLAG of record i after MarkDuplicates (input order).  Optional outputs (may be NULL):
 * upos_out/score_out (adapted values for candidate records, 0 otherwise). */
int orc_mark_duplicates(const orc_batch *b, const orc_header *h, uint16_t *flag_out,
                        int32_t *upos_out, int32_t *score_out);

/* ---- optical duplicates + DuplicationMetrics (filters/mark-optical-duplicates.go) ---- */
#define ORC_NCTR 7   /* UnpairedReadsExamined, ReadPairsExamined, SecondaryOrSupplementary, UnmappedReads,
                        UnpairedReadDuplicates, ReadPairDuplicates, ReadPairOpticalDuplicates */
/* Runs MarkDuplicates(alsoOpticals=true) then MarkOpticalDuplicates over the records in the order `perm`
 * (sorted order; NULL = input order).  counters is [(n_lib+1)][ORC_NCTR] int64, row n_lib = "Unknown Library".
 * hist (may be NULL) is [(n_lib+1)][3][hist_len] int64: duplicatesCount / nonOptical / optical set-size
 * histograms (index clamped to hist_len-1). */
int orc_dup_metrics(const orc_batch *b, const orc_header *h, const uint32_t *perm, int pixel_dist,
                    uint16_t *flag_out, int64_t *counters, int64_t *hist, int hist_len);
void orc_tile_info(const uint8_t *qname, uint32_t len, int64_t *t, int64_t *x, int64_t *y);
int64_t orc_estimate_library_size(int64_t n_pairs, int64_t n_unique_pairs);

/* ---- known sites (intervals/intervals.go) ---- */
typedef struct { int32_t start, end; } orc_interval;
size_t orc_flatten(orc_interval *iv, size_t n);                       /* intervals.go:103 */
int orc_overlap(const orc_interval *iv, size_t n, int32_t start, int32_t end);   /* :139 */
void orc_intersect(const orc_interval *iv, size_t n, int32_t start, int32_t end, size_t *lo, size_t *hi); /* :166 */
void orc_sort_by_start(orc_interval *iv, size_t n);

/* ---- BQSR (filters/bqsr.go, filters/utils.go) ---- */
typedef struct orc_bqsr_ref {
  /* per refid: reference bases (raw .elfasta ASCII) and flattened known-site intervals (1-based, inclusive) */
  const uint8_t *const *ref_seq; const int64_t *ref_seq_len;
  const orc_interval *const *sites; const int64_t *n_sites;
} orc_bqsr_ref;

/* dense count tables: index helpers (all int64 pairs {observations, mismatches}) */
#define ORC_NQUAL 94
#define ORC_NCTX 16
/* qual_tbl  [n_cov][94][2]
 * cycle_tbl [n_cov][94][2*max_cycle+1][2]   cycle c stored at c+max_cycle
 * ctx_tbl   [n_cov][94][16][2]              context key k stored at (k>>4)&15 */
int orc_bqsr_gather(const orc_batch *b, const orc_header *h, const orc_bqsr_ref *r, const uint16_t *flags /* may be NULL: b->flag */,
                    int max_cycle, int64_t *qual_tbl, int64_t *cycle_tbl, int64_t *ctx_tbl);

/* helper pieces exposed for KATs */
int orc_read_coordinate_for_reference_coordinate(const uint32_t *cigar, uint32_t n_cigar, int soft_start, int ref_index,
                                                 int right_tail, int *ok);
void orc_context_with(const uint8_t *bases, int n, int32_t *keys_out);
int orc_cycle(uint16_t flag, int l_seq, int index);
/* runs hardClipAdaptorSequence + hardClipSoftClippedBases on one record; returns the surviving base range
 * [*a,*b) in original read coordinates, the new POS, and the rewritten CIGAR (BAM-encoded, cap ops). */
int orc_clip_for_bqsr(const orc_batch *b, uint64_t i, int *a, int *bnd, int32_t *new_pos, uint32_t *cigar_out, int cap);
int orc_recalibrate_aln(const orc_batch *b, const orc_header *h, const uint16_t *flags, uint64_t i);

/* finalized tables (float64 host math, filters/bqsr.go:553-919) */
typedef struct orc_bqsr_final orc_bqsr_final;
orc_bqsr_final *orc_bqsr_finalize(int n_cov, int max_cycle, const int64_t *qual_tbl, const int64_t *cycle_tbl, const int64_t *ctx_tbl);
void orc_bqsr_final_free(orc_bqsr_final *f);
/* EmpiricalQuality per entry (same shapes as the count tables without the trailing [2]); 255 = entry absent */
void orc_bqsr_final_empirical(const orc_bqsr_final *f, uint8_t *qual_emp, uint8_t *cycle_emp, uint8_t *ctx_emp);
/* per-cov combined entry: reportedQuality, empirical quality, obs, mism; present[c]=0 if RG absent from tables */
void orc_bqsr_final_combined(const orc_bqsr_final *f, double *reported, uint8_t *emp, int64_t *obs, int64_t *mism, uint8_t *present);
/* quantized quality map for `levels` (0 = identity): counts[94], scores[94]  (bqsr.go:863-899) */
void orc_bqsr_quantize(const orc_bqsr_final *f, int levels, int64_t *counts, uint8_t *scores);
void orc_static_quantized_scores(const uint8_t *quals, int n, uint8_t *out254);   /* bqsr.go:710-744 */
/* recalibrated quality for one key, bqsr.go:979-999 (cov must be present) */
uint8_t orc_bqsr_recal_qual(const orc_bqsr_final *f, int cov, int qual, int cycle, int ctx_key,
                            const uint8_t *quantized, const uint8_t *static_q /* NULL if no --sqq */);
/* ApplyBQSR over a batch: qual_out gets the full qual column (same offsets as b->qual_off) */
int orc_bqsr_apply(const orc_batch *b, const orc_header *h, const orc_bqsr_final *f, int quantize_levels,
                   const uint8_t *sqq, int n_sqq, int max_cycle, uint8_t *qual_out);
/* GATK-report text (filters/print-bqsr.go:269-298).  cov_names[c] = covariate string.  Returns malloc'd string. */
char *orc_bqsr_report(const orc_bqsr_final *f, const char *const *cov_names, const char *prefix);
void orc_free(void *p);

/* float helpers exposed for KATs */
double orc_go_log10(double x);
double orc_go_pow10(double y);  /* math.Pow(10, y) */
double orc_go_log(double x);     /* math.Log as Go computes it on amd64 (math/log.go), orc_gomath.c */
double orc_go_exp(double x);     /* math.Exp, the pure-Go function (math/exp.go); amd64 builds of Go run an assembly kernel instead: orc_gomath.c */
double orc_go_lgamma(double x);  /* math.Lgamma for x > 0 (math/lgamma.go) */
int orc_gomath_selfcheck(void);  /* 0 if the constants of orc_go_log have the bit patterns the Go source prints */
uint8_t orc_bayesian_estimate(int64_t observations, int64_t mismatches, double prior);

/* BAM alignment records (sam/bam-files.go:443-468, 481-737): the records order[0 .. n_order) of b (NULL: all, in input order) behind
 * each other, FLAG / QUAL optionally replaced; the optional fields are a deterministic function of the record (orc_bam.c), written
 * in arbitrary integer types (normalize_tags = 0: a BAM as other tools write it) or as elPrep re-encodes them (1).  out NULL: size only. */
size_t orc_bam_encode(const orc_batch *b, const char *const *rg_ids, const uint32_t *order, uint64_t n_order, const uint16_t *flags,
                      const uint8_t *qual, int normalize_tags, uint8_t *out);

/* the same operators on n_threads host cores (<= 0: all): bench.py's CPU baseline ("port" of the reference's parallel CPU path:
 * parallel merge sort, sharded maps, thread-private tables).  Same results as the sequential functions above. */
int orc_sort_coordinate_mt(const orc_batch *b, uint32_t *perm_out, int n_threads);
int orc_dup_metrics_mt(const orc_batch *b, const orc_header *h, const uint32_t *perm, int pixel_dist, uint16_t *flag_out, int64_t *counters,
                       int n_threads);
int orc_bqsr_gather_mt(const orc_batch *b, const orc_header *h, const orc_bqsr_ref *r, const uint16_t *flags, int max_cycle,
                       int64_t *qual_tbl, int64_t *cycle_tbl, int64_t *ctx_tbl, int n_threads);
int orc_bqsr_apply_mt(const orc_batch *b, const orc_header *h, const orc_bqsr_final *f, int quantize_levels, const uint8_t *sqq, int n_sqq,
                      int max_cycle, uint8_t *qual_out, int n_threads);

void orc_bam_offsets(const orc_batch *b, const char *const *rg_ids, const uint32_t *order, uint64_t n_order, int normalize_tags, uint64_t *off_out);

/* sfm contig groups (sam/split-merge.go:178-213): group_of_ref[n_ref] gets 1-based group index; returns #groups (excl. unmapped) */
int orc_contig_groups(const int32_t *ref_len, int n_ref, int contig_group_size, int32_t *group_of_ref);

#ifdef __cplusplus
}
#endif
#endif
