/*
 * elprep_hip.h — C ABI of libelprep_hip.so: the MI355X (gfx950) implementation of elPrep's
 * coordinate-sort -> mark-duplicates -> optical-duplicate metrics -> BQSR gather -> BQSR apply hot path.
 *
 * This is the drop-in boundary a host program (elPrep's Go code through cgo; the C++
 * host layer in elprep_amd/host; the Python test/bench harness through ctypes) binds to.  Plain C types
 * only.  Every entry point cites the reference interface it replaces (paths relative to
 * ExaScience/elprep v5.1.3).
 *
 * Conventions
 *  - Every function returns 0 on success and a negative elp_status on error; elp_last_error(ctx) holds a
 *    message.  The reference reports all errors by log.Panic (internal/misc.go:31-43); a Go adapter turns a
 *    non-zero status into log.Panic(elp_last_error(ctx)).
 *  - One elp_ctx per GPU.  A ctx owns one HIP stream, the staged column store in HBM and all scratch.
 *    Calls on one ctx must not overlap in time, except elp_stage, which may be called from many host
 *    threads (it serialises internally; it mirrors the reference's LimitedPar batch nodes,
 *    sam/filter-pipeline.go:290-292).
 *  - Records are identified by their staging index (order of arrival = input order), 0-based.
 *  - Records that carry the sr:i tag (has_sr; the copies `elprep split` leaves in a contig-group file, sam/split-merge.go:286-293)
 *    take part in duplicate marking and are then dropped by filters.RemoveOptionalReads (filters/simple-filters.go:146-152, appended
 *    behind the mark-duplicates filter, cmd/filter.go:773,803): they never reach Sam.Alignments.  Here they stay staged, but the sort
 *    puts them behind every other record (elp_num_sorted() records are output), and the duplication metrics and BQSR ignore them.
 *  - Limits (ELP_ERR_UNSUPPORTED): 2^32-16 records per context, 4194303 bases per read, QNAMEs of at most 1000 bytes.
 *  - There is NO CPU fallback: if no gfx950 device is usable, elp_create fails.
 *  - Environment (read once per process): ELP_SYNC_SPIN=0 makes the library wait for its stream with hipStreamSynchronize instead of
 *    polling it (the default: the path waits ~15 times per pass for a few words that size the next launches, and an interrupt wake-up
 *    costs 100-200 us on a busy host; a host that runs many contexts per core may prefer to give the core up).  The host library
 *    (elprep_host.h) takes ELP_HOST_THREADS=<n>: worker threads of its table path per process (default: the hardware's, at most 16).
 */
#ifndef ELPREP_HIP_H
#define ELPREP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ELP_NIL16 0xFFFFu /* "no value" for 16-bit dictionary ids (Go nil interface{} LIBID / missing RG tag) */

typedef enum elp_status {
  ELP_OK = 0,
  ELP_ERR_ARG = -1,       /* invalid argument / call order */
  ELP_ERR_HIP = -2,       /* HIP runtime error */
  ELP_ERR_NOMEM = -3,     /* out of device memory */
  ELP_ERR_DATA = -4,      /* input the reference would panic on (invalid QUAL, cycle > max_cycle, read without RG in ApplyBQSR ...) */
  ELP_ERR_UNSUPPORTED = -5 /* valid for the reference but outside this implementation's stated limits */
} elp_status;

typedef struct elp_ctx elp_ctx;

/* ---------------------------------------------------------------------------------------------------
 * Record batches: the SoA restaging of sam.Alignment (sam/sam-types.go:289-331) produced where the
 * reference batches records (InputFile.RunPipeline, sam/filter-pipeline.go:282-296; BytesToAlignment :92).
 * Fixed fields are exactly BAM's (sam/bam-files.go:299-312) after AddREFID (filters/simple-filters.go:208-231).
 * --------------------------------------------------------------------------------------------------- */
typedef struct elp_batch {
  uint64_t n;                 /* records in this batch */
  const int32_t *refid;       /* REFID temp: index of RNAME in @SQ, -1 for '*' / unknown */
  const int32_t *pos;         /* POS, 1-based */
  const int32_t *next_refid;  /* NextREFID temp ('=' already resolved) */
  const int32_t *pnext;       /* PNEXT, 1-based */
  const int32_t *tlen;        /* TLEN */
  const uint16_t *flag;       /* FLAG */
  const uint8_t *mapq;        /* MAPQ */
  const uint16_t *rgid;       /* dense id of the RG:Z tag string (index into elp_header.rg_*), ELP_NIL16 = no RG tag */
  const uint8_t *has_sr;      /* 1 = record carries the sr:i tag written by `elprep split` (sam/split-merge.go:291); may be NULL */
  const uint32_t *l_seq;      /* SEQ length in bases */
  const uint64_t *qname_off;  /* n+1 byte offsets into qname */
  const uint8_t *qname;       /* QNAME bytes, no terminator */
  const uint64_t *cigar_off;  /* n+1 offsets (in ops) into cigar */
  const uint32_t *cigar;      /* BAM-encoded ops: len<<4 | op, op indexes "MIDNSHP=X" (sam/bam-files.go:289) */
  const uint64_t *seq_off;    /* n+1 byte offsets into seq4 */
  const uint8_t *seq4;        /* BAM 4-bit bases, high nibble first, "=ACMGRSVTWYHKDBN" (utils/nibbles, sam/sam-types.go:228) */
  const uint64_t *qual_off;   /* n+1 byte offsets into qual */
  const uint8_t *qual;        /* raw phred, no +33 (sam/sam-files.go:400-402) */
  const uint16_t *split;      /* optional (NULL = all 0): id of the `elprep split` file the record belongs to.  Records with different
                                 split ids are duplicate-marked as if by separate `elprep filter` runs (their fragment / mate / pair
                                 keys never match), so that one context can hold several contig-group splits of an `sfm` run
                                 (cmd/sfm.go:129-805 runs one filter process per split file) */
} elp_batch;

/* Header facts read by the filters: @SQ LN (alignmentAgreesWithHeader, filters/utils.go:130-139) and the @RG
 * dictionaries (lbTable, filters/mark-duplicates.go:413-423; readGroupCovariate, filters/bqsr.go:35-51). */
typedef struct elp_header {
  int32_t n_ref;
  const int32_t *ref_len;   /* @SQ LN per refid */
  int32_t n_rg;
  const uint16_t *rg_lib;   /* per rgid: dense id of the LB string, ELP_NIL16 if the RG has no LB / is not in the header */
  const uint16_t *rg_cov;   /* per rgid: dense id of the BQSR read-group covariate string (PU if present, else ID) */
  int32_t n_lib;
  int32_t n_cov;
} elp_header;

/* ---- context ---- */
int elp_create(int device_ordinal, elp_ctx **out);
void elp_destroy(elp_ctx *ctx);
const char *elp_last_error(const elp_ctx *ctx);
int elp_sync(elp_ctx *ctx);                 /* wait for the ctx stream */
void *elp_stream(elp_ctx *ctx);             /* the hipStream_t all kernels of this ctx are launched on */

/* ---- staging (replaces Sam.AddNodes' Slice(&alns) collection, sam/filter-pipeline.go:108-124) ---- */
int elp_set_header(elp_ctx *ctx, const elp_header *hdr);
int elp_reserve(elp_ctx *ctx, uint64_t n_records, uint64_t qname_bytes, uint64_t cigar_ops, uint64_t seq_bytes, uint64_t qual_bytes);
int elp_stage(elp_ctx *ctx, const elp_batch *batch); /* appends; host buffers may be reused when the call returns */
int elp_reset(elp_ctx *ctx);                         /* drops all staged records and results */
/* The same two calls with every array as an argument of its own - the forms a cgo binding uses with plain Go slices.  cgo's pointer
 * rules ("Passing pointers", cmd/cgo) let Go code pass a Go pointer to C only if the memory it points to contains no Go pointers: an
 * elp_header / elp_batch that lives in Go memory and points at Go slices breaks that rule (run-time panic "cgo argument has Go pointer
 * to Go pointer"), a call that passes &slice[0] of every slice does not (filters/pedantic.go:28-41 is the reference's own precedent
 * for a cgo call with scalar / single-pointer arguments).  A host that keeps its column buffers in C memory (elp_pinned_alloc: page-
 * locked, read in place by the DMA engine - the fast route over PCIe) may use either form.  has_sr and split may be NULL. */
int elp_set_header_columns(elp_ctx *ctx, int32_t n_ref, const int32_t *ref_len, int32_t n_rg, const uint16_t *rg_lib, const uint16_t *rg_cov,
                           int32_t n_lib, int32_t n_cov);
int elp_stage_columns(elp_ctx *ctx, uint64_t n, const int32_t *refid, const int32_t *pos, const int32_t *next_refid, const int32_t *pnext,
                      const int32_t *tlen, const uint16_t *flag, const uint8_t *mapq, const uint16_t *rgid, const uint8_t *has_sr,
                      const uint32_t *l_seq, const uint64_t *qname_off, const uint8_t *qname, const uint64_t *cigar_off, const uint32_t *cigar,
                      const uint64_t *seq_off, const uint8_t *seq4, const uint64_t *qual_off, const uint8_t *qual, const uint16_t *split);
uint64_t elp_num_records(const elp_ctx *ctx);
uint64_t elp_num_qual_bytes(const elp_ctx *ctx);     /* size of the staged QUAL column (elp_get_qual) */

/* ---- staging straight from BAM, and BAM out: the data formats either side of the path (sam/bam-files.go) ----
 * elp_set_read_group_ids: the @RG ID strings in the order of elp_header.rg_* (RG:Z tags are looked up in them).
 * elp_stage_bam: `bytes` = inflated BGZF payload holding whole alignment records (block_size, fixed fields, read name, CIGAR, bases,
 *   qualities, optional fields; SAMv1 4.2) as bamReader hands them to parseBamAlignment (:299-400).  The records cross PCIe as
 *   they are and are cut into the columns on the device; all of them get `split_id`.  Appends like elp_stage and may be mixed with
 *   it only in separate contexts.  rec_off (optional): byte offset of every record's block_size field, rec_off[n_records] = n_bytes -
 *   a BAM reader knows them; without them the call walks the block_size chain on the host (one dependent load per record).
 *   Fast path: `bytes` in page-locked memory (elp_pinned_alloc) - the DMA engine reads it in place;
 *   pageable memory goes through a pinned double buffer.  Returns when `bytes` may be reused.
 * elp_emit_sorted_bam: formatBamAlignment (:635-737) of the elp_num_sorted() records of the sort's output, in that order, into
 *   `out` (host memory, `cap` bytes; NULL = just compute *n_bytes_out): FLAG and QUAL as the path left them, bin() recomputed
 *   (:443-468), optional fields re-encoded as formatBamTag does (:481-632).  BGZF deflate stays with the host. */
int elp_set_read_group_ids(elp_ctx *ctx, const char *const *ids);
/* cgo form (no array of pointers): the ids behind each other, id_off[g] .. id_off[g + 1] = the bytes of read group g (n_rg + 1 offsets) */
int elp_set_read_group_ids_flat(elp_ctx *ctx, const uint8_t *ids, const uint32_t *id_off);
void *elp_pinned_alloc(size_t bytes);
void elp_pinned_free(void *p);
int elp_stage_bam(elp_ctx *ctx, const uint8_t *bytes, uint64_t n_bytes, const uint64_t *rec_off /* n_records + 1, may be NULL */,
                  uint64_t n_records, uint16_t split_id);
int elp_emit_sorted_bam(elp_ctx *ctx, uint8_t *out, uint64_t cap, uint64_t *n_bytes_out);
/* BGZF on the device (utils/bgzf/bgzf-files.go).
 * elp_stage_bgzf = the reader (:95-221) + elp_stage_bam: `bgzf` holds whole BGZF blocks of a BAM file (the file, or a part of it that
 * starts at a block and ends with a complete alignment record; an end-of-file block is skipped).  The compressed bytes are copied to the
 * device (in chunks, while the decoder works on the chunks in front), every block is inflated (RFC 1951: stored, fixed and dynamic Huffman
 * blocks; round 6, csrc/bgzf.hip: a wavefront turns the block's bit stream into literals in place and match tokens - 64 candidate symbols
 * per step -, a workgroup resolves the matches by pointer jumping in LDS), its CRC-32 is checked
 * ("invalid CRC-32 value for a data block in a BGZF file"), the starts of the alignment records are found on the device (every block
 * guesses its first record start, walks its records, and the guesses are proven by checking that every block's chain ends where the
 * next one's begins; wrong guesses are repaired in order), and the records are staged as elp_stage_bam stages them.  first_record = the
 * offset of the first alignment record in the inflated stream (behind magic, header text and reference dictionary, which the host
 * parses from the first block(s) itself); 0 for a part that starts with a record.
 * elp_emit_sorted_bgzf = elp_emit_sorted_bam + the writer (:324-383): the sorted records as BGZF blocks of <= 65280 payload bytes, each
 * COMPRESSED on the device (DEFLATE over a parallel LZ77 parse with the block's own Huffman codes - round 6; fixed codes where those are
 * not longer -, csrc/deflate_core.hpp; a block that would not shrink is stored), CRC-32 and ISIZE computed on the device.  Inflating the blocks gives elp_emit_sorted_bam's bytes (parity
 * of a BAM file is defined on the inflated stream: the reference's own bytes are whatever Go's compress/flate emits); a BAM file = the
 * host's header block(s) + these blocks + the 28-byte end-of-file block (:53-62).  A size query (out = NULL) returns an UPPER BOUND - the
 * size of the stored form, which is also the room `out` must offer; *n_bytes_out of the real call is the actual size. */
int elp_stage_bgzf(elp_ctx *ctx, const uint8_t *bgzf, uint64_t n_bytes, uint64_t first_record, uint16_t split_id);
int elp_emit_sorted_bgzf(elp_ctx *ctx, uint8_t *out, uint64_t cap, uint64_t *n_bytes_out);
/* The merge of `elprep merge` / `sfm` phase 3 with payloads (MergeSortedFilesSplitPerChromosome, sam/split-merge.go:410-576): the sorted
 * outputs of a context that holds group splits and of the context that holds the spread split as ONE stream of BAM records in the merge's
 * order (elp_merge_spread's slots), gathered in HBM.  Both contexts staged with elp_stage_bam, coordinate-sorted, on one device. */
int elp_emit_merged_bam(elp_ctx *groups, elp_ctx *spread, uint8_t *out, uint64_t cap, uint64_t *n_bytes_out);

/* ---- fused per-record predicates: filters/simple-filters.go ----
 * The filters that stand in front of MarkDuplicates in filters1 (cmd/filter.go:696-803), evaluated in one pass over the staged
 * records: a rejected record takes part in nothing that follows (duplicate marking, sort output, metrics, BQSR tables) - it is
 * behind the elp_num_sorted() records of the permutation.  Call before the other operators; may be called again (predicates add up).
 *   remove_unmapped          RemoveUnmappedReads          :73-75    FLAG & 0x4
 *   remove_unmapped_strict   RemoveUnmappedReadsStrict    :79-83    ... or POS == 0 or RNAME == "*"
 *   min_mapq                 RemoveMappingQualityLessThan :332-347  keep MAPQ >= min_mapq (0 = off, > 255 rejects everything)
 *   remove_non_exact         RemoveNonExactMappingReads   :90-99    CIGAR may hold M and S only
 *   remove_duplicates        RemoveDuplicateReads         :136-138  FLAG & 0x400 as the column is now
 *   use_regions              RemoveNonOverlappingReads    :310-328  regions[refid] = [n_regions[refid]][2] {Start, End} as
 *                            intervals.FromBed makes them, sorted by Start and flattened; intervals.Overlap's comparisons */
typedef struct elp_predicates {
  int remove_unmapped, remove_unmapped_strict, min_mapq, remove_non_exact, remove_duplicates, use_regions;
  const int32_t *const *regions;
  const int64_t *n_regions;
} elp_predicates;
int elp_filter_records(elp_ctx *ctx, const elp_predicates *p, uint64_t *n_rejected_out /* may be NULL */);
/* cgo form (no pointers to pointers): the target regions of all contigs behind each other, region_off[r] .. region_off[r + 1] = the
 * {Start, End} rows of refid r (n_ref + 1 offsets, counted in intervals); regions NULL = no target-region test */
int elp_filter_records_flat(elp_ctx *ctx, int remove_unmapped, int remove_unmapped_strict, int min_mapq, int remove_non_exact, int remove_duplicates,
                            const int32_t *regions, const int64_t *region_off, uint64_t *n_rejected_out /* may be NULL */);

/* ---- `elprep split` / `merge` without touching payloads on the CPU: sam/split-merge.go ----
 * elp_split_classify: SplitFilePerChromosome's routing rule (:280-293) for every staged record: split_out[i] = 0 for RNAME "*",
 *   else group_of_ref[refid] (1..n_groups, computeContigGroups :178-213 on the host); spread_out[i] = 1 if the read also goes to the
 *   spread file (mate in another group); counts_out[n_groups + 2] = records per split (unmapped, groups, spread).  The split ids are
 *   also written to the context's split-id column (elp_batch.split), so that elp_copy_records / elp_exchange_records with new_split = -1
 *   deliver every record with the id of its split file.
 * elp_merge_spread: MergeSortedFilesSplitPerChromosome (:410-576) as ranks: both contexts coordinate-sorted; slot_of_spread_out[j] =
 *   output slot of the j-th record of `spread`'s sorted output among `groups`' sorted output (behind every group read of its
 *   (refid, POS) and in front of the first greater one; group reads fill the remaining slots in order). */
int elp_split_classify(elp_ctx *ctx, const int32_t *group_of_ref, int32_t n_groups, uint16_t *split_out, uint8_t *spread_out, uint64_t *counts_out);
/* elp_copy_records: the write side of the same routing (:280-293 writes the record into the file of its split, and a copy tagged sr:i:1
 *   into the group file if the original goes to the spread file): appends the records idx[0 .. n) of `src` (staging indices, any order)
 *   to `dst` - two contexts of this process on one GPU or on two (device-to-device / peer copies of gathered column slices; nothing
 *   passes through the host).  new_split >= 0: the split id the copies get (else they keep theirs); tag_sr = 1: live records arrive as
 *   sr-tagged copies; tag_sr = 2: only the records whose index has bit 31 set do (the index is its low 31 bits: one call delivers a
 *   split file's records in input order, tagged copies among the others, as :280-293 writes them).  FLAG and QUAL travel as they are now; the inflated BAM records travel too if both contexts hold them
 *   (elp_stage_bam).  Large sets move in pieces inside the call; on an error the pieces already appended stay (elp_num_records tells). */
int elp_copy_records(elp_ctx *dst, elp_ctx *src, const uint32_t *idx, uint64_t n, int new_split, int tag_sr);
int elp_merge_spread(elp_ctx *groups, elp_ctx *spread, uint64_t *slot_of_spread_out);

/* ---- coordinate sort: By(CoordinateLess).ParallelStableSort (sam/sam-types.go:425-473, 639-641) ----
 * Builds the permutation on device: perm[k] = staging index of the record at sorted position k; records equal
 * under all nine keys of CoordinateLess keep staging order.  Payload permutation is the caller's (host) work.
 * The comparator's modFlag(FLAG) tie-break reads the FLAG column as it is at the time of the call.  In `elprep filter` the sort is
 * the Finalize step of the phase-1 pipeline (sam/filter-pipeline.go:116): it runs BEHIND the filters, so with --mark-duplicates it
 * sees the duplicate bits.  A drop-in host therefore calls elp_mark_duplicates first, then elp_sort_coordinate.
 * Concurrency (round 6): the sort runs on a SIDE LANE of the context - a stream, scratch pool and error words of its own; it reads the
 * key column and the comparator's columns and writes the permutation, nothing the BQSR calls touch.  Once elp_mark_duplicates has
 * returned, a host may therefore call elp_sort_coordinate from a thread of its own (a goroutine locked to its OS thread) WHILE another
 * thread calls elp_dup_metrics (side lane of its own, below) and a third drives elp_bqsr_gather(_device) -> finalize -> elp_bqsr_apply
 * on the same context: the three chains need nothing of each other (the reference runs them one after the other,
 * cmd/filter.go:162-196).  The call returns when the permutation is complete; calls that read it afterwards (elp_get_permutation,
 * elp_emit_*) are ordered behind it.  Calls that CHANGE staged records (staging, reset, rollback, filters, exchange) must not overlap
 * with any other call on the context. */
int elp_sort_coordinate(elp_ctx *ctx);
/* on != 0: the host announces that elp_sort_coordinate will follow elp_mark_duplicates on this context (what `elprep filter
 * --mark-duplicates --sorting-order coordinate` does, sam/filter-pipeline.go:116).  The sort's key passes read the coordinate keys only -
 * not the duplicate bits, which only the comparator's tail (modFlag, sam/sam-types.go:447-452) looks at - so elp_mark_duplicates queues
 * them on the sort lane as soon as it has made the keys, and they run while it finishes; elp_sort_coordinate then breaks the ties on the
 * final FLAGs.  Same permutation either way; without a sort behind it the option costs the passes' time.  (Measured with bench.py's step,
 * where the gather follows mark duplicates at once: no gain - the GPU is busy either way, profiles/round6_sort_ahead_ab.txt; it is for a
 * host that has work of its own between the two calls.) */
int elp_sort_ahead(elp_ctx *ctx, int on);
int elp_get_permutation(elp_ctx *ctx, uint32_t *perm_out /* n */);
/* number of records that survive RemoveOptionalReads = staged records without the sr tag: the first elp_num_sorted() entries of
 * the permutation are the output of the run, the tagged copies follow behind them */
uint64_t elp_num_sorted(const elp_ctx *ctx);

/* ---- mark duplicates: filters.MarkDuplicates (filters/mark-duplicates.go:398-445) ----
 * Sets FLAG |= 0x400 on the staged flag column exactly as the reference's fragment/pair tournaments do
 * (deterministic total order: score, then QNAME, then later arrival — the single-threaded execution of the
 * reference).  also_opticals is accepted for signature parity (it only changes which reads carry a LIBID temp). */
int elp_mark_duplicates(elp_ctx *ctx, int also_opticals);
int elp_get_flags(elp_ctx *ctx, uint16_t *flag_out /* n, staging order */);
/* adapted values of filters/mark-duplicates.go:79-110 (unclipped 5' position) and :57-68 (Phred-sum score);
 * 0 for records that are not duplicate-marking candidates.  Either pointer may be NULL. */
int elp_get_adapted(elp_ctx *ctx, int32_t *upos_out, int32_t *score_out);

/* ---- DuplicationMetrics counters: filters.MarkOpticalDuplicates (filters/mark-optical-duplicates.go:469-525) ----
 * counters: [(n_lib + 1)][7] int64 in the order UnpairedReadsExamined, ReadPairsExamined, SecondaryOrSupplementaryReads,
 * UnmappedReads, UnpairedReadDuplicates, ReadPairDuplicates, ReadPairOpticalDuplicates; row n_lib = "Unknown Library".
 * Requires elp_mark_duplicates.  Derived float metrics (PERCENT_DUPLICATION, ESTIMATED_LIBRARY_SIZE, :527-569) and the
 * Picard text (:608-699) stay on the host.  Runs on a side lane of the context (see elp_sort_coordinate): it reads what mark duplicates
 * left and writes nothing another call reads, so it may be called from a second host thread while the context sorts or gathers. */
#define ELP_NCTR 7
int elp_dup_metrics(elp_ctx *ctx, int optical_pixel_distance, int64_t *counters);
/* The same plus the three set-size histograms the reference keeps per library (duplicatesCountHistogram,
 * nonOpticalDuplicatesCountHistogram, opticalDuplicatesCountHistogram, filters/mark-optical-duplicates.go:104-174, 310-324):
 * hist = [(n_lib + 1)][3][hist_len] int64; a set of n listed reads with k optical duplicates counts into bin n of the first,
 * bin n - k (if > 0) of the second and bin k + 1 (if k > 0) of the third histogram; indices >= hist_len count into the last bin
 * (the reference's maps are unbounded); hist_len >= 2. */
int elp_dup_metrics_hist(elp_ctx *ctx, int optical_pixel_distance, int64_t *counters, int64_t *hist, int hist_len);

/* ---- BQSR inputs: NewBaseRecalibrator(knownSites, referenceFasta) (filters/bqsr.go:424-443) ----
 * bases = raw .elfasta bytes of one contig (fasta.MappedFasta.Seq, fasta/fasta-files.go:355);
 * start_end = known-site intervals [n][2], 1-based inclusive, ALREADY sorted by start and flattened
 * (intervals.ParallelSortByStart + ParallelFlatten, intervals/intervals.go:77-132) — the host keeps doing that. */
int elp_bqsr_set_reference(elp_ctx *ctx, int32_t refid, const uint8_t *bases, int64_t len);
int elp_bqsr_set_known_sites(elp_ctx *ctx, int32_t refid, const int32_t *start_end, int64_t n);

/* ---- BQSR gather: BaseRecalibrator.Recalibrate (filters/bqsr.go:467-551) ----
 * Walks the records in any order (integer sums) and fills dense {observations, mismatches} int64 tables:
 *   qual_tbl  [n_cov][94][2]
 *   cycle_tbl [n_cov][94][2*max_cycle+1][2]     cycle c at index c + max_cycle
 *   ctx_tbl   [n_cov][94][16][2]                context key k (filters/bqsr.go:64-76) at index (k >> 4) & 15
 * A Go map entry of the reference exists iff observations > 0.  Uses the staged flag column (duplicates excluded,
 * recalibrateAln filters/bqsr.go:225-244), so call after elp_mark_duplicates. */
#define ELP_NQUAL 94
#define ELP_NCTX 16
int elp_bqsr_gather(elp_ctx *ctx, int max_cycle, int64_t *qual_tbl, int64_t *cycle_tbl, int64_t *ctx_tbl);

/* The same, but the tables stay in HBM (ctx-owned) for the device group's all-reduce below; elp_bqsr_tables_fetch copies them out.
 * The copy runs on a stream of its own behind the last writer of the tables, and it is the ONE call that may run on a second host
 * thread while the context is busy with another call (elp_dup_metrics): the host fetches and finalises the tables while the GPU
 * counts optical duplicates - the two steps the reference runs one after the other (cmd/filter.go:162-196). */
int elp_bqsr_gather_device(elp_ctx *ctx, int max_cycle);
int elp_bqsr_tables_fetch(elp_ctx *ctx, int64_t *qual_tbl, int64_t *cycle_tbl, int64_t *ctx_tbl);
/* The rows form of the device tables (round 5): with many read groups the dense tables are tens of megabytes around a few hundred rows
 * that hold anything (a Go map of the reference has those entries only, filters/bqsr.go:445-459).  elp_bqsr_quals_counted: bit q of
 * bits[q / 64] = quality q had table slots in the gather that made the tables (no other quality's rows hold anything).
 * elp_bqsr_tables_fetch_rows: the rows of `quals` only - q_rows [n_cov][n_quals][2], c_rows [n_cov][n_quals][2*max_cycle+1][2], x_rows
 * [n_cov][n_quals][16][2]; returns 1 and copies nothing if a quality that was not asked for has observations (tables summed with
 * another context's or rank's: fetch the dense tables then).  Same threading rule as elp_bqsr_tables_fetch. */
int elp_bqsr_quals_counted(elp_ctx *ctx, uint64_t *bits /* [2] */);
int elp_bqsr_tables_fetch_rows(elp_ctx *ctx, const uint8_t *quals, int n_quals, int64_t *q_rows, int64_t *c_rows, int64_t *x_rows);

/* ---- device group and its one collective: the table / metrics combination of `elprep sfm`'s merge phase ----
 * LoadAndCombineBQSRTables (filters/print-bqsr.go:310-329) and LoadAndCombineDuplicateMetrics
 * (filters/mark-optical-duplicates.go:711-731) sum what the per-split `filter --bqsr-tables-only` runs produced.  One process per
 * GPU; rank 0 makes the group id (RCCL's ncclUniqueId) and the host hands it to the other ranks by whatever channel it has (the Go
 * host: a file next to the split files, or its parent `sfm` process); every rank then joins with its context.  A group of one needs
 * no id and makes every collective a no-op.
 *   elp_bqsr_tables_add        dst += src on the device: the contexts (splits) of ONE rank are summed before the collective
 *   elp_bqsr_tables_allreduce  ONE ncclAllReduce(ncclInt64, ncclSum) over xGMI, in place on the tables in HBM; `counters`
 *                              (n_counters <= 4096 int64 in host memory, e.g. the duplication counters) ride behind the tables
 *                              through the same call and come back summed
 *   elp_allreduce_i64          the same collective for a plain host buffer */
#define ELP_GROUP_ID_BYTES 128
int elp_group_probe(void); /* 0 if the communication library (RCCL) can be loaded and has every entry point the group needs; no GPU needed */
int elp_group_unique_id(uint8_t *id_out /* ELP_GROUP_ID_BYTES */);
int elp_group_init(elp_ctx *ctx, int rank, int world, const uint8_t *id /* ELP_GROUP_ID_BYTES; may be NULL if world == 1 */);
/* The same group over a transport of the caller's (a host without RCCL, or a host program that already has its own communicator - MPI,
 * gloo, the Go side's net/rpc): `allreduce` must sum `n` int64 values over all ranks, in place, and return 0; it is called on the calling
 * thread of elp_bqsr_tables_allreduce / elp_allreduce_i64 with a page-locked host buffer (the device tables make one round trip over PCIe).
 * Replaces: nothing in the reference (elprep sfm merges per-split tables through files, cmd/split-filter-merge.go:413-470). */
typedef int (*elp_allreduce_fn)(void *user, int64_t *values, size_t n);
int elp_group_init_transport(elp_ctx *ctx, int rank, int world, elp_allreduce_fn allreduce, void *user);
/* Point-to-point messages of such a group: `sendrecv` must deliver send_bytes from send_buf to rank send_peer and fill recv_buf with the
 * recv_bytes that rank recv_peer sends to this rank in its matching call (a peer of -1: nothing that way), then return 0; page-locked
 * host buffers.  An RCCL group (elp_group_init) needs no callback: ncclSend / ncclRecv over xGMI, device to device. */
typedef int (*elp_sendrecv_fn)(void *user, int send_peer, const void *send_buf, size_t send_bytes, int recv_peer, void *recv_buf, size_t recv_bytes);
int elp_group_set_p2p(elp_ctx *ctx, elp_sendrecv_fn sendrecv, void *user);
/* The split phase of `elprep sfm` between PROCESSES (sam/split-merge.go:280-293 writes a record to the file of the split it belongs
 * to; with one process per GPU the "file" is a context of another rank): one step of the all-to-all.  The records `idx` (staging
 * indices, n of them) of `src` are gathered on its GPU as elp_copy_records gathers them - new_split / tag_sr as there - and sent to rank
 * send_peer of the device group; the records that rank recv_peer selected for this rank in ITS matching call are appended to `dst`.  Either
 * direction may be absent (peer -1; then src resp. dst may be NULL).  Every rank calls it world - 1 times, step s with send_peer =
 * (rank + s) % world and recv_peer = (rank - s + world) % world; src or dst must belong to the group.  No host copy of a record: the
 * records move in pieces of at most 4 M records (columns below 4 GiB each); per piece and direction a 160-byte header (with the sender's
 * status), an 8-byte verdict back from the receiver, then three device buffers (fixed columns, variable-length pools, scans).  A failure
 * on either side of a direction reaches the other side in the header or the verdict: both calls return an error, neither waits in a
 * message its peer will not post. */
int elp_exchange_records(elp_ctx *src, int send_peer, const uint32_t *idx, uint64_t n, int new_split, int tag_sr, elp_ctx *dst, int recv_peer);
/* elp_group_share: `ctx` joins the device group `member` belongs to by BORROWING its communicator / transport (same process, same
 * device; `member` must outlive the use, calls of the two contexts on the group must not overlap): the reader context of the split phase
 * exchanges records through the group the rank's processing context reduces its tables in - one communicator per rank. */
int elp_group_share(elp_ctx *ctx, elp_ctx *member);
int elp_group_rank(const elp_ctx *ctx);
int elp_group_size(const elp_ctx *ctx);
int elp_bqsr_tables_add(elp_ctx *dst, elp_ctx *src);
int elp_bqsr_tables_allreduce(elp_ctx *ctx, int64_t *counters, size_t n_counters);
int elp_allreduce_i64(elp_ctx *ctx, int64_t *buf, size_t n);

/* ---- BQSR apply: BaseRecalibratorTables.ApplyBQSR (filters/bqsr.go:936-1005) ----
 * lut: [n_cov][94][2*max_cycle+1][17] bytes = the reference's memo map applyKey{rg, qual, cycle, context} -> uint8,
 * densely tabulated by the host from the finalized tables (context index 16 = key -1); cov_present[c] = 0 means the read
 * group is absent from the tables (read left untouched, :953-955).  Rewrites the staged qual column in place. */
int elp_bqsr_apply(elp_ctx *ctx, int max_cycle, const uint8_t *lut, const uint8_t *cov_present);
/* The LUT's way to the device ahead of the call: from the thread that built it (Go: the goroutine that ran FinalizeBQSRTables), on a copy
 * stream of the context, while other calls (sort, metrics) run on the context from another thread; elp_bqsr_apply(ctx, max_cycle, NULL,
 * NULL) then uses it.  The only other call that may share a context with running calls is elp_bqsr_tables_fetch.  Ordering the
 * caller must keep: no staging call (elp_stage*, elp_reset, elp_rollback) and no elp_bqsr_gather* runs on the context at the same time -
 * the upload reads the staged read length and the gather's quality hint without a lock; sort, metrics, emit and tables_fetch may.
 * A `lut` in page-locked memory (elp_pinned_alloc) is copied from where it lies (no staging copy: with 16 read groups the LUT is 25 MB):
 * the caller then leaves it unchanged until the elp_bqsr_apply that uses it has been called and the context synchronised. */
int elp_bqsr_lut_upload(elp_ctx *ctx, int max_cycle, const uint8_t *lut, const uint8_t *cov_present);
/* The same for the LUT in rows form: rows [n_cov][n_quals][2*max_cycle+1][17] for the qualities `quals`, defaults [n_cov][94] = the one byte
 * every other (read group, quality) row consists of (a row without cycle and context entries is its prior whatever the cycle and the
 * context); the dense LUT is made from them on the device.  With 16 read groups 1.9 MB instead of 25.6 MB cross PCIe. */
int elp_bqsr_lut_upload_rows(elp_ctx *ctx, int max_cycle, const uint8_t *quals, int n_quals, const uint8_t *rows, const uint8_t *defaults,
                             const uint8_t *cov_present);
int elp_get_qual(elp_ctx *ctx, uint8_t *qual_out /* qual_bytes, staging order and offsets */);

/* ---- CleanSam (filters/simple-filters.go:292-306, `elprep filter --clean-sam`: a filter of the phase-1 pipeline, cmd/filter.go:747) ----
 * On the staged records, in front of elp_mark_duplicates: MAPQ of unmapped reads becomes 0; an alignment whose End() lies behind the
 * LN of its reference sequence is soft-clipped there by softClipEndOfRead (filters/utils.go:102-119) - with that function's arithmetic as
 * it stands in the reference (the running position accumulates, the clip's length is ReadLengthFromCigar + clipFrom): a drop-in writes
 * what the reference writes.  The CIGAR column is rebuilt if any record is rewritten (a rewritten CIGAR can be one operation longer).
 * n_clipped_out (may be NULL): records rewritten.  ELP_ERR_DATA where the reference panics ("Unexpected non-0 relative clipping
 * position in CleanSam."), ELP_ERR_UNSUPPORTED for a clip length that does not fit the 28 bits of a BAM CIGAR field. */
int elp_clean_sam(elp_ctx *ctx, uint64_t *n_clipped_out);

/* Measurement and test harness entry points of the same library (snapshot / rollback of the mutable columns, pinned kernel choices, per-kernel
 * timing) are declared in elprep_hip_debug.h: they are not part of the reference's interface and a drop-in host does not need them. */

#ifdef __cplusplus
}
#endif
#endif /* ELPREP_HIP_H */
