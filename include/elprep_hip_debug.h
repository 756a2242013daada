/* elprep_hip_debug.h - harness entry points of libelprep_hip.so that are NOT part of the drop-in boundary (include/elprep_hip.h).
 *
 * The reference has no counterpart for any of them: a Go host binds elprep_hip.h only.  bench.py, the parity tests and the A/B tools use
 * these to re-run the path on resident input (elp_snapshot / elp_rollback), to pin a kernel choice per context (elp_set_tuning) and to
 * read per-kernel times (elp_profile_*).  Same shared object, separate header.
 *
 * Two process-wide switches for fault hunts, read from the environment when the library first allocates / launches:
 *   ELP_DEBUG_POISON=<byte>  every new device buffer of a context is filled with that byte before its first use: a kernel that reads
 *                            memory nothing wrote shows as a parity failure instead of depending on what the allocator hands out;
 *                            a buffer that is regrown is overwritten with 0xDD before it is freed (a stale pointer into it reads that)
 *   ELP_DEBUG_TRACE=1        every kernel launch is named on stderr and waited for (which kernel faulted; ~100 x slower)
 *   ELP_DEBUG_GUARD=1        4 KB of pattern behind every device buffer of a context, checked when the buffer is released or regrown and by
 *                            elp_debug_check_guards: a kernel that writes past the end of its buffer aborts the process (buffer size on stderr) */
#ifndef ELPREP_HIP_DEBUG_H
#define ELPREP_HIP_DEBUG_H
#include "elprep_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- snapshot of the two columns the path mutates (FLAG by elp_mark_duplicates, QUAL by elp_bqsr_apply) ----
 * elp_snapshot copies them aside in HBM; elp_rollback restores them and invalidates derived state (sort keys, scores,
 * duplicate tables).  Lets a host re-run the path on identical input (bench.py's timed steps; `--bqsr-tables-only`
 * style what-if runs) without re-staging over PCIe. */
/* number of live device buffers whose guard pattern was overwritten (0 without ELP_DEBUG_GUARD) */
int elp_debug_check_guards(void);

int elp_snapshot(elp_ctx *ctx);
int elp_rollback(elp_ctx *ctx);

/* ---- kernel choices ----
 * The library picks its kernels from the staged data (read sets of one length, number of distinct qualities, order of the
 * mates).  Tests and A/B measurements pin a choice per context with elp_set_tuning instead of process-wide environment
 * variables; value 0 (or -1 where 0 is a value) gives the choice back to the library.  The reference has no counterpart: its
 * one code path per operator is what every choice here must reproduce bit for bit.
 *   "count_kernel"     1: general BQSR count kernel even for read sets of one length; 2: the one-length kernel with one table for all
 *                      covariates (never the covariate split); 3: the one-length kernel split by covariate wherever it applies
 *   "apply_kernel"     1: general ApplyBQSR kernel; 3: the one-length kernel split by covariate even where one table holds every covariate
 *   "exchange_piece"   > 0: records per piece of elp_exchange_records (default 4 M, less for long records: a piece's columns stay below 4 GiB)
 *   "bgzf_stored"      1: elp_emit_sorted_bgzf frames stored DEFLATE blocks (BTYPE 00) instead of compressing
 *   "bgzf_piece"       inflated bytes per record-scan pass of elp_stage_bgzf (default 1 GiB)
 *   "bgzf_inflate_piece"  inflated bytes whose blocks one launch of the decoder takes (default 2 GiB; the scan passes of
 *                      "bgzf_piece" bytes run inside it; token scratch: 171 KB per 64 KB block)
 *   "bgzf_fixed"       1: elp_emit_sorted_bgzf writes fixed Huffman codes only (round 5) instead of the blocks' own codes
 *   "bgzf_copy_chunk"  blocks per H2D chunk and decoder launch of elp_stage_bgzf (0: the blocks that fill the chip once);
 *   "bgzf_first_chunk_div"  the first chunk is 1/div of that (default 4: the decoder starts early)
 *   "bgzf_tok_fail_above"  (tests) the decoder's token scratch "does not fit" for more than this many blocks: the launch halves
 *   "bgzf_tok_lds"     (experiments) unused LDS bytes per decoder wave: fewer waves per CU
 *   "bgzf_inflate"     1: round 5's decoder (one kernel: a wave decodes and copies a block, window in LDS; separate CRC pass)
 *                      instead of round 6's two phases (tokens: 64 candidate symbols per wave and step; matches + CRC: a workgroup)
 *   "bgzf_weak_guess"  1: elp_stage_bgzf's blocks guess their first record start blindly (every guess is then repaired: same result)
 *   "score_kernel"     1: general Phred-score / low-quality-tail kernel even for read sets of one length
 *   "count3_rlog"      >= 0: log2 of the context-cell replication of the one-length count kernel (measurements)
 *   "qual_hint"        1: no sampled quality hint (the gather sizes its tables on the report-and-retry path)
 *   "qual_hint_drop"   q >= 0: quality q is removed from the sampled hint (the kernels' no-slot paths)
 *   "pair_table_slots" cap on the LDS table slots per pair bucket of elp_mark_duplicates (a power of two >= 2; 0 = no cap): a
 *                      small value sends every bucket through the overflow path
 *   "presort_tile"     1 / 2 / 3: radix tile (4096 / 8192 / 16384 keys) of the key passes elp_sort_ahead queues from inside mark duplicates (default 2)
 *   "side_priority"    1: the side lanes' streams (sort, metrics) get the highest stream priority instead of the default - set before their first use
 *   "apply_wgs"        1 .. 3: workgroups per CU of the one-length ApplyBQSR kernel (default: what its LDS allows, at most 3)
 *   "md_fused"         1: mark duplicates by the separate passes of rounds 2-5 (adapt_fixed, md_keys, md_mate_scan, md_mate_pairs) instead of
 *                      the fused front pass of round 6 (md_front) - same flags; A/B timing and the tests run both
 *   "mate_path"        1: every mate candidate is matched by the partitioned pass (hash partition + LDS tables), no neighbour
 *                      shortcut - what coordinate-ordered or shuffled input takes by itself; 2: ... by the table in HBM
 *   "radix_tile"       1: every radix pass in tiles of 4096 keys; 2: of 8192 keys; 3: of 16384 (default: by the array's length)
 *   "sort_pairs"       1: the coordinate sort moves (key, index) pairs through its passes even where key << b | index fits one word
 *   "tie_rounds"       1: the coordinate sort orders its long runs of equal coordinates (the unmapped block, pile-ups) by radix rounds
 *                      over every live name position - the path a group of > 1024 names that agree in their leading positions takes
 *                      by itself - instead of one round on the leading positions + comparison of what it leaves equal
 * Returns ELP_ERR_ARG for an unknown key or a value out of range. */
int elp_set_tuning(elp_ctx *ctx, const char *key, int64_t value);

/* ---- measurement ----
 * With profiling on, every kernel launch is bracketed by hipEvents on the ctx stream; elp_profile_get returns, per
 * kernel name, the launch count and the summed duration in milliseconds. */
int elp_profile_enable(elp_ctx *ctx, int on);
int elp_profile_reset(elp_ctx *ctx);
int elp_profile_count(elp_ctx *ctx);
int elp_profile_get(elp_ctx *ctx, int index, const char **name, uint64_t *launches, double *total_ms);

#ifdef __cplusplus
}
#endif
#endif /* ELPREP_HIP_DEBUG_H */
