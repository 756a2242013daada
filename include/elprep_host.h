/*
 * elprep_host.h — C ABI of libelprep_host.so: the host-side (CPU, float64) pieces that sit between the device
 * kernels of libelprep_hip.so, mirroring what elPrep's Go host code keeps doing in the drop-in design
 * ("Float finalisation and report printing deliberately stay on the host side").
 *
 * A Go adapter does NOT need this library: it keeps calling the reference's own FinalizeBQSRTables /
 * PrintBQSRTables / PrintDuplicatesMetrics and only needs elp_bqsr_lut_* semantics (see INTEGRATION.md for the
 * Go restatement of the LUT decomposition).  The C++ host layer and the Python harness of this repository use it.
 */
#ifndef ELPREP_HOST_H
#define ELPREP_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct elp_bqsr_tables elp_bqsr_tables;

/* BaseRecalibratorTables (filters/bqsr.go:445-459) as dense arrays; shapes as in elp_bqsr_gather. Copies the inputs. */
elp_bqsr_tables *elp_bqsr_tables_new(int n_cov, int max_cycle, const int64_t *qual_tbl, const int64_t *cycle_tbl, const int64_t *ctx_tbl);
/* The same from the rows of the qualities that can hold anything (elp_bqsr_tables_fetch_rows): quals ascending; q_rows [n_cov][n_quals][2],
 * c_rows [n_cov][n_quals][2*max_cycle+1][2], x_rows [n_cov][n_quals][16][2]; every other row is empty. */
elp_bqsr_tables *elp_bqsr_tables_new_rows(int n_cov, int max_cycle, const uint8_t *quals, int n_quals, const int64_t *q_rows, const int64_t *c_rows,
                                          const int64_t *x_rows);
void elp_bqsr_tables_free(elp_bqsr_tables *t);
/* bqsrTable.merge (filters/bqsr.go:210-223) / LoadAndCombineBQSRTables (filters/print-bqsr.go:310-329): t += other */
int elp_bqsr_tables_merge(elp_bqsr_tables *t, const int64_t *qual_tbl, const int64_t *cycle_tbl, const int64_t *ctx_tbl);
/* FinalizeBQSRTables (filters/bqsr.go:677-694) + initializeCombinedBQSRTable (:655-674) */
int elp_bqsr_tables_finalize(elp_bqsr_tables *t);
/* EmpiricalQuality per entry, 255 = entry absent.  Shapes [n_cov][94], [n_cov][94][2*max_cycle+1], [n_cov][94][16]. */
int elp_bqsr_tables_empirical(const elp_bqsr_tables *t, uint8_t *qual_emp, uint8_t *cycle_emp, uint8_t *ctx_emp);
/* per read-group covariate: reportedQuality, EmpiricalQuality, Observations, Mismatches, present flag */
int elp_bqsr_tables_combined(const elp_bqsr_tables *t, double *reported, uint8_t *emp, int64_t *obs, int64_t *mism, uint8_t *present);
/* initializeQuantizedQualityScores (filters/bqsr.go:863-899): counts[94], scores[94]; levels 0 = identity */
int elp_bqsr_tables_quantize(const elp_bqsr_tables *t, int levels, int64_t *counts, uint8_t *scores);
/* Dense tabulation of ApplyBQSR's memo map (filters/bqsr.go:936-1003): lut [n_cov][94][2*max_cycle+1][17],
 * cov_present [n_cov].  sqq = --sqq list (may be NULL, n_sqq 0). */
int elp_bqsr_tables_build_lut(const elp_bqsr_tables *t, int quantize_levels, const uint8_t *sqq, int n_sqq, uint8_t *lut, uint8_t *cov_present);
/* The same LUT in rows form: rows [n_cov][n_quals][2*max_cycle+1][17] for the qualities `quals`, defaults [n_cov][94] = the one byte every
 * other (covariate, quality) row consists of; -2 if a quality outside `quals` has table entries.  (elp_bqsr_lut_upload_rows expands it.) */
int elp_bqsr_tables_build_lut_rows(const elp_bqsr_tables *t, int quantize_levels, const uint8_t *sqq, int n_sqq, const uint8_t *quals, int n_quals, uint8_t *rows,
                                   uint8_t *defaults, uint8_t *cov_present);
/* PrintBQSRTables (filters/print-bqsr.go:269-298) into a malloc'd string (free with elp_host_free) */
char *elp_bqsr_tables_report(const elp_bqsr_tables *t, const char *const *cov_names, const char *tablename_prefix);
void elp_host_free(void *p);

/* calculateDerivedDuplicateMetrics (filters/mark-optical-duplicates.go:527-569) for one library row of elp_dup_metrics */
int elp_dup_derived(const int64_t *ctr7, double *percent_duplication, int64_t *estimated_library_size);
/* PrintDuplicatesMetrics (filters/mark-optical-duplicates.go:608-699) without the timestamp line's clock value and without set-size
 * histograms (elp_dup_metrics_report_hist adds them; the reference loses them in sfm mode, :702-731).  lib_names has n_lib entries;
 * row n_lib is "Unknown Library".  Returns a malloc'd string. */
char *elp_dup_metrics_report(const int64_t *counters, int n_lib, const char *const *lib_names, const char *command_line);
/* The same with the "## HISTOGRAM" block (:628-697): written when exactly one library has pairs - the return-on-investment column
 * (histogramRoi :580-588) and, per set size 1..100 and every larger size that occurs, the number of duplicate sets / sets with optical
 * duplicates / sets with non-optical duplicates from elp_dup_metrics_hist's histograms [(n_lib + 1)][3][hist_len] (sizes >= hist_len - 1
 * are merged into the last bin there: pass a hist_len above the largest set to get the reference's rows). */
char *elp_dup_metrics_report_hist(const int64_t *counters, const int64_t *hist, int hist_len, int n_lib, const char *const *lib_names,
                                  const char *command_line);

#ifdef __cplusplus
}
#endif
#endif
