// apply_rec.hpp - the 8-byte per-read record of ApplyBQSR for read sets of one length (apply3.hip).  It holds nothing the LUT decides:
// since round 6 the score kernel of the adapt stage writes it as it goes (sort.hip: k_score_uniform has the read's flags and low-quality-tail
// bounds at hand) and k_bqsr_apply3 tests the read group's presence in the tables itself - a pass of its own over four columns
// (k_apply_records, 0.3 ms per 50 M reads) is only made where the adapt stage ran another score kernel.
#pragma once
#include "common.hpp"

namespace elp {

enum : uint32_t { AR_ON = 1u << 8, AR_REV = 1u << 9, AR_NEG = 1u << 10 };
constexpr uint32_t AR_NO_RG = 0xFFFFFFFFu;  // x of the record of a read without a read group ({AR_NO_RG, 0}): readGroupCovariate panics (filters/bqsr.go:38)

// per read: x = context window lo | hi << 16 (bases whose context covariate is valid), y = covariate | AR_* | (cf + lmax) << 16
__device__ __forceinline__ uint2 apply_record(uint32_t len, int lmax, uint16_t f, uint64_t qb, uint32_t cov) {
  const bool rev = f & F_REVERSED;
  const uint32_t hi1 = (uint32_t)qb;
  const int left = hi1 ? (int)(qb >> 32) : (int)len, right = hi1 ? (int)hi1 - 1 : (int)len - 1;
  int cl = left + (rev ? 0 : 1), cr1 = right - (rev ? 1 : 0) + 1;
  cl = cl < 0 ? 0 : cl;
  cr1 = cr1 > (int)len ? (int)len : cr1;
  cr1 = cr1 < cl ? cl : cr1;
  const int rof = (f & F_LAST) ? -1 : 1;
  const int cf = rof + (rev ? ((int)len - 1) * rof : 0), ci = rev ? -rof : rof;
  return make_uint2((uint32_t)cl | ((uint32_t)cr1 << 16), (cov & 0xFFu) | AR_ON | (rev ? AR_REV : 0u) | (ci < 0 ? AR_NEG : 0u) | ((uint32_t)(cf + lmax) << 16));
}

}  // namespace elp
