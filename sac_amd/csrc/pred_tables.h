// sac_amd/csrc/pred_tables.h -- NLMS step-size / power-normalisation tables.
// Reference: NLMS_Stream constructor, /root/reference/src/pred/ls.h:34-43:
//   powtab[i] = 1/(1+i)^pow_decay, mutab[i] = mu_decay^i, sum_powtab = sum_i powtab[i] (in order).
#pragma once
#include "libm_port.h"
#include "simt.h"

namespace sacamd {

SA_HD void lms_table_entry(int i, double mu_decay, double pow_decay, double *mutab, double *powtab) {
  *powtab = 1.0 / (sa_pow((double)(1 + i), pow_decay));
  *mutab = sa_pow(mu_decay, (double)i);
}

// ---- lane-major copies of the tables for the canonical-order cascade layouts (pred_lms.h, CANON 2) ----
// The canonical layouts read tap 8k + c (dot) resp. 4k + c4 (power sum) in lane (c, m), slot j with k = r*LPC*J + m*J + j:
// from the natural tables that is a gather with a lane stride of 8J doubles (one cache line per lane).  k_tables therefore
// also writes, per stage, mutab in dot order and powtab in power-sum order, element (round r, slot j, lane l) at
// (r*J + j)*256 + l: every load of the kernel is then one contiguous 2 KB row per slot.
constexpr int kCanonNL = 256;
SA_HD constexpr int canon_slots(int s) { return s == 0 ? 9 : s == 1 ? 5 : s == 2 ? 3 : 1; }
SA_HD int canon_rounds_of_class(int lms_class) { return lms_class == 7 ? 1 : (lms_class == 8 ? 2 : 4); }
// doubles of one stage's block pair {mutab (dot order), powtab (power-sum order)} and of all four stages
SA_HD long long canon_stage_doubles(int s, int rounds) { return 2LL * rounds * canon_slots(s) * kCanonNL; }
SA_HD long long canon_tab_doubles(int rounds) { long long d = 0; for (int s = 0; s < 4; s++) d += canon_stage_doubles(s, rounds); return d; }
SA_HD int canon_mt_index(int J, int tap) {     // tap < 8 * floor(n / 8)
  const int k = tap >> 3, c = tap & 7, r = k / (32 * J), q = k % (32 * J), m = q / J, j = q % J;
  return (r * J + j) * kCanonNL + (c >> 1) * 64 + (c & 1) * 32 + m;
}
SA_HD int canon_pt_index(int J, int tap) {     // tap < 4 * floor(n / 4)
  const int k = tap >> 2, c4 = tap & 3, r = k / (64 * J), q = k % (64 * J), lw = q / J, j = q % J;
  return (r * J + j) * kCanonNL + c4 * 64 + lw;
}

}  // namespace sacamd
