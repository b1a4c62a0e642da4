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

}  // namespace sacamd
