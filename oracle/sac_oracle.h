/* oracle/sac_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * C-ABI of this repo's own CPU restatement of the Sac encode hot path (oracle/sac_oracle.cpp).
 * It is the checker for the HIP path; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  Nothing under sac_amd/ links, loads or calls it.
 *
 * Function-for-function it mirrors oracle/ref_driver.cpp (the genuine reference classes), so
 * tests can run the same call against both and compare.
 */
#ifndef SAC_ORACLE_H
#define SAC_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_frame_cfg {
  int optimize;
  double fraction;
  int maxnfunc;
  int num_threads;
  double sigma;
  int optk;
  int cost; /* 0 L1, 1 RMS, 2 Entropy, 3 Golomb, 4 Bitplane */
  int reset;
  int sparse_pcm;
  int zero_mean;
} orc_frame_cfg;

int orc_profile(float *out /*58*3: vmin,vmax,vdef*/);
void orc_domain_tables(int *fwd /*32768*/, int *inv /*4095*/);
int orc_predict_frame(int nch, int framesize, int total, const int32_t *samples,
                      const int32_t *stats, const float *coefs, int from, int n, int optimize,
                      int optk, int32_t *error, int32_t *pred);
int orc_predict_trace(int nch, int total, const int32_t *samples, const int32_t *stats,
                      const float *coefs, int from, int n, int optimize, int optk,
                      double *pd_out, double *plpc_out, double *plms_out, int32_t *error);
double orc_cost(int kind, const int32_t *buf, int n);
int orc_bitplane_encode(const int32_t *s2u, int n, int maxbpn, uint8_t *out, int cap);
int orc_bitplane_trace(const int32_t *s2u, int n, int maxbpn, uint16_t *p1s, uint8_t *bits,
                       int maxdec);
int orc_bitplane_decode(const uint8_t *in, int len, int n, int maxbpn, int32_t *err_out);
int orc_rangecoder_encode(const uint16_t *p1s, const uint8_t *bits, int n, uint8_t *out, int cap);
double orc_remap(const int32_t *raw, int n, const int32_t *pred, const int32_t *error,
                 int32_t *s2u_map, int *maxbpn_map, uint8_t *usedl, uint8_t *usedh);
int orc_mapencode(const uint8_t *usedl, const uint8_t *usedh, uint8_t *out, int cap);
void orc_analyse(const int32_t *raw, int n, int32_t *out /*mean,min,max*/);
void orc_rng(int n, const int *kinds, const double *args, double *out);
void orc_gen_norm(double x, double xmin, double xmax, double r, int n, double *out);
double orc_reflect(double x, double xmin, double xmax);
void orc_ssc(int which, int n, const double *lambdas, double sigma0, double *out);
double orc_dds_quadratic(int ndim, const double *xmin, const double *xmax, const double *xstart,
                         const double *center, int nfunc_max, int num_threads, double sigma,
                         double *xbest, double *trace_cost);
int orc_encode_frame(int nch, int framesize, int n, const int32_t *raw, const orc_frame_cfg *rc,
                     float *profile_io, uint8_t *out, int cap, double *trace_cost,
                     float *trace_coefs, int *info);
/* FrameCoder::SearchMethod for the following orc_encode_frame calls: 0 DDS (default), 1 DE (opt/de.cpp), 2 CMA (opt/cma.cpp);
 * with 1 / 2 the trace arrays must hold maxnfunc + 32 entries */
void orc_set_search_method(int search);
int orc_decode_frame(const uint8_t *rec, int len, int nch, int framesize, int32_t *out,
                     int cap_samples, float *coefs_out);
double orc_dot(const double *x, const double *y, int n);
double orc_s2pow(const double *x, const double *p, int n);
int orc_ldlt(const double *A, int n, double nu, const double *b, double *w);
/* PredictLaplace table for one plane: out[avg_sum] for avg_sum < count (vle.cpp:70-79) */
void orc_laplace_table(int bpn, int count, uint16_t *out);
int orc_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
