/*
 * gnss_oracle.c — CPU oracle (TEST INFRASTRUCTURE ONLY): scalar float64 C restatement of the
 * reference's tracking hot path, written independently of oracle/gnss_oracle.py so the two
 * can be checked against each other, and used as the timed "MATLAB-equivalent CPU restatement"
 * baseline (bench.py cpu_baseline, kind "port", 1 core — the reference processes channels and
 * epochs strictly serially, tracking.m:133,184).
 *
 * PARITY STATUS: "parity unpinned by the reference" (no MATLAB/Octave here, no golden vectors in
 * the reference).  Never linked into or called by the product library.
 *
 * Citations are file:line under /root/reference/GPS/GPS_L1CA/ unless a path is given.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_PI 3.141592653589793

/* ---- MATLAB colon operator (MathWorks' published algorithm), element access ---------------- */
typedef struct {
  double a, d, c;
  long n; /* number of intervals; vector has n+1 elements */
} orc_colon;

static double orc_round(double x) { return x >= 0 ? floor(x + 0.5) : -floor(-x + 0.5); }

static orc_colon orc_colon_make(double a, double d, double b) {
  orc_colon r;
  double tol = 2.0 * 2.220446049250313e-16 * fmax(fabs(a), fabs(b));
  double sig = d > 0 ? 1.0 : -1.0;
  long n;
  if (a == floor(a) && d == 1.0)
    n = (long)(floor(b) - a);
  else if (a == floor(a) && d == floor(d))
    n = (long)floor((b - a) / d);
  else {
    n = (long)orc_round((b - a) / d);
    if (sig * (a + n * d - b) > tol) n -= 1;
  }
  double c = a + n * d;
  if (sig * (c - b) > -tol) c = b;
  r.a = a;
  r.d = d;
  r.c = c;
  r.n = n;
  return r;
}

static inline double orc_colon_at(const orc_colon* r, long i) {
  /* first half forwards from a, second half backwards from c, exact middle averaged */
  if (2 * i < r->n) return r->a + (double)i * r->d;
  if (2 * i > r->n) return r->c - (double)(r->n - i) * r->d;
  return (r->a + r->c) / 2.0;
}

/* ---- generateCAcode.m:42-90 ------------------------------------------------------------------ */
static const int kG2s[51] = {5,   6,   7,   8,   17,  18,  139, 140, 141, 251, 252, 254, 255,
                             256, 257, 258, 469, 470, 471, 472, 473, 474, 509, 512, 513, 514,
                             515, 516, 859, 860, 861, 862, 145, 175, 52,  21,  237, 235, 886,
                             657, 634, 762, 355, 1012, 176, 603, 130, 359, 595, 68,  386};

void orc_generate_ca(int prn, double* out /*1023*/) {
  double g1[1023], g2[1023], reg[10];
  int i, k;
  for (k = 0; k < 10; ++k) reg[k] = -1;
  for (i = 0; i < 1023; ++i) {
    g1[i] = reg[9];
    double save = reg[2] * reg[9];
    for (k = 9; k > 0; --k) reg[k] = reg[k - 1];
    reg[0] = save;
  }
  for (k = 0; k < 10; ++k) reg[k] = -1;
  for (i = 0; i < 1023; ++i) {
    g2[i] = reg[9];
    double save = reg[1] * reg[2] * reg[5] * reg[7] * reg[8] * reg[9];
    for (k = 9; k > 0; --k) reg[k] = reg[k - 1];
    reg[0] = save;
  }
  int sh = kG2s[prn - 1];
  for (i = 0; i < 1023; ++i) {
    double g2s = g2[(i + 1023 - sh) % 1023]; /* [g2(1023-sh+1:1023) g2(1:1023-sh)] */
    out[i] = -(g1[i] * g2s);
  }
}

/* ---- one correlator block, tracking.m:247-300 (R-scaled: GAL_E1C/include/tracking.m:236-303) -- */
/* iq: int8 interleaved I,Q (fileType 2); tables: arms padded tables of table_len entries each. */
void orc_correlate_block(const int8_t* iq, int64_t first_sample, int n, const double* tables, int arms,
                         int table_len, double rem, double step, double d, double r, double mult,
                         double carr_freq, double rem_carr, double fs, double code_length, int swap_iq,
                         double* sums /*arms*6*/, double* rem_code_new, double* rem_carr_new) {
  orc_colon te = orc_colon_make((rem - d) * r, step * r, ((n - 1) * step + rem - d) * r);
  orc_colon tl = orc_colon_make((rem + d) * r, step * r, ((n - 1) * step + rem + d) * r);
  orc_colon tp = orc_colon_make(rem * r, step * r, ((n - 1) * step + rem) * r);
  double tlast = orc_colon_at(&tp, n - 1);
  *rem_code_new = (r != 1.0) ? (tlast / r + step) - code_length : (tlast + step) - code_length; /* :273 */
  double w = carr_freq * 2.0 * ORC_PI;
  *rem_carr_new = fmod((w * ((double)n / fs)) + rem_carr, 2 * ORC_PI); /* :280-283 */
  for (int a = 0; a < arms * 6; ++a) sums[a] = 0.0;
  const int8_t* p = iq + 2 * first_sample;
  for (long i = 0; i < n; ++i) {
    double trig = (w * ((double)i / fs)) + rem_carr;    /* :280-281 */
    double cs = cos(trig), sn = sin(trig);              /* exp(-1i*trig) = cs - 1i*sn, :287 */
    double x0 = (double)p[2 * i], x1 = (double)p[2 * i + 1];
    double re = swap_iq ? x1 : x0, im = swap_iq ? x0 : x1;
    double ib = cs * re + sn * im;                      /* real(carrsig .* rawSignal), :291 */
    double qb = cs * im - sn * re;                      /* imag(...), :292 */
    long ke = (long)ceil(orc_colon_at(&te, i) * mult);  /* ceil(tcode)+1, 0-based here, :255 */
    long kl = (long)ceil(orc_colon_at(&tl, i) * mult);
    long kp = (long)ceil(orc_colon_at(&tp, i) * mult);
    for (int a = 0; a < arms; ++a) {
      const double* t = tables + (size_t)a * table_len;
      double* s = sums + a * 6;
      s[0] += t[ke] * ib; /* I_E :295 */
      s[1] += t[ke] * qb; /* Q_E */
      s[2] += t[kp] * ib; /* I_P */
      s[3] += t[kp] * qb; /* Q_P */
      s[4] += t[kl] * ib; /* I_L */
      s[5] += t[kl] * qb; /* Q_L */
    }
  }
}

/* Common/calcLoopCoef.m:41-45 */
static void orc_loop_coef(double lbw, double zeta, double k, double* tau1, double* tau2) {
  double wn = lbw * 8 * zeta / (4 * zeta * zeta + 1);
  *tau1 = k / (wn * wn);
  *tau2 = 2.0 * zeta / wn;
}

/* Field order of the output rows (n_epochs doubles each), mirrors the product's gc_track_field
 * for the first 15 entries so tests can compare row by row. */
enum { F_ABS = 0, F_CODEF, F_CARRF, F_IE, F_QE, F_IP, F_QP, F_IL, F_QL, F_DLL, F_DLLF, F_PLL, F_PLLF, F_REMC, F_REMP, F_N };

/* tracking.m:133-368 for GPS L1 C/A, channels strictly serial like the reference.
 * out: [nch][F_N][n_epochs]; epochs_done[nch].  Returns 0, or 1 on the short-read early return. */
int orc_track_l1ca(const int8_t* iq, int64_t n_samples, int nch, const int* prn, const double* acquired_freq,
                   const int64_t* code_phase, double fs, double code_freq_basis, double code_length,
                   double el_spacing, double int_time, double dll_bw, double dll_zeta, double pll_bw,
                   double pll_zeta, int64_t skip_samples, int n_epochs, double* out, int* epochs_done) {
  double tau1code, tau2code, tau1carr, tau2carr;
  orc_loop_coef(dll_bw, dll_zeta, 1.0, &tau1code, &tau2code);   /* :100-102 */
  orc_loop_coef(pll_bw, pll_zeta, 0.25, &tau1carr, &tau2carr);  /* :109-110 */
  memset(out, 0, sizeof(double) * (size_t)nch * F_N * n_epochs);
  for (int c = 0; c < nch; ++c) epochs_done[c] = 0;
  for (int c = 0; c < nch; ++c) {
    if (prn[c] == 0) continue; /* :136 */
    double* o = out + (size_t)c * F_N * n_epochs;
    double ca[1023], table[1025];
    orc_generate_ca(prn[c], ca);
    table[0] = ca[1022];
    memcpy(table + 1, ca, sizeof ca);
    table[1024] = ca[0]; /* :158 */
    int64_t pos = skip_samples + code_phase[c] - 1; /* :150-152 */
    double code_freq = code_freq_basis, rem_code = 0.0;
    double carr_freq = acquired_freq[c], carr_basis = acquired_freq[c], rem_carr = 0.0;
    double old_code_nco = 0, old_code_err = 0, old_carr_nco = 0, old_carr_err = 0;
    for (int e = 0; e < n_epochs; ++e) {
      o[F_ABS * n_epochs + e] = (double)pos;                       /* :212-216 */
      double step = code_freq / fs;                                 /* :219 */
      int n = (int)ceil((code_length - rem_code) / step);           /* :222 */
      if (pos + n > n_samples) return 1;                            /* :241-245 */
      o[F_REMC * n_epochs + e] = rem_code;                          /* :249 */
      o[F_REMP * n_epochs + e] = rem_carr;                          /* :277 */
      double s[6], rc, rp;
      orc_correlate_block(iq, pos, n, table, 1, 1025, rem_code, step, el_spacing, 1.0, 1.0, carr_freq, rem_carr,
                          fs, code_length, 0, s, &rc, &rp);
      pos += n;
      rem_code = rc;
      rem_carr = rp;
      double i_e = s[0], q_e = s[1], i_p = s[2], q_p = s[3], i_l = s[4], q_l = s[5];
      double carr_err = atan(q_p / i_p) / (2.0 * ORC_PI);           /* :305 */
      double carr_nco = old_carr_nco + (tau2carr / tau1carr) * (carr_err - old_carr_err) +
                        carr_err * (int_time / tau1carr);           /* :308-309 */
      old_carr_nco = carr_nco;
      old_carr_err = carr_err;
      o[F_CARRF * n_epochs + e] = carr_freq;                        /* :314 */
      carr_freq = carr_basis + carr_nco;                            /* :317 */
      double em = sqrt(i_e * i_e + q_e * q_e), lm = sqrt(i_l * i_l + q_l * q_l);
      double code_err = (em - lm) / (em + lm);                      /* :322-323 */
      double code_nco = old_code_nco + (tau2code / tau1code) * (code_err - old_code_err) +
                        code_err * (int_time / tau1code);           /* :326-327 */
      old_code_nco = code_nco;
      old_code_err = code_err;
      o[F_CODEF * n_epochs + e] = code_freq;                        /* :332 */
      code_freq = code_freq_basis - code_nco;                       /* :335 */
      o[F_DLL * n_epochs + e] = code_err;
      o[F_DLLF * n_epochs + e] = code_nco;
      o[F_PLL * n_epochs + e] = carr_err;
      o[F_PLLF * n_epochs + e] = carr_nco;
      o[F_IE * n_epochs + e] = i_e;
      o[F_QE * n_epochs + e] = q_e;
      o[F_IP * n_epochs + e] = i_p;
      o[F_QP * n_epochs + e] = q_p;
      o[F_IL * n_epochs + e] = i_l;
      o[F_QL * n_epochs + e] = q_l;
      epochs_done[c] = e + 1;
    }
  }
  return 0;
}
