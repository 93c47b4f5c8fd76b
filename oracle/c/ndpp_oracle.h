/*
 * ndpp_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C (gcc, IEEE double, no contraction, no fast-math) restatement of the
 * reference's scattering-moment hot path.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product
 * (ndpp_amd/, libndpp_hip.so) never links, imports or calls it.
 *
 * Parity status: PINNED -- every function here is checked in
 * tests/test_oracle_vs_ref.py against the real reference Fortran compiled by
 * flang (oracle/_ref/libndpp_ref.so) when that library is present, and in
 * tests/test_oracle_golden.py against committed golden vectors generated from
 * it (tests/golden/, generator tests/golden/make_golden.py).
 *
 * All citations are file:line under /root/reference/src.
 */
#ifndef NDPP_ORACLE_H
#define NDPP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* module global's hidden numerics (global.F90:32-59; defaults
 * constants.F90:70-100) plus order / mu_bins.  Same layout as the product's
 * ndpp_params in include/ndpp_hip.h so one ctypes.Structure serves both. */
typedef struct {
  int    order;              /* L = scatt_order + 1 moments                   */
  int    mu_bins;            /* M, points of the uniform mu grid              */
  double sab_threshold;      /* SAB_THRESHOLD        (1e-6)                   */
  double brent_mu_thresh;    /* BRENT_MU_THRESH      (1e-6)                   */
  double adaptive_mu_tol;    /* ADAPTIVE_MU_TOL      (1e-7)                   */
  double adaptive_eout_tol;  /* ADAPTIVE_EOUT_TOL    (1e-8)                   */
  int    adaptive_mu_its;    /* ADAPTIVE_MU_ITS      (15)                     */
  int    adaptive_eout_its;  /* ADAPTIVE_EOUT_ITS    (15)                     */
  int    ne_per_grp;         /* NE_PER_GRP           (20)                     */
  int    sab_epts_per_bin;   /* SAB_EPTS_PER_BIN     (10)                     */
  int    extend_pts;         /* EXTEND_PTS           (50)                     */
  int    inel_extend_pts;    /* INEL_EXTEND_PTS      (30)                     */
} oracle_params;

void   oracle_default_params(oracle_params *p);

/* legendre.F90:349 */
double oracle_calc_pn(int n, double x);
/* legendre.F90:22 (orders 0..10 restated; the reference allows <=10) */
void   oracle_calc_int_pn_tablelin(int n, double xlo, double xhi, double flo,
                                   double fhi, double *integrals);
/* search.F90:21 ; returns 1-based index, or -1 where the reference aborts */
int    oracle_binary_search(const double *a, int n, double v);
/* scattdata_header.F90:251-257 */
void   oracle_mu_grid(int M, double *mu);

/* freegas.F90:154,188,235,356,415,482,563,18 */
void   oracle_calc_fg_eout_bounds(double A, double kT, double Ein,
                                  double *lo, double *hi);
double oracle_calc_sab(double A, double kT, double Ein, double Eout,
                       double beta, double mu);
double oracle_brent_mu(const oracle_params *p, double A, double kT, double Ein,
                       double Eout, double beta, double thresh, double lo,
                       double hi);
void   oracle_find_fg_mu(const oracle_params *p, double A, double kT,
                         double Ein, double Eout, double mu2[2]);
double oracle_calc_fgk(double A, double kT, double Ein, double Eout, int l,
                       double mu, const double *fEmu, const double *gmu, int M);
double oracle_adaptive_simpsons_mu(const oracle_params *p, double A, double kT,
                                   double Ein, double Eout, int l,
                                   const double *fEmu, const double *gmu, int M,
                                   double a, double b);
double oracle_adaptive_simpsons_eout(const oracle_params *p, double A,
                                     double kT, double Ein, int l,
                                     const double *fEmu, const double *gmu,
                                     int M, double a, double b);
/* distro is [G][L] (L fastest) == Fortran distro(order, groups) */
void   oracle_integrate_freegas_leg(const oracle_params *p, double Ein,
                                    double A, double kT, const double *fEmu,
                                    const double *gmu, const double *E_bins,
                                    int nbins, double *distro);

/* scattdata_header.F90:1466, :956.  distro must be pre-zeroed by the caller,
 * as the reference's callers do (scattdata_header.F90:545-546,569). */
double oracle_tolab(double R, double w);
void   oracle_integrate_file4_cm_leg(const oracle_params *p, const double *fw,
                                     double Ein, double awr, double Q,
                                     const double *E_bins, int nbins,
                                     const double *w, double *distro);

/* Batched elastic moments = the adist-only branch of integrate_distro
 * (scattdata_header.F90:533-591) for n_ein points: both bracketing rows are
 * integrated at the same E_in and blended (1-f)*lo + f*hi.  E_in below
 * freegas_cutoff use integrate_freegas_leg, the rest integrate_file4_cm_leg
 * with Q.  Same argument meaning as ndpp_elastic_leg_batch in
 * include/ndpp_hip.h.  out is [n_ein][G][L].  nthreads<=0: all cores.
 * counters (may be NULL): [0] += number of calc_fgk evaluations.            */
int    oracle_elastic_leg_batch(const oracle_params *p, double A, double kT,
                                double freegas_cutoff, double Q, int n_ein,
                                const double *ein, const int *row_lo,
                                const double *w_hi, int n_rows,
                                const double *f_tab, int G,
                                const double *e_bins, double *out, int nthreads,
                                unsigned long long *counters);

/* ---- file 6 family (oracle/c/file6.c) ---- */
/* array_merge.F90:13 */
int    oracle_merge(const double *a, int na, const double *b, int nb, double *res);
/* interpolation.F90:24 */
double oracle_interpolate_tab1(const double *data, double x);
/* scattdata_header.F90:1554 / :1521+:1616 */
int    oracle_cast_to_unitbase(const double *Eout, int np, double *ub);
int    oracle_unitbase(double Ein, int M, int np1, const double *eout1, const double *pdf1,
                       int intt1, const double *f1, double Ei1, int np2,
                       const double *eout2, const double *pdf2, int intt2,
                       const double *f2, double Ei2, double *Eout, double *pdf, int *INTT,
                       double *fEmu);
/* scattdata_header.F90:1085, :1334, :1274 (distro [G][L], pre-zeroed) */
void   oracle_integrate_file6_cm_leg(const oracle_params *p, const double *fEmu, int np,
                                     const double *mu, double Ein, double awr,
                                     const double *Eout, int INTT, const double *thispdf,
                                     const double *E_bins, int nbins, double *distro);
void   oracle_integrate_file6_lab_leg(const oracle_params *p, const double *fEmu, int np,
                                      const double *mu, const double *Eout, int INTT,
                                      const double *thispdf, const double *E_bins,
                                      int nbins, double *distro);
void   oracle_law9_scatter_lab_leg(const oracle_params *p, const double *fmu,
                                   const double *edata, double Ein, const double *E_bins,
                                   int nbins, const double *mu, double *distro);
/* edist branches of integrate_distro (:593-656), batched; see file6.c */
int    oracle_file6_leg_batch(const oracle_params *p, double awr, int frame_cm, int n_ein,
                              const double *ein, const int *row_lo, int n_rows,
                              const double *e_grid, const int *row_ptr, const double *eout,
                              const double *pdf, const int *intt, const double *f, int G,
                              const double *e_bins, double *out, int nthreads);
int    oracle_law9_leg_batch(const oracle_params *p, int n_ein, const double *ein,
                             const int *row_lo, const double *w_hi, int n_rows,
                             const double *f_tab, const double *edata, int G,
                             const double *e_bins, double *out, int nthreads);

/* ---- thermal S(alpha,beta) tables (oracle/c/sab.c) ----
 * Flattened SAlphaBeta (ace_header.F90:201-235); identical layout to
 * ndpp_sab_flat of include/ndpp_hip.h.  Arrays keep the Fortran element order. */
typedef struct {
  double threshold_inelastic, threshold_elastic;
  int n_inelastic_e_in, n_inelastic_e_out, n_inelastic_mu, secondary_mode;
  const double *inelastic_e_in;   /* [NEi]                                      */
  const double *inelastic_sigma;  /* [NEi]                                      */
  const double *inelastic_e_out;  /* [NEi][NEo]       == (NEo, NEi)   modes 0,1 */
  const double *inelastic_mu;     /* [NEi][NEo][NMU]  == (NMU,NEo,NEi) modes 0,1 */
  const int    *cont_ptr;         /* [NEi+1]                          mode 2    */
  const double *cont_e_out;       /* [sum]                                      */
  const double *cont_pdf;         /* [sum]                                      */
  const double *cont_mu;          /* [sum][NMU]                                 */
  int elastic_mode, n_elastic_e_in, n_elastic_mu;
  const double *elastic_e_in;     /* [NEe]                                      */
  const double *elastic_P;        /* [NEe]                                      */
  const double *elastic_mu;       /* [NEe][NMUe]      == (NMUe, NEe)            */
} oracle_sab_flat;

void oracle_sab_el(const oracle_params *p, const oracle_sab_flat *t, int NE, const double *ein,
                   int G, const double *e_bins, double *out);
int  oracle_sab_inel_disc(const oracle_params *p, const oracle_sab_flat *t, int NE,
                          const double *ein, int G, const double *e_bins, double *out);
void oracle_sab_inel_cont(const oracle_params *p, const oracle_sab_flat *t, int NE,
                          const double *ein, int G, const double *e_bins, double *out);
void oracle_sab_combine(int L, int G, int NE, const double *el, const double *inel, double *mat);
int  oracle_calc_scattsab(const oracle_params *p, const oracle_sab_flat *t, int NE,
                          const double *ein, int G, const double *e_bins, double *el,
                          double *inel, double *mat);
/* scatt.F90:786 */
void oracle_apply_tol_scatt(int L, int G, int n, double *data, double tol);
int  oracle_sab_egrid(const oracle_params *p, const oracle_sab_flat *t, int nb,
                      const double *e_bins, double *out, int cap);

/* ---- fission spectrum chi (oracle/c/chi.c) ----
 * One energy distribution of a fission reaction (prompt; one per nested edist)
 * or of a delayed precursor group; identical layout to ndpp_chi_spectrum. */
typedef struct {
  int law, n_data;           /* ACE law (4, 61, 7, 9, 11) and edist%data          */
  const double *data;
  int threshold, n_sigma;    /* prompt: rxn%threshold and the reaction's sigma     */
  const double *sigma;       /*   (nuc%fission itself for MT 18, chi.F90:72-76)    */
  int has_next;              /* associated(edist%next)                             */
  int pv_n_regions, pv_n_pairs; /* edist%p_valid (Tab1)                            */
  const int *pv_nbt, *pv_int;
  const double *pv_x, *pv_y;
} oracle_chi_spectrum;

typedef struct {
  int n_grid;
  const double *energy, *fission;          /* nuc%energy, nuc%fission              */
  int nu_t_type, n_nu_t;                   /* 1 polynomial, 2 tabular (TAB1)       */
  const double *nu_t_data;
  int nu_d_type, n_nu_d;                   /* 0 none, 2 tabular                    */
  const double *nu_d_data;
  int n_precursor, n_prec_data;
  const double *nu_d_precursor_data;
} oracle_chi_nuclide;

int  oracle_chi_egrid(int n_prompt, const oracle_chi_spectrum *prompt, int n_delay,
                      const oracle_chi_spectrum *delay, double *out, int cap);
void oracle_calc_chi(const oracle_chi_nuclide *n, int n_prompt,
                     const oracle_chi_spectrum *prompt, int n_delay,
                     const oracle_chi_spectrum *delay, int G, const double *E_bins, int NE,
                     const double *E_grid, double *chi_t, double *chi_p, double *chi_d);

/* ---- ACE -> tabular conversion (oracle/c/convert.c); data arrays are the Fortran
 * arrays, addressed with the reference's 1-based index arithmetic ---- */
/* scattdata_header.F90:669 for one incoming energy; out[M] pre-zeroed */
void oracle_convert_file4_row(int type, int lc, const double *data, const double *mu, int M,
                              double *out);
void oracle_convert_file4(int M, int n_rows, const int *type, const int *location,
                          const double *data, double *f_tab);
int  oracle_file6_ne(const double *data);
int  oracle_file6_np(const double *data, int iE);
/* scattdata_header.F90:769 for incoming energy iE (1-based); distro [NP][M] */
int  oracle_convert_file6_row(int law, const double *data, int iE, const double *mu, int M,
                              double *eouts, double *pdf, double *cdf, int *INTT, double *distro);
/* scatt_convert_distro (:325) for a law 4/44/61 ScattData -> CSR tables */
int  oracle_convert_file6(int M, int law, const double *data, int n_adist,
                          const double *adist_energy, const int *adist_type,
                          const int *adist_location, const double *adist_data, double *e_grid,
                          int *row_ptr, double *eout, double *pdf, double *cdf, int *intt,
                          double *f);

#ifdef __cplusplus
}
#endif
#endif
