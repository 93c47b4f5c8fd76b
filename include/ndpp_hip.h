/*
 * ndpp_hip.h -- C ABI of libndpp_hip.so, the MI355X (gfx950) implementation of
 * NDPP's scattering-moment integration hot path.
 *
 * The reference (ndpp/ndpp, Fortran) has no FFI; its seams are module
 * procedures.  Each entry point below names the reference procedure it
 * replaces (file:line under the reference's src/).  Conventions:
 *   - all reals are IEEE double, all integers 32-bit;
 *   - arrays are 0-based but keep the Fortran (column-major) element order, so
 *     a Fortran caller can pass c_loc(array) unchanged:  distro(order,groups)
 *     is double[G][L] with the Legendre index fastest;
 *   - pointers are HOST pointers unless the function name ends in _d;
 *   - the caller allocates outputs; outputs are fully written;
 *   - return 0 on success, a negative errno-style code otherwise; the library
 *     never calls exit()/stop (the reference's fatal_error -> stop,
 *     error.F90:79-154, becomes a return code + per-point status word);
 *   - the caller selects the device (hipSetDevice) before calling; calls are
 *     re-entrant per device; there is NO CPU fallback: without a usable HIP
 *     device every compute entry point returns NDPP_EDEVICE.
 * INTEGRATION.md shows the ISO_C_BINDING interface block for each function.
 */
#ifndef NDPP_HIP_H
#define NDPP_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NDPP_OK         0
#define NDPP_EDEVICE   (-5)    /* HIP runtime error / no device              */
#define NDPP_ENOMEM    (-12)   /* workspace allocation failed                */
#define NDPP_EINVAL    (-22)   /* bad argument                               */
#define NDPP_EOVERFLOW (-75)   /* adaptive tree outgrew the workspace        */

/* per-E_in status bits */
#define NDPP_ST_OK        0
#define NDPP_ST_NONFINITE 1    /* NaN/Inf in the result row                  */
#define NDPP_ST_RANGE     2    /* E_in / row index outside the tables        */
#define NDPP_ST_ORDER_NOISE 4  /* retired (never set since 0.4).  Rounds 2-3 raised it on file-6 /
                                  law-9 batches with order > 8: their moments of orders 8-10 agreed
                                  with the reference's only to its own rounding noise (1e-10 ...
                                  3e-10, legendre.F90:46-140).  Those moments are now evaluated in the
                                  reference's operation order and hold the 1e-10 bar up to P10. */

#define NDPP_MAX_ORDER 11      /* L = scatt_order+1 <= 11 (ndpp.F90:290-301) */

/* Module `global`'s hidden numerics (global.F90:32-59, defaults
 * constants.F90:70-100, set from ndpp.xml at ndpp.F90:355-423) plus the two
 * sizes every kernel needs.  Read-only during a call. */
typedef struct ndpp_params {
  int    order;              /* L = scatt_order + 1 Legendre moments          */
  int    mu_bins;            /* M, points of the uniform mu grid (2001)       */
  double sab_threshold;      /* SAB_THRESHOLD        1e-6                     */
  double brent_mu_thresh;    /* BRENT_MU_THRESH      1e-6                     */
  double adaptive_mu_tol;    /* ADAPTIVE_MU_TOL      1e-7                     */
  double adaptive_eout_tol;  /* ADAPTIVE_EOUT_TOL    1e-8                     */
  int    adaptive_mu_its;    /* ADAPTIVE_MU_ITS      15                       */
  int    adaptive_eout_its;  /* ADAPTIVE_EOUT_ITS    15                       */
  int    ne_per_grp;         /* NE_PER_GRP           20                       */
  int    sab_epts_per_bin;   /* SAB_EPTS_PER_BIN     10  (grid builders)      */
  int    extend_pts;         /* EXTEND_PTS           50  (grid builders)      */
  int    inel_extend_pts;    /* INEL_EXTEND_PTS      30  (grid builders)      */
} ndpp_params;

/* Device-side work counters of the last batch call (optional out-param). */
typedef struct ndpp_stats {
  unsigned long long k_evals;      /* free-gas kernel evaluations (exp count) */
  unsigned long long mu_visits;    /* joint mu-tree node visits               */
  unsigned long long mu_integrals; /* inner (mu) adaptive integrals           */
  unsigned long long eout_nodes;   /* outer (E_out) adaptive tree nodes       */
  double             mu_kernel_ms; /* sum of hipEvent spans of fg_mu_kernel   */
  int                mu_kernel_launches;
  double             total_ms;     /* hipEvent time of the whole batch        */
  unsigned long long wave_iters;   /* fg_mu_kernel loop trips, summed over waves */
  unsigned long long lane_iters;   /* ... of which lanes doing a node (<= 64x)   */
  unsigned long long order_visits; /* sum over node visits of active orders      */
  double             mu_level_ms[32]; /* fg_mu_kernel time per outer-tree level  */
  double             mu_busy_ms;   /* time with at least one inner-integration launch
                                      (fg_mu_kernel, fg_gauss_kernel) in flight (the
                                      pipeline contexts of a batch overlap theirs)       */
  int                contexts;     /* pipeline contexts that ran side by side (1 or 2)    */
  unsigned long long gauss_integrals; /* inner integrals (every row of the job) done by the
                                      certified Gauss rule instead of the walk           */
  double             gauss_ms;     /* sum of hipEvent spans of fg_gauss_kernel            */
} ndpp_stats;

void        ndpp_default_params(ndpp_params *p);
const char *ndpp_version(void);
/* message of the last failing call on this thread ("" if none) */
const char *ndpp_last_error(void);
/* Device time (ms, hipEvent bracket) of the kernels of this thread's last batch
 * call; host<->device staging is outside the bracket.  Measurement aid.        */
float ndpp_last_gpu_ms(void);
/* Device time (ms) the calling thread's batch calls have spent per kernel family since the last
 * reset -- what a whole-nuclide call (ndpp_scatt_nuclide) is made of.  Families, in order:
 * free-gas inner walk (fg_mu_kernel), free-gas other stages, file4-CM (+ classification),
 * file6-CM, file6-lab, law 9, S(alpha,beta), chi, ACE -> tabular conversion.  ndpp_profile_get
 * fills min(n, #families) entries and returns the number of families.  Measurement aid.     */
#define NDPP_PROFILE_FAMILIES 9
void        ndpp_profile_reset(void);
int         ndpp_profile_get(double *ms, int n);
/* number of visible HIP devices (0 if none); never fails */
int         ndpp_device_count(void);
/* Which free-gas moments this build integrates in the reference's own arithmetic (about 1.5x the
 * cost of the product arithmetic; replaces nothing in the reference -- it is the reference's
 * freegas.F90:415-553 operation for operation): every incoming energy whose bracketing table
 * rows are not linear in mu (ndpp_freegas_rough_rows: rough[row] = 1; host-side mirror of what
 * the batch calls decide on the device), and, on linear rows, the energies below
 * ndpp_freegas_strict_below (MeV; 0 = never, +inf = always).  For cost models that balance work
 * across GPUs (ndpp_amd/dist.py) and for tests.                                              */
double      ndpp_freegas_strict_below(int groups, double A, double kT);
int         ndpp_freegas_rough_rows(int mu_bins, int n_rows, const double *f_tab, int *rough);
/* Select / query the calling thread's device, for hosts that do not link HIP themselves (a
 * Fortran host with one MPI rank or OpenMP thread per GPU; the reference's ranks each take a
 * block of nuclides, ndpp.F90:934-950).  Every entry point works on the calling thread's
 * current device; the cached workspace and the lock that serialises batch calls are per
 * device, so threads that selected different devices run concurrently.                  */
int         ndpp_set_device(int device);
int         ndpp_get_device(void);      /* ordinal >= 0, or a negative error code */
/* free the cached workspace of the current device */
int         ndpp_release_workspace(void);
/* Allocate the cached device workspace ahead of time: bytes = 0 reserves what the largest
 * batch may take (min(60 % of free HBM, 128 GB)).  Without it the first large batch pays
 * for the allocation (up to seconds for ~100 GB).                                      */
int         ndpp_reserve_workspace(size_t bytes);
/* Device buffers for hosts that call the *_d entry points without linking HIP themselves
 * (e.g. a Fortran host that keeps its tables resident across calls).  ndpp_dev_alloc
 * returns NULL on failure (message in ndpp_last_error); copies are synchronous.          */
void       *ndpp_dev_alloc(size_t bytes);
int         ndpp_dev_free(void *p);
int         ndpp_dev_upload(void *dst_d, const void *src, size_t bytes);
int         ndpp_dev_download(void *dst, const void *src_d, size_t bytes);
int         ndpp_dev_synchronize(void);   /* hipDeviceSynchronize on the current device */

/* ---- B-fine: replaces `subroutine integrate_freegas_leg(Ein, A, kT, fEmu,
 * mu, E_bins, order, distro)` freegas.F90:18-146.  fEmu[M] is f(mu) on the
 * uniform grid mu[M] (scattdata_header.F90:251-257); mu may be NULL, otherwise
 * it must equal that grid bit for bit (NDPP_EINVAL if not -- the reference's
 * calc_fgk indexing, freegas.F90:437-448, is only meaningful on it).
 * distro[G][L] is fully written (normalised so that sum_g P0 = 1).          */
int ndpp_integrate_freegas_leg(const ndpp_params *p, double Ein, double A,
                               double kT, const double *fEmu, const double *mu,
                               const double *E_bins, int n_bins, double *distro);

/* ---- B-fine: replaces `subroutine integrate_file4_cm_leg(fw, Ein, awr, Q,
 * E_bins, w, order, distro)` scattdata_header.F90:956-1078.  Unlike the
 * Fortran (whose callers pre-zero distro, :545-546,:569) distro[G][L] is fully
 * written: groups the routine does not reach are 0.                          */
int ndpp_integrate_file4_cm_leg(const ndpp_params *p, const double *fw,
                                double Ein, double awr, double Q,
                                const double *E_bins, int n_bins,
                                const double *w, double *distro);

/* ---- B-batch: the adist-only branch of `integrate_distro`
 * (scattdata_header.F90:533-591) evaluated for n_ein incoming energies of ONE
 * ScattData (one reaction): for every E_in both bracketing rows of the
 * tabulated angular distribution are integrated at that same E_in and blended
 *     out = lo*(1-f) + hi*f                     (:566,:589)
 * with integrate_freegas_leg where ein < freegas_cutoff (MT=2 only: pass
 * freegas_cutoff = 0 for other reactions) and integrate_file4_cm_leg (with Q)
 * elsewhere (:548-564).
 *   f_tab  [n_rows][M]  this%distro(iE)%data(:,1) rows, contiguous
 *                       (n_rows * M * 8 bytes < 4 GiB per call: NDPP_EINVAL otherwise)
 *   row_lo [n_ein]      0-based lower bracketing row (upper = row_lo+1)
 *   w_hi   [n_ein]      f of :542
 *   out    [n_ein][G][L]
 *   status [n_ein]      NDPP_ST_* bits, may be NULL
 * This is the loop body of calc_elastic_grid (scatt.F90:633-672) minus the
 * sigma/threshold bookkeeping of scatt_interp_distro, which stays host-side. */
int ndpp_elastic_leg_batch(const ndpp_params *p, double A, double kT,
                           double freegas_cutoff, double Q, int n_ein,
                           const double *ein, const int *row_lo,
                           const double *w_hi, int n_rows, const double *f_tab,
                           int G, const double *e_bins, double *out,
                           int *status, ndpp_stats *stats);

/* Same, with ein/row_lo/w_hi/f_tab/e_bins/out/status already resident in
 * device memory (HBM) of the current device; work is enqueued on `stream`
 * (a hipStream_t, NULL = default stream) and the call returns after the batch
 * has completed on the device (it synchronises the stream).                 */
int ndpp_elastic_leg_batch_d(const ndpp_params *p, double A, double kT,
                             double freegas_cutoff, double Q, int n_ein,
                             const double *ein_d, const int *row_lo_d,
                             const double *w_hi_d, int n_rows,
                             const double *f_tab_d, int G,
                             const double *e_bins_d, double *out_d,
                             int *status_d, void *stream, ndpp_stats *stats);

/* Mixed-nuclide form of the two calls above: ONE call for the elastic grids of a whole
 * library shard.  Incoming energy i belongs to nuclide nuc_of_ein[i] (0-based), whose
 * awr, kT, free-gas cutoff and Q are A[k], kT[k], freegas_cutoff[k], Q[k]; row_lo[i]
 * indexes the concatenated table f_tab[n_rows][mu_bins] of all nuclides.  The group
 * structure is common.  Results are bit-identical to per-nuclide calls; the point is
 * occupancy: a nuclide has only 10^2-10^3 free-gas points and every level of the
 * breadth-first pipeline lasts as long as its slowest inner integral, so small batches
 * leave most of the GPU idle (measured: 423 nuclides, 2.1e5 points, 273 s as 423
 * calls vs one call).  _d: all arrays are device pointers.      */
int ndpp_elastic_leg_multi(const ndpp_params *p, int n_nuc, const double *A, const double *kT,
                           const double *freegas_cutoff, const double *Q, int n_ein,
                           const double *ein, const int *nuc_of_ein, const int *row_lo,
                           const double *w_hi, int n_rows, const double *f_tab, int G,
                           const double *e_bins, double *out, int *status, ndpp_stats *stats);
int ndpp_elastic_leg_multi_d(const ndpp_params *p, int n_nuc, const double *A_d,
                             const double *kT_d, const double *cutoff_d, const double *Q_d,
                             int n_ein, const double *ein_d, const int *nuc_of_ein_d,
                             const int *row_lo_d, const double *w_hi_d, int n_rows,
                             const double *f_tab_d, int G, const double *e_bins_d,
                             double *out_d, int *status_d, void *stream, ndpp_stats *stats);

/* ---- B-batch: the edist branches of `integrate_distro`
 * (scattdata_header.F90:593-656) for n_ein incoming energies of ONE ScattData
 * whose secondary distribution is a correlated energy-angle table (ACE laws
 * 4 / 44 / 61 after convert_distro):  unit-base interpolation between the two
 * bracketing incoming energies (`unitbase` :1521, `cast_to_unitbase` :1554,
 * `interp_unitbase` :1616 -- fused, the fEmu(M,|ub|) table is never
 * materialised) followed by
 *   frame_cm = 1: `integrate_file6_cm_leg`  (:1085-1266, uses awr, ne_per_grp)
 *   frame_cm = 0: `integrate_file6_lab_leg` (:1334-1450).
 * The ScattData is passed flattened (CSR over incoming energies):
 *   e_grid  [n_rows]           this%E_grid
 *   row_ptr [n_rows+1]         offsets of each row's outgoing-energy block
 *   eout,pdf[row_ptr[n_rows]]  this%Eouts(k)%data, this%pdfs(k)%data
 *   intt    [n_rows]           this%INTT (1 histogram, 2 lin-lin, 3..5 log forms)
 *   f       [row_ptr[n_rows]][M]  this%distro(k)%data(:, j): one f(mu) column per
 *                              (row, outgoing energy), contiguous in mu
 *   row_lo  [n_ein]            0-based iE of scatt_interp_distro (:471-482)
 *   out     [n_ein][G][L]      fully written; NOT multiplied by sigma*p_valid
 * Every row needs >= 2 outgoing energies.  Bit-identical to the Fortran.      */
int ndpp_file6_leg_batch(const ndpp_params *p, double awr, int frame_cm, int n_ein,
                         const double *ein, const int *row_lo, int n_rows,
                         const double *e_grid, const int *row_ptr,
                         const double *eout, const double *pdf, const int *intt,
                         const double *f, int G, const double *e_bins, double *out,
                         int *status);

/* ---- B-batch: the law-9 branch of `integrate_distro` (:605-638): for every
 * E_in `law9_scatter_lab_leg` (:1274-1326) on both bracketing rows of the
 * tabulated angular distribution f_tab[n_rows][M], blended (1-f)*lo + f*hi.
 * edata[n_edata] is edist%data: the TAB1 of the nuclear temperature T(E)
 * followed by the restriction energy U.                                       */
int ndpp_law9_leg_batch(const ndpp_params *p, int n_ein, const double *ein,
                        const int *row_lo, const double *w_hi, int n_rows,
                        const double *f_tab, int n_edata, const double *edata,
                        int G, const double *e_bins, double *out, int *status);

/* ---- thermal S(alpha,beta) tables ------------------------------------------
 * Flattened `type(SAlphaBeta)` (ace_header.F90:201-235).  Arrays keep the
 * Fortran element order (first index fastest), so a Fortran host passes
 * c_loc(sab%inelastic_mu) etc. unchanged; the continuous secondary mode's
 * per-E_in jagged `inelastic_data(:)` is concatenated (CSR via cont_ptr).     */
typedef struct ndpp_sab_flat {
  double threshold_inelastic, threshold_elastic;   /* 0 elastic: no elastic data */
  int n_inelastic_e_in, n_inelastic_e_out, n_inelastic_mu;
  int secondary_mode;             /* 0 equal, 1 skewed, 2 continuous (constants.F90:144-147) */
  const double *inelastic_e_in;   /* [NEi]                                       */
  const double *inelastic_sigma;  /* [NEi]                                       */
  const double *inelastic_e_out;  /* (NEo, NEi)       modes 0,1                  */
  const double *inelastic_mu;     /* (NMU, NEo, NEi)  modes 0,1                  */
  const int    *cont_ptr;         /* [NEi+1]          mode 2                     */
  const double *cont_e_out;       /* [cont_ptr[NEi]]  inelastic_data(k)%e_out    */
  const double *cont_pdf;         /*                  inelastic_data(k)%e_out_pdf */
  const double *cont_mu;          /* (NMU, sum)       inelastic_data(k)%mu       */
  int elastic_mode;               /* 3 discrete, 4 exact (constants.F90:150-152) */
  int n_elastic_e_in, n_elastic_mu;
  const double *elastic_e_in;     /* [NEe]                                       */
  const double *elastic_P;        /* [NEe]                                       */
  const double *elastic_mu;       /* (NMUe, NEe)                                 */
} ndpp_sab_flat;

/* Replaces the Legendre path of `calc_scattsab` (scatt.F90:543-596):
 * `integrate_sab_el` (sab.F90:21) + `integrate_sab_inel` (:117; discrete :142 or
 * continuous :253) + `combine_sab_grid` (:415) on the incoming grid ein[n_ein]
 * (built by sab_egrid + add_one_more_point on the host).
 *   scatt_mat [n_ein][G][L]   normalised so that sum_g P0 = 1; last point = its
 *                             neighbour (sab.F90:452)
 *   el, inel  [n_ein][G][L]   optional (NULL): the un-normalised sigma-weighted
 *                             parts (sab_int_el / sab_int_inel)
 * Bit-identical to the Fortran.                                               */
int ndpp_sab_batch(const ndpp_params *p, const ndpp_sab_flat *t, int n_ein,
                   const double *ein, int G, const double *e_bins, double *el,
                   double *inel, double *scatt_mat);

/* ---- fission spectrum chi ---------------------------------------------------
 * One secondary energy distribution: a fission reaction's `edist` (one entry per
 * nested distribution, in `edist%next` chain order -- calc_chi, chi.F90:49-87) or
 * a delayed precursor group's (`nuc%nu_d_edist(j)`, :90-93).                    */
typedef struct ndpp_chi_spectrum {
  int law, n_data;              /* ACE law: 4/61 tabular, 7 Maxwell, 9 evaporation,
                                   11 Watt; other laws yield zeros like the reference */
  const double *data;           /* edist%data                                       */
  int threshold, n_sigma;       /* prompt only: rxn%threshold and the reaction's     */
  const double *sigma;          /*   sigma (nuc%fission for MT 18, chi.F90:72-76)    */
  int has_next;                 /* associated(edist%next)                            */
  int pv_n_regions, pv_n_pairs; /* edist%p_valid (Tab1, endf_header.F90:9-20)        */
  const int *pv_nbt, *pv_int;
  const double *pv_x, *pv_y;
} ndpp_chi_spectrum;

typedef struct ndpp_chi_nuclide {
  int n_grid;
  const double *energy, *fission;        /* nuc%energy, nuc%fission [n_grid]          */
  int nu_t_type, n_nu_t;                 /* 1 polynomial, 2 tabular (constants.F90:187) */
  const double *nu_t_data;
  int nu_d_type, n_nu_d;                 /* 0 none, 2 tabular                          */
  const double *nu_d_data;
  int n_precursor, n_prec_data;
  const double *nu_d_precursor_data;     /* per group: lambda, TAB1 of the yield       */
} ndpp_chi_nuclide;

/* Replaces the incoming-energy loop of `calc_chi` (chi.F90:124-159):
 * ChiData%beta / %prob / %integrate (chidata_header.F90:139-493) for every
 * spectrum, their combination (including the overwrite at chi.F90:135) and the
 * three normalisations.  e_grid[n_ein] is the union grid of the spectra
 * (chi.F90:97-113; ndpp_amd.grid.chi_egrid on the host).
 *   chi_t, chi_p [n_ein][G]            == chi_total/chi_prompt(groups, NE)
 *   chi_d        [n_delay][n_ein][G]   == chi_delay(groups, NE, n_precursor)       */
int ndpp_chi_batch(const ndpp_chi_nuclide *nuc, int n_prompt,
                   const ndpp_chi_spectrum *prompt, int n_delay,
                   const ndpp_chi_spectrum *delay, int G, const double *e_bins,
                   int n_ein, const double *e_grid, double *chi_t, double *chi_p,
                   double *chi_d);

/* ---- ACE -> tabular conversion ------------------------------
 * One reaction and the (nested) energy distribution in hand, as ScattData%init
 * receives them (scattdata_header.F90:78): pointers into the Nuclide's own arrays. */
typedef struct ndpp_ace_reaction {
  int MT;                        /* rxn%MT                                            */
  int law;                       /* edist%law; 0: no energy distribution passed       */
  int has_angle_dist;            /* rxn%has_angle_dist                                */
  int n_adist;                   /* rxn%adist%n_energy                                */
  const double *adist_energy;    /* rxn%adist%energy  [n_adist]                       */
  const int    *adist_type;      /* rxn%adist%type    [n_adist] (constants.F90:138-141) */
  const int    *adist_location;  /* rxn%adist%location[n_adist]                       */
  int n_adist_data;
  const double *adist_data;      /* rxn%adist%data                                    */
  int n_edata;
  const double *edata;           /* edist%data (law 4 / 44 / 61 / 9 blocks)           */
  double threshold_energy;       /* nuc%energy(rxn%threshold)                         */
} ndpp_ace_reaction;

/* Replaces the shape decisions of `ScattData%init` (scattdata_header.F90:78-271):
 * is_init = 0 for a non-scattering MT or an unsupported law (the reference leaves
 * the object uninitialised, :101,:105-109); sd_law = this%law; NE incoming
 * energies; total_np = sum over them of the number of outgoing energies.  Host only. */
int ndpp_scattdata_shape(const ndpp_ace_reaction *r, int *is_init, int *sd_law, int *NE,
                         int *total_np);

/* Replaces `ScattData%init` + `scatt_convert_distro` (scattdata_header.F90:78,:325;
 * convert_file4 :669, convert_file6 :769): evaluates every ACE angular
 * distribution of the reaction on the uniform mu grid.  Outputs are the tables
 * ndpp_elastic_leg_batch / ndpp_file6_leg_batch / ndpp_law9_leg_batch take:
 *   e_grid [NE], row_ptr [NE+1], eout/pdf/cdf [total_np] (zeros where the reference
 *   has none), intt [NE], f [total_np][mu_bins] (== distro(iE)%data(mu_bins, NP)).
 * Reference behaviour kept on purpose: a law-4 table receives the angular
 * distribution in its first two outgoing-energy columns only (:364-366, sic);
 * unknown angular types / file-4 interpolation codes give zero columns.  Where the
 * reference aborts or reads out of bounds this returns NDPP_EINVAL.              */
int ndpp_convert_distro(int mu_bins, const ndpp_ace_reaction *r, int G, const double *e_bins,
                        int NE, int total_np, double *e_grid, int *row_ptr, double *eout,
                        double *pdf, double *cdf, int *intt, double *f);

/* ---- incoming-energy grids of one nuclide, host only --------
 * What the grid builders read of a ScattData: is_init, rxn%MT, rxn%Q_value and its
 * incoming-energy grid (after ndpp_convert_distro / ScattData%init).             */
typedef struct ndpp_sd_grid {
  int is_init;
  int MT;
  double Q_value;
  int n;
  const double *e_grid;
} ndpp_sd_grid;

/* Replaces `merge(a, b, result)` array_merge.F90:13-107 (a zero becomes MIN_EIN
 * 1e-14 unless it meets an equal value).  *n_out is always set; out is written
 * when cap suffices.                                                             */
int ndpp_merge_grids(int na, const double *a, int nb, const double *b, int cap, double *out,
                     int *n_out);

/* Replaces `create_Ein_grid` scatt.F90:166-243 (combine_Eins :252, add_elastic_Eins
 * :313, add_one_more_point :426, add_inelastic_Eins :456): the elastic grid and,
 * if any non-elastic ScattData is initialised, the inelastic grid (else
 * *n_inel = 0).  cutoff = the elastic ScattData's freegas_cutoff (MeV; 0 when the
 * free-gas treatment is off), thresh = the lowest non-elastic threshold energy
 * (calc_scatt, scatt.F90:103-123).  Uses p->extend_pts / p->inel_extend_pts.
 * Two-call protocol: counts are always returned, arrays are written when their
 * capacity suffices.                                                             */
int ndpp_create_ein_grid(const ndpp_params *p, int n_sd, const ndpp_sd_grid *sds, int n_bins,
                         const double *e_bins, int n_nuc, const double *nuc_grid, double awr,
                         double kT, double cutoff, double thresh, int cap_el, double *ein_el,
                         int *n_el, int cap_inel, double *ein_inel, int *n_inel);

/* ---- whole nuclide (SURVEY 8a row H1, 8b "B-batch convenience") ----------------
 * The parts of the ACE data model calc_scatt reads (ace_header.F90:14-164), as
 * pointers into the host's own arrays.                                           */
typedef struct ndpp_ace_edist {     /* one DistEnergy of the rxn%edist -> %next chain */
  int law, n_data;
  const double *data;               /* edist%data                                     */
  int pv_n_regions, pv_n_pairs;     /* edist%p_valid (Tab1)                           */
  const int *pv_nbt, *pv_int;
  const double *pv_x, *pv_y;
} ndpp_ace_edist;

typedef struct ndpp_ace_rxn {
  int MT;
  double Q_value;
  int multiplicity;                 /* rxn%multiplicity                               */
  int threshold;                    /* rxn%threshold: 1-based index into nuc%energy   */
  int scatter_in_cm;
  int n_sigma;
  const double *sigma;              /* rxn%sigma (from the threshold index on)        */
  int has_mult_E;                   /* rxn%multiplicity_with_E; then multiplicity_E:  */
  int mE_n_regions, mE_n_pairs;
  const int *mE_nbt, *mE_int;
  const double *mE_x, *mE_y;
  int has_angle_dist, n_adist;      /* rxn%adist, as in ndpp_ace_reaction             */
  const double *adist_energy;
  const int *adist_type, *adist_location;
  int n_adist_data;
  const double *adist_data;
  int n_edist;                      /* length of the edist chain (0: none)            */
  const ndpp_ace_edist *edist;
} ndpp_ace_rxn;

typedef struct ndpp_ace_nuclide {
  double awr, kT, freegas_cutoff;   /* nuc%awr, %kT, %freegas_cutoff (MeV)            */
  int n_grid;
  const double *energy, *elastic;   /* nuc%energy, nuc%elastic [n_grid]               */
  int n_reaction;
  const ndpp_ace_rxn *reactions;
} ndpp_ace_nuclide;

typedef struct ndpp_scatt_result {  /* allocated by the library                       */
  int n_el, n_inel, L, G;
  double *ein_el, *ein_inel;        /* Ein_el(n_el), Ein_inel(n_inel)                 */
  double *el_mat, *inel_mat;        /* (L, G, n_el), (L, G, n_inel) Fortran order     */
  double *nuinel_mat;               /* NULL unless nuscatt                            */
} ndpp_scatt_result;

/* Replaces `calc_scatt(nuc, energy_bins, scatt_type=Legendre, order, mu_bins, nuscatt,
 * Ein_el, Ein_inel, el_mat, inel_mat, nuinel_mat)` scatt.F90:33-157 (order and mu_bins
 * come in p): ScattData%init + convert_distro for every reaction and nested
 * distribution, create_Ein_grid, calc_elastic_grid, calc_inelastic_grid.  The result
 * arrays are owned by the library until ndpp_free_scatt_result.                    */
int  ndpp_scatt_nuclide(const ndpp_params *p, const ndpp_ace_nuclide *nuc, int n_bins,
                        const double *e_bins, int nuscatt, ndpp_scatt_result *out);
void ndpp_free_scatt_result(ndpp_scatt_result *r);

/* calc_scatt for n_nuclides nuclides (one group structure): identical results to
 * n_nuclides ndpp_scatt_nuclide calls, but the elastic grids of all nuclides -- where
 * > 99 % of the time goes -- are integrated by ONE ndpp_elastic_leg_multi call.
 * out[n_nuclides]; on error every result is freed.                                  */
int  ndpp_scatt_library(const ndpp_params *p, int n_nuclides, const ndpp_ace_nuclide *nuclides,
                        int n_bins, const double *e_bins, int nuscatt, ndpp_scatt_result *out);

/* ---- wire format, host only ---------------------------------------
 * The byte stream the reference's BINARY writers produce (stream access: raw
 * little-endian int32 / float64).  Each function returns the number of bytes of the
 * section (or -1) and writes it when cap suffices, so a host can size, then fill.   */
/* group indices of the energy-bin edges in an incoming grid, ndpp.F90:648-679 (1-based) */
int  ndpp_group_index(int n_bins, const double *e_bins, int n_ein, const double *ein, int *index);
/* print_scatt_bin scatt.F90:1139-1258: NE, Ein, group indices, then per E_in gmin, gmax and
 * the moments of groups gmin..gmax (range found on P0 > 0); elastic, inelastic, nu-inelastic */
long ndpp_scatt_wire(const ndpp_scatt_result *r, int n_bins, const double *e_bins, long cap,
                     unsigned char *buf);
/* print_chi_bin chi.F90:319-353; arrays as ndpp_chi_batch returns them */
long ndpp_chi_wire(int G, int n_ein, int n_prec, const double *e_grid, const double *chi_t,
                   const double *chi_p, const double *chi_d, long cap, unsigned char *buf);
/* file header ndpp.F90:1314-1329 */
long ndpp_header_wire(const char *name, int name_len, double kT, int G, const double *e_bins,
                      int scatt_type, int scatt_order, int nuscatter, int chi_present,
                      int mu_bins, double thin_tol, long cap, unsigned char *buf);

/* ---- formatted outputs, host only ----------------------------------
 * The ASCII library format and ndpp_lib.xml, character for character as the reference's
 * formatted writes produce them (lines end with '\n'); same size-then-fill convention.   */
enum { NDPP_FMT_ASCII = 1, NDPP_FMT_BINARY = 2, NDPP_FMT_HDF5 = 3, NDPP_FMT_NONE = 4,
       NDPP_FMT_HUMAN = 5 };            /* constants.F90:56-58,193-194 */
/* to_str(real(8)) string.F90:408-455 (6 significant digits); out16 holds <= 15 chars + NUL;
 * returns the length */
int  ndpp_real_to_str(double x, char *out16);
/* print_ascii_array output.F90:221-253: four 1PE20.12 fields per line, lines trimmed */
long ndpp_ascii_array(int n, const double *a, long cap, char *buf);
/* print_scatt_ascii scatt.F90:881-997, print_chi_ascii chi.F90:203-244, ASCII header
 * ndpp.F90:1283-1304: the text counterparts of ndpp_scatt_wire / _chi_wire / _header_wire */
long ndpp_scatt_ascii(const ndpp_scatt_result *r, int n_bins, const double *e_bins, long cap,
                      char *buf);
long ndpp_chi_ascii(int G, int n_ein, int n_prec, const double *e_grid, const double *chi_t,
                    const double *chi_p, const double *chi_d, long cap, char *buf);
long ndpp_header_ascii(const char *name, int name_len, double kT, int G, const double *e_bins,
                       int scatt_type, int scatt_order, int nuscatter, int chi_present,
                       int mu_bins, double thin_tol, long cap, char *buf);
/* ndpp_lib.xml: print_ndpp_lib_xml_header / _nuclide / _closer ndpp.F90:958-1110
 * (directory = the reference's path_input; strings are trimmed like trim()) */
long ndpp_lib_xml_header(const char *directory, int lib_format, int n_listings, int nuscatter,
                         int chi_present, int scatt_type, int scatt_order, double print_tol,
                         double thin_tol, int mu_bins, int n_bins, const double *e_bins,
                         long cap, char *buf);
long ndpp_lib_xml_nuclide(const char *alias, double awr, const char *name, const char *path,
                          double kT, int zaid, int metastable, double freegas_cutoff,
                          int lib_format, long cap, char *buf);
long ndpp_lib_xml_closer(int lib_format, long cap, char *buf);

/* ---- one table from matrices to file: ndpp.F90:594-717 / :755-813 -------------------- */
typedef struct ndpp_output_options {   /* the run-level settings of nuclearDataPreProc */
  int lib_format;                      /* NDPP_FMT_ASCII or NDPP_FMT_BINARY             */
  int scatt_type, scatt_order;         /* as written to the header (order, not order+1) */
  int nuscatter, integrate_chi, mu_bins;
  double print_tol, thin_tol;
} ndpp_output_options;
/* apply_tol_scatt on every matrix, then (thin_tol > 0) thin_grid on the elastic grid and on
 * the inelastic grid with nu-inelastic riding along (ndpp.F90:611-646); in place, r->n_el /
 * r->n_inel shrink.  thin_report (NULL or 4 doubles): compression and max error, elastic
 * then inelastic.                                                                         */
int  ndpp_finish_scatt(const ndpp_output_options *o, ndpp_scatt_result *r, int n_bins,
                       const double *e_bins, double *thin_report);
/* The table's library file: header (init_library ndpp.F90:1246), scatter section
 * (print_scatt, group indices ndpp.F90:648-679), chi section when integrate_chi and
 * n_chi > 0 (print_chi); is_sab: thermal table (nuscatter flag written as 0).            */
long ndpp_nuclide_file(const ndpp_output_options *o, const char *name, int name_len, double kT,
                       int is_sab, const ndpp_scatt_result *r, int n_bins, const double *e_bins,
                       int n_chi, int n_prec, const double *e_chi, const double *chi_t,
                       const double *chi_p, const double *chi_d, long cap, unsigned char *buf);

/* ---- thinning: replaces `thin_grid(xout, yout, tokeep, tol, compression, maxerr[, yout2
 * [, yout3]])` thin.F90:17-501, host only, in place: x[n] and the matrices y[n][G][L]
 * (y2 of the same shape and the 1-D y3[n] optional, NULL when absent) keep their first
 * *n_out entries.  tokeep[n_keep]: abscissae never removed (the group edges).              */
int ndpp_thin_grid(int n, double *x, int L, int G, double *y, double *y2, double *y3, int n_keep,
                   const double *tokeep, double tol, int *n_out, double *compression,
                   double *maxerr);

/* Replaces `sab_egrid(sab, energy_bins, Ein)` sab.F90:460-568 (host only; the caller then
 * appends the extra top point, scatt.F90:426) and the union grid of calc_chi, chi.F90:97-113.
 * Counts are always returned, arrays written when cap suffices.                             */
int ndpp_sab_egrid(const ndpp_params *p, const ndpp_sab_flat *t, int n_bins, const double *e_bins,
                   int cap, double *ein, int *n_out);
int ndpp_chi_egrid(int n_prompt, const ndpp_chi_spectrum *prompt, int n_delay,
                   const ndpp_chi_spectrum *delay, int cap, double *e_grid, int *n_out);

/* ---- epilogue: replaces `apply_tol_scatt(data, tol)` scatt.F90:786-818, in place
 * on data[n][G][L]: groups whose P0 lies in (0, tol) are zeroed and every row is
 * renormalised to its original sum_g P0.  Bit-identical to the Fortran.        */
int ndpp_apply_tol_scatt(int L, int G, int n, double *data, double tol);

#ifdef __cplusplus
}
#endif
#endif /* NDPP_HIP_H */
