/*
 * ddamg_oracle.h -- TEST INFRASTRUCTURE ONLY.  Never linked into or called by the product
 * (ddalphaamg_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Plain-C CPU restatement of the DDalphaAMG V-cycle hot path (reference mrottmann/DDalphaAMG,
 * citations as file:line relative to the reference tree at each function).  All data are in
 * the reference's own outer storage: lexicographic sites (T,Z,Y,X; X fastest,
 * src/data_layout.h:30-32), interleaved (re,im); D = [V][4][9] complex (U/2, src/dirac.c:80),
 * clover = [V][42] complex (src/dirac.c:386-398), fine vectors [V][12] complex.
 *
 * Parity pinning: checked against golden vectors dumped from the real reference
 * (oracle/_ref, oracle/ref_dump.c -> tests/golden/*.npz) by tests/test_oracle_golden.py.
 */
#ifndef DDAMG_ORACLE_H
#define DDAMG_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

/* dirac_setup (src/dirac.c:60-168): returns average plaquette in [0,3] */
double orc_gauge_to_operator(const int L[4], const double *gauge, int anti_pbc, double m0, double csw,
                             double *D, double *clover);

/* d_plus_clover_double / d_plus_clover_float (src/dirac_generic.c:159-277).
 * The f32 variant rounds D, clover and phi to float first (as schwarz_PRECISION_setup,
 * src/schwarz_generic.c:1037-1074, and trans_float do) and computes in float. */
void orc_dirac_apply_f64(const int L[4], const double *D, const double *clover, const double *phi, double *eta);
void orc_dirac_apply_f32(const int L[4], const double *D, const double *clover, const double *phi, double *eta);

/* timing helper for bench.py's cpu_baseline: `reps` fp32 applies on pre-converted float data,
 * OpenMP over sites; returns seconds per apply and the number of threads used */
double orc_dirac_time_f32(const int L[4], const double *D, const double *clover, const double *phi, int reps, int *threads);

#ifdef __cplusplus
}
#endif
void orc_set_threads(int n);

#endif
