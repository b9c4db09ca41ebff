// gauge.h -- see gauge.cpp
#pragma once
#include <hip/hip_runtime.h>
namespace ddamg {
// gauge_in: [V][4][9] complex fp64 lexicographic (T,Z,Y,X; X fastest).  Writes D_out [V][36] complex
// (= U/2, after the optional anti-periodic sign) and clover_out [V][42] complex in the reference's
// storage; returns the average plaquette in [0,3].
double gauge_to_operator(const int L[4], const double* gauge_in, int anti_pbc, double m0, double csw,
                         double* D_out, double* clover_out);
// the same computed on the device (single process): links up, D / clover / plaquette down; gauge_device.hip
double gauge_to_operator_device(const int L[4], const double* gauge_in, int anti_pbc, double m0, double csw, double* D_out, double* clover_out,
                                hipStream_t st);
// clover term of the own sites and the sum of their plaquette traces from the links of the lattice extended by `halo`
// sites in every direction (U_ext_host: [prod(L+2 halo)][4][9] complex, lexicographic in the extended lattice), on the device
double clover_and_plaquette_extended_device(const int L[4], const int halo[4], const double* U_ext_host, double m0, double csw, double* clover_out, hipStream_t st);
// the same on a process grid: gauge_in is the process's own part; the links of the neighbouring processes that the
// clover leaves reach (one site deep, corners included) are fetched first (the reference exchanges the ghost shell of
// the gauge field in dirac_setup, src/dirac.c:88-120); the plaquette is the global average
struct Geometry;
struct Comm;
double gauge_to_operator_dist(const Geometry& g, Comm* comm, const double* gauge_in, int anti_pbc, double m0, double csw,
                              double* D_out, double* clover_out, hipStream_t st);
}
