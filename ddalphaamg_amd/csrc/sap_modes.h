// sap_modes.h -- residual-update modes of the Schwarz block-solve kernels (sap.hip, sap_pair.hip)
#pragma once
namespace ddamg {
// MODE_NONE: r is up to date; MODE_NBOUNDARY: r_b -= D_{b,ext} delta_ext (n_block_PRECISION_boundary_op);
// MODE_FULLRES: r_b = eta_b - (D x)_b with the whole operator (block_op + boundary_op, src/schwarz_generic.c:1296-1312)
enum { MODE_NONE = 0, MODE_NBOUNDARY = 1, MODE_FULLRES = 2 };
}
