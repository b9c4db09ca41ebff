/* ddamg_hip_io.h -- the reference's on-disk formats for gauge configurations and spinor / test-vector files, read and
 * written directly by every process for its own part of the lattice (host code only: no GPU, no MPI).
 *
 * Replaces, for the data either side of the hot path:
 *   read_conf                       src/io.c:459-563   one file, rank 0 reads and scatters rows of X
 *   vector_io (_READ / _WRITE)      src/io.c:704-846   one spinor, optional text header, lexicographic sites
 *   vector_io_single_file           src/io.c:951-1124  n spinors behind one header (test vectors: setup persistence)
 *   write_header                    src/io.c:671-702
 *   read_conf_multi                 src/io.c:566-668   one file per process of the grid, <base>.pt<T>pz<Z>py<Y>px<X>
 * Not covered: the LIME/HDF5 builds.
 *
 * Layouts (all little endian unless `big_endian` is set, the reference's -DBIG_ENDIAN_CNFG / -DBIG_ENDIAN_TV builds):
 *   configuration: int32 T,Z,Y,X ; double plaquette ; then for t,z,y,x (x fastest): 4 directions (T,Z,Y,X) x 3x3 complex
 *                  doubles, row major = 72 doubles per site.  The anti-periodic sign of the last time slice is NOT applied
 *                  here (ddamg_hip_set_gauge takes it as an argument).
 *   vectors:       optional "<header>\n ... </header>\n" text block, then per vector, for t,z,y,x: 12 complex doubles.
 * A process with coordinates `process_coords` on the grid `process_grid` (T,Z,Y,X; rank order as in ddamg_hip.h) owns the
 * sub-lattice global/process_grid at that position; its part is returned / taken in local lexicographic order. */
#ifndef DDAMG_HIP_IO_H
#define DDAMG_HIP_IO_H
#ifdef __cplusplus
extern "C" {
#endif

/* text of the last failure of a call below (static buffer, per thread) */
const char* ddamg_hip_io_last_error(void);

/* lattice extents and plaquette stored in a configuration file.  Returns 0, or -1 on failure. */
int ddamg_hip_conf_info(const char* path, int big_endian, int lattice_out[4], double* plaq_out);

/* gauge_local: [V_local][4][9][2] doubles.  The file's extents must equal global_lattice (ASSERT in src/io.c:497-498). */
int ddamg_hip_read_conf(const char* path, const int global_lattice[4], const int process_grid[4], const int process_coords[4],
                        int big_endian, double* gauge_local, double* plaq_out);
/* the same layout written out (single file; every process writes its own rows, the process at the origin the header) */
int ddamg_hip_write_conf(const char* path, const int global_lattice[4], const int process_grid[4], const int process_coords[4],
                         int big_endian, const double* gauge_local, double plaq);

/* read_conf_multi (src/io.c:566-668): the part of the process at process_coords from the file <base>.pt<T>pz<Z>py<Y>px<X>, which
 * carries the header of the global lattice and that process's links in local lexicographic order; the writer is its inverse
 * (every process writes its own file). */
int ddamg_hip_read_conf_multi(const char* base, const int global_lattice[4], const int process_grid[4], const int process_coords[4],
                              int big_endian, double* gauge_local, double* plaq_out);
int ddamg_hip_write_conf_multi(const char* base, const int global_lattice[4], const int process_grid[4], const int process_coords[4],
                               int big_endian, const double* gauge_local, double plaq);

/* fields of write_header; strings may be NULL (written as empty) */
typedef struct ddamg_hip_vector_header {
  const char* vector_type;        /* second line of the header: "test vectors", or the file name for single spinors */
  double m0, csw, clov_plaq, hopp_plaq;
  const char* clov_conf_name;
  const char* hopp_conf_name;
  int has_eigenvalues;            /* write the "eigenvalues:" line from `eigenvalues` (2 n doubles) */
  const double* eigenvalues;
} ddamg_hip_vector_header;

/* vectors_local: [n][V_local][12][2] doubles.  A file without header is accepted when n == 1 (src/io.c:735-743). */
int ddamg_hip_read_vectors(const char* path, const int global_lattice[4], const int process_grid[4], const int process_coords[4],
                           int n, int big_endian, double* vectors_local);
/* header == NULL writes the bare data (readable as a single spinor only) */
int ddamg_hip_write_vectors(const char* path, const int global_lattice[4], const int process_grid[4], const int process_coords[4],
                            int n, int big_endian, const ddamg_hip_vector_header* header, const double* vectors_local);

#ifdef __cplusplus
}
#endif
#endif
