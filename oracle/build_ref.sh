#!/usr/bin/env bash
# Build the REAL reference (danshapero/sigma, Fortran 2003) from the sources
# where they lie under /root/reference, with amdflang.  TEST INFRASTRUCTURE
# ONLY: outputs go to oracle/_ref/ (git-ignored); no reference source is copied.
#
# Only the modules on / below the hot path are compiled, in the order of
# /root/reference/src/CMakeLists.txt:1-40.  Not compiled: wrapper.f90 (dead code), sigma.f90
# (umbrella).  eigensolver.f90 IS compiled for its lanczos / generalized_lanczos routines
# (eigensolver.f90:27-155); its eigensolve routines call LAPACK dstev, which stays unresolved
# exactly like dgetrf below (never called).
# util.f90 holds one off-path routine (`determinant`, util.f90:59) that calls
# LAPACK dgetrf; the image has no LAPACK, no stand-in is written, and the
# symbol is simply left unresolved in the shared object (never called on the
# matvec / solver path; lazy binding).
set -euo pipefail
REF=${SIGMA_REFERENCE:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
FC=${FC:-/opt/rocm/bin/amdflang}
[ -d "$REF/src" ] || { echo "reference sources not present at $REF - skipping"; exit 0; }
command -v "$FC" >/dev/null || { echo "no Fortran compiler ($FC) - skipping"; exit 0; }
mkdir -p "$OUT/obj"
SRCS="types.f90 util.f90 vectors.f90
 linear_operator/linear_operator_interface.f90 linear_operator/linear_operator_sums.f90
 linear_operator/linear_operator_products.f90 linear_operator/linear_operator_adjoints.f90
 linear_operator/linear_operators.f90
 graph/graph_interfaces.f90 graph/formats/coo_graphs.f90 graph/formats/cs_graphs.f90
 graph/formats/ellpack_graphs.f90 graph/formats/ll_graphs.f90 graph/graph_factory.f90
 graph/permutations.f90 graph/graphs.f90
 matrix/sparse_matrix_interfaces.f90 matrix/formats/default_sparse_matrix_kernels.f90
 matrix/formats/default_matrices.f90 matrix/formats/cs_matrices.f90
 matrix/formats/ellpack_matrices.f90 matrix/sparse_matrix_factory.f90
 matrix/sparse_matrix_composites.f90 matrix/sparse_matrix_algebra.f90
 matrix/sparse_matrices.f90 eigensolver.f90
 solver/bicgstab_solvers.f90 solver/cg_solvers.f90 solver/jacobi_solvers.f90
 solver/ldu_solvers.f90"
cd "$OUT/obj"
OBJS=""
for s in $SRCS; do
  o="$(basename "${s%.f90}").o"
  if [ ! -f "$o" ] || [ "$REF/src/$s" -nt "$o" ]; then
    "$FC" -O2 -fPIC -c "$REF/src/$s" -o "$o"
  fi
  OBJS="$OBJS $o"
done
# the driver is OUR code (oracle/ref_driver.f90); it only `use`s reference modules
"$FC" -O2 -fPIC -c "$HERE/ref_driver.f90" -o ref_driver.o
"$FC" -O2 -o "$OUT/sigma_ref_driver" ref_driver.o $OBJS \
    -Wl,-z,execstack -Wl,--unresolved-symbols=ignore-all
echo "built $OUT/sigma_ref_driver"

# the reference-side binding (oracle/hip_binding.f90: types that EXTEND the reference's csr_matrix /
# ellpack_matrix / linear_solver) compiled against the reference's .mod files and linked with the
# product library; the test program re-runs the reference's two deterministic solver tests through it
LIBDIR="$HERE/../sigma_amd"
if [ -f "$LIBDIR/libsigma_hip.so" ]; then
  "$FC" -O2 -fPIC -c "$HERE/hip_binding.f90" -o hip_binding.o
  "$FC" -O2 -fPIC -c "$HERE/hip_binding_test.f90" -o hip_binding_test.o
  "$FC" -O2 -o "$OUT/hip_binding_test" hip_binding_test.o hip_binding.o $OBJS \
      -L"$LIBDIR" -lsigma_hip -Wl,-rpath,'$ORIGIN/../../sigma_amd' \
      -Wl,-z,execstack -Wl,--unresolved-symbols=ignore-all
  echo "built $OUT/hip_binding_test"
  # one rank of a row-partitioned solve through the same binding (hip_comm + hip_dist_csr_matrix); started once per rank
  # by tests/test_gpu_multirank.py
  "$FC" -O2 -fPIC -c "$HERE/hip_dist_test.f90" -o hip_dist_test.o
  "$FC" -O2 -o "$OUT/hip_dist_test" hip_dist_test.o hip_binding.o $OBJS \
      -L"$LIBDIR" -lsigma_hip -Wl,-rpath,'$ORIGIN/../../sigma_amd' \
      -Wl,-z,execstack -Wl,--unresolved-symbols=ignore-all
  echo "built $OUT/hip_dist_test"
else
  echo "libsigma_hip.so not built yet - skipping hip_binding_test"
fi
