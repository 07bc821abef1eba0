// Shared between svd.hip (row-pair Jacobi, driver, epilogue) and svd_block.hip (block Jacobi sweeps).
#pragma once
#include "nd4hip_internal.h"

struct JacState {            // per-matrix device state
  unsigned rotations;        // rotations applied in the current sweep
  unsigned done;             // converged in an earlier sweep
};

// round-robin tournament: n2 players (even), step s in [0, n2-1), slot i in [0, n2/2) -> p < q
__host__ __device__ __forceinline__ void nd4_rr_pair(int n2, int s, int i, int& p, int& q) {
  const int m = n2 - 1;
  if (i == 0) { p = m; q = s; }
  else { p = (s + i) % m; q = (s - i + m) % m; }
  if (p > q) { const int t = p; p = q; q = t; }
}

// One sweep of block Jacobi over all block pairs (svd_block.hip). Requires N % 64 == 0.
// W, Ut: [batch, N, N]; st/floor2: per matrix; offmax: max cos^2 seen (bits of a double);
// scratch must hold nd4_jacobi_block_scratch_doubles(batch, N) doubles.
size_t nd4_jacobi_block_scratch_doubles(int batch, int N);
int nd4_jacobi_block_sweep(nd4hip_handle* h, int batch, int N, double* W, double* Ut, JacState* st,
                           const double* floor2, double tol2, unsigned long long* offmax, double* scratch, bool dense_phase);
// dense_phase: the previous sweep rotated most pairs (the driver knows from the rotation count): skip the per-visit pre-check
