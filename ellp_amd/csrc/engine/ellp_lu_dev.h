// ellp_lu_dev.h — internal: LU with partial pivoting of a device-resident square matrix stored by rows (ellp_lu.hip), used by
// the certificate of the certified hybrid above 1,024 rows (ellp_exact.inc).  Not part of the C ABI.
#ifndef ELLP_LU_DEV_H
#define ELLP_LU_DEV_H
#include <hip/hip_runtime.h>

#include <cstdint>

struct EllpLuWork {
    double *M;      // m x m by rows: in: the matrix; out: L (multipliers, below the diagonal) and U, rows in pivoted order
    double *prow, *irow, *udiag;
    int64_t *piv;   // piv[i] = the row exchanged with row i at step i (i itself: none, also for a skipped zero column)
    void *cands, *st;
    int64_t m;
};
hipError_t ellp_lu_rows_alloc(EllpLuWork *w, int64_t m);
void ellp_lu_rows_free(EllpLuWork *w);
void ellp_lu_rows_factor(EllpLuWork *w, hipStream_t stream);  // enqueues 2 m launches; no synchronisation
#endif
