/* ORACLE (test infrastructure, see orc.h) — small dense linear algebra helpers. */
#ifndef ORC_LINALG_H
#define ORC_LINALG_H
#define ORC_LA_MAX 12
void orc_jacobi_svd(double* At, int m, int n, double* W, double* Vt, int n1);
void orc_jacobi_svd_ord(double* At, int m, int n, double* W, double* Vt, int n1, int ordering);
void orc_svd(const double* A, int m, int n, double* W, double* Ut, double* Vt);
void orc_svd_solve(const double* A, int m, int n, const double* b, double* x);
void orc_inv3_svd(const double A[9], double Ainv[9]);
int  orc_qr_solve(double* A, int m, int n, double* b, double* x);
#endif
