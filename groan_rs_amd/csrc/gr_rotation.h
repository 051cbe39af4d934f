// gr_rotation.h -- 3x3 rotation algebra of the Kabsch step (fp64), usable from device code and from the host tests.
//
// Reference: rotation = U diag(1, 1, sign det(U V^T)) V^T of the SVD H = U S V^T (src/system/rmsd.rs:573-583).
//   gr_kabsch_rotation  -- general path: eigenvectors of H^T H by cyclic Jacobi (handles reflections and rank loss)
//   gr_polar_rotation   -- fast path for det H > 0 and a not-too-flat H: then the answer is the orthogonal polar factor
//                          U V^T of H, which the scaled Newton iteration X <- (g X + X^-T / g) / 2 reaches in 5-7 steps
//                          of plain 3x3 arithmetic (no eigen-decomposition): ~6x shorter on the GPU, where one lane
//                          closes a frame (on the tail of the sums kernel, or in a finalizer workgroup of the resident pass, where its latency
//                          is what the parked frames have to cover)
//   gr_best_rotation    -- polar when it applies and converges, otherwise Jacobi
#pragma once
#include <math.h>
#if defined(__HIPCC__)
#define GR_ROT_HD __host__ __device__
#else
#define GR_ROT_HD
#endif

GR_ROT_HD inline void gr_jacobi_eig3(double A[3][3], double V[3][3]) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 32; ++sweep) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        const double dg = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (off <= 1e-60 || off <= 1e-32 * dg) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (A[p][q] == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq; }
                for (int k = 0; k < 3; ++k) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk; }
                for (int k = 0; k < 3; ++k) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq; }
            }
    }
}

// R = U diag(1,1,sign det(U V^T)) V^T of H = U S V^T (rmsd.rs:573-583), from the eigenvectors of H^T H:
// u_k = H v_k / |H v_k| (k = 1,2), u_3' = u_1 x u_2, R = u_1 v_1^T + u_2 v_2^T + det(V) u_3' v_3^T.
GR_ROT_HD inline void gr_kabsch_rotation(const double H[3][3], double R[3][3]) {
    double HtH[3][3], V[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) HtH[i][j] = H[0][i] * H[0][j] + H[1][i] * H[1][j] + H[2][i] * H[2][j];
    gr_jacobi_eig3(HtH, V);
    const double w[3] = { HtH[0][0], HtH[1][1], HtH[2][2] };
    int o0 = 0, o1 = 1, o2 = 2;
    if (w[o1] > w[o0]) { int t = o0; o0 = o1; o1 = t; }
    if (w[o2] > w[o0]) { int t = o0; o0 = o2; o2 = t; }
    if (w[o2] > w[o1]) { int t = o1; o1 = o2; o2 = t; }
    double v[3][3];
    for (int i = 0; i < 3; ++i) { v[0][i] = V[i][o0]; v[1][i] = V[i][o1]; v[2][i] = V[i][o2]; }
    double u[3][3];
    for (int k = 0; k < 2; ++k)
        for (int i = 0; i < 3; ++i) u[k][i] = H[i][0] * v[k][0] + H[i][1] * v[k][1] + H[i][2] * v[k][2];
    double n0 = sqrt(u[0][0] * u[0][0] + u[0][1] * u[0][1] + u[0][2] * u[0][2]);
    if (!(n0 >= 1e-300)) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i][j] = (i == j) ? 1.0 : 0.0; return; }
    for (int i = 0; i < 3; ++i) u[0][i] /= n0;
    const double d01 = u[1][0] * u[0][0] + u[1][1] * u[0][1] + u[1][2] * u[0][2];
    for (int i = 0; i < 3; ++i) u[1][i] -= d01 * u[0][i];
    double n1 = sqrt(u[1][0] * u[1][0] + u[1][1] * u[1][1] + u[1][2] * u[1][2]);
    if (n1 < 1e-12 * n0) {
        const double a0 = fabs(u[0][0]), a1 = fabs(u[0][1]), a2 = fabs(u[0][2]);
        const int m = a0 < a1 ? (a0 < a2 ? 0 : 2) : (a1 < a2 ? 1 : 2);
        double e[3] = { 0, 0, 0 }; e[m] = 1.0;
        const double d = e[0] * u[0][0] + e[1] * u[0][1] + e[2] * u[0][2];
        for (int i = 0; i < 3; ++i) u[1][i] = e[i] - d * u[0][i];
        n1 = sqrt(u[1][0] * u[1][0] + u[1][1] * u[1][1] + u[1][2] * u[1][2]);
    }
    for (int i = 0; i < 3; ++i) u[1][i] /= n1;
    u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
    u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
    u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
    const double detV = v[0][0] * (v[1][1] * v[2][2] - v[1][2] * v[2][1]) - v[0][1] * (v[1][0] * v[2][2] - v[1][2] * v[2][0]) +
                        v[0][2] * (v[1][0] * v[2][1] - v[1][1] * v[2][0]);
    const double sg = detV < 0 ? -1.0 : 1.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[i][j] = u[0][i] * v[0][j] + u[1][i] * v[1][j] + sg * u[2][i] * v[2][j];
}

// Orthogonal polar factor of H by Newton's iteration with Frobenius scaling (Higham).  Returns false -- R untouched --
// when H is (close to) a reflection or flat (det of the normalised H <= 1e-5: sigma_3 may be ~0 or negative, where
// U V^T is not the Kabsch answer or the iteration is ill-conditioned) or when 12 steps do not converge.
GR_ROT_HD inline bool gr_polar_rotation(const double H[3][3], double R[3][3]) {
    double n2 = 0.0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) n2 += H[i][j] * H[i][j];
    if (!(n2 > 1e-280) || !(n2 < 1e280)) return false;
    const double inv = 1.0 / sqrt(n2);
    double X[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) X[i][j] = H[i][j] * inv;
    for (int it = 0; it < 12; ++it) {
        double C[3][3];   // cofactors: X^-T = C / det
        C[0][0] = X[1][1] * X[2][2] - X[1][2] * X[2][1]; C[0][1] = X[1][2] * X[2][0] - X[1][0] * X[2][2]; C[0][2] = X[1][0] * X[2][1] - X[1][1] * X[2][0];
        C[1][0] = X[2][1] * X[0][2] - X[2][2] * X[0][1]; C[1][1] = X[2][2] * X[0][0] - X[2][0] * X[0][2]; C[1][2] = X[2][0] * X[0][1] - X[2][1] * X[0][0];
        C[2][0] = X[0][1] * X[1][2] - X[0][2] * X[1][1]; C[2][1] = X[0][2] * X[1][0] - X[0][0] * X[1][2]; C[2][2] = X[0][0] * X[1][1] - X[0][1] * X[1][0];
        const double det = X[0][0] * C[0][0] + X[0][1] * C[0][1] + X[0][2] * C[0][2];
        if (it == 0 && !(det > 1e-5)) return false;
        if (!(det > 0.0)) return false;
        double nx = 0.0, nc = 0.0;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { nx += X[i][j] * X[i][j]; nc += C[i][j] * C[i][j]; }
        // g^2 = |X^-1|_F / |X|_F; any positive g leaves the fixed point alone, so single precision is plenty
        const float g = sqrtf(sqrtf((float)(nc / nx)) / (float)det);
        const double a = 0.5 * (double)g, b = 0.5 / ((double)g * det);
        double delta = 0.0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                const double xn = a * X[i][j] + b * C[i][j];
                const double d = xn - X[i][j];
                delta += d * d;
                X[i][j] = xn;
            }
        if (delta <= 1e-13) {
            // |dX|_F <= 3e-7 with quadratic convergence: one unscaled step lands on the fixed point to ~1e-13
            C[0][0] = X[1][1] * X[2][2] - X[1][2] * X[2][1]; C[0][1] = X[1][2] * X[2][0] - X[1][0] * X[2][2]; C[0][2] = X[1][0] * X[2][1] - X[1][1] * X[2][0];
            C[1][0] = X[2][1] * X[0][2] - X[2][2] * X[0][1]; C[1][1] = X[2][2] * X[0][0] - X[2][0] * X[0][2]; C[1][2] = X[2][0] * X[0][1] - X[2][1] * X[0][0];
            C[2][0] = X[0][1] * X[1][2] - X[0][2] * X[1][1]; C[2][1] = X[0][2] * X[1][0] - X[0][0] * X[1][2]; C[2][2] = X[0][0] * X[1][1] - X[0][1] * X[1][0];
            const double d2 = X[0][0] * C[0][0] + X[0][1] * C[0][1] + X[0][2] * C[0][2];
            const double h = 0.5 / d2;
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i][j] = 0.5 * X[i][j] + h * C[i][j];
            return true;
        }
    }
    return false;
}

GR_ROT_HD inline void gr_best_rotation(const double H[3][3], double R[3][3]) {
    if (!gr_polar_rotation(H, R)) gr_kabsch_rotation(H, R);
}
