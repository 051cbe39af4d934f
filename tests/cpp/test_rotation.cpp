// Host test of groan_rs_amd/csrc/gr_rotation.h: the Newton polar fast path must agree with the Jacobi/Kabsch general
// path (the restatement of src/system/rmsd.rs:573-583) wherever it accepts the matrix, over planar, linear, noisy,
// mirrored and badly scaled covariance matrices.  Exit code 0 = pass.
#include "../../groan_rs_amd/csrc/gr_rotation.h"
#include <cstdio>
#include <random>
#include <algorithm>
int main() {
    std::mt19937_64 rng(7);
    std::normal_distribution<double> N(0, 1);
    double worst = 0, worst_orth = 0; long fast = 0, total = 0;
    for (int trial = 0; trial < 200000; ++trial) {
        int n = 4 + (int)(rng() % 200);
        double noise = std::pow(10.0, -4 + 5.0 * (rng() % 1000) / 1000.0);   // 1e-4 .. 10
        double flat[3] = {1, 1, 1};
        if (trial % 7 == 0) flat[2] = 1e-3 * (rng() % 10);   // near-planar groups
        if (trial % 31 == 0) { flat[1] = 1e-4; flat[2] = 1e-4; }   // near-linear
        // random rotation via QR-ish (Gram-Schmidt)
        double Q[3][3];
        for (auto &r : Q) for (auto &v : r) v = N(rng);
        auto norm = [](double *v) { double s = std::sqrt(v[0]*v[0]+v[1]*v[1]+v[2]*v[2]); for (int i=0;i<3;++i) v[i]/=s; };
        norm(Q[0]);
        double d = Q[1][0]*Q[0][0]+Q[1][1]*Q[0][1]+Q[1][2]*Q[0][2]; for (int i=0;i<3;++i) Q[1][i]-=d*Q[0][i]; norm(Q[1]);
        Q[2][0]=Q[0][1]*Q[1][2]-Q[0][2]*Q[1][1]; Q[2][1]=Q[0][2]*Q[1][0]-Q[0][0]*Q[1][2]; Q[2][2]=Q[0][0]*Q[1][1]-Q[0][1]*Q[1][0];
        if (trial % 13 == 0) for (int i=0;i<3;++i) Q[2][i] = -Q[2][i];   // improper: reflection case
        double H[3][3] = {};
        for (int k = 0; k < n; ++k) {
            double p[3] = { N(rng)*flat[0], N(rng)*flat[1], N(rng)*flat[2] }, q[3];
            for (int i=0;i<3;++i) q[i] = Q[i][0]*p[0]+Q[i][1]*p[1]+Q[i][2]*p[2] + noise*N(rng);
            for (int a=0;a<3;++a) for (int c=0;c<3;++c) H[a][c] += p[a]*q[c];
        }
        double scale = std::pow(10.0, (double)(rng()%9) - 2);
        for (auto &r : H) for (auto &v : r) v *= scale;
        double Rj[3][3], Rp[3][3];
        gr_kabsch_rotation(H, Rj);
        ++total;
        if (gr_polar_rotation(H, Rp)) {
            ++fast;
            double m = 0;
            for (int i=0;i<3;++i) for (int j=0;j<3;++j) m = std::max(m, std::fabs(Rj[i][j]-Rp[i][j]));
            // objective gap: tr(R^T H) must agree (near-degenerate sigma_2 ~ sigma_3 ~ 0 allows different R with equal objective)
            double tj=0,tp=0,nh=0; for (int i=0;i<3;++i) for (int j=0;j<3;++j) { tj+=Rj[i][j]*H[i][j]; tp+=Rp[i][j]*H[i][j]; nh+=H[i][j]*H[i][j]; }
            double og = std::fabs(tj-tp)/std::sqrt(nh);
            double orth = 0;
            for (int i=0;i<3;++i) for (int j=0;j<3;++j) { double s=0; for (int k=0;k<3;++k) s+=Rp[k][i]*Rp[k][j]; orth=std::max(orth,std::fabs(s-(i==j))); }
            worst_orth = std::max(worst_orth, orth);
            (void)og;
            if (m > worst) worst = m;
        }
    }
    printf("fast path taken %ld / %ld, worst |dR| = %.3g, worst orthogonality defect %.3g\n", fast, total, worst, worst_orth);
    const bool ok = worst <= 1e-11 && worst_orth <= 1e-14 && fast > total / 2;
    printf(ok ? "PASS\n" : "FAIL\n");
    return ok ? 0 : 1;
}
