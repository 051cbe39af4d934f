// abi_latency.cpp -- one-frame calls through the C ABI itself (no Python in the way), BASELINE configs[1]'s shape: a 363-atom group in a
// 32 817-atom orthorhombic system.  µs per call of gr_group_center (naive / Bai-Breen estimate / COM) and gr_rmsd_batch with one frame,
// with the single-wave kernels (default) and with GR_TUNE_SMALL_CALLS = 0.
// Build: g++ -O2 -std=c++17 -Iinclude -o tools/bin/abi_latency tools/microbench/abi_latency.cpp -Lgroan_rs_amd -lgroan_hip -Wl,-rpath,'$ORIGIN/../../groan_rs_amd'
#include "groan_hip.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const float sigma = argc > 1 ? (float)atof(argv[1]) : 0.3f;   // spread of the group (nm): 0.6 makes it wider than half the cell -- the image proof fails, every RMSD call takes the literal redo
    const uint64_t n = 32817;
    int st = 0;
    gr_ctx *cur = gr_ctx_create(0, n, 2, &st), *ref = gr_ctx_create(0, n, 1, &st);
    if (!cur || !ref) { printf("no context: %d\n", st); return 1; }
    std::mt19937 rng(7); std::normal_distribution<float> g(0.f, sigma); std::uniform_real_distribution<float> u(1.f, 16.f);
    std::vector<float> m(n), x0(3 * n), x1(3 * n);
    const float box9[9] = { 6.44f, 6.76f, 7.26f, 0, 0, 0, 0, 0, 0 };
    for (uint64_t i = 0; i < n; ++i) { m[i] = u(rng); for (int a = 0; a < 3; ++a) { x0[3 * i + a] = 3.2f + g(rng); x1[3 * i + a] = x0[3 * i + a] + 0.05f * g(rng); } }
    for (float &v : x0) v = v < 0.f ? v + 6.44f : (v > 6.44f ? v - 6.44f : v);
    for (float &v : x1) v = v < 0.f ? v + 6.44f : (v > 6.44f ? v - 6.44f : v);
    const uint64_t s0 = 0, e0 = 362;
    for (gr_ctx *c : { cur, ref }) { gr_set_masses(c, m.data(), n); gr_group_create_from_ranges(c, "Peptide", &s0, &e0, 1); }
    gr_frame_upload(ref, 0, x0.data(), box9); gr_frame_upload_wait(ref, 0);
    gr_frame_upload(cur, 0, x1.data(), box9); gr_frame_upload_wait(cur, 0);
    gr_rmsd_plan *plan = gr_rmsd_plan_create(ref, 0, cur, "Peptide", &st);
    if (!plan) { printf("no plan: %d\n", st); return 1; }
    for (int small : { 4096, 0 }) {
        gr_ctx_set_tuning(cur, GR_TUNE_SMALL_CALLS, small);
        printf("GR_TUNE_SMALL_CALLS = %d\n", small);
        auto run = [&](const char *name, auto body) {
            for (int i = 0; i < 20; ++i) body();
            const int reps = 1000; const double t0 = now();
            for (int i = 0; i < reps; ++i) body();
            printf("  %-44s %7.2f us per call\n", name, (now() - t0) / reps);
        };
        float out[3], r; int fs;
        run("gr_group_center naive COM", [&] { gr_group_center(cur, 0, "Peptide", GR_CENTER_NAIVE, 1, out); });
        run("gr_group_center Bai-Breen estimate (COM)", [&] { gr_group_center(cur, 0, "Peptide", GR_CENTER_ESTIMATE, 1, out); });
        run("gr_group_center COM (estimate + unwrapped mean)", [&] { gr_group_center(cur, 0, "Peptide", GR_CENTER_PBC, 1, out); });
        run("gr_rmsd_batch, 1 frame", [&] { gr_rmsd_batch(plan, 0, 1, &r, &fs, nullptr); });
    }
    gr_rmsd_plan_destroy(plan); gr_ctx_destroy(cur); gr_ctx_destroy(ref);
    return 0;
}
