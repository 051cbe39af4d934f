// tests/cpp/test_host.cpp -- the C++ host mirror (include/groan_hip.hpp) exercised the way the reference's own
// unit tests exercise the Rust API.  Built by __graft_entry__.build() (g++, links libgroan_hip.so), run on the
// GPU box by tests/test_gpu_cpp_host.py which passes a directory of raw little-endian fixtures:
//   gro_keep.f32 [261][3]  frames.f32 [11][261][3]  boxes.f32 [11][9]  box0.f32 [9]  masses.f32 [261]
#include <cmath>
#include <cstdio>
#include <fstream>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/groan_hip.hpp"

using namespace groan;
static int failures = 0;
#define CHECK(...) do { if (!(__VA_ARGS__)) { printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #__VA_ARGS__); ++failures; } } while (0)
#define NEAR(a, b, eps) CHECK(std::fabs((double)(a) - (double)(b)) <= (eps))

static std::vector<float> load(const std::string &path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) { printf("cannot open %s\n", path.c_str()); exit(2); }
    std::vector<float> v((size_t)f.tellg() / 4);
    f.seekg(0);
    f.read(reinterpret_cast<char *>(v.data()), (std::streamsize)v.size() * 4);
    return v;
}

struct RmsdData {   // ParallelTrajData (parallel.rs:31-49)
    std::vector<std::pair<uint64_t, float>> values;
    size_t worker = 0;
    void initialize(size_t thread_id) { worker = thread_id; }
    static RmsdData reduce(std::vector<RmsdData> data) {
        RmsdData out;
        for (auto &d : data) out.values.insert(out.values.end(), d.values.begin(), d.values.end());
        return out;
    }
};

int main(int argc, char **argv) {
    if (argc < 2) { printf("usage: test_host <fixture dir>\n"); return 2; }
    const std::string dir = argv[1];
    // ---- analysis.rs:895-929 get_com_two_atoms_pbc
    {
        System s(2);
        s.set_masses({12.8f, 0.4f});
        const float xyz[6] = {4.5f, 3.2f, 1.7f, 9.8f, 9.5f, 3.0f};
        const Box9 box = {10, 10, 10, 0, 0, 0, 0, 0, 0};
        s.set_frame(xyz, &box);
        Vector3D c = s.group_get_com("all");
        NEAR(c[0], 4.35757, 1e-4); NEAR(c[1], 3.08788, 1e-4); NEAR(c[2], 1.7393947, 1e-4);
        c = s.group_get_center("all");
        NEAR(c[0], 2.15, 1e-5); NEAR(c[1], 1.35, 1e-5); NEAR(c[2], 2.35, 1e-5);
        NEAR(s.atoms_distance(0, 1, Dimension::X), 4.7, 1e-5);   // 4.5 - 9.8 = -5.3 -> +10
        try { s.group_get_center("Nonexistent"); CHECK(false); } catch (const Error &e) { CHECK(e.kind == "GroupError" && e.variant == "NotFound"); }
        try { s.atoms_distance(0, 7, Dimension::XYZ); CHECK(false); } catch (const Error &e) { CHECK(e.variant == "OutOfRange" && e.index == 7); }
        s.set_box(nullptr);
        try { s.group_get_com("all"); CHECK(false); } catch (const Error &e) { CHECK(e.variant == "InvalidSimBox(DoesNotExist)"); }
    }
    // ---- container.rs:517-573
    {
        AtomContainer c = AtomContainer::from_indices({11, 1, 2, 3, 20, 5, 0, 5, 4, 18, 6, 19, 1, 13, 20, 27}, 20);
        CHECK(c.blocks.size() == 4 && c.blocks[0] == std::make_pair<uint64_t, uint64_t>(0, 6) && c.blocks[3] == std::make_pair<uint64_t, uint64_t>(18, 19));
        CHECK(c.get_n_atoms() == 11 && c.isin(5) && !c.isin(12));
        AtomContainer r = AtomContainer::from_ranges({{543, 1020}, {1000, 1432}}, 1028);
        CHECK(r.blocks.size() == 1 && r.blocks[0].second == 1027);
    }
    // ---- rmsd.rs:795-820 + 1202-1226: RMSD of the Protein over short_trajectory.xtc
    const float expected[11] = {0.23669721f, 0.2634763f, 0.26021627f, 0.21364464f, 0.22166993f, 0.19383307f, 0.26422343f,
                                0.27013618f, 0.26398134f, 0.23475659f, 0.24208021f};
    std::vector<float> gro = load(dir + "/gro_keep.f32"), frames = load(dir + "/frames.f32"), boxes = load(dir + "/boxes.f32"),
                       box0 = load(dir + "/box0.f32"), masses = load(dir + "/masses.f32");
    const uint64_t n = masses.size();
    CHECK(gro.size() == 3 * n && frames.size() == 11 * 3 * n && boxes.size() == 99);
    Box9 b0; for (int k = 0; k < 9; ++k) b0[k] = box0[k];
    System reference(n);
    reference.set_masses(masses);
    reference.group_create_from_ranges("Protein", {{0, 60}});
    reference.set_frame(gro.data(), &b0);
    {   // ---- geometry selection (groups.rs:94-188, shape.rs): device result == the host predicate applied atom by atom
        const Shape sphere = Shape::sphere({0.5f, 4.5f, 3.5f}, 4.6f), cyl = Shape::cylinder({5.0f, 8.0f, 3.0f}, 3.0f, 6.0f, Dimension::Y);
        CHECK(Shape::sphere({1.0f, 2.0f, 4.5f}, 1.5f).inside({4.8f, 2.1f, 0.3f}, Box9{5, 5, 5, 0, 0, 0, 0, 0, 0}));          // shape.rs:532-539
        CHECK(!Shape::sphere({1.0f, 2.0f, 4.5f}, 1.5f).inside_naive({4.8f, 2.1f, 0.3f}));                                     // :583-586
        try { Shape::cylinder({1, 2, 3}, 1, 1, Dimension::XY); CHECK(false); } catch (const std::invalid_argument &) {}      // Cylinder::new panics
        try { Shape::triangular_prism({3, 4, 2}, {7, 5, 1.8f}, {4, 3, 2}, 4.3f); CHECK(false); } catch (const std::invalid_argument &) {}   // :948-957
        CHECK(!reference.group_create_from_geometries("Picked", "all", {sphere}));
        CHECK(!reference.group_create_from_geometries("Picked2", "all", {sphere, cyl}));
        uint64_t want = 0, want2 = 0;
        for (uint64_t i = 0; i < n; ++i) {
            const Vector3D p{gro[3 * i], gro[3 * i + 1], gro[3 * i + 2]};
            if (sphere.inside(p, b0)) ++want;
            if (sphere.inside(p, b0) && cyl.inside(p, b0)) ++want2;
        }
        CHECK(reference.group_get_n_atoms("Picked") == want && want > 0 && want < n);
        CHECK(reference.group_get_n_atoms("Picked2") == want2);
        try { reference.group_create_from_geometry("Pi>cked", "all", sphere); CHECK(false); } catch (const Error &e) { CHECK(e.variant == "InvalidName"); }
        try { reference.group_create_from_geometry("Picked", "nothing", sphere); CHECK(false); } catch (const Error &e) { CHECK(e.variant == "InvalidQuery"); }
    }
    {
        System system(n, 0, 11);
        system.set_masses(masses);
        system.group_create_from_ranges("Protein", {{0, 60}});
        RMSDConverterAnalyzer analyzer(reference, system, "Protein");
        uint64_t next = 0;
        std::vector<float> got;
        auto source = [&](Frame &f) {
            if (next >= 11) return false;
            static Box9 bx; for (int k = 0; k < 9; ++k) bx[k] = boxes[9 * next + k];
            f = Frame{frames.data() + next * 3 * n, &bx, next, (float)next}; ++next; return true;
        };
        for_each_frame_analyze<float>(system, source, analyzer, [&](const Frame &, float r) { got.push_back(r); });
        CHECK(got.size() == 11);
        for (int f = 0; f < 11 && f < (int)got.size(); ++f) NEAR(got[f], expected[f], 5e-7);
        // calc_rmsd_and_fit via the System method, then the fitted frame is optimally superposed
        Box9 bx; for (int k = 0; k < 9; ++k) bx[k] = boxes[k];
        system.set_frame(frames.data(), &bx);
        NEAR(system.calc_rmsd_and_fit(reference, "Protein"), expected[0], 5e-7);
        NEAR(system.calc_rmsd(reference, "Protein"), expected[0], 2e-6);
        // error variants (rmsd.rs:1075-1275)
        try { system.calc_rmsd(reference, "Nonexistent"); CHECK(false); } catch (const Error &e) { CHECK(e.variant == "NonexistentGroup"); }
        system.group_create_from_ranges("Protein", {{0, 59}});
        try { system.calc_rmsd(reference, "Protein"); CHECK(false); } catch (const Error &e) { CHECK(e.variant == "InconsistentGroup" && e.counts[0] == 61 && e.counts[1] == 60); }
    }
    // ---- atom iterators (iterators.rs:1053-1554, iterating.rs:43,108): anonymous selections through gr_sel_*
    {
        System s(n, 0, 1);
        s.set_masses(masses);
        Box9 bx; for (int k = 0; k < 9; ++k) bx[k] = boxes[k];
        s.set_frame(frames.data(), &bx);
        s.group_create_from_ranges("Protein", {{0, 60}});
        AtomIterator it = s.group_iter("Protein");
        CHECK(it.get_n_atoms() == 61);
        Vector3D a = it.get_com(), b = s.group_get_com("Protein");
        for (int k = 0; k < 3; ++k) CHECK(a[k] == b[k]);                                   // the same kernels, the same bits
        AtomContainer none;
        Vector3D e = AtomIterator(s, none).get_center();
        CHECK(std::isnan(e[0]) && std::isnan(e[1]) && std::isnan(e[2]));                   // an empty iterator is NaN, not an error (:1186-1188)
        AtomIterator near = s.atoms_iter().filter_geometry(Shape::sphere(b, 1.0f));
        CHECK(near.get_n_atoms() > 0 && near.get_n_atoms() < n);
        std::vector<float> before = s.get_positions();
        it.translate({1.0f, 0.0f, 0.0f});
        std::vector<float> after = s.get_positions();
        CHECK(after[3 * 100] == before[3 * 100]);                                          // atom 100 is not in the iterator
        CHECK(after[0] != before[0]);
        AtomContainer out; out.blocks.emplace_back(n - 1, n + 3);
        try { AtomIterator(s, out).get_center_naive(); CHECK(false); } catch (const Error &err) { CHECK(err.variant == "OutOfRange"); }
    }
    // ---- traj_iter_map_reduce (parallel.rs:208-481): two workers (both on device 0), frames round-robin
    {
        auto make_system = [&](int device) {
            System s(n, device, 1);
            s.set_masses(masses);
            s.group_create_from_ranges("Protein", {{0, 60}});
            return s;
        };
        std::vector<Box9> bxs(11);
        for (int f = 0; f < 11; ++f) for (int k = 0; k < 9; ++k) bxs[f][k] = boxes[9 * f + k];
        auto read_frame = [&](uint64_t f, Frame &out) { out = Frame{frames.data() + f * 3 * n, &bxs[f], f, (float)f}; return true; };
        // a context is not re-entrant, and calc_rmsd uses the REFERENCE context's workspace to extract the reference
        // side: workers that share one reference System serialise on it (or build one RMSDConverterAnalyzer each)
        std::mutex ref_mutex;
        auto body = [&](System &s, RmsdData &d) {
            std::lock_guard<std::mutex> lock(ref_mutex);
            d.values.emplace_back(d.values.size() * 2 + d.worker, s.calc_rmsd(reference, "Protein"));
        };
        RmsdData all = traj_iter_map_reduce<RmsdData>({0, 0}, 11, make_system, read_frame, body, RmsdData{});
        CHECK(all.values.size() == 11);
        for (auto &kv : all.values) NEAR(kv.second, expected[kv.first], 5e-7);
        // an error in one worker surfaces as an error of the whole call
        auto bad_body = [&](System &s, RmsdData &) { std::lock_guard<std::mutex> lock(ref_mutex); s.calc_rmsd(reference, "Nonexistent"); };
        try { traj_iter_map_reduce<RmsdData>({0, 0}, 11, make_system, read_frame, bad_body, RmsdData{}); CHECK(false); } catch (const std::runtime_error &) {}
    }
    // ---- XtcReader as a frame source + RMSDConverterAnalyzer + XtcWriter: `system.xtc_iter(f)?.convert_and_analyze(..)` and
    // traj_write_frame on the reference's 50-atom triclinic trajectory (argv[2]); decode(encode(decode)) is a fixed point
    if (argc >= 3) {
        XtcReader xr(std::string(argv[2]) + "/triclinic_trajectory.xtc");
        CHECK(xr.n_atoms() == 50 && xr.n_frames() == 11);
        System sys50(50, 0, 4);
        std::vector<float> m50(50, 1.0f);
        sys50.set_masses(m50);
        std::vector<uint64_t> steps;
        std::vector<std::vector<float>> decoded;
        {
            XtcWriter xw(std::string(argv[1]) + "/rewritten.xtc");
            auto src = xr.frames();
            Frame f;
            while (src(f)) {
                steps.push_back(f.step);
                decoded.emplace_back(f.xyz, f.xyz + 150);
                xw.write_frame(f.xyz, 50, f.box, (int64_t)f.step, f.time, 100.0f);
            }
        }
        CHECK(steps.size() == 11 && steps[0] == 0 && steps[1] == 5000 && steps[10] == 50000);   // xtc_io/mod.rs tests
        XtcReader again(std::string(argv[1]) + "/rewritten.xtc");
        CHECK(again.n_frames() == 11);
        for (uint64_t i = 0; i < 11; ++i) { Frame f = again.read(i); for (int k = 0; k < 150; ++k) CHECK(f.xyz[k] == decoded[i][k]); }
        auto strided = xr.frames(1, 8, 3);   // with_range + with_step: frames 1, 4, 7
        Frame f; int cnt = 0; uint64_t want[3] = {5000, 20000, 35000};
        while (strided(f)) { CHECK(cnt < 3 && f.step == want[cnt]); ++cnt; }
        CHECK(cnt == 3);
        xr.read_frames_device(sys50, 0, 4);
        for (uint32_t k = 0; k < 4; ++k) { std::vector<float> got = sys50.get_positions(k); for (int j = 0; j < 150; ++j) CHECK(got[j] == decoded[k][j]); }
        TrrReader tr(std::string(argv[2]) + "/triclinic_trajectory.trr");
        CHECK(tr.n_atoms() == 50 && tr.n_frames() == 13);
        try { XtcReader bad(std::string(argv[2]) + "/no_such_file.xtc"); CHECK(false); } catch (const Error &e) { CHECK(e.variant == "FileNotFound"); }
    }
    printf(failures ? "test_host: %d FAILED\n" : "test_host: all passed\n", failures);
    return failures ? 1 : 0;
}
