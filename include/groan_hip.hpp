// groan_hip.hpp -- C++17 host-side mirror of the groan_rs interface for the per-frame geometry path, on top of
// the C ABI (groan_hip.h).  Header-only and thin: same names, argument meaning and error behaviour as the
// reference's Rust API so that host code (and the tests in tests/cpp/) read like the reference's own:
//
//   reference (Rust)                                     here (C++)
//   System::new / clone per worker                       groan::System(n_atoms, device, n_slots)
//   TrajRead::update_system                              System::set_frame(xyz, box9)
//   system.group_create_from_ranges / _from_indices      same
//   Sphere / Rectangular / Cylinder / TriangularPrism    groan::Shape factories; system.group_create_from_geometry(-ies)
//   system.group_get_center / _com / estimate_* / naive  same (return Vector3D)
//   system.group_distance / atoms_distance / group_all_distances
//   system.atoms_translate / atoms_wrap / group_* / atoms_center(_mass)
//   system.calc_rmsd(&reference, group) / calc_rmsd_and_fit
//   FrameAnalyze / FrameConvert / FrameConvertAnalyze    abstract classes with the same single method
//   TrajAnalyzer / TrajConverter / TrajConverterAnalyzer for_each_frame_* adapters over any frame source
//   RMSDConverterAnalyzer                                 groan::RMSDConverterAnalyzer (cached plan)
//   XtcReader / TrrReader (+ with_range / with_step)      groan::XtcReader / TrrReader::frames(start, end, step) as frame sources
//   XtcWriter                                             groan::XtcWriter (host frames and device slots)
//   ParallelTrajData + traj_iter_map_reduce              groan::traj_iter_map_reduce (one worker thread per GPU,
//                                                         frames round-robin, shared error flag, reduce())
//   Result<T, GroupError|AtomError|RMSDError>             exceptions carrying the variant + payload
#pragma once
#include <array>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <algorithm>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "groan_hip.h"

namespace groan {

using Vector3D = std::array<float, 3>;
using Box9 = std::array<float, 9>;   // gro order (simbox.rs:13-26)

enum class Dimension : int { None = 0, X, Y, Z, XY, XZ, YZ, XYZ };   // dimension.rs:13-23

// ---- errors (src/errors.rs): variant names as in the reference
struct Error : std::runtime_error {
    std::string kind, variant; int status; uint64_t index; uint64_t counts[2];
    Error(std::string k, std::string v, int st, uint64_t idx = 0, uint64_t c0 = 0, uint64_t c1 = 0)
        : std::runtime_error(k + "::" + v), kind(std::move(k)), variant(std::move(v)), status(st), index(idx), counts{c0, c1} {}
};

inline std::string simbox_variant(int st) { return st == GR_E_NOT_ORTHOGONAL ? "NotOrthogonal" : (st == GR_E_NO_BOX ? "DoesNotExist" : "ZeroLength"); }

// ---- AtomContainer (container.rs): bit-exact block logic lives in the library
struct AtomContainer {
    std::vector<std::pair<uint64_t, uint64_t>> blocks;
    static AtomContainer from_indices(const std::vector<uint64_t> &idx, uint64_t n_atoms) {
        std::vector<uint64_t> s(idx.size() + 1), e(idx.size() + 1);
        size_t nb = gr_container_from_indices(idx.data(), idx.size(), n_atoms, s.data(), e.data());
        return make(s, e, nb);
    }
    static AtomContainer from_ranges(const std::vector<std::pair<uint64_t, uint64_t>> &r, uint64_t n_atoms) {
        std::vector<uint64_t> a(r.size() + 1), b(r.size() + 1), s(r.size() + 1), e(r.size() + 1);
        for (size_t i = 0; i < r.size(); ++i) { a[i] = r[i].first; b[i] = r[i].second; }
        size_t nb = gr_container_from_ranges(a.data(), b.data(), r.size(), n_atoms, s.data(), e.data());
        return make(s, e, nb);
    }
    uint64_t get_n_atoms() const { auto [s, e] = split(); return gr_container_n_atoms(s.data(), e.data(), blocks.size()); }
    bool isin(uint64_t i) const { auto [s, e] = split(); return gr_container_isin(s.data(), e.data(), blocks.size(), i) != 0; }
    std::vector<uint64_t> indices() const {
        auto [s, e] = split();
        std::vector<uint64_t> out(get_n_atoms() + 1);
        out.resize(gr_container_expand(s.data(), e.data(), blocks.size(), out.data()));
        return out;
    }
    static AtomContainer set_union(const AtomContainer &x, const AtomContainer &y) {
        auto [s1, e1] = x.split(); auto [s2, e2] = y.split();
        std::vector<uint64_t> s(x.blocks.size() + y.blocks.size() + 1), e(s.size());
        size_t nb = gr_container_union(s1.data(), e1.data(), x.blocks.size(), s2.data(), e2.data(), y.blocks.size(), s.data(), e.data());
        return make(s, e, nb);
    }
  private:
    static AtomContainer make(const std::vector<uint64_t> &s, const std::vector<uint64_t> &e, size_t nb) {
        AtomContainer c; for (size_t i = 0; i < nb; ++i) c.blocks.emplace_back(s[i], e[i]); return c;
    }
    std::pair<std::vector<uint64_t>, std::vector<uint64_t>> split() const {
        std::vector<uint64_t> s(blocks.size() + 1), e(blocks.size() + 1);
        for (size_t i = 0; i < blocks.size(); ++i) { s[i] = blocks[i].first; e[i] = blocks[i].second; }
        return {s, e};
    }
};

// ---- System (src/system/mod.rs:38-73)
// Shape (src/structures/shape.rs): a value type over gr_shape; the factories throw where the reference panics.
struct Shape {
    gr_shape c{};
    static Shape sphere(const Vector3D &pos, float radius) { Shape s; check(gr_shape_sphere(&s.c, pos.data(), radius), "Sphere::new"); return s; }
    static Shape rectangular(const Vector3D &pos, float x, float y, float z) { Shape s; check(gr_shape_rectangular(&s.c, pos.data(), x, y, z), "Rectangular::new"); return s; }
    static Shape cylinder(const Vector3D &pos, float radius, float height, Dimension orientation) {
        Shape s; check(gr_shape_cylinder(&s.c, pos.data(), radius, height, (int)orientation), "Cylinder::new | Unsupported orientation dimension"); return s;
    }
    static Shape triangular_prism(const Vector3D &b1, const Vector3D &b2, const Vector3D &b3, float height) {
        Shape s; check(gr_shape_triangular_prism(&s.c, b1.data(), b2.data(), b3.data(), height), "TriangularPrism::new | invalid base"); return s;
    }
    bool inside(const Vector3D &point, const Box9 &box) const { int in = 0; check(gr_shape_inside(&c, point.data(), box.data(), 0, &in), "Shape::inside"); return in != 0; }
    bool inside_naive(const Vector3D &point) const { int in = 0; check(gr_shape_inside(&c, point.data(), nullptr, 1, &in), "NaiveShape::inside_naive"); return in != 0; }
private:
    static void check(int st, const char *what) { if (st != GR_OK) throw std::invalid_argument(std::string("FATAL GROAN ERROR | ") + what); }
};

class System {
  public:
    System(uint64_t n_atoms, int device = 0, uint32_t n_slots = 1) : n_(n_atoms), device_(device) {
        int st = 0;
        ctx_ = gr_ctx_create(device, n_atoms, n_slots, &st);
        if (!ctx_) throw Error("DeviceError", gr_status_string(st), st);
    }
    ~System() { if (ctx_) gr_ctx_destroy(ctx_); }
    System(const System &) = delete;
    System &operator=(const System &) = delete;
    System(System &&o) noexcept : ctx_(o.ctx_), n_(o.n_), device_(o.device_) { o.ctx_ = nullptr; }

    gr_ctx *raw() const { return ctx_; }
    // System::atoms_iter / group_iter (src/system/iterating.rs:43,108); defined below AtomIterator
    class AtomIterator atoms_iter(uint32_t slot = 0);
    class AtomIterator group_iter(const std::string &name, uint32_t slot = 0);
    friend class AtomIterator;
    uint64_t get_n_atoms() const { return n_; }
    int device() const { return device_; }
    void set_masses(const std::vector<float> &m) { check_plain(gr_set_masses(ctx_, m.data(), m.size())); }
    void set_strict_orthogonal(bool on) { gr_ctx_set_strict_orthogonal(ctx_, on ? 1 : 0); }

    // TrajRead::update_system (traj_read.rs:160-186): rvec[n] + box as the xtc readers deliver them
    void set_frame(const float *xyz, const Box9 *box, uint32_t slot = 0) {
        check_plain(gr_frame_upload(ctx_, slot, xyz, box ? box->data() : nullptr));
        check_plain(gr_sync(ctx_));
    }
    std::vector<float> get_positions(uint32_t slot = 0) const {
        std::vector<float> out(3 * n_);
        check_plain(gr_frame_download(ctx_, slot, out.data()));
        return out;
    }
    void set_box(const Box9 *box, uint32_t slot = 0) { check_plain(gr_frame_set_box(ctx_, slot, box ? box->data() : nullptr)); }

    bool group_create_from_ranges(const std::string &name, const std::vector<std::pair<uint64_t, uint64_t>> &r) {
        std::vector<uint64_t> s(r.size() + 1), e(r.size() + 1);
        for (size_t i = 0; i < r.size(); ++i) { s[i] = r[i].first; e[i] = r[i].second; }
        int st = gr_group_create_from_ranges(ctx_, name.c_str(), s.data(), e.data(), r.size());
        if (st != GR_OK && st != GR_E_GROUP_EXISTS) group_error(st, name);
        return st == GR_E_GROUP_EXISTS;   // AlreadyExistsWarning
    }
    bool group_create_from_indices(const std::string &name, const std::vector<uint64_t> &idx) {
        int st = gr_group_create_from_indices(ctx_, name.c_str(), idx.data(), idx.size());
        if (st != GR_OK && st != GR_E_GROUP_EXISTS) group_error(st, name);
        return st == GR_E_GROUP_EXISTS;
    }
    // System::group_create_from_geometry / _geometries (groups.rs:94-188) with a source group in place of the query
    bool group_create_from_geometries(const std::string &name, const std::string &source, const std::vector<Shape> &shapes, uint32_t slot = 0, bool naive = false) {
        std::vector<gr_shape> raw;
        for (const Shape &s : shapes) raw.push_back(s.c);
        int st = gr_group_create_from_geometries(ctx_, slot, name.c_str(), source.c_str(), raw.data(), raw.size(), naive ? 1 : 0);
        if (st == GR_E_GROUP_NOT_FOUND) throw Error("GroupError", "InvalidQuery", st);
        if (st == GR_E_INVALID_NAME) throw Error("GroupError", "InvalidName", st);
        if (st != GR_OK && st != GR_E_GROUP_EXISTS) group_error(st, name);
        return st == GR_E_GROUP_EXISTS;
    }
    bool group_create_from_geometry(const std::string &name, const std::string &source, const Shape &shape, uint32_t slot = 0) {
        return group_create_from_geometries(name, source, { shape }, slot);
    }
    uint64_t group_get_n_atoms(const std::string &name) const {
        uint64_t n = 0;
        if (gr_group_n_atoms(ctx_, name.c_str(), &n) != GR_OK) throw Error("GroupError", "NotFound", GR_E_GROUP_NOT_FOUND);
        return n;
    }

    // analysis.rs:52-320
    Vector3D group_get_center_naive(const std::string &g, uint32_t slot = 0) const { return center(g, GR_CENTER_NAIVE, 0, slot); }
    Vector3D group_estimate_center(const std::string &g, uint32_t slot = 0) const { return center(g, GR_CENTER_ESTIMATE, 0, slot); }
    Vector3D group_get_center(const std::string &g, uint32_t slot = 0) const { return center(g, GR_CENTER_PBC, 0, slot); }
    Vector3D group_get_com_naive(const std::string &g, uint32_t slot = 0) const { return center(g, GR_CENTER_NAIVE, 1, slot); }
    Vector3D group_estimate_com(const std::string &g, uint32_t slot = 0) const { return center(g, GR_CENTER_ESTIMATE, 1, slot); }
    Vector3D group_get_com(const std::string &g, uint32_t slot = 0) const { return center(g, GR_CENTER_PBC, 1, slot); }

    // analysis.rs:348-471
    float group_distance(const std::string &g1, const std::string &g2, Dimension dim, uint32_t slot = 0) const {
        float out = 0;
        int st = gr_group_distance(ctx_, slot, g1.c_str(), g2.c_str(), (int)dim, &out);
        if (st) group_error(st, g1);
        return out;
    }
    float atoms_distance(uint64_t i, uint64_t j, Dimension dim, uint32_t slot = 0) const {
        float out = 0;
        int st = gr_atoms_distance(ctx_, slot, i, j, (int)dim, &out);
        if (st) atom_error(st);
        return out;
    }
    // row-major n1 x n2 (ndarray::Array2<f32>)
    std::vector<float> group_all_distances(const std::string &g1, const std::string &g2, Dimension dim, uint32_t slot = 0) const {
        std::vector<float> out((size_t)group_get_n_atoms(g1) * group_get_n_atoms(g2));
        int st = gr_group_all_distances(ctx_, slot, g1.c_str(), g2.c_str(), (int)dim, out.data(), out.size());
        if (st) group_error(st, g1);
        return out;
    }

    // modifying.rs:45-75,201-222 ; utility.rs:109-185
    void atoms_translate(const Vector3D &v, uint32_t slot = 0) { int st = gr_group_translate(ctx_, slot, nullptr, v.data()); if (st) atom_error(st); }
    void group_translate(const std::string &g, const Vector3D &v, uint32_t slot = 0) { int st = gr_group_translate(ctx_, slot, g.c_str(), v.data()); if (st) group_error(st, g); }
    void atoms_wrap(uint32_t slot = 0) { int st = gr_group_wrap(ctx_, slot, nullptr); if (st) atom_error(st); }
    void group_wrap(const std::string &g, uint32_t slot = 0) { int st = gr_group_wrap(ctx_, slot, g.c_str()); if (st) group_error(st, g); }
    void atoms_center(const std::string &g, Dimension dim, uint32_t slot = 0) { int st = gr_atoms_center(ctx_, slot, g.c_str(), (int)dim, 0); if (st) group_error(st, g); }
    void atoms_center_mass(const std::string &g, Dimension dim, uint32_t slot = 0) { int st = gr_atoms_center(ctx_, slot, g.c_str(), (int)dim, 1); if (st) group_error(st, g); }

    // rmsd.rs:75-166
    float calc_rmsd(const System &reference, const std::string &group, uint32_t slot = 0, uint32_t ref_slot = 0) const {
        float r = 0;
        int st = gr_calc_rmsd(ctx_, slot, reference.ctx_, ref_slot, group.c_str(), &r, nullptr);
        if (st) rmsd_error(st, group);
        return r;
    }
    float calc_rmsd_and_fit(const System &reference, const std::string &group, uint32_t slot = 0, uint32_t ref_slot = 0) {
        float r = 0;
        int st = gr_calc_rmsd_and_fit(ctx_, slot, reference.ctx_, ref_slot, group.c_str(), &r);
        if (st) rmsd_error(st, group);
        return r;
    }

    [[noreturn]] void rmsd_error(int st, const std::string &group) const {
        uint64_t idx = gr_last_error_index(ctx_);
        switch (st) {
        case GR_E_GROUP_NOT_FOUND: throw Error("RMSDError", "NonexistentGroup", st);
        case GR_E_EMPTY_GROUP: throw Error("RMSDError", "EmptyGroup", st);
        case GR_E_NO_POSITION: throw Error("RMSDError", "InvalidPosition", st, idx);
        case GR_E_NO_MASS: throw Error("RMSDError", "InvalidMass", st, idx);
        case GR_E_INCONSISTENT_GROUP: { uint64_t c[2]; gr_last_error_counts(ctx_, c); throw Error("RMSDError", "InconsistentGroup", st, 0, c[0], c[1]); }
        case GR_E_NO_BOX: case GR_E_NOT_ORTHOGONAL: case GR_E_ZERO_BOX: throw Error("RMSDError", "InvalidSimBox(" + simbox_variant(st) + ")", st);
        default: throw Error("DeviceError", std::string(gr_status_string(st)) + ": " + gr_last_error(ctx_) + " [" + group + "]", st);
        }
    }

  private:
    Vector3D center(const std::string &g, int kind, int weighted, uint32_t slot) const {
        Vector3D out{};
        int st = gr_group_center(ctx_, slot, g.c_str(), kind, weighted, out.data());
        if (st) group_error(st, g);
        return out;
    }
    void check_plain(int st) const { if (st) throw Error("DeviceError", std::string(gr_status_string(st)) + ": " + gr_last_error(ctx_), st); }
    [[noreturn]] void group_error(int st, const std::string &g) const {
        uint64_t idx = gr_last_error_index(ctx_);
        switch (st) {
        case GR_E_GROUP_NOT_FOUND: throw Error("GroupError", "NotFound", st);
        case GR_E_EMPTY_GROUP: throw Error("GroupError", "EmptyGroup", st);
        case GR_E_NO_POSITION: throw Error("GroupError", "InvalidPosition", st, idx);
        case GR_E_NO_MASS: throw Error("GroupError", "InvalidMass", st, idx);
        case GR_E_NO_BOX: case GR_E_NOT_ORTHOGONAL: case GR_E_ZERO_BOX: throw Error("GroupError", "InvalidSimBox(" + simbox_variant(st) + ")", st);
        default: throw Error("DeviceError", std::string(gr_status_string(st)) + ": " + gr_last_error(ctx_) + " [" + g + "]", st);
        }
    }
    [[noreturn]] void atom_error(int st) const {
        uint64_t idx = gr_last_error_index(ctx_);
        switch (st) {
        case GR_E_OUT_OF_RANGE: throw Error("AtomError", "OutOfRange", st, idx);
        case GR_E_NO_POSITION: throw Error("AtomError", "InvalidPosition", st, idx);
        case GR_E_NO_MASS: throw Error("AtomError", "InvalidMass", st, idx);
        case GR_E_NO_BOX: case GR_E_NOT_ORTHOGONAL: case GR_E_ZERO_BOX: throw Error("AtomError", "InvalidSimBox(" + simbox_variant(st) + ")", st);
        default: throw Error("DeviceError", std::string(gr_status_string(st)) + ": " + gr_last_error(ctx_), st);
        }
    }
    gr_ctx *ctx_ = nullptr;
    uint64_t n_;
    int device_;
};

// ---- atom iterators (src/structures/iterators.rs:28-46,350-402,1053-1604; src/system/iterating.rs:43-140)
// An AtomContainer + the System (and slot) it walks.  Each method is one gr_sel_* call: the kernels take the container's
// blocks directly.  Error behaviour is the iterator traits': box first, then the first atom without position / mass; an EMPTY
// iterator is not an error -- its centre is (NaN, NaN, NaN) (iterators.rs:1186-1188).
class AtomIterator {
  public:
    AtomIterator(System &system, AtomContainer container, uint32_t slot = 0) : sys_(&system), c_(std::move(container)), slot_(slot) {}
    const AtomContainer &container() const { return c_; }
    uint64_t get_n_atoms() const { return c_.get_n_atoms(); }
    Vector3D get_center_naive() const { return center(GR_CENTER_NAIVE, 0); }
    Vector3D get_com_naive() const { return center(GR_CENTER_NAIVE, 1); }
    Vector3D estimate_center() const { return center(GR_CENTER_ESTIMATE, 0); }
    Vector3D get_center() const { return center(GR_CENTER_PBC, 0); }
    Vector3D estimate_com() const { return center(GR_CENTER_ESTIMATE, 1); }
    Vector3D get_com() const { return center(GR_CENTER_PBC, 1); }
    AtomIterator filter_geometry(const Shape &shape) const { return filter(shape, false); }          // :1094-1105
    AtomIterator filter_geometry_naive(const Shape &shape) const { return filter(shape, true); }     // :994-1004
    void translate(const Vector3D &v) { auto se = split(); int st = gr_sel_translate(sys_->raw(), slot_, se.first.data(), se.second.data(), c_.blocks.size(), v.data()); if (st) sys_->atom_error(st); }
    void wrap() { auto se = split(); int st = gr_sel_wrap(sys_->raw(), slot_, se.first.data(), se.second.data(), c_.blocks.size()); if (st) sys_->atom_error(st); }
    AtomIterator set_union(const AtomIterator &o) const { return AtomIterator(*sys_, AtomContainer::set_union(c_, o.c_), slot_); }   // OrderedAtomIterator::union :1572
    std::vector<float> all_distances(const AtomIterator &o, Dimension dim) const {
        auto a = split(); auto b = o.split();
        std::vector<float> out((size_t)get_n_atoms() * o.get_n_atoms());
        int st = gr_sel_all_distances(sys_->raw(), slot_, a.first.data(), a.second.data(), c_.blocks.size(), b.first.data(), b.second.data(), o.c_.blocks.size(), (int)dim, out.data(), out.size());
        if (st) sys_->atom_error(st);
        return out;
    }
  private:
    std::pair<std::vector<uint64_t>, std::vector<uint64_t>> split() const {
        std::vector<uint64_t> s(c_.blocks.size() + 1), e(c_.blocks.size() + 1);
        for (size_t i = 0; i < c_.blocks.size(); ++i) { s[i] = c_.blocks[i].first; e[i] = c_.blocks[i].second; }
        return {s, e};
    }
    Vector3D center(int kind, int weighted) const {
        auto se = split(); Vector3D out{};
        int st = gr_sel_center(sys_->raw(), slot_, se.first.data(), se.second.data(), c_.blocks.size(), kind, weighted, out.data());
        if (st) sys_->atom_error(st);
        return out;
    }
    AtomIterator filter(const Shape &shape, bool naive) const {
        auto se = split();
        const size_t cap = (size_t)get_n_atoms() + 1;
        std::vector<uint64_t> os(cap), oe(cap); size_t nb = 0; uint64_t na = 0;
        int st = gr_sel_filter_geometry(sys_->raw(), slot_, se.first.data(), se.second.data(), c_.blocks.size(), &shape.c, 1, naive ? 1 : 0, os.data(), oe.data(), cap, &nb, &na);
        if (st) sys_->atom_error(st);
        AtomContainer out; for (size_t i = 0; i < nb; ++i) out.blocks.emplace_back(os[i], oe[i]);
        return AtomIterator(*sys_, out, slot_);
    }
    System *sys_; AtomContainer c_; uint32_t slot_;
};
inline AtomIterator System::atoms_iter(uint32_t slot) { AtomContainer c; c.blocks.emplace_back(0, n_ - 1); return AtomIterator(*this, c, slot); }
inline AtomIterator System::group_iter(const std::string &name, uint32_t slot) {
    size_t nb = 0;
    if (gr_group_n_blocks(ctx_, name.c_str(), &nb) != GR_OK) throw Error("GroupError", "NotFound", GR_E_GROUP_NOT_FOUND);
    std::vector<uint64_t> s(nb + 1), e(nb + 1);
    gr_group_blocks(ctx_, name.c_str(), s.data(), e.data());
    AtomContainer c; for (size_t i = 0; i < nb; ++i) c.blocks.emplace_back(s[i], e[i]);
    return AtomIterator(*this, c, slot);
}

// ---- plug-in traits (src/structures/traj_convert.rs:30-36,76-83,125-132)
template <typename R> struct FrameAnalyze { virtual ~FrameAnalyze() = default; virtual R analyze(const System &system) = 0; };
struct FrameConvert { virtual ~FrameConvert() = default; virtual void convert(System &system) = 0; };
template <typename R> struct FrameConvertAnalyze { virtual ~FrameConvertAnalyze() = default; virtual R convert_analyze(System &system) = 0; };

// ---- RMSDConverterAnalyzer (src/system/rmsd.rs:170-251)
class RMSDConverterAnalyzer : public FrameAnalyze<float>, public FrameConvertAnalyze<float> {
  public:
    RMSDConverterAnalyzer(const System &reference, System &target, const std::string &group, uint32_t ref_slot = 0)
        : target_(target), group_(group) {
        int st = 0;
        plan_ = gr_rmsd_plan_create(reference.raw(), ref_slot, target.raw(), group.c_str(), &st);
        if (!plan_) reference.rmsd_error(st, group);
    }
    ~RMSDConverterAnalyzer() override { if (plan_) gr_rmsd_plan_destroy(plan_); }
    float analyze(const System &) override { return batch(0, 1, false)[0]; }
    float convert_analyze(System &) override { return batch(0, 1, true)[0]; }
    // many resident frames per call: the shape the GPU wants
    std::vector<float> batch(uint32_t first_slot, uint32_t n, bool fit) {
        std::vector<float> r(n);
        std::vector<int> st(n);
        int s = fit ? gr_rmsd_fit_batch(plan_, first_slot, n, r.data(), st.data()) : gr_rmsd_batch(plan_, first_slot, n, r.data(), st.data(), nullptr);
        if (s) target_.rmsd_error(s, group_);
        return r;
    }
    uint32_t last_fallbacks() const { return gr_rmsd_plan_last_fallbacks(plan_); }
  private:
    gr_rmsd_plan *plan_ = nullptr;
    System &target_;
    std::string group_;
};

// ---- a decoded frame as the xtc readers hand it to update_system
struct Frame { const float *xyz; const Box9 *box; uint64_t step; float time; };

// TrajAnalyzer::next (traj_convert.rs:95-105): update the system with each frame, run the analyzer
template <typename R, typename Source, typename Sink>
void for_each_frame_analyze(System &system, Source &&next_frame, FrameAnalyze<R> &analyzer, Sink &&sink) {
    Frame f;
    while (next_frame(f)) { system.set_frame(f.xyz, f.box); sink(f, analyzer.analyze(system)); }
}
template <typename R, typename Source, typename Sink>
void for_each_frame_convert_analyze(System &system, Source &&next_frame, FrameConvertAnalyze<R> &ca, Sink &&sink) {
    Frame f;
    while (next_frame(f)) { system.set_frame(f.xyz, f.box); sink(f, ca.convert_analyze(system)); }
}

// ---- XtcReader / TrrReader as frame sources (src/io/xtc_io/mod.rs, src/io/trr_io.rs; `with_range` / `with_step` of
// traj_read.rs:609-697 are the `start, end, step` of frames()), XtcWriter (xtc_io/mod.rs:256-331)
class XtcReader {
  public:
    explicit XtcReader(const std::string &path) {
        int st = 0;
        x_ = gr_xtc_open(path.c_str(), &st);
        if (!x_) throw Error("ReadTrajError", st == GR_E_IO ? "FileNotFound" : "NotXtc", st);
        xyz_.resize(3 * (size_t)n_atoms());
    }
    ~XtcReader() { if (x_) gr_xtc_close(x_); }
    XtcReader(const XtcReader &) = delete;
    XtcReader &operator=(const XtcReader &) = delete;
    uint64_t n_atoms() const { return gr_xtc_n_atoms(x_); }
    uint64_t n_frames() const { return gr_xtc_n_frames(x_); }
    const gr_xtc *raw() const { return x_; }
    // frame i decoded into this reader's buffer (valid until the next read)
    Frame read(uint64_t i) {
        float b[9]; uint64_t step = 0; float time = 0, prec = 0;
        const int st = gr_xtc_read_frame(x_, i, xyz_.data(), b, &step, &time, &prec);
        if (st != GR_OK) throw Error("ReadTrajError", "FrameNotFound", st, i);
        for (int k = 0; k < 9; ++k) box_[k] = b[k];
        return Frame{xyz_.data(), &box_, step, time};
    }
    // a Source for for_each_frame_*: frames start, start + step, ... < end
    std::function<bool(Frame &)> frames(uint64_t start = 0, uint64_t end = UINT64_MAX, uint64_t step = 1) {
        auto next = std::make_shared<uint64_t>(start);
        return [this, next, end, step](Frame &out) {
            if (*next >= std::min<uint64_t>(end, n_frames())) return false;
            out = read(*next); *next += step;
            return true;
        };
    }
    // a batch of frames unpacked on the GPU straight into the system's slots (compressed bytes over PCIe)
    void read_frames_device(System &system, uint64_t first_frame, uint32_t n, uint32_t first_slot = 0, uint64_t frame_step = 1, int host_threads = 0) {
        const int st = gr_xtc_read_frames_device(x_, first_frame, n, frame_step, system.raw(), first_slot, host_threads, nullptr, nullptr);
        if (st != GR_OK) throw Error("ReadTrajError", "FrameNotFound", st, first_frame);
    }
  private:
    gr_xtc *x_ = nullptr;
    std::vector<float> xyz_;
    Box9 box_{};
};

class TrrReader {
  public:
    explicit TrrReader(const std::string &path) {
        int st = 0;
        t_ = gr_trr_open(path.c_str(), &st);
        if (!t_) throw Error("ReadTrajError", st == GR_E_IO ? "FileNotFound" : "NotTrr", st);
        xyz_.resize(3 * (size_t)n_atoms());
    }
    ~TrrReader() { if (t_) gr_trr_close(t_); }
    TrrReader(const TrrReader &) = delete;
    TrrReader &operator=(const TrrReader &) = delete;
    uint64_t n_atoms() const { return gr_trr_n_atoms(t_); }
    uint64_t n_frames() const { return gr_trr_n_frames(t_); }
    // positions of frame i (an all-zero position = the reference's "no position": NaN in x, trr_io.rs:108-112)
    Frame read(uint64_t i) {
        float b[9]; uint64_t step = 0; float time = 0, lambda = 0;
        const int st = gr_trr_read_frame(t_, i, xyz_.data(), nullptr, nullptr, b, &step, &time, &lambda);
        if (st != GR_OK) throw Error("ReadTrajError", "FrameNotFound", st, i);
        for (size_t a = 0; a < xyz_.size(); a += 3) if (xyz_[a] == 0.0f && xyz_[a + 1] == 0.0f && xyz_[a + 2] == 0.0f) xyz_[a] = NAN;
        for (int k = 0; k < 9; ++k) box_[k] = b[k];
        return Frame{xyz_.data(), &box_, step, time};
    }
    std::function<bool(Frame &)> frames(uint64_t start = 0, uint64_t end = UINT64_MAX, uint64_t step = 1) {
        auto next = std::make_shared<uint64_t>(start);
        return [this, next, end, step](Frame &out) {
            if (*next >= std::min<uint64_t>(end, n_frames())) return false;
            out = read(*next); *next += step;
            return true;
        };
    }
  private:
    gr_trr *t_ = nullptr;
    std::vector<float> xyz_;
    Box9 box_{};
};

class XtcWriter {
  public:
    explicit XtcWriter(const std::string &path) {
        int st = 0;
        w_ = gr_xtc_writer_open(path.c_str(), &st);
        if (!w_) throw Error("WriteTrajError", "CouldNotCreate", st);
    }
    ~XtcWriter() { if (w_) gr_xtc_writer_close(w_); }
    XtcWriter(const XtcWriter &) = delete;
    XtcWriter &operator=(const XtcWriter &) = delete;
    void write_frame(const float *xyz, uint64_t n_atoms, const Box9 *box, int64_t step, float time, float precision = 1000.0f) {
        const int st = gr_xtc_write_frame(w_, n_atoms, xyz, box ? box->data() : nullptr, step, time, precision);
        if (st != GR_OK) throw Error("WriteTrajError", "CouldNotWrite", st);
    }
    // device frames (e.g. after calc_rmsd_and_fit on a batch of slots), all atoms or a group
    void write_slots(System &system, uint32_t first_slot, uint32_t n_frames, const char *group = nullptr, float precision = 1000.0f, int host_threads = 0) {
        const int st = gr_xtc_write_slots(w_, system.raw(), first_slot, n_frames, group, nullptr, nullptr, precision, host_threads);
        if (st == GR_E_GROUP_NOT_FOUND) throw Error("WriteTrajError", "GroupNotFound", st);
        if (st != GR_OK) throw Error("WriteTrajError", "CouldNotWrite", st);
    }
  private:
    gr_xtc_writer *w_ = nullptr;
};

// ---- ParallelTrajData + traj_iter_map_reduce (src/system/parallel.rs:31-49,208-481)
// One worker thread per device; worker n takes frames n, n+T, ... (parallel.rs:424-448); a shared AtomicBool
// polled every ERROR_FLAG_FREQ = 10 frames stops the others after the first failure (parallel.rs:28,453-475).
// start_frame / end_frame / step = the reference's start_time / end_time / step in frame-index form (a reader's index turns times
// into frame numbers): the frames visited are start_frame + k * step below end_frame, worker n skips n * step of them and then
// advances by step * T (parallel.rs:425-448).  progress = the ProgressPrinter: called from worker 0 after every frame it completes
// (ProgressStatus::Running; the reference attaches the printer to the master thread only, :417-422) and once at the end with
// Completed and the last frame ANY worker read, or Failed and the frame that failed (:288-321).
enum class ProgressStatus { Running, Completed, Failed };
using ProgressPrinter = std::function<void(ProgressStatus, uint64_t frame)>;
template <typename Data>
Data traj_iter_map_reduce(const std::vector<int> &devices, uint64_t n_frames,
                          const std::function<System(int device)> &make_system,                       // System clone per worker
                          const std::function<bool(uint64_t frame_index, Frame &out)> &read_frame,    // random access frame source
                          const std::function<void(System &, Data &)> &body, const Data &init_data,
                          uint64_t start_frame = 0, uint64_t end_frame = UINT64_MAX, uint64_t step = 1,
                          const ProgressPrinter &progress = nullptr) {
    const size_t T = devices.size();
    if (T == 0) throw std::invalid_argument("Number of threads to spawn must be > 0.");
    if (step == 0) throw std::invalid_argument("ReadTrajError::InvalidStep");
    const uint64_t end = std::min(end_frame, n_frames);
    std::vector<Data> data(T, init_data);
    std::vector<std::string> errors(T);
    std::vector<uint64_t> last_frame(T, start_frame), failed_at(T, 0);
    std::atomic<bool> error_flag{false};
    std::vector<std::thread> workers;
    for (size_t n = 0; n < T; ++n) {
        data[n].initialize(n);
        workers.emplace_back([&, n] {
            uint64_t f = start_frame + n * step;
            try {
                System system = make_system(devices[n]);
                for (uint64_t i = 0; f < end; f += step * T, ++i) {
                    if (i % 10 == 0 && error_flag.load(std::memory_order_relaxed)) return;
                    Frame fr;
                    if (!read_frame(f, fr)) return;
                    system.set_frame(fr.xyz, fr.box);
                    body(system, data[n]);
                    last_frame[n] = f;
                    if (n == 0 && progress) progress(ProgressStatus::Running, f);
                }
            } catch (const std::exception &e) {
                error_flag.store(true, std::memory_order_relaxed);
                errors[n] = e.what(); failed_at[n] = f;
            }
        });
    }
    for (auto &w : workers) w.join();
    for (size_t n = 0; n < T; ++n)
        if (!errors[n].empty()) { if (progress) progress(ProgressStatus::Failed, failed_at[n]); throw std::runtime_error(errors[n]); }
    if (progress) progress(ProgressStatus::Completed, *std::max_element(last_frame.begin(), last_frame.end()));
    return Data::reduce(std::move(data));
}

}  // namespace groan
