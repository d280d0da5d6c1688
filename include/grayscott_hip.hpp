// grayscott_hip.hpp -- header-only C++17 mirror of the reference's backend interface over the
// C ABI (gs_hip.h).  Same names, argument meaning and error behaviour as the Rust items:
//
//   gs::Parameters        data/src/parameters.rs:13-33, Default :72-83
//   gs::HipConcentration  Concentration trait, data/src/concentration/mod.rs:198-296
//   gs::Evolving/Species  data/src/concentration/mod.rs:17-187 (Species::new :36-59)
//   gs::Simulation        SimulateBase/SimulateCreate/Simulate, compute/shared/src/lib.rs:19-58
//
// Errors: a non-zero gs_status becomes gs::HipError (the Rust shim's HipError); programming
// errors that panic in the reference (shape mismatch in write_scalar_view,
// concentration/mod.rs:291-295) throw std::logic_error.
#pragma once
#include "gs_hip.h"

#include <array>
#include <cstddef>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace gs {

using Precision = float; // data/src/lib.rs:11

struct HipError : std::runtime_error {
    int32_t code;
    HipError(int32_t c, const std::string &m)
        : std::runtime_error("gs_hip error " + std::to_string(c) + ": " + m), code(c) {}
};

inline void check(int32_t status)
{
    if (status != GS_OK) throw HipError(status, gs_last_error());
}

struct Parameters {
    std::array<std::array<Precision, 3>, 3> weights{{{0.25f, 0.5f, 0.25f}, {0.5f, 0.0f, 0.5f}, {0.25f, 0.5f, 0.25f}}};
    Precision diffusion_rate_u = 0.1f, diffusion_rate_v = 0.05f;
    Precision feed_rate = 0.014f, kill_rate = 0.054f, time_step = 1.0f;

    // the reference's compile-time stencil choices (data/Cargo.toml:28-58, parameters.rs:91-122) as
    // run-time values: "oono-puri" (default), "5points", "patrakarttunen", "pretty"
    static Parameters with_stencil(const std::string &name)
    {
        Parameters p;
        if (name == "5points") p.weights = {{{0.f, 1.f, 0.f}, {1.f, 0.f, 1.f}, {0.f, 1.f, 0.f}}};
        else if (name == "patrakarttunen") {
            const Precision a = 1.0f / 6.0f, b = 4.0f / 6.0f;
            p.weights = {{{a, b, a}, {b, 0.f, b}, {a, b, a}}};
        } else if (name == "pretty") p.weights = {{{1.f, 1.f, 1.f}, {1.f, 1.f, 1.f}, {1.f, 1.f, 1.f}}};
        else if (name != "oono-puri") throw std::invalid_argument("unknown stencil " + name);
        return p;
    }

    gs_params to_c() const
    {
        gs_params p;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) p.w[i][j] = weights[i][j];
        p.du = diffusion_rate_u;
        p.dv = diffusion_rate_v;
        p.feed = feed_rate;
        p.kill = kill_rate;
        p.dt = time_step;
        return p;
    }
};

// Backend CLI arguments (every field defaulted, compute/shared/src/lib.rs:20-25).
struct HipArgs {
    std::vector<int32_t> devices{0};
    int32_t math = GS_MATH_STRICT, kernel = GS_KERNEL_AUTO, rows_per_block = 0, fuse_steps = 0, cols_per_lane = 0;
    int32_t boundary = GS_BOUNDARY_CLIPPED, no_tune = 0, share_taps = 0, general_kernels = 0;
    // not a gs_options field: the most extra blocks gs_fields_place may draw when make_species places a Species by
    // measurement -- every Species of >= 2^26 cells on a context with one slab (0 = planes as hipMalloc hands them out)
    int32_t place_candidates = 12;
};

// Concentration::Context: owner of the gs_ctx.
class HipContext {
  public:
    HipContext(const Parameters &params, const HipArgs &args)
    {
        gs_params p = params.to_c();
        gs_options o;
        gs_default_options(&o);
        o.math = args.math;
        o.kernel = args.kernel;
        o.rows_per_block = args.rows_per_block;
        o.fuse_steps = args.fuse_steps;
        o.cols_per_lane = args.cols_per_lane;
        o.boundary = args.boundary;
        o.no_tune = args.no_tune;
        o.share_taps = args.share_taps;
        o.general_kernels = args.general_kernels;
        check(gs_ctx_create(&ctx_, &p, &o, args.devices.data(), (int32_t)args.devices.size(), 0, 1, nullptr));
        place_candidates_ = args.devices.size() == 1 ? args.place_candidates : 0;
    }
    ~HipContext() { gs_ctx_destroy(ctx_); }
    int32_t place_candidates() const { return place_candidates_; }
    HipContext(const HipContext &) = delete;
    HipContext &operator=(const HipContext &) = delete;
    gs_ctx *get() const { return ctx_; }
    void sync() const { check(gs_sync(ctx_)); }
    // waits for the asynchronous downloads enqueued so far (not for later steps)
    void download_wait() const { check(gs_download_wait(ctx_)); }
    // ... for all but the newest `in_flight` (0 or 1): two images on their way, the PCIe link never idles between them
    void download_wait_but(int32_t in_flight) const { check(gs_download_wait_but(ctx_, in_flight)); }
    // counters of the context (passes, steps, launches, blocking ghost refreshes) and, on slab chains, the
    // halo-stream / interior times of the passes timed with set_pass_timing (gs_ctx_stats)
    gs_stats stats() const
    {
        gs_stats st;
        check(gs_ctx_stats(ctx_, &st));
        return st;
    }
    void set_pass_timing(int32_t passes) const { check(gs_ctx_set_pass_timing(ctx_, passes)); }


  private:
    gs_ctx *ctx_ = nullptr;
    int32_t place_candidates_ = 0;
};
using Context = std::shared_ptr<HipContext>;

using Shape = std::array<std::size_t, 2>;
using Range = std::pair<std::size_t, std::size_t>; // half-open, like Rust's Range<usize>

class HipConcentration {
  public:
    static HipConcentration default_(Context &c, Shape s) { return HipConcentration(c, s); }
    static HipConcentration zeros(Context &c, Shape s) { return HipConcentration(c, s); }
    static HipConcentration ones(Context &c, Shape s)
    {
        HipConcentration x(c, s);
        check(gs_field_fill(c->get(), x.f_, 1.0f));
        return x;
    }
    HipConcentration(HipConcentration &&o) noexcept : ctx_(std::move(o.ctx_)), f_(o.f_), shape_(o.shape_) { o.f_ = nullptr; }
    HipConcentration &operator=(HipConcentration &&o) noexcept
    {
        std::swap(ctx_, o.ctx_);
        std::swap(f_, o.f_);
        std::swap(shape_, o.shape_);
        return *this;
    }
    ~HipConcentration()
    {
        if (f_) gs_field_destroy(ctx_->get(), f_);
    }
    Shape shape() const { return shape_; }
    Shape raw_shape() const
    {
        uint64_t r = 0, p = 0;
        check(gs_field_raw_shape(f_, &r, &p));
        return {(std::size_t)r, (std::size_t)p};
    }
    void fill_slice(Context &c, std::array<Range, 2> slice, Precision value)
    {
        check(gs_field_fill_slice(c->get(), f_, slice[0].first, slice[0].second, slice[1].first,
                                  slice[1].second, value));
    }
    void finalize(Context &c) { check(gs_field_finalize(c->get(), f_)); }
    // make_scalar_view: owned dense copy [rows * cols]
    std::vector<Precision> make_scalar_view(Context &c)
    {
        std::vector<Precision> out(shape_[0] * shape_[1]);
        check(gs_field_download(c->get(), f_, out.data()));
        return out;
    }
    // write_scalar_view: the target must have exactly this table's shape (validate_write)
    void write_scalar_view(Context &c, Precision *target, Shape target_shape)
    {
        if (target_shape != shape_) throw std::logic_error("write_scalar_view: target shape mismatch");
        check(gs_field_download(c->get(), f_, target));
    }
    // write_scalar_view_after (data/src/concentration/gpu/image/mod.rs:196-206): enqueue the
    // download behind the steps already enqueued; `target` (ideally from PinnedImage) is valid
    // after HipContext::download_wait()
    void write_scalar_view_after(Context &c, Precision *target, Shape target_shape)
    {
        if (target_shape != shape_) throw std::logic_error("write_scalar_view_after: target shape mismatch");
        check(gs_field_download_async(c->get(), f_, target));
    }
    gs_field *raw() const { return f_; }
    // a zero-copy producer wrote cells through gs_field_device_ptr: ghost rows of neighbouring slabs are stale
    void mark_written(Context &c) { check(gs_field_mark_written(c->get(), f_)); }

  private:
    HipConcentration(Context &c, Shape s) : ctx_(c), shape_(s)
    {
        check(gs_field_create(c->get(), &f_, s[0], s[1]));
    }
    Context ctx_;
    gs_field *f_ = nullptr;
    Shape shape_{};
};

// Page-locked host image for overlapped downloads (gs_host_alloc / gs_host_free).
class PinnedImage {
  public:
    explicit PinnedImage(Shape s) : shape_(s)
    {
        void *p = nullptr;
        check(gs_host_alloc(&p, (uint64_t)s[0] * s[1] * sizeof(Precision)));
        data_ = static_cast<Precision *>(p);
    }
    ~PinnedImage() { gs_host_free(data_); }
    PinnedImage(const PinnedImage &) = delete;
    PinnedImage &operator=(const PinnedImage &) = delete;
    Precision *data() { return data_; }
    const Precision *data() const { return data_; }
    Shape shape() const { return shape_; }

  private:
    Precision *data_ = nullptr;
    Shape shape_;
};

// Pair of concentrations, slot 0 = input, slot 1 = output (concentration/mod.rs:140-187).
class Evolving {
  public:
    static Evolving zeros_out(Context &c, Shape s)
    {
        return Evolving(HipConcentration::default_(c, s), HipConcentration::zeros(c, s));
    }
    static Evolving ones_out(Context &c, Shape s)
    {
        return Evolving(HipConcentration::default_(c, s), HipConcentration::ones(c, s));
    }
    HipConcentration &in() { return pair_[0]; }
    HipConcentration &out() { return pair_[1]; }
    Shape shape() const { return pair_[0].shape(); }
    void flip(Context &c)
    {
        pair_[1].finalize(c);
        std::swap(pair_[0], pair_[1]);
    }
    void swap_slots() { std::swap(pair_[0], pair_[1]); }

  private:
    Evolving(HipConcentration a, HipConcentration b) : pair_{std::move(a), std::move(b)} {}
    std::array<HipConcentration, 2> pair_;
};

class Species {
  public:
    // Species::new (concentration/mod.rs:36-59)
    static Species new_(Context context, Shape shape)
    {
        Evolving u = Evolving::ones_out(context, shape);
        Evolving v = Evolving::zeros_out(context, shape);
        const std::size_t num_range[2] = {7, 8}, frac = 16, row_shift = 4;
        std::array<Range, 2> center;
        for (int i = 0; i < 2; ++i) {
            const std::size_t shift = (i == 0) ? row_shift : 0;
            auto edge = [&](int j) {
                const std::size_t x = shape[i] * num_range[j] / frac;
                return x > shift ? x - shift : 0; // saturating_sub
            };
            center[i] = {edge(0), edge(1)};
        }
        u.out().fill_slice(context, center, 0.0f);
        v.out().fill_slice(context, center, 1.0f);
        Species s(std::move(context), std::move(u), std::move(v));
        s.flip();
        return s;
    }
    Context &context() { return context_; }
    Shape shape() const { return u_.shape(); }
    // placement by measurement (gs_fields_place; not in the reference): U's and V's planes are given blocks of different
    // physical regions of HBM (at most `candidates` extra blocks drawn); planes that move keep their contents
    void place(int32_t candidates)
    {
        gs_field *planes[4] = {u_.in().raw(), v_.in().raw(), u_.out().raw(), v_.out().raw()};
        check(gs_fields_place(context_->get(), planes, candidates, nullptr, nullptr));
    }
    void flip()
    {
        u_.flip(context_);
        v_.flip(context_);
    }
    Evolving &u() { return u_; }
    Evolving &v() { return v_; }
    std::vector<Precision> make_result_view() { return v_.in().make_scalar_view(context_); }
    void write_result_view(Precision *target, Shape target_shape)
    {
        v_.in().write_scalar_view(context_, target, target_shape);
    }
    void write_result_view_after(PinnedImage &image)
    {
        v_.in().write_scalar_view_after(context_, image.data(), image.shape());
    }

  private:
    Species(Context c, Evolving u, Evolving v) : context_(std::move(c)), u_(std::move(u)), v_(std::move(v)) {}
    Context context_;
    Evolving u_, v_;
};

class Simulation {
  public:
    using CliArgs = HipArgs;
    // SimulateCreate::new
    static Simulation new_(const Parameters &params, const HipArgs &args = HipArgs())
    {
        return Simulation(std::make_shared<HipContext>(params, args));
    }
    // SimulateBase::make_species
    Species make_species(Shape shape) const
    {
        Species s = Species::new_(context_, shape);
        // planes of >= 256 MiB: below, they largely stay in the last-level cache and where they lie does not show
        if (context_->place_candidates() > 0 && (uint64_t)shape[0] * (uint64_t)shape[1] >= (1ull << 26))
            s.place(context_->place_candidates());
        return s;
    }
    // Simulate::perform_steps: the steps are DONE on return (as in every backend of the reference:
    // compute/shared/src/gpu/mod.rs:77-91 ends in a fence wait); results end up in the input slots
    void perform_steps(Species &species, std::size_t steps) const
    {
        prepare_steps(species, steps);
        check(gs_sync(context_->get()));
    }
    // SimulateGpu::prepare_steps (compute/shared/src/gpu/mod.rs:70-75): enqueue only; a download or
    // gs_sync waits.  HIP streams order the work, so no future object is passed along.
    void prepare_steps(Species &species, std::size_t steps) const
    {
        int32_t slot = 0;
        check(gs_run(context_->get(), species.u().in().raw(), species.v().in().raw(), species.u().out().raw(),
                     species.v().out().raw(), steps, &slot));
        if (slot == 1) {
            species.u().swap_slots();
            species.v().swap_slots();
        }
    }
    // SimulateStep::perform_step (compute/shared/src/cpu.rs:21-28): one gs_step, then flip
    void perform_step(Species &species) const
    {
        check(gs_step(context_->get(), species.u().in().raw(), species.v().in().raw(), species.u().out().raw(),
                      species.v().out().raw()));
        species.flip();
    }
    const Context &context() const { return context_; }

  private:
    explicit Simulation(Context c) : context_(std::move(c)) {}
    Context context_;
};

} // namespace gs
