// Exercises include/grayscott_hip.hpp (the C++ mirror of the reference's backend interface)
// the way the reference's `simulate` binary drives a backend (simulate/src/main.rs:56-59,
// 113-115): Simulation::new -> make_species -> perform_steps -> write_result_view.
// Usage: host_mirror ROWS COLS STEPS OUT.bin   (writes U then V as raw f32)
// Built and run by tests/test_cpp_host_mirror.py; plain g++, links libgs_hip.so.
#include "grayscott_hip.hpp"

#include <cstdio>
#include <cstdlib>

int main(int argc, char **argv)
{
    if (argc != 5) {
        std::fprintf(stderr, "usage: %s rows cols steps out.bin\n", argv[0]);
        return 2;
    }
    const std::size_t rows = std::strtoull(argv[1], nullptr, 10), cols = std::strtoull(argv[2], nullptr, 10);
    const std::size_t steps = std::strtoull(argv[3], nullptr, 10);
    try {
        gs::HipArgs args;
        args.place_candidates = 2; // every Species placed by measurement (gs_fields_place), as a large run would ask for
        gs::Simulation sim = gs::Simulation::new_(gs::Parameters(), args);
        gs::Species species = sim.make_species({rows, cols});
        // the driver's pattern (simulate/src/main.rs:99-106): steps, asynchronous image, more steps
        const std::size_t first = steps / 2;
        sim.perform_steps(species, first);
        species.place(3); // ... and again in the middle of the run: the planes keep their contents
        gs::PinnedImage image({rows, cols});
        species.write_result_view_after(image);
        if (steps - first > 0) sim.perform_steps(species, steps - first - 1);
        if (steps - first > 0) sim.perform_step(species);
        species.context()->download_wait();
        std::vector<float> v(rows * cols);
        species.write_result_view(v.data(), {rows, cols});
        std::vector<float> u = species.u().in().make_scalar_view(species.context());
        bool threw = false;
        try {
            species.write_result_view(v.data(), {rows, cols + 1}); // must be rejected
        } catch (const std::logic_error &) {
            threw = true;
        }
        if (!threw) return 3;
        // the library's own counters: every step requested was enqueued, in at least one pass
        const gs_stats st = species.context()->stats();
        if (st.steps != steps || (steps > 0 && st.passes == 0) || st.ghost_refreshes != 0) return 5;
        std::FILE *f = std::fopen(argv[4], "wb");
        if (!f) return 4;
        std::fwrite(u.data(), sizeof(float), u.size(), f);
        std::fwrite(v.data(), sizeof(float), v.size(), f);
        std::fwrite(image.data(), sizeof(float), rows * cols, f); // V after steps/2 steps
        std::fclose(f);
    } catch (const gs::HipError &e) {
        std::fprintf(stderr, "HipError: %s\n", e.what());
        return 10 - e.code; // GS_ERR_NO_DEVICE (-4) -> 14
    }
    return 0;
}
