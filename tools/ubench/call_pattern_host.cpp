// The reference's driver loop (simulate/src/main.rs:99-115: prepare_steps(32), the V image behind it, two images in flight)
// from a COMPILED host, through the C++ mirror of the reference interface (include/grayscott_hip.hpp): what a Rust caller's
// loop costs per call, without the Python harness's ctypes overhead (tools/call_pattern.py measures the same loop from Python).
//   g++ -std=c++17 -O2 -I include tools/ubench/call_pattern_host.cpp -o tools/ubench/call_pattern_host \
//       -L grayscott_amd -lgs_hip -Wl,-rpath,$PWD/grayscott_amd -Wl,-rpath,/opt/rocm/lib
//   tools/ubench/call_pattern_host [rows=1080] [cols=1920] [steps per image=32] [images=400]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "grayscott_hip.hpp"

using namespace gs;

int main(int argc, char **argv)
{
    const std::size_t rows = argc > 1 ? std::atoi(argv[1]) : 1080, cols = argc > 2 ? std::atoi(argv[2]) : 1920;
    const std::size_t n = argc > 3 ? std::atoi(argv[3]) : 32;
    const int images = argc > 4 ? std::atoi(argv[4]) : 400;
    try {
        Simulation sim = Simulation::new_(Parameters());
        Species sp = sim.make_species({rows, cols});
        sim.perform_steps(sp, 4000); // on-line tuning done (calls of < 32 steps run the marching kernel)
        std::vector<std::unique_ptr<PinnedImage>> pinned;
        for (int i = 0; i < 3; ++i) pinned.emplace_back(new PinnedImage({rows, cols}));
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        double best[2] = {0, 0};
        for (int rep = 0; rep < 4; ++rep) {
            // steps only, one synchronisation per call (Simulate::perform_steps)
            double t0 = now();
            for (int i = 0; i < images; ++i) sim.perform_steps(sp, n);
            const double only = rows * cols * (double)n * images / (now() - t0) / 1e6;
            // the async-gpu loop: two images in flight
            t0 = now();
            for (int i = 0; i < images; ++i) {
                sim.prepare_steps(sp, n);
                sp.write_result_view_after(*pinned[i % 3]);
                if (i) sp.context()->download_wait_but(1);
            }
            sp.context()->download_wait();
            sp.context()->sync();
            const double overlapped = rows * cols * (double)n * images / (now() - t0) / 1e6;
            if (only > best[0]) best[0] = only;
            if (overlapped > best[1]) best[1] = overlapped;
        }
        std::printf("{\"host\": \"C++ (include/grayscott_hip.hpp)\", \"rows\": %zu, \"cols\": %zu, \"steps_per_image\": %zu, \"images\": %d, "
                    "\"steps_only_Mcells_steps_per_s\": %.0f, \"overlapped_image_per_call_Mcells_steps_per_s\": %.0f}\n",
                    rows, cols, n, images, best[0], best[1]);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
