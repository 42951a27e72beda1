// rayz — command-line driver with the reference's interface: `rayz <img_w> [out.ppm]`
// (main, src/rayz.zig:12-43 of jlucier/rayz): build the randomBouncing scene, time render(), print the
// rate line to stderr, write a P3 PPM to the file or to stdout.
//
// Extras the reference does not have (environment variables, so the positional interface stays the
// reference's): RAYZ_SPP, RAYZ_BOUNCES, RAYZ_SEED, RAYZ_GRID (half-width of the sphere grid, 11 in the
// reference), RAYZ_PRECISION=f32|f64, RAYZ_TRAVERSAL=linear|bvh|auto (default auto), RAYZ_DEVICES=0,1,...
// (render on several GPUs of the node through rayz_hip_render_multi; default: device 0).
#include "rayz.hpp"

#include <chrono>
#include <cstdlib>

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s <img_w> [out.ppm]\n", argv[0]); // the reference panics on `.?`, src/rayz.zig:16
        return 2;
    }
    char* end = nullptr;
    const unsigned long img_w = std::strtoul(argv[1], &end, 10);
    if (!img_w || (end && *end)) {
        std::fprintf(stderr, "error: InvalidCharacter\n");
        return 1;
    }
    const char* e;
    uint64_t seed = 0;
    const bool has_seed = (e = std::getenv("RAYZ_SEED")) != nullptr;
    if (has_seed) seed = std::strtoull(e, nullptr, 10);
    int grid = 11;
    if ((e = std::getenv("RAYZ_GRID"))) grid = std::atoi(e);

    rayz::Tracer tracer = rayz::randomBouncing(img_w, -grid, grid, has_seed ? &seed : nullptr);
    if ((e = std::getenv("RAYZ_SPP"))) tracer.samples_per_px = std::strtoul(e, nullptr, 10);
    if ((e = std::getenv("RAYZ_BOUNCES"))) tracer.max_bounces = std::strtoul(e, nullptr, 10);
    if ((e = std::getenv("RAYZ_PRECISION")) && std::string(e) == "f64") tracer.gpu.precision = RAYZ_PRECISION_F64;
    if ((e = std::getenv("RAYZ_TRAVERSAL")))
        tracer.gpu.traversal = std::string(e) == "bvh" ? RAYZ_TRAVERSAL_BVH : std::string(e) == "linear" ? RAYZ_TRAVERSAL_LINEAR : RAYZ_TRAVERSAL_AUTO;

    if ((e = std::getenv("RAYZ_DEVICES"))) { // "0,1,2": anything else is an error, not a silent fall-back to device 0
        for (const char* q = e;;) {
            char* next = nullptr;
            const long d = std::strtol(q, &next, 10);
            if (next == q || d < 0 || d >= RAYZ_MAX_DEVICES || (*next != ',' && *next != '\0')) {
                std::fprintf(stderr, "error: RAYZ_DEVICES=\"%s\" is not a comma-separated list of device ordinals\n", e);
                return 2;
            }
            tracer.gpu.devices.push_back((int)d);
            if (*next == '\0') break;
            q = next + 1;
        }
    }

    if (rayz_hip_init(tracer.gpu.devices.empty() ? 0 : tracer.gpu.devices[0]) != RAYZ_OK) {
        std::fprintf(stderr, "error: GpuRenderFailed: %s\n", rayz_hip_last_error());
        return 1;
    }
    const auto st = std::chrono::steady_clock::now();
    double rays_traced;
    try {
        rays_traced = (double)tracer.render();
    } catch (const rayz::GpuRenderFailed& ex) {
        std::fprintf(stderr, "error: GpuRenderFailed: %s\n", ex.what());
        return 1;
    }
    const double durr = std::chrono::duration<double>(std::chrono::steady_clock::now() - st).count();
    std::fprintf(stderr, "Finished render (%.2fs): %.2f rps and %.2f us per ray\n", durr, rays_traced / durr,
                 1e6 * durr / rays_traced); // src/rayz.zig:30-34

    if (argc > 2) {
        FILE* f = std::fopen(argv[2], "w");
        if (!f) {
            std::fprintf(stderr, "error: cannot create %s\n", argv[2]);
            return 1;
        }
        tracer.img.writePPM(f);
        std::fclose(f);
    } else {
        tracer.img.writePPM(stdout);
    }
    rayz_hip_shutdown();
    return 0;
}
