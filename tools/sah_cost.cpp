// sah_cost.cpp — CPU only: the surface-area cost (expected box / leaf tests of a long random ray) of the walked tree under both
// split rules, with depth and build time:  g++ -O2 -std=c++17 -Iinclude tools/sah_cost.cpp -o /tmp/sah_cost && /tmp/sah_cost
#include "../rayz_amd/csrc/bvh_build.hpp"
#include "../rayz_amd/host/rayz.hpp"
#include <chrono>
#include <cstdio>
extern "C" int rayz_hip_render(const RayzSceneDesc*, const RayzCameraDesc*, const RayzRenderParams*, float*, RayzRenderStats*) { return 1; }
extern "C" int rayz_hip_render_f64(const RayzSceneDesc*, const RayzCameraDesc*, const RayzRenderParams*, double*, RayzRenderStats*) { return 1; }
extern "C" int rayz_hip_render_multi(const int*, int, const RayzSceneDesc*, const RayzCameraDesc*, const RayzRenderParams*, float*, RayzRenderStats*) { return 1; }
extern "C" int rayz_hip_render_multi_f64(const int*, int, const RayzSceneDesc*, const RayzCameraDesc*, const RayzRenderParams*, double*, RayzRenderStats*) { return 1; }
extern "C" int rayz_hip_multi_create(const int*, int, const RayzSceneDesc*, uint32_t, RayzMulti** out) { return 1; }
extern "C" int rayz_hip_multi_destroy(RayzMulti*) { return 0; }
extern "C" int rayz_hip_multi_render(RayzMulti*, const RayzCameraDesc*, const RayzRenderParams*, float*, RayzRenderStats*) { return 1; }
extern "C" int rayz_hip_multi_render_f64(RayzMulti*, const RayzCameraDesc*, const RayzRenderParams*, double*, RayzRenderStats*) { return 1; }
extern "C" const char* rayz_hip_last_error(void) { return ""; }
static void report(const char* name, rayz::Tracer& t) {
    const rayz::Tracer::Flat f = t.flatten();
    for (int sah = 0; sah < 2; ++sah) {
        auto t0 = std::chrono::steady_clock::now();
        const rayz_bvh::FlatBvh b = rayz_bvh::build(f.spheres, f.triangles, true, sah);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        double inner = 0, leaf = 0;
        const double root = rayz_bvh::detail::halfArea(b.nodes[0].box);
        for (const auto& n : b.nodes) (n.count ? leaf : inner) += rayz_bvh::detail::halfArea(n.box) / root * (n.count ? n.count : 1);
        std::printf("%-12s %s: %zu nodes depth %u big %zu  sum area inner %.2f  leaf-tests %.2f  build %.0f ms\n", name, sah ? "sah   " : "median", b.nodes.size(), b.depth, b.big.size(), inner, leaf, ms);
    }
}
int main() {
    const uint64_t seed = 42;
    rayz::Tracer a = rayz::randomBouncing(192, -11, 11, &seed); report("config2", a);
    rayz::Tracer b = rayz::randomBouncing(192, -50, 50, &seed); report("config3", b);
    const uint64_t s1 = 1;
    rayz::Tracer c = rayz::triangleMesh(192, 224, &s1); report("config5", c);
}
