// sanitize_main.cpp — CPU-only AddressSanitizer / UBSan run over the host-side C++ of the repository:
// the oracle (modes A and B, flat list and BVH, spheres and triangles), the host mirror (scene generators,
// flatten, PPM writer) and the product's BVH builder.  Built and run by tests/test_sanitizers.py.
#include "../oracle/rayz_oracle.cpp"
#include "../rayz_amd/csrc/bvh_build.hpp"
#include "../rayz_amd/host/rayz.hpp"

#include <cstdio>

// The host mirror's render() calls the HIP library; this CPU build has none.
extern "C" int rayz_hip_render(const RayzSceneDesc*, const RayzCameraDesc*, const RayzRenderParams*, float*, RayzRenderStats*) { return RAYZ_ERR_NO_DEVICE; }
extern "C" int rayz_hip_render_f64(const RayzSceneDesc*, const RayzCameraDesc*, const RayzRenderParams*, double*, RayzRenderStats*) { return RAYZ_ERR_NO_DEVICE; }
extern "C" int rayz_hip_render_multi(const int*, int, const RayzSceneDesc*, const RayzCameraDesc*, const RayzRenderParams*, float*, RayzRenderStats*) { return RAYZ_ERR_NO_DEVICE; }
extern "C" int rayz_hip_render_multi_f64(const int*, int, const RayzSceneDesc*, const RayzCameraDesc*, const RayzRenderParams*, double*, RayzRenderStats*) { return RAYZ_ERR_NO_DEVICE; }
extern "C" int rayz_hip_multi_create(const int*, int, const RayzSceneDesc*, uint32_t, RayzMulti** out) { if (out) *out = nullptr; return RAYZ_ERR_NO_DEVICE; }
extern "C" int rayz_hip_multi_destroy(RayzMulti*) { return RAYZ_OK; }
extern "C" int rayz_hip_multi_render(RayzMulti*, const RayzCameraDesc*, const RayzRenderParams*, float*, RayzRenderStats*) { return RAYZ_ERR_NO_DEVICE; }
extern "C" int rayz_hip_multi_render_f64(RayzMulti*, const RayzCameraDesc*, const RayzRenderParams*, double*, RayzRenderStats*) { return RAYZ_ERR_NO_DEVICE; }
extern "C" const char* rayz_hip_last_error(void) { return "no device in the sanitizer build"; }

static int run(rayz::Tracer& t, const char* name) {
    t.samples_per_px = 3;
    t.max_bounces = 6;
    const rayz::Tracer::Flat f = t.flatten();
    const RayzSceneDesc sd = f.desc();
    const RayzCameraDesc cd = t.camera.desc();
    RayzRenderParams p = t.params(7);
    p.chunk_spp = 2;
    std::vector<float> a((size_t)p.width * p.height * 3), b(a.size());
    std::vector<double> c(a.size());
    RayzRenderStats st{};
    int rc = rayz_oracle_render_b_f32(&sd, &cd, &p, nullptr, 0, a.data(), &st, 2);
    p.traversal = RAYZ_TRAVERSAL_BVH;
    rc |= rayz_oracle_render_b_f32(&sd, &cd, &p, nullptr, 0, b.data(), &st, 2);
    p.precision = RAYZ_PRECISION_F64;
    rc |= rayz_oracle_render_b_f64(&sd, &cd, &p, nullptr, 0, c.data(), &st, 2);
    uint64_t rng[4];
    std::memcpy(rng, t.rng.s, 32);
    p.tmin = 1e-10;
    rc |= rayz_oracle_render_a(&sd, &cd, &p, 0, p.height, rng, 0, c.data(), nullptr, &st);
    rc |= rayz_oracle_render_a(&sd, &cd, &p, 0, p.height, rng, 1, c.data(), nullptr, &st);
    const rayz_bvh::FlatBvh tree = rayz_bvh::build(f.spheres, f.triangles);
    size_t diff = 0;
    for (size_t i = 0; i < a.size(); ++i) diff += a[i] != b[i];
    std::printf("%s: rc %d, %zu nodes, depth %u, flat-vs-bvh differing values %zu\n", name, rc, tree.nodes.size(), tree.depth, diff);
    bool threw = false;
    try {
        t.render();
    } catch (const rayz::GpuRenderFailed&) {
        threw = true;
    }
    return rc != 0 || !threw || diff > a.size() / 100;
}

int main() {
    const uint64_t seed = 5;
    int bad = 0;
    rayz::Tracer t1 = rayz::randomBouncing(48, -4, 4, &seed);
    bad |= run(t1, "randomBouncing");
    rayz::Tracer t2 = rayz::threeSpheres(40, &seed);
    bad |= run(t2, "threeSpheres");
    rayz::Tracer t3 = rayz::triangleMesh(40, 6, &seed);
    bad |= run(t3, "triangleMesh");
    rayz::Tracer t4 = rayz::Tracer::init(24, 30, 1, 0, rayz::V3{0, 0, 2}, rayz::V3{}, rayz::V3::y_hat(), &seed);
    bad |= run(t4, "empty pool");
    { // which hittables the walked tree keeps OUT of it (bvh_build.hpp, peel_oversized): an outlier only
        rayz::Tracer rb = rayz::randomBouncing(48, -11, 11, &seed);
        const rayz::Tracer::Flat fr = rb.flatten();
        const rayz_bvh::FlatBvh peeled = rayz_bvh::build(fr.spheres, fr.triangles, true);
        const bool ground_only = peeled.big.size() == 1 && fr.spheres[peeled.big[0]].radius == 1000.0; // src/rayz.zig:58-74
        // a compact cluster of 40 similar spheres, each a good part of the cluster's extent: nothing is an outlier
        std::vector<RayzSphere> cl(40);
        for (size_t i = 0; i < cl.size(); ++i) {
            cl[i] = RayzSphere{};
            cl[i].center[0] = 0.9 * (double)(i % 4), cl[i].center[1] = 0.9 * (double)((i / 4) % 4), cl[i].center[2] = 0.9 * (double)(i / 16);
            cl[i].radius = 0.5 + 0.01 * (double)i;
        }
        const rayz_bvh::FlatBvh cluster = rayz_bvh::build(cl, {}, true);
        std::printf("peel: randomBouncing keeps %zu out (ground: %d), compact cluster keeps %zu out\n", peeled.big.size(), (int)ground_only,
                    cluster.big.size());
        bad |= !ground_only || !cluster.big.empty() || cluster.order.size() != cl.size();
    }
    { // the tree the GPU walks (surface-area split): every hittable in exactly one leaf of <= 2, skip links consistent, depth
      // within the budget the LDS stacks are sized for — small pools, the 10k grid (sweep) and a pool above kSweepMax (bins)
        for (int g : {2, 11, 50, 40, -1, -2}) {
            rayz::Tracer rb = g == 40 ? rayz::triangleMesh(32, 60, &seed) : rayz::randomBouncing(32, g < 0 ? -2 : -g, g < 0 ? 2 : g, &seed);
            rayz::Tracer::Flat fr = rb.flatten();
            if (g < 0) { // degenerate pools: 5,000 coincident spheres (no split position is better than another: halved), and
                         // 3,000 on a line with one far outlier (every SAH split is lopsided: the depth budget must hold)
                fr.spheres.assign(g == -1 ? 5000 : 3000, fr.spheres[1]);
                if (g == -2) {
                    for (size_t i = 0; i < fr.spheres.size(); ++i) fr.spheres[i].center[0] = std::pow(1.01, (double)i);
                    fr.spheres.back().center[0] = 1e15;
                }
                fr.triangles.clear();
            }
            const rayz_bvh::FlatBvh tr = rayz_bvh::build(fr.spheres, fr.triangles, true, true);
            const size_t n = fr.spheres.size() + fr.triangles.size();
            std::vector<int> seen(n, 0);
            for (uint32_t b : tr.big) seen[b]++;
            bool ok = tr.depth <= rayz_bvh::detail::levelsFor(n - tr.big.size()) + rayz_bvh::detail::kSahExtraDepth;
            for (size_t i = 0; i < tr.nodes.size(); ++i) {
                const rayz_bvh::FlatNode& nd = tr.nodes[i];
                ok = ok && nd.skip > i && nd.skip <= tr.nodes.size() && nd.count <= 2;
                for (uint32_t k = 0; k < nd.count; ++k) {
                    const uint32_t h = tr.order[nd.first + k];
                    seen[h]++;
                    const rayz_bvh::Box hb = h < fr.spheres.size() ? rayz_bvh::sphereBox(fr.spheres[h]) : rayz_bvh::triangleBox(fr.triangles[h - fr.spheres.size()]);
                    for (int a = 0; a < 3; ++a) ok = ok && hb.lo[a] >= nd.box.lo[a] && hb.hi[a] <= nd.box.hi[a];
                }
                if (nd.count == 0) { // children inside the parent, right child where the left subtree ends
                    const rayz_bvh::FlatNode &l = tr.nodes[i + 1], &r = tr.nodes[l.skip];
                    ok = ok && l.skip < nd.skip && r.skip == nd.skip;
                    for (int a = 0; a < 3; ++a) ok = ok && std::fmin(l.box.lo[a], r.box.lo[a]) == nd.box.lo[a] && std::fmax(l.box.hi[a], r.box.hi[a]) == nd.box.hi[a];
                }
            }
            for (int c : seen) ok = ok && c == 1;
            std::printf("sah tree over %zu hittables: %zu nodes, depth %u, %s\n", n, tr.nodes.size(), tr.depth, ok ? "ok" : "BROKEN");
            bad |= !ok;
        }
    }
    { // the 16-bit plane indices of the walked tree's node records (bvh_build.hpp: PlaneGrid): every quantised box contains
      // the padded box it stands for, over grids of very different position and extent
        uint64_t st = 88172645463325252ull;
        auto rnd = [&]() { st ^= st << 13, st ^= st >> 7, st ^= st << 17; return (double)(st >> 11) * (1.0 / 9007199254740992.0); };
        size_t checked = 0, broken = 0;
        for (int g = 0; g < 200; ++g) {
            const double scale = std::pow(10.0, -3.0 + 9.0 * rnd()), shift = (rnd() - 0.5) * scale * std::pow(10.0, 3.0 * rnd());
            double lo[3], hi[3];
            for (int k = 0; k < 3; ++k) lo[k] = shift - scale * rnd(), hi[k] = shift + scale * rnd();
            const double pad = 1e-6 * scale * rnd();
            const rayz_bvh::PlaneGrid grid = rayz_bvh::PlaneGrid::over(lo, hi, 2.0 * pad);
            for (int b = 0; b < 500; ++b) {
                rayz_bvh::Box bx;
                for (int k = 0; k < 3; ++k) {
                    const double a = lo[k] + (hi[k] - lo[k]) * rnd(), c = lo[k] + (hi[k] - lo[k]) * rnd();
                    bx.lo[k] = std::fmin(a, c), bx.hi[k] = b % 7 == 0 ? bx.lo[k] : std::fmax(a, c); // (flat boxes too)
                }
                uint32_t w[3];
                grid.quantize(bx, pad, w);
                for (int k = 0; k < 3; ++k) {
                    const uint32_t il = w[k] & 0xffffu, ih = w[k] >> 16;
                    ++checked;
                    broken += !(grid.plane(k, il) <= bx.lo[k] - pad && grid.plane(k, ih) >= bx.hi[k] + pad && il <= ih);
                }
            }
        }
        std::printf("plane grid: %zu quantised planes pairs checked, %zu not conservative\n", checked, broken);
        bad |= broken != 0;
    }
    FILE* f = std::fopen("/dev/null", "w");
    t1.img.writePPM(f);
    std::fclose(f);
    std::printf(bad ? "FAILED\n" : "sanitizer run ok\n");
    return bad;
}
