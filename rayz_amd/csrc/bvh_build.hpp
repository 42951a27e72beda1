// bvh_build.hpp — host builds of the BVH, flattened (pre-order + skip links): the REFERENCE's tree, node for node (what
// rayz_hip_scene_bvh exports and tests check against the oracle's own build), and the tree the GPU walks — the same
// hittables and boxes, organised for fewer box tests (oversized hittables kept out, surface-area split: build()'s flags;
// the nearest hit does not depend on the tree).
//
// The reference's tree is the one `BVH.build` makes (src/hit.zig:130-161 of jlucier/rayz): hittables in pool order
// (src/ecs.zig:43-51) with `Sphere.boundingBox` boxes (src/geom.zig:24-31: union of the boxes at time 0 and
// 1), node box = union of its hittables' boxes, leaves of ≤ 2, otherwise a STABLE sort (std.mem.sort) of the
// node's range by `bbox.low[axis]` on the box's longest axis (`amax` tie rule, src/vec.zig:150-156) and a
// split at nobjs/2.  It is then laid out in depth-first pre-order (left subtree first — the order
// `findHit` visits, src/hit.zig:195-204) with a skip link per node: "hit → next node in memory, miss → skip"
// walks the same nodes as the reference's recursion, with no stack.
#pragma once

#include "../../include/rayz_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace rayz_bvh {

struct Box {
    double lo[3], hi[3];
    Box() {
        for (int k = 0; k < 3; ++k) lo[k] = std::numeric_limits<double>::infinity(), hi[k] = -lo[k];
    }
    void enclose(const Box& o) { // AABB.enclose, src/hit.zig:55-60
        for (int k = 0; k < 3; ++k) lo[k] = std::fmin(lo[k], o.lo[k]), hi[k] = std::fmax(hi[k], o.hi[k]);
    }
    int longestAxis() const { // src/hit.zig:62-64 + V3.amax
        const double x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
        if (x > y) return x > z ? 0 : 2;
        return y > z ? 1 : 2;
    }
};

inline Box sphereBox(const RayzSphere& s) { // Sphere.boundingBox, src/geom.zig:24-31
    Box b;
    for (int k = 0; k < 3; ++k) {
        const double o1 = s.center[k], o2 = s.center[k] + s.velocity[k] * 1.0; // center.at(1)
        const double a1 = std::fmin(o1 - s.radius, o1 + s.radius), b1 = std::fmax(o1 - s.radius, o1 + s.radius);
        const double a2 = std::fmin(o2 - s.radius, o2 + s.radius), b2 = std::fmax(o2 - s.radius, o2 + s.radius);
        b.lo[k] = std::fmin(a1, a2);
        b.hi[k] = std::fmax(b1, b2);
    }
    return b;
}

// Build-defined triangle hittable: vertex bounds padded by 1e-4 per side (a flat box can never pass the
// reference's strict `t1 > t0`, src/hit.zig:97).
inline Box triangleBox(const RayzTriangle& t) {
    Box b;
    for (int k = 0; k < 3; ++k) {
        b.lo[k] = std::fmin(std::fmin(t.v0[k], t.v1[k]), t.v2[k]) - 1e-4;
        b.hi[k] = std::fmax(std::fmax(t.v0[k], t.v1[k]), t.v2[k]) + 1e-4;
    }
    return b;
}

struct FlatNode {
    Box box;
    uint32_t skip;  // index of the next node when this subtree is done or culled (== n_nodes at the end)
    uint32_t first; // leaves: first entry in `order`
    uint32_t count; // leaves: 1 or 2; inner nodes: 0
};

struct FlatBvh {
    std::vector<FlatNode> nodes;  // depth-first pre-order
    std::vector<uint32_t> order;  // hittable indices after the in-place sorts, i.e. leaf order
    uint32_t depth = 0;
    std::vector<uint32_t> big;    // hittables kept OUT of the tree (build(.., peel_oversized = true) only)
};

namespace detail {
struct Item {
    Box box;
    uint32_t pool;
};
inline void build(std::vector<Item>& h, size_t si, size_t ei, FlatBvh& out, uint32_t depth) {
    const size_t me = out.nodes.size();
    out.nodes.push_back(FlatNode{});
    out.depth = std::max(out.depth, depth);
    Box bb;
    for (size_t i = si; i < ei; ++i) bb.enclose(h[i].box);
    const size_t nobjs = ei - si;
    if (nobjs <= 2) {
        out.nodes[me].box = bb;
        out.nodes[me].first = (uint32_t)si;
        out.nodes[me].count = (uint32_t)nobjs;
    } else {
        const int ax = bb.longestAxis();
        std::stable_sort(h.begin() + si, h.begin() + ei,
                         [ax](const Item& a, const Item& b) { return a.box.lo[ax] < b.box.lo[ax]; });
        const size_t mid = nobjs / 2 + si;
        out.nodes[me].box = bb;
        out.nodes[me].count = 0;
        build(h, si, mid, out, depth + 1);
        build(h, mid, ei, out, depth + 1);
    }
    out.nodes[me].skip = (uint32_t)out.nodes.size();
}

// The tree the GPU walks need not be the reference's — the nearest hit does not depend on how the hittables are
// organised — so it is built for FEWER BOX TESTS: greedy surface-area heuristic, cost(split) = area(L)·n(L) + area(R)·n(R)
// over the hittables sorted by box centre on each axis (every split position for up to kSweepMax hittables, 64 bins of
// the centre range above), leaves of ≤ 2 as the device's leaf descriptor holds them.  Deterministic (stable sorts, ties
// to the lower axis and the lower position).  `budget` = levels left: the per-lane stacks in LDS are sized by the depth, so
// a node that could no longer finish with halvings alone is split at its median instead (never deeper than the reference's
// tree + kSahExtraDepth).
#ifndef RAYZ_SAH_SWEEP_MAX
#define RAYZ_SAH_SWEEP_MAX 4096
#endif
constexpr size_t kSweepMax = RAYZ_SAH_SWEEP_MAX;
constexpr uint32_t kSahBins = 64, kSahExtraDepth = 4;
inline double halfArea(const Box& b) {
    const double x = b.hi[0] - b.lo[0], y = b.hi[1] - b.lo[1], z = b.hi[2] - b.lo[2];
    return x * y + y * z + z * x;
}
inline uint32_t levelsFor(size_t n) { // depth of a halving tree over n hittables with leaves of <= 2 (root = 1)
    uint32_t d = 1;
    while (n > 2) n = n - n / 2, ++d;
    return d;
}
inline void buildSah(std::vector<Item>& h, size_t si, size_t ei, FlatBvh& out, uint32_t depth, uint32_t budget) {
    const size_t me = out.nodes.size();
    out.nodes.push_back(FlatNode{});
    out.depth = std::max(out.depth, depth);
    Box bb;
    for (size_t i = si; i < ei; ++i) bb.enclose(h[i].box);
    const size_t n = ei - si;
    out.nodes[me].box = bb;
    if (n <= 2) {
        out.nodes[me].first = (uint32_t)si;
        out.nodes[me].count = (uint32_t)n;
        out.nodes[me].skip = (uint32_t)out.nodes.size();
        return;
    }
    auto centre = [](const Item& it, int ax) { return it.box.lo[ax] + it.box.hi[ax]; }; // (twice the centre)
    size_t mid = si + n / 2;
    int best_ax = -1;
    if (levelsFor(n) < budget) { // room for an uneven split
        double best = std::numeric_limits<double>::infinity();
        size_t best_pos = 0;
        std::vector<double> right_area;
        for (int ax = 0; ax < 3; ++ax) {
            double c0 = std::numeric_limits<double>::infinity(), c1 = -c0;
            for (size_t i = si; i < ei; ++i) c0 = std::fmin(c0, centre(h[i], ax)), c1 = std::fmax(c1, centre(h[i], ax));
            if (!(c1 > c0)) continue;
            if (n <= kSweepMax) {
                std::stable_sort(h.begin() + (ptrdiff_t)si, h.begin() + (ptrdiff_t)ei,
                                 [&](const Item& a, const Item& b) { return centre(a, ax) < centre(b, ax); });
                right_area.assign(n + 1, 0.0);
                Box acc;
                for (size_t i = n; i-- > 1;) acc.enclose(h[si + i].box), right_area[i] = halfArea(acc);
                acc = Box();
                for (size_t i = 1; i < n; ++i) { // split: [0, i) | [i, n)
                    acc.enclose(h[si + i - 1].box);
                    const double cost = halfArea(acc) * (double)i + right_area[i] * (double)(n - i);
                    if (cost < best) best = cost, best_ax = ax, best_pos = i;
                }
            } else {
                Box bins[kSahBins];
                size_t cnt[kSahBins] = {};
                const double scale = (double)kSahBins / (c1 - c0);
                auto bin_of = [&](const Item& it) { return std::min<size_t>(kSahBins - 1, (size_t)((centre(it, ax) - c0) * scale)); };
                for (size_t i = si; i < ei; ++i) bins[bin_of(h[i])].enclose(h[i].box), cnt[bin_of(h[i])]++;
                double ra[kSahBins + 1] = {};
                size_t rn[kSahBins + 1] = {};
                Box acc;
                for (size_t b = kSahBins; b-- > 1;) acc.enclose(bins[b]), rn[b] = rn[b + 1] + cnt[b], ra[b] = rn[b] ? halfArea(acc) : 0.0;
                acc = Box();
                size_t ln = 0;
                for (size_t b = 1; b < kSahBins; ++b) { // split: bins [0, b) | [b, kSahBins)
                    acc.enclose(bins[b - 1]), ln += cnt[b - 1];
                    if (ln == 0 || rn[b] == 0) continue;
                    const double cost = halfArea(acc) * (double)ln + ra[b] * (double)rn[b];
                    if (cost < best) best = cost, best_ax = ax, best_pos = kSweepMax + b; // (a bin, not a position)
                }
            }
        }
        if (best_ax >= 0) {
            const int ax = best_ax;
            if (best_pos > kSweepMax && n > kSweepMax) {
                double c0 = std::numeric_limits<double>::infinity(), c1 = -c0;
                for (size_t i = si; i < ei; ++i) c0 = std::fmin(c0, centre(h[i], ax)), c1 = std::fmax(c1, centre(h[i], ax));
                const double scale = (double)kSahBins / (c1 - c0);
                const size_t b = best_pos - kSweepMax;
                mid = (size_t)(std::stable_partition(h.begin() + (ptrdiff_t)si, h.begin() + (ptrdiff_t)ei,
                                                     [&](const Item& it) {
                                                         return std::min<size_t>(kSahBins - 1, (size_t)((centre(it, ax) - c0) * scale)) < b;
                                                     }) -
                               h.begin());
            } else {
                std::stable_sort(h.begin() + (ptrdiff_t)si, h.begin() + (ptrdiff_t)ei,
                                 [&](const Item& a, const Item& b) { return centre(a, ax) < centre(b, ax); });
                mid = si + best_pos;
            }
        }
    }
    if (best_ax < 0) { // out of levels, or every centre coincides: halve along the longest axis of the centres' range
        Box cb;
        for (size_t i = si; i < ei; ++i)
            for (int k = 0; k < 3; ++k) cb.lo[k] = std::fmin(cb.lo[k], centre(h[i], k)), cb.hi[k] = std::fmax(cb.hi[k], centre(h[i], k));
        const int ax = cb.longestAxis();
        std::stable_sort(h.begin() + (ptrdiff_t)si, h.begin() + (ptrdiff_t)ei,
                         [&](const Item& a, const Item& b) { return centre(a, ax) < centre(b, ax); });
        mid = si + n / 2;
    }
    out.nodes[me].count = 0;
    buildSah(h, si, mid, out, depth + 1, budget - 1);
    buildSah(h, mid, ei, out, depth + 1, budget - 1);
    out.nodes[me].skip = (uint32_t)out.nodes.size();
}
} // namespace detail

constexpr size_t kMaxBig = 8;   // oversized hittables kept out of the tree, at most
constexpr size_t kMinTree = 32; // .. and only while the tree keeps more than this many (small pools are flat-list territory)

// Hittables are numbered spheres first, then triangles (src/ecs.zig:43-51 order, triangles appended).
//
// peel_oversized = false: the reference's tree, node for node (what rayz_hip_scene_bvh exports and the oracle's
// independent build is checked against).
// peel_oversized = true (the tree the GPU walks): the same build over the pool MINUS its oversized hittables — those
// whose box is longer than a quarter of the box of everything else (at most kMaxBig, largest first).  One such hittable
// (the r = 1000 ground sphere of randomBouncing, src/rayz.zig:58-74) makes the box of every one of its ~14 ancestors
// cover the whole scene, so every ray visits them all; kept out of the tree it is tested once per segment instead and
// the remaining boxes are tight.  The nearest hit does not depend on how the hittables are organised.
// sah = true (the tree the GPU walks, again): split by surface-area heuristic instead (detail::buildSah above).
inline FlatBvh build(const std::vector<RayzSphere>& spheres, const std::vector<RayzTriangle>& triangles,
                     bool peel_oversized = false, bool sah = false) {
    FlatBvh out;
    if (spheres.empty() && triangles.empty()) return out;
    std::vector<detail::Item> h(spheres.size() + triangles.size());
    for (size_t i = 0; i < spheres.size(); ++i) h[i] = {sphereBox(spheres[i]), (uint32_t)i};
    for (size_t i = 0; i < triangles.size(); ++i)
        h[spheres.size() + i] = {triangleBox(triangles[i]), (uint32_t)(spheres.size() + i)};
    if (peel_oversized) {
        auto extent = [](const Box& b) { return std::fmax(std::fmax(b.hi[0] - b.lo[0], b.hi[1] - b.lo[1]), b.hi[2] - b.lo[2]); };
        // an OUTLIER only: longer than a quarter of everything else AND more than kOutlier times the median hittable — a
        // compact cluster of similar hittables, each spanning a good part of the cluster, stays whole (it would be
        // tested once per segment, f64 roots included, ahead of a walk that could have culled it)
        constexpr double kOutlier = 8.0;
        double median = 0;
        {
            std::vector<double> ext(h.size());
            for (size_t i = 0; i < h.size(); ++i) ext[i] = extent(h[i].box);
            std::nth_element(ext.begin(), ext.begin() + (ptrdiff_t)(ext.size() / 2), ext.end());
            median = ext[ext.size() / 2];
        }
        while (out.big.size() < kMaxBig && h.size() > kMinTree) {
            size_t worst = 0;
            for (size_t i = 1; i < h.size(); ++i)
                if (extent(h[i].box) > extent(h[worst].box)) worst = i;
            Box rest;
            for (size_t i = 0; i < h.size(); ++i)
                if (i != worst) rest.enclose(h[i].box);
            if (!(extent(h[worst].box) > 0.25 * extent(rest) && extent(h[worst].box) > kOutlier * median)) break;
            out.big.push_back(h[worst].pool);
            h.erase(h.begin() + (ptrdiff_t)worst); // keeps pool order among the rest
        }
    }
    if (sah) detail::buildSah(h, 0, h.size(), out, 1, detail::levelsFor(h.size()) + detail::kSahExtraDepth);
    else detail::build(h, 0, h.size(), out, 1);
    out.order.resize(h.size());
    for (size_t i = 0; i < h.size(); ++i) out.order[i] = h[i].pool;
    return out;
}

// ---- the boxes as the device holds them: 16-bit plane indices on a scene-wide grid ------------------------------------
// plane(i) = glo[k] + i·cell[k] with glo, cell f32 (products of a 16-bit index and a 24-bit mantissa are exact in f64).  A
// lower plane gets the LARGEST index whose plane lies at or below (true plane − pad), an upper one the SMALLEST at or above
// (true plane + pad): the held box always contains the padded true one.  `pad` covers the rounding of the device's slab
// test (rayz_device.hpp: bvh_box_hit), which therefore needs no slack of its own.
struct PlaneGrid {
    float glo[3] = {0, 0, 0}, cell[3] = {1, 1, 1};
    double extent = 0; // largest span of the grid (part of the padding's scale)
    // the grid over [lo - margin, hi + margin]: 65,535 cells per axis, the cell size rounded UP so that the last plane reaches
    static PlaneGrid over(const double* lo, const double* hi, double margin) {
        PlaneGrid g;
        for (int k = 0; k < 3; ++k) {
            const double a = lo[k] - margin, b = hi[k] + margin;
            float f = (float)a;
            if ((double)f > a) f = std::nextafter(f, -std::numeric_limits<float>::infinity());
            g.glo[k] = f;
            float c = (float)((b - (double)f) / 65535.0);
            if (!(c > 0.0f)) c = std::numeric_limits<float>::min();
            while ((double)f + 65535.0 * (double)c < b) c = std::nextafter(c, std::numeric_limits<float>::infinity());
            g.cell[k] = c;
            g.extent = std::fmax(g.extent, 65535.0 * (double)c);
        }
        return g;
    }
    double plane(int k, uint32_t i) const { return (double)glo[k] + (double)i * (double)cell[k]; }
    uint32_t lower(int k, double v) const { // largest index with plane <= v (0 if even the first plane lies above: never, by `over`)
        double x = std::floor((v - (double)glo[k]) / (double)cell[k]);
        x = x < 0 ? 0 : (x > 65535.0 ? 65535.0 : x);
        uint32_t i = (uint32_t)x;
        while (i > 0 && plane(k, i) > v) --i;
        while (i < 65535u && plane(k, i + 1) <= v) ++i;
        return i;
    }
    uint32_t upper(int k, double v) const { // smallest index with plane >= v
        double x = std::ceil((v - (double)glo[k]) / (double)cell[k]);
        x = x < 0 ? 0 : (x > 65535.0 ? 65535.0 : x);
        uint32_t i = (uint32_t)x;
        while (i < 65535u && plane(k, i) < v) ++i;
        while (i > 0 && plane(k, i - 1) >= v) --i;
        return i;
    }
    // {lo.x | hi.x << 16, lo.y | hi.y << 16, lo.z | hi.z << 16} of a box padded by `pad` per side
    void quantize(const Box& b, double pad, uint32_t out[3]) const {
        for (int k = 0; k < 3; ++k) out[k] = lower(k, b.lo[k] - pad) | (upper(k, b.hi[k] + pad) << 16);
    }
};

// Conservative narrowing of a box bound to R: never shrinks the box.
template <class R> inline R roundDown(double v) {
    R r = (R)v;
    if ((double)r > v) r = std::nextafter(r, -std::numeric_limits<R>::infinity());
    return r;
}
template <class R> inline R roundUp(double v) {
    R r = (R)v;
    if ((double)r < v) r = std::nextafter(r, std::numeric_limits<R>::infinity());
    return r;
}

} // namespace rayz_bvh
