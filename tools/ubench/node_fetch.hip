// node_fetch.hip — what a divergent BVH step costs the vector-memory path: every lane reads ONE random 64-B record
// (L2-resident table, far larger than L1), dependent chain (the next index comes out of the data).
//   mode 0: 4 x global_load_dwordx4 per lane, own record (64 different lines per instruction)
//   mode 1: quad-cooperative: instruction j reads the record of quad-lane j, lane l takes quarter (l & 3), so each
//           instruction touches 16 lines instead of 64; no transpose (the sum is order-free) — the memory side only
//   mode 2: mode 1 + the 4x4 transpose through DPP quad_perm so that each lane ends up with its own record
// Prints ns per wave-step and the implied 16-B lane-requests per clock per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int CTRL> __device__ __forceinline__ unsigned qp(unsigned v) { // quad_perm broadcast / permute
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
template <int J> __device__ __forceinline__ unsigned quad_bcast(unsigned v) { return qp<J | (J << 2) | (J << 4) | (J << 6)>(v); }

template <int MODE> __global__ __launch_bounds__(256) void k(const u4* __restrict__ tab, unsigned mask, int steps, unsigned* out,
                                                             unsigned long long* cyc) {
    const unsigned l = threadIdx.x & 63, i4 = l & 3;
    unsigned cur = (blockIdx.x * 256 + threadIdx.x) * 2654435761u & mask, acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < steps; ++s) {
        u4 r0, r1, r2, r3;
        if (MODE == 0) {
            const u4* p = tab + 4 * (size_t)cur;
            r0 = p[0], r1 = p[1], r2 = p[2], r3 = p[3];
        } else {
            const unsigned c0 = quad_bcast<0>(cur), c1 = quad_bcast<1>(cur), c2 = quad_bcast<2>(cur), c3 = quad_bcast<3>(cur);
            r0 = tab[4 * (size_t)c0 + i4], r1 = tab[4 * (size_t)c1 + i4], r2 = tab[4 * (size_t)c2 + i4], r3 = tab[4 * (size_t)c3 + i4];
            if (MODE == 2) { // lane i wants quarter q of its own record = register set i of quad-lane q
                u4 w[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    // from quad-lane q take its register set `i4` (mine): select by the RECEIVER's index
                    unsigned a0 = quad_bcast<0>(r0[c]), a1 = quad_bcast<0>(r1[c]), a2 = quad_bcast<0>(r2[c]), a3 = quad_bcast<0>(r3[c]);
                    w[0][c] = i4 == 0 ? a0 : i4 == 1 ? a1 : i4 == 2 ? a2 : a3;
                    a0 = quad_bcast<1>(r0[c]), a1 = quad_bcast<1>(r1[c]), a2 = quad_bcast<1>(r2[c]), a3 = quad_bcast<1>(r3[c]);
                    w[1][c] = i4 == 0 ? a0 : i4 == 1 ? a1 : i4 == 2 ? a2 : a3;
                    a0 = quad_bcast<2>(r0[c]), a1 = quad_bcast<2>(r1[c]), a2 = quad_bcast<2>(r2[c]), a3 = quad_bcast<2>(r3[c]);
                    w[2][c] = i4 == 0 ? a0 : i4 == 1 ? a1 : i4 == 2 ? a2 : a3;
                    a0 = quad_bcast<3>(r0[c]), a1 = quad_bcast<3>(r1[c]), a2 = quad_bcast<3>(r2[c]), a3 = quad_bcast<3>(r3[c]);
                    w[3][c] = i4 == 0 ? a0 : i4 == 1 ? a1 : i4 == 2 ? a2 : a3;
                }
                r0 = w[0], r1 = w[1], r2 = w[2], r3 = w[3];
            }
        }
        const unsigned x = r0.x ^ r1.y ^ r2.z ^ r3.w ^ r0.w ^ r1.x ^ r2.y ^ r3.z;
        acc += x;
        // stand-in for the slab tests: ~40 dependent-free VALU ops
        float f = __uint_as_float((x & 0x7fffffu) | 0x3f800000u), g = f;
#pragma unroll
        for (int q = 0; q < 20; ++q) f = __builtin_fmaf(f, 0.999f, 0.001f), g = __builtin_fmaf(g, 1.001f, -0.001f);
        acc += (unsigned)(f + g);
        cur = (x ^ ((unsigned)s * 2654435761u)) & mask; // step-dependent: no short cycles in the chain
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (l == 0) cyc[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
}

template <int MODE> void run(const char* name, const u4* tab, unsigned mask) {
    const int steps = 4000;
    printf("%-48s", name);
    for (int wps : {1, 2, 4, 5, 8}) {
        const int blocks = 256 * wps;
        unsigned* out;
        unsigned long long* cyc;
        (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
        (void)hipMalloc(&cyc, (size_t)blocks * 4 * 8);
        hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, tab, mask, steps, out, cyc);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> c((size_t)blocks * 4);
        (void)hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto x : c) mean += (double)x;
        mean /= c.size();                       // 100 MHz ticks per wave for all steps
        const double ns_step = mean * 10.0 / steps; // per wave-step
        const double req_per_clk = (wps * 4.0) * 256.0 / (ns_step * 2.4); // 16-B lane requests / clk / CU at 2.4 GHz
        printf("  %dw: %6.0f ns (%.2f)", wps, ns_step, req_per_clk);
        if (wps == 1) { std::vector<unsigned> o((size_t)blocks * 256); (void)hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost); unsigned long long h = 0; for (auto x : o) h = h * 1000003ull + x; printf(" [out hash %016llx]", h); }
        (void)hipFree(out);
        (void)hipFree(cyc);
    }
    printf("   per wave-step (16-B lane requests / clk / CU)\n");
}

int main() {
    const unsigned n = 16384; // records of 64 B = 1 MiB: L2-resident, 32x the L1
    std::vector<unsigned> h((size_t)n * 16);
    srand(3);
    for (auto& x : h) x = (unsigned)rand() * 2654435761u + (unsigned)rand();
    u4* tab;
    (void)hipMalloc(&tab, h.size() * 4);
    (void)hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0>("own record, 4 x dwordx4 per lane", tab, n - 1);
    run<1>("quad-cooperative lines, no transpose", tab, n - 1);
    run<2>("quad-cooperative lines + DPP transpose", tab, n - 1);
    return 0;
}
