// scan_loop.hip — which part of the flat-list scan's inner loop costs the time?  Replicates one group of 4
// y-moving sphere tests (the "basis-fma" body: 35 VALU reading 20 different SGPRs) and adds the loop's other
// ingredients one at a time.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CONSTANT __attribute__((address_space(4)))

// MODE 0: VALU body only, sphere data loaded once      1: + s_load of the next group each iteration (no wait needed: data reused)
// MODE 2: + real dependency: wait for the loads issued in the previous iteration (the kernel's ping-pong)
// MODE 3: MODE 2 + the group-reject branch (never taken slow path)
template <int MODE> __global__ void k(const f4* __restrict__ sph, const float* __restrict__ vy, int n, float* out,
                                      unsigned long long* cyc, const float* in) {
    const CONSTANT f4* g = (const CONSTANT f4*)sph;
    const CONSTANT float* gv = (const CONSTANT float*)vy;
    const float e1x = in[threadIdx.x & 63], e1z = in[64 + (threadIdx.x & 63)], e2x = in[128 + (threadIdx.x & 63)],
                e2y = in[192 + (threadIdx.x & 63)], e2z = in[256 + (threadIdx.x & 63)], k1 = in[320], k2 = in[321], t2y = in[322];
    float acc = 0;
    f4 c[4] = {g[0], g[1], g[2], g[3]};
    float v[4] = {gv[0], gv[1], gv[2], gv[3]};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i += 4) {
        f4 cn[4];
        float vn[4];
        if (MODE >= 1) {
            if (MODE >= 2) asm volatile("" ::"s"(c[0].x), "s"(v[0]));
#pragma unroll
            for (int q = 0; q < 4; ++q) cn[q] = g[i + 4 + q], vn[q] = gv[i + 4 + q];
            __builtin_amdgcn_sched_barrier(0);
        }
        float d[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float p1 = __builtin_fmaf(c[q].z, e1z, __builtin_fmaf(c[q].x, e1x, k1));
            const float p2 = __builtin_fmaf(v[q], t2y, __builtin_fmaf(c[q].z, e2z, __builtin_fmaf(c[q].y, e2y, __builtin_fmaf(c[q].x, e2x, k2))));
            d[q] = __builtin_fmaf(-p1, p1, __builtin_fmaf(-p2, p2, c[q].w));
        }
        const float m = __builtin_fmaxf(__builtin_fmaxf(d[0], d[1]), __builtin_fmaxf(d[2], d[3]));
        if (MODE >= 3) {
            if (m >= 0.f) acc += __builtin_sqrtf(m) / (d[0] + 3.0f); // never taken with the data used
        } else {
            acc += m;
        }
        if (MODE >= 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) c[q] = cn[q], v[q] = vn[q];
        } else if (MODE == 1) {
            asm volatile("" ::"s"(cn[0].x), "s"(vn[0]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int MODE> void run(const char* name, const f4* sph, const float* vy, int n, const float* in) {
    printf("%-58s", name);
    for (int wps : {1, 2, 3, 4}) {
        const int threads = wps * 4 * 64, blocks = 256;
        float* out;
        unsigned long long* cyc;
        hipMalloc(&out, (size_t)threads * blocks * 4);
        hipMalloc(&cyc, (size_t)threads * blocks / 64 * 8);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, sph, vy, n, out, cyc, in);
        hipDeviceSynchronize();
        std::vector<unsigned long long> c((size_t)threads * blocks / 64);
        hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto x : c) mean += (double)x;
        mean /= c.size();
        printf("  %dw: %6.2f", wps, mean / ((double)n * wps)); // ticks per wave-test per SIMD
        hipFree(out);
        hipFree(cyc);
    }
    printf("   ticks / wave-test / SIMD\n");
}

int main() {
    const int n = 8192 * 40; // 40 passes over an 8k-record (160 KB) list
    std::vector<f4> h(n + 8, f4{100.f, 100.f, 100.f, 0.01f});
    std::vector<float> hv(n + 8, 0.1f), hin(1024, 0.5f);
    f4* sph;
    float *vy, *in;
    hipMalloc(&sph, h.size() * 16);
    hipMalloc(&vy, hv.size() * 4);
    hipMalloc(&in, 4096);
    hipMemcpy(sph, h.data(), h.size() * 16, hipMemcpyHostToDevice);
    hipMemcpy(vy, hv.data(), hv.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(in, hin.data(), 4096, hipMemcpyHostToDevice);
    run<0>("0: VALU body only (8.75 instr/test -> 17.5 cyc ideal)", sph, vy, n, in);
    run<1>("1: + scalar loads issued, not consumed", sph, vy, n, in);
    run<2>("2: + consume the previous iteration's loads (ping-pong)", sph, vy, n, in);
    run<3>("3: + group-reject branch", sph, vy, n, in);
    return 0;
}
