// scan_body_pk.hip — the same 4-sphere loop body, two spheres per v_pk_fma_f32 (one SGPR-PAIR read per stage for
// two spheres), against the scalar-FMA body.  No loads; values opaque.   ticks per wave-test per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CONSTANT __attribute__((address_space(4)))

template <int MODE, int G> __global__ void k(int n, float* out, unsigned long long* cyc, const float* in) {
    const int l = threadIdx.x & 63;
    const float e1x = in[l], e1z = in[64 + l], e2x = in[128 + l], e2y = in[192 + l], e2z = in[256 + l], k1 = in[320 + l],
                k2 = in[384 + l], t2y = in[448 + l];
    const f2 E1x{e1x, e1x}, E1z{e1z, e1z}, E2x{e2x, e2x}, E2y{e2y, e2y}, E2z{e2z, e2z}, K1{k1, k1}, K2{k2, k2}, T2y{t2y, t2y};
    const CONSTANT f2* ci = (const CONSTANT f2*)in;
    f2 cx[G / 2], cy[G / 2], cz[G / 2], r2[G / 2], vy[G / 2]; // SoA pairs: {sphere 2j, sphere 2j+1}
    for (int q = 0; q < G / 2; ++q) cx[q] = ci[256 + q], cy[q] = ci[260 + q], cz[q] = ci[264 + q], r2[q] = ci[268 + q], vy[q] = ci[272 + q];
    float acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i += G) {
#pragma unroll
        for (int q = 0; q < G / 2; ++q)
            asm volatile("" : "+s"(cx[q]), "+s"(cy[q]), "+s"(cz[q]), "+s"(r2[q]), "+s"(vy[q]));
        f2 d[G / 2];
#pragma unroll
        for (int q = 0; q < G / 2; ++q) {
            if (MODE == 0) { // packed: 8 v_pk_fma per 2 spheres
                f2 p1 = __builtin_elementwise_fma(cx[q], E1x, K1);
                f2 p2 = __builtin_elementwise_fma(cx[q], E2x, K2);
                p1 = __builtin_elementwise_fma(cz[q], E1z, p1);
                p2 = __builtin_elementwise_fma(cy[q], E2y, p2);
                p2 = __builtin_elementwise_fma(cz[q], E2z, p2);
                p2 = __builtin_elementwise_fma(vy[q], T2y, p2);
                f2 t = __builtin_elementwise_fma(-p2, p2, r2[q]);
                d[q] = __builtin_elementwise_fma(-p1, p1, t);
            } else { // scalar FMAs on the same SGPRs
                for (int h = 0; h < 2; ++h) {
                    const float p1 = __builtin_fmaf(cz[q][h], e1z, __builtin_fmaf(cx[q][h], e1x, k1));
                    const float p2 = __builtin_fmaf(vy[q][h], t2y, __builtin_fmaf(cz[q][h], e2z, __builtin_fmaf(cy[q][h], e2y, __builtin_fmaf(cx[q][h], e2x, k2))));
                    d[q][h] = __builtin_fmaf(-p1, p1, __builtin_fmaf(-p2, p2, r2[q][h]));
                }
            }
        }
        float m = __builtin_fmaxf(d[0].x, d[0].y);
#pragma unroll
        for (int q = 1; q < G / 2; ++q) m = __builtin_fmaxf(m, __builtin_fmaxf(d[q].x, d[q].y));
        if (m >= 0.f) acc += __builtin_sqrtf(m) / (d[0].x + 3.0f); // never taken
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (l == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int MODE, int G> void run(const char* name, const float* in) {
    const int n = 400000;
    printf("%-52s", name);
    for (int wps : {1, 2, 3, 4}) {
        const int threads = wps * 4 * 64, blocks = 256;
        float* out;
        unsigned long long* cyc;
        hipMalloc(&out, (size_t)threads * blocks * 4);
        hipMalloc(&cyc, (size_t)threads * blocks / 64 * 8);
        hipLaunchKernelGGL((k<MODE, G>), dim3(blocks), dim3(threads), 0, 0, n, out, cyc, in);
        hipDeviceSynchronize();
        std::vector<unsigned long long> c((size_t)threads * blocks / 64);
        hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto x : c) mean += (double)x;
        mean /= c.size();
        printf("  %dw: %6.2f", wps, mean / ((double)n * wps));
        hipFree(out);
        hipFree(cyc);
    }
    printf("   ticks / wave-test / SIMD\n");
}

int main() {
    std::vector<float> h(1024, 0.5f);
    for (int q = 0; q < 16; ++q) h[512 + q] = 100.f + q, h[520 + q] = 100.f, h[528 + q] = 100.f, h[536 + q] = 0.01f, h[544 + q] = 0.1f;
    float* in;
    hipMalloc(&in, 4096);
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    run<1, 4>("scalar FMAs, SGPR operands, branch, G=4", in);
    run<0, 4>("packed FMAs (2 spheres / instr), branch, G=4", in);
    run<1, 8>("scalar FMAs, SGPR operands, branch, G=8", in);
    run<0, 8>("packed FMAs (2 spheres / instr), branch, G=8", in);
    return 0;
}
