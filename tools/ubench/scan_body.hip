// scan_body.hip — the flat-list scan's loop body WITHOUT any loads: where do 3.3 cycles per VALU go?
// BODY: 4 y-moving sphere tests (35 VALU).  Variants: sphere values in SGPRs vs VGPRs; with / without the
// group-reject branch (v_cmp + s_and_saveexec + s_cbranch_execz, slow path never taken); 4 or 8 tests per branch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <bool SGPR, int BRANCH, int G> __global__ void k(int n, float* out, unsigned long long* cyc, const float* in) {
    const int l = threadIdx.x & 63;
    const float e1x = in[l], e1z = in[64 + l], e2x = in[128 + l], e2y = in[192 + l], e2z = in[256 + l], k1 = in[320 + l],
                k2 = in[384 + l], t2y = in[448 + l];
    float cx[G], cy[G], cz[G], r2[G], vy[G];
    const __attribute__((address_space(4))) float* ci = (const __attribute__((address_space(4))) float*)in;
    for (int q = 0; q < G; ++q) {
        if (SGPR) cx[q] = ci[512 + q], cy[q] = ci[520 + q], cz[q] = ci[528 + q], r2[q] = ci[536 + q], vy[q] = ci[544 + q];
        else cx[q] = in[512 + q + (l >> 6)], cy[q] = in[520 + q + (l >> 6)], cz[q] = in[528 + q + (l >> 6)],
             r2[q] = in[536 + q + (l >> 6)], vy[q] = in[544 + q + (l >> 6)];
    }
    float acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i += G) {
#pragma unroll
        for (int q = 0; q < G; ++q) { // keep the values opaque so nothing is hoisted
            if (SGPR) asm volatile("" : "+s"(cx[q]), "+s"(cy[q]), "+s"(cz[q]), "+s"(r2[q]), "+s"(vy[q]));
            else asm volatile("" : "+v"(cx[q]), "+v"(cy[q]), "+v"(cz[q]), "+v"(r2[q]), "+v"(vy[q]));
        }
        float d[G];
#pragma unroll
        for (int q = 0; q < G; ++q) {
            const float p1 = __builtin_fmaf(cz[q], e1z, __builtin_fmaf(cx[q], e1x, k1));
            const float p2 = __builtin_fmaf(vy[q], t2y, __builtin_fmaf(cz[q], e2z, __builtin_fmaf(cy[q], e2y, __builtin_fmaf(cx[q], e2x, k2))));
            d[q] = __builtin_fmaf(-p1, p1, __builtin_fmaf(-p2, p2, r2[q]));
        }
        float m = d[0];
#pragma unroll
        for (int q = 1; q < G; ++q) m = __builtin_fmaxf(m, d[q]);
        if (BRANCH == 1) {
            if (m >= 0.f) acc += __builtin_sqrtf(m) / (d[0] + 3.0f); // never taken: all tests miss
        } else if (BRANCH == 2) {
            if (__builtin_amdgcn_ballot_w64(m >= 0.f) != 0) acc += __builtin_sqrtf(m) / (d[0] + 3.0f); // scalar branch
        } else {
            acc += m;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (l == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <bool SGPR, int BRANCH, int G> void run(const char* name, const float* in) {
    const int n = 400000;
    printf("%-60s", name);
    for (int wps : {1, 2, 3, 4}) {
        const int threads = wps * 4 * 64, blocks = 256;
        float* out;
        unsigned long long* cyc;
        hipMalloc(&out, (size_t)threads * blocks * 4);
        hipMalloc(&cyc, (size_t)threads * blocks / 64 * 8);
        hipLaunchKernelGGL((k<SGPR, BRANCH, G>), dim3(blocks), dim3(threads), 0, 0, n, out, cyc, in);
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
    printf("   ticks / wave-test / SIMD (8.75 VALU)\n");
}

int main() {
    std::vector<float> h(1024, 0.5f);
    for (int q = 0; q < 8; ++q) h[512 + q] = 100.f + q, h[520 + q] = 100.f, h[528 + q] = 100.f, h[536 + q] = 0.01f, h[544 + q] = 0.1f;
    float* in;
    hipMalloc(&in, 4096);
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    run<true, 0, 4>("SGPR operands, no branch, G=4", in);
    run<false, 0, 4>("VGPR operands, no branch, G=4", in);
    run<true, 1, 4>("SGPR operands, reject branch (exec-mask form), G=4", in);
    run<true, 2, 4>("SGPR operands, reject branch (ballot + scalar branch), G=4", in);
    run<true, 1, 8>("SGPR operands, reject branch (exec-mask form), G=8", in);
    run<false, 1, 4>("VGPR operands, reject branch (exec-mask form), G=4", in);
    return 0;
}
