// fma64_sgpr.hip — does a v_fma_f64 whose first operand is an SGPR pair issue as fast as the all-VGPR form?  (The f64
// flat-list scan feeds its sphere fields that way.)  Whole chip, 4 waves per SIMD like the trace kernel.
//   hipcc --offload-arch=gfx950 -O3 -o fma64_sgpr fma64_sgpr.hip && ./fma64_sgpr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
struct S { double s[16]; };

template <int FORM> __global__ __launch_bounds__(256, 4) void k(float* out, const float* in, S sv, int iters) {
    double a[16], b = in[threadIdx.x & 63], c = in[64 + (threadIdx.x & 63)];
    for (int i = 0; i < 16; ++i) a[i] = in[128 + i];
    for (int it = 0; it < iters; ++it) {
        if (FORM == 0) {
#define X(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (FORM == 1) { // a different SGPR pair each time
#define X(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "s"(sv.s[i]), "v"(c));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (FORM == 2) { // the same SGPR pair every time
#define X(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "s"(sv.s[0]), "v"(c));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else { // the scan's dependent form: acc = fma(s, v, acc) chains of 3, then the two squares
#define X(i) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(a[i]) : "s"(sv.s[i]), "v"(c), "v"(b));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        }
    }
    double acc = 0;
    for (int i = 0; i < 16; ++i) acc += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)acc;
}

template <int FORM> void run(const char* name) {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * 4, iters = 20000;
    float *out, *in;
    (void)hipMalloc(&out, sizeof(float) * blocks * 256);
    (void)hipMalloc(&in, sizeof(float) * 256);
    std::vector<float> h(256, 1.0f);
    for (int i = 0; i < 256; ++i) h[i] = 1.0f + 1e-3f * (float)(i % 7);
    h[0] = 0.999f;
    (void)hipMemcpy(in, h.data(), sizeof(float) * 256, hipMemcpyHostToDevice);
    S sv;
    for (int i = 0; i < 16; ++i) sv.s[i] = 1.0 + 1e-4 * i;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(256), 0, 0, out, in, sv, 100);
    (void)hipDeviceSynchronize();
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(256), 0, 0, out, in, sv, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double instr = (double)blocks * 4 * 64.0 * iters;
    std::printf("{\"form\": \"%s\", \"ms\": %.3f, \"TFLOP_per_s\": %.2f, \"cycles_per_wave_instr_per_SIMD_at_2.4GHz\": %.3f}\n", name, best,
                instr * 128.0 / (best * 1e-3) / 1e12, best * 1e-3 * 2.4e9 / (instr / (prop.multiProcessorCount * 4.0)));
    (void)hipFree(out);
    (void)hipFree(in);
}

int main() {
    run<0>("v_fma_f64 vgpr operands");
    run<1>("v_fma_f64 sgpr pair, different each time");
    run<2>("v_fma_f64 sgpr pair, the same each time");
    run<3>("v_fma_f64 sgpr pair, non-accumulating");
    return 0;
}
