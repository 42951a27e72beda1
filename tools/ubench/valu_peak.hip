// valu_peak.hip — sustained vector FMA rate of one MI355X (gfx950): v_fma_f32, v_pk_fma_f32 and v_fma_f64, whole chip,
// 8 waves per SIMD, 16 independent accumulators per lane.  Gives the FP64 / FP32 vector peaks bench.py prices the
// trace kernels against (SURVEY.md §8d asks for a measurement: the local guide quotes only the FP32 figure).
//   hipcc --offload-arch=gfx950 -O3 -o valu_peak valu_peak.hip && ./valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
typedef float f2 __attribute__((ext_vector_type(2)));

template <int FORM> __global__ __launch_bounds__(256, 8) void k(float* out, const float* in, int iters) {
    if (FORM == 0) {
        float a[16], b = in[threadIdx.x & 63], c = in[64 + (threadIdx.x & 63)];
        for (int i = 0; i < 16; ++i) a[i] = in[128 + i];
        for (int it = 0; it < iters; ++it) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        }
        float acc = 0;
        for (int i = 0; i < 16; ++i) acc += a[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    } else if (FORM == 1) {
        f2 a[16], b = {in[threadIdx.x & 63], in[1]}, c = {in[64 + (threadIdx.x & 63)], in[2]};
        for (int i = 0; i < 16; ++i) a[i] = f2{in[128 + i], in[144 + i]};
        for (int it = 0; it < iters; ++it) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        }
        float acc = 0;
        for (int i = 0; i < 16; ++i) acc += a[i].x + a[i].y;
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    } else {
        double a[16], b = in[threadIdx.x & 63], c = in[64 + (threadIdx.x & 63)];
        for (int i = 0; i < 16; ++i) a[i] = in[128 + i];
        for (int it = 0; it < iters; ++it) {
#define X(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        }
        double acc = 0;
        for (int i = 0; i < 16; ++i) acc += a[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = (float)acc;
    }
}

template <int FORM> void run(const char* name, double flop_per_instr_lane) {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * 8, iters = 20000;
    float *out, *in;
    (void)hipMalloc(&out, sizeof(float) * blocks * 256);
    (void)hipMalloc(&in, sizeof(float) * 256);
    std::vector<float> h(256, 1.0f);
    for (int i = 0; i < 256; ++i) h[i] = 1.0f + 1e-3f * (float)(i % 7);
    h[0] = 0.999f;
    (void)hipMemcpy(in, h.data(), sizeof(float) * 256, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(256), 0, 0, out, in, 100);
    (void)hipDeviceSynchronize();
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double instr = (double)blocks * 4 /*waves*/ * 64.0 * iters; // wave-instructions
    const double tflops = instr * 64.0 * flop_per_instr_lane / (best * 1e-3) / 1e12;
    std::printf("{\"form\": \"%s\", \"ms\": %.3f, \"TFLOP_per_s\": %.2f, \"cycles_per_wave_instr_per_SIMD_at_2.4GHz\": %.3f}\n", name, best, tflops,
                best * 1e-3 * 2.4e9 / (instr / (prop.multiProcessorCount * 4.0)));
    (void)hipFree(out);
    (void)hipFree(in);
}

int main() {
    run<0>("v_fma_f32", 2.0);
    run<1>("v_pk_fma_f32", 4.0);
    run<2>("v_fma_f64", 2.0);
    return 0;
}
