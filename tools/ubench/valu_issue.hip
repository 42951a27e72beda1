// valu_issue.hip — issue cost of the wave64 VALU forms the flat-list scan uses, on gfx950.
// Each kernel runs N iterations of 32 instructions of one form over 8 independent accumulators and reports
// cycles per instruction per SIMD (s_memtime), for 1..8 waves per SIMD.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int FORM> __global__ void k(float* out, const float* in, int iters, unsigned long long* cyc, float s0f, float s1f) {
    float a[8], b = in[threadIdx.x & 63], c = in[64 + (threadIdx.x & 63)];
    for (int i = 0; i < 8; ++i) a[i] = in[128 + i];
    float s0 = __builtin_amdgcn_readfirstlane(s0f), s1 = __builtin_amdgcn_readfirstlane(s1f);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (FORM == 0) { // VOP2 v_fmac, all VGPR
#define X(i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (FORM == 1) { // VOP2 v_fmac, src0 = SGPR
#define X(i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "s"(s0), "v"(c));
                REP8(X)
#undef X
            } else if (FORM == 2) { // VOP3 v_fma, all VGPR, distinct dst chain
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (FORM == 3) { // VOP3 v_fma, SGPR src0, neg on src2
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, -%0" : "+v"(a[i]) : "s"(s0), "v"(c));
                REP8(X)
#undef X
            } else if (FORM == 4) { // VOP3 v_fma, SGPR as src2 (addend), neg src0
#define X(i) asm volatile("v_fma_f32 %0, -%0, %0, %1" : "+v"(a[i]) : "s"(s1));
                REP8(X)
#undef X
            } else if (FORM == 5) { // VOP2 v_mul with SGPR
#define X(i) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[i]) : "s"(s0));
                REP8(X)
#undef X
            } else if (FORM == 6) { // v_max3 (VOP3)
#define X(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (FORM == 7) { // VOP2 v_sub with SGPR src0
#define X(i) asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(a[i]) : "s"(s0));
                REP8(X)
#undef X
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0;
    for (int i = 0; i < 8; ++i) acc += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int FORM> void run(const char* name) {
    const int iters = 20000;
    float *out, *in;
    unsigned long long* cyc;
    hipMalloc(&in, 4096);
    std::vector<float> h(1024, 1.0001f);
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    printf("%-44s", name);
    for (int wps : {1, 2, 4, 6, 8}) { // waves per SIMD: one block of wps*4 waves per CU, 256 CUs
        const int threads = wps * 4 * 64, blocks = 256;
        hipMalloc(&out, (size_t)threads * blocks * 4);
        hipMalloc(&cyc, (size_t)threads * blocks / 64 * 8);
        hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(threads), 0, 0, out, in, iters, cyc, 1.5f, 0.25f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> c((size_t)threads * blocks / 64);
        hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto v : c) mean += (double)v;
        mean /= c.size();
        // each wave issued iters*32 instructions in `mean` cycles while sharing its SIMD with wps-1 others
        printf("  %dw: %5.2f", wps, mean / ((double)iters * 32 * wps));
        hipFree(out);
        hipFree(cyc);
    }
    printf("   cycles/instr/SIMD\n");
    hipFree(in);
}

int main() {
    run<0>("VOP2 v_fmac  v,v");
    run<1>("VOP2 v_fmac  s,v");
    run<5>("VOP2 v_mul   s,v");
    run<7>("VOP2 v_sub   s,v");
    run<2>("VOP3 v_fma   v,v,v");
    run<3>("VOP3 v_fma   s,v,-v");
    run<4>("VOP3 v_fma   -v,v,s");
    run<6>("VOP3 v_max3  v,v,v");
    return 0;
}
