// mfma_prefilter.hip — cost model of a conservative broad phase on the matrix pipe: per tile of 32 spheres a wave
// runs 4 v_mfma_f32_32x32x16_bf16 (p1, p2 for its two 32-ray column blocks; operands split hi+lo into bf16 slots),
// then q = p1² + p2² per accumulator element, one compare per element, and parks the rare survivors in an LDS
// list.  Prints ticks per wave per "64 tests" (= one sphere against the wave's 64 rays), comparable with
// scan_body_pk's ticks per wave-test.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef short bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE> __global__ __launch_bounds__(256) void k(int tiles, int stream_tiles, const bf8* __restrict__ A, const bf8* __restrict__ B,
                                                             float* out, unsigned long long* cyc, float bound) {
    __shared__ unsigned lds[32 * 256];
    const int l = threadIdx.x & 63;
    unsigned* list = lds + threadIdx.x;
    const bf8 b0p1 = B[l], b0p2 = B[64 + l], b1p1 = B[128 + l], b1p2 = B[192 + l];
    unsigned cnt = 0, total = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int t = 0;
    bf8 a = A[(size_t)0 * 64 + l];
    for (int i = 0; i < tiles; ++i) {
        int tn = t + 1;
        if (tn == stream_tiles) tn = 0;
        const bf8 an = A[(size_t)tn * 64 + l]; // prefetch the next tile's fragment
        const f16v z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const f16v d01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0p1, z, 0, 0, 0);
        const f16v d02 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0p2, z, 0, 0, 0);
        const f16v d11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1p1, z, 0, 0, 0);
        const f16v d12 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1p2, z, 0, 0, 0);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const f16v& p1 = blk ? d11 : d01;
            const f16v& p2 = blk ? d12 : d02;
            if (MODE == 0) {
                bool sv[16];
#pragma unroll
                for (int v = 0; v < 16; ++v) sv[v] = __builtin_fmaf(p2[v], p2[v], p1[v] * p1[v]) <= bound;
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    if (sv[v]) {
                        list[256 * (cnt & 31u)] = (unsigned)(i * 32 + blk * 16 + v);
                        cnt++;
                    }
            } else { // min-reduce first, then per-element in the slow path
                float mn = 3.0e38f;
                float q[16];
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    q[v] = __builtin_fmaf(p2[v], p2[v], p1[v] * p1[v]);
                    mn = __builtin_fminf(mn, q[v]);
                }
                if (mn <= bound) {
#pragma unroll
                    for (int v = 0; v < 16; ++v)
                        if (q[v] <= bound) {
                            list[256 * (cnt & 31u)] = (unsigned)(i * 32 + blk * 16 + v);
                            cnt++;
                        }
                }
            }
        }
        if (__ballot(cnt > 16u) != 0ull) { // stand-in for the flush
            total += cnt;
            cnt = 0;
        }
        a = an;
        t = tn;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(total + cnt);
    if (list[0] == 0xfffffffeu) out[0] = 0.f;
    if (l == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}


template <int MODE> __global__ __launch_bounds__(256) void k2(int tiles, int stream_tiles, const bf8* __restrict__ A, const bf8* __restrict__ B,
                                                              float* out, unsigned long long* cyc, float bound) {
    __shared__ unsigned lds[32 * 256];
    const int l = threadIdx.x & 63;
    unsigned* list = lds + threadIdx.x;
    const bf8 b0p1 = B[l], b0p2 = B[64 + l], b1p1 = B[128 + l], b1p2 = B[192 + l];
    unsigned cnt = 0, total = 0;
    const f16v z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto process = [&](const f16v& p1, const f16v& p2, unsigned id) {
        if (MODE == 0) {
            unsigned long long m[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) m[v] = __ballot(__builtin_fmaf(p2[v], p2[v], p1[v] * p1[v]) <= bound);
#pragma unroll
            for (int v = 0; v < 16; ++v)
                if (m[v] != 0ull) {
                    if (__builtin_amdgcn_inverse_ballot_w64(m[v])) {
                        list[256 * (cnt & 31u)] = id + v;
                        cnt++;
                    }
                }
        } else if (MODE >= 2) { // MODE 2: square test max(|p1|,|p2|) <= b; MODE 3: disc test; groups of 4 rows, park (tile, block, group)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float q[4];
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    q[v] = MODE == 2 ? __builtin_fmaxf(__builtin_fabsf(p1[4 * g + v]), __builtin_fabsf(p2[4 * g + v]))
                                     : __builtin_fmaf(p2[4 * g + v], p2[4 * g + v], p1[4 * g + v] * p1[4 * g + v]);
                const float mn = __builtin_fminf(__builtin_fminf(q[0], q[1]), __builtin_fminf(q[2], q[3]));
                const unsigned long long m = __ballot(mn <= bound);
                if (m != 0ull) {
                    if (__builtin_amdgcn_inverse_ballot_w64(m)) {
                        list[256 * (cnt & 31u)] = id + 4 * g;
                        cnt++;
                    }
                }
            }
        } else {
            float q[16];
            float mn = 3.0e38f;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                q[v] = __builtin_fmaf(p2[v], p2[v], p1[v] * p1[v]);
                mn = __builtin_fminf(mn, q[v]);
            }
            if (__ballot(mn <= bound) != 0ull) {
                if (mn <= bound) { // park the whole (tile, block): the flush re-tests its 16 spheres exactly
                    list[256 * (cnt & 31u)] = id;
                    cnt++;
                }
            }
        }
        if (__ballot(cnt > 16u) != 0ull) { // stand-in for the flush
            total += cnt;
            cnt = 0;
        }
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int t = 0;
    bf8 a = A[(size_t)0 * 64 + l];
    f16v x1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0p1, z, 0, 0, 0);
    f16v x2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0p2, z, 0, 0, 0);
    for (int i = 0; i < tiles; ++i) {
        int tn = t + 1;
        if (tn == stream_tiles) tn = 0;
        const bf8 an = A[(size_t)tn * 64 + l];
        const f16v y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1p1, z, 0, 0, 0);
        const f16v y2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1p2, z, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        process(x1, x2, (unsigned)(i * 32));
        __builtin_amdgcn_sched_barrier(0);
        x1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(an, b0p1, z, 0, 0, 0);
        x2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(an, b0p2, z, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        process(y1, y2, (unsigned)(i * 32 + 16));
        __builtin_amdgcn_sched_barrier(0);
        a = an;
        t = tn;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(total + cnt) + x1[0] * 0.f;
    if (list[0] == 0xfffffffeu) out[0] = 0.f;
    if (l == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

static unsigned short bf16(float f) {
    unsigned u;
    __builtin_memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

template <int MODE, int PIPE> void run(const char* name, const bf8* A, const bf8* B, int stream_tiles) {
    const int tiles = 20000;
    printf("%-44s", name);
    for (int wps : {1, 2, 3, 4}) {
        const int threads = 256, blocks = 256 * wps;
        float* out;
        unsigned long long* cyc;
        hipMalloc(&out, (size_t)threads * blocks * 4);
        hipMalloc(&cyc, (size_t)threads * blocks / 64 * 8);
        if (PIPE) hipLaunchKernelGGL((k2<MODE>), dim3(blocks), dim3(threads), 0, 0, tiles, stream_tiles, A, B, out, cyc, 1.0f);
        else hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), 0, 0, tiles, stream_tiles, A, B, out, cyc, 1.0f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> c((size_t)threads * blocks / 64);
        hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
        std::vector<float> o((size_t)threads * blocks);
        hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost);
        double mean = 0, surv = 0;
        for (auto x : c) mean += (double)x;
        for (auto x : o) surv += x;
        mean /= c.size();
        printf("  %dw: %6.2f (%.2f%%)", wps, mean / ((double)tiles * 32 * wps), 100.0 * surv / ((double)o.size() * tiles * 32));
        hipFree(out);
        hipFree(cyc);
    }
    printf("   ticks / wave / 64 tests / SIMD (survivor rate)\n");
}

int main() {
    const int stream_tiles = 313;
    std::vector<unsigned short> a((size_t)stream_tiles * 64 * 8, 0), b(4 * 64 * 8, 0);
    srand(1);
    for (int t = 0; t < stream_tiles; ++t)
        for (int r = 0; r < 32; ++r) { // sphere row r: slot 0 = X, slot 3 = Y in [-10, 10)
            a[((size_t)t * 64 + r) * 8 + 0] = bf16(20.f * rand() / RAND_MAX - 10.f);
            a[((size_t)t * 64 + r) * 8 + 3] = bf16(20.f * rand() / RAND_MAX - 10.f);
        }
    for (int blk = 0; blk < 2; ++blk)
        for (int c = 0; c < 32; ++c) { // ray column c: p1 = X + shift, p2 = Y + shift
            b[((size_t)(2 * blk) * 64 + c) * 8 + 0] = bf16(1.f);
            b[((size_t)(2 * blk + 1) * 64 + c) * 8 + 3] = bf16(1.f);
        }
    bf8 *A, *B;
    hipMalloc(&A, a.size() * 2);
    hipMalloc(&B, b.size() * 2);
    hipMemcpy(A, a.data(), a.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(B, b.data(), b.size() * 2, hipMemcpyHostToDevice);
    run<0, 0>("4 MFMA + mul/fma/cmp per element, masks", A, B, stream_tiles);
    run<1, 0>("4 MFMA + mul/fma/min, slow path per lane", A, B, stream_tiles);
    run<0, 1>("pipelined, per-element uniform branch", A, B, stream_tiles);
    run<1, 1>("pipelined, min-reduce, park (tile,block)", A, B, stream_tiles);
    run<2, 1>("pipelined, square test, park groups of 4", A, B, stream_tiles);
    run<3, 1>("pipelined, disc test, park groups of 4", A, B, stream_tiles);
    return 0;
}
