// fp32 MFMA issue-rate probe (dev tool): how close to 64 flop/clk/SIMD does a stream of v_mfma_f32_16x16x4_f32 /
// v_mfma_f32_32x32x2_f32 get with 1 or 2 waves per SIMD, with or without interleaved VALU moves.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_probe.hip -o abl/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NACC, int MOVS>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
    extern __shared__ float lds[];
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f4){0, 0, 0, 0};
    float a[8], b[4];
    for (int i = 0; i < 8; ++i) a[i] = a0 + threadIdx.x + i;
    for (int i = 0; i < 4; ++i) b[i] = b0 + i;
    if (a0 != 1.0f) {   // "real" data: full-mantissa values in [-1, 1)
        for (int i = 0; i < 8; ++i) a[i] = (float)((threadIdx.x * 2654435761u + i * 40503u + blockIdx.x * 7919u) & 0xffffff) / 8388608.0f - 1.0f;
        for (int i = 0; i < 4; ++i) b[i] = (float)((threadIdx.x * 2246822519u + i * 3266489917u) & 0xffffff) / 8388608.0f * 0.01f - 0.005f;
    }
    float m[8];
    for (int i = 0; i < 8; ++i) m[i] = a0 * i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 256 / NACC / 4; ++g) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < NACC; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(t + g) & 7], b[j], acc[t], 0, 0, 0);
            if (MOVS) {
#pragma unroll
                for (int x = 0; x < MOVS; ++x) asm volatile("v_mov_b32 %0, %1" : "=v"(m[x & 7]) : "v"(m[(x + 1) & 7]));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += m[i];
    if (s == 123.456f) out[threadIdx.x] = s + lds[0];
}

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
    extern __shared__ float lds[];
    f16v acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int v = 0; v < 16; ++v) acc[i][v] = 0;
    float a[8], b[4];
    for (int i = 0; i < 8; ++i) a[i] = a0 + threadIdx.x + i;
    for (int i = 0; i < 4; ++i) b[i] = b0 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 128 / NACC / 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < NACC; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(t + g) & 7], b[j], acc[t], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int v = 0; v < 16; ++v) s += acc[i][v];
    if (s == 123.456f) out[threadIdx.x] = s + lds[0];
}

template <typename F>
static void run(const char* name, F launch, double flop_per_iter_per_wave, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch(blocks, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(blocks, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = flop_per_iter_per_wave * iters * 4.0 * blocks;
    printf("%-44s blocks=%4d  %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, fl / ms / 1e9);
}

int main() {
    float* out;
    hipMalloc(&out, 4096);
    const int iters = 2000;
    for (int real = 0; real < 2; ++real) {
        const float av = real ? 0.5f : 1.0f;
        hipFuncSetAttribute((const void*)k16<8, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        run(real ? "16x16x4 acc=8 REAL data, 2 blocks/CU" : "16x16x4 acc=8 trivial data, 2 blocks/CU",
            [&](int b, int it) { hipLaunchKernelGGL((k16<8, 0>), dim3(b), dim3(256), 60 * 1024, 0, out, it, av, 2.0f); }, 256.0 * 2048, 2048, iters);
        run(real ? "16x16x4 acc=8 REAL data, 2 blocks/CU (again)" : "16x16x4 acc=8 trivial data, 2 blocks/CU (again)",
            [&](int b, int it) { hipLaunchKernelGGL((k16<8, 0>), dim3(b), dim3(256), 60 * 1024, 0, out, it, av, 2.0f); }, 256.0 * 2048, 2048, iters * 4);
    }
    return 0;
    const double f16 = 256.0 * 2048, f32 = 128.0 * 4096;
    for (int occ = 1; occ <= 2; ++occ) {
        const size_t lds = occ == 1 ? 100 * 1024 : 60 * 1024;
        const int blocks = 256 * occ * 4;
#define L16(NACC, MOVS)                                                                                    \
    hipFuncSetAttribute((const void*)k16<NACC, MOVS>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024); \
    run("16x16x4 acc=" #NACC " movs/32mfma=" #MOVS, [&](int b, int it) { hipLaunchKernelGGL((k16<NACC, MOVS>), dim3(b), dim3(256), lds, 0, out, it, 1.0f, 2.0f); }, f16, blocks, iters);
        printf("--- %d block(s) of 4 waves per CU\n", occ);
        L16(8, 0)
        L16(4, 0)
        L16(2, 0)
        L16(1, 0)
        L16(8, 4)
        L16(8, 10)
        L16(8, 20)
#define L32(NACC)                                                                                       \
    hipFuncSetAttribute((const void*)k32<NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024); \
    run("32x32x2 acc=" #NACC, [&](int b, int it) { hipLaunchKernelGGL((k32<NACC>), dim3(b), dim3(256), lds, 0, out, it, 1.0f, 2.0f); }, f32, blocks, iters);
        L32(4)
        L32(2)
        L32(1)
    }
    return 0;
}
