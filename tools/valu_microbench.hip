// tools/valu_microbench.hip -- measures wave64 VALU issue cost on gfx950 for the instruction kinds
// the 2-opt filter sweep uses (plain/packed fp32, v_sqrt_f32, min3), at 1/2/4/8 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_microbench.hip -o tools/valu_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
#define ITERS 4096
#define UNROLL 16

template <int KIND>
__global__ void k(float* out, float seed)
{
    float a[8]; v2f p[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 1e-3f + i; p[i] = (v2f){a[i], a[i] + 0.5f}; }
    const float c = seed * 0.999f; const v2f pc = {c, c + 1e-3f};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int i = u & 7;
            if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
            if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
            if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pc));
            if (KIND == 4) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 5) asm volatile("v_min3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
            if (KIND == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
            if (KIND == 7) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 8) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (KIND == 9) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (KIND == 10) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a[i]) : "v"(c));
            if (KIND == 11) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(p[i]) : "v"(pc));
            if (KIND == 12) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (KIND == 13) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(p[i]) : "v"(pc));
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, float* d)
{
    for (int wps : {1, 2, 4, 8}) {           // waves per SIMD: block = 256 threads (1 wave/SIMD), wps blocks per CU
        const int blocks = 256 * wps;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)ITERS * UNROLL * wps;     // wave-instructions per SIMD
        printf("%-28s waves/SIMD=%d  %.3f ms  %.2f ns/wave-instr/SIMD  (= %.2f cyc @2.4GHz)\n", name, wps, ms,
               ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    }
}

int main()
{
    float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    run<0>("v_add_f32", d); run<9>("v_sub_f32", d); run<8>("v_mul_f32", d); run<1>("v_fma_f32", d);
    run<10>("v_fmac_f32", d); run<12>("v_min_f32", d); run<5>("v_min3_f32", d);
    run<2>("v_pk_add_f32", d); run<11>("v_pk_add_f32 neg", d); run<13>("v_pk_add_f32 op_sel", d);
    run<6>("v_pk_mul_f32", d); run<3>("v_pk_fma_f32", d);
    run<4>("v_sqrt_f32", d); run<7>("v_rsq_f32", d);
    return 0;
}
