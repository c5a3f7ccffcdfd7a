// tools/valu_chain_microbench.hip -- the filter sweep's per-pair dependency chain
// (sub, sub, mul, fmac, sqrt, add, sub, min) as 16 independent chains per loop trip, registers only,
// to calibrate the VALU cost model of the hot loop.  Variants: plain, pk_fma for the squares.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define ITERS 2048

template <int KIND>
__global__ void k(float* out, float seed)
{
    float cx[8], cy[8], dc[4], ec[8];
    for (int i = 0; i < 8; ++i) { cx[i] = seed * (i + 1) + threadIdx.x * 1e-3f; cy[i] = seed * (i + 2) * 0.7f; ec[i] = 0.01f * i; }
    for (int i = 0; i < 4; ++i) dc[i] = 0.1f * i;
    float rx = seed * 0.3f, ry = seed * 0.2f, best = 1e9f;
    for (int it = 0; it < ITERS; ++it) {
        float mm = 1e9f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float t[4], dn[4];
            if (KIND == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float dx = rx - cx[a + q], dy = ry - cy[a + q];
                    float s = __builtin_fmaf(dy, dy, dx * dx);
                    dn[q] = __builtin_amdgcn_sqrtf(s);
                    t[q] = (dc[q] + dn[q]) - ec[a + q];
                }
            } else if (KIND == 1) {     // squares through v_pk_mul / v_pk_fma on 2 pairs at a time
#pragma unroll
                for (int q = 0; q < 4; q += 2) {
                    v2f dx = {rx - cx[a + q], rx - cx[a + q + 1]};
                    v2f dy = {ry - cy[a + q], ry - cy[a + q + 1]};
                    v2f s = dx * dx;
                    s = __builtin_elementwise_fma(dy, dy, s);
                    dn[q] = __builtin_amdgcn_sqrtf(s.x); dn[q + 1] = __builtin_amdgcn_sqrtf(s.y);
                    t[q] = (dc[q] + dn[q]) - ec[a + q]; t[q + 1] = (dc[q + 1] + dn[q + 1]) - ec[a + q + 1];
                }
            } else if (KIND == 2) {     // no sqrt (to price it)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float dx = rx - cx[a + q], dy = ry - cy[a + q];
                    float s = __builtin_fmaf(dy, dy, dx * dx);
                    dn[q] = s;
                    t[q] = (dc[q] + dn[q]) - ec[a + q];
                }
            } else {                    // mul+mul+add instead of mul+fma
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float dx = rx - cx[a + q], dy = ry - cy[a + q];
                    float s;
                    asm volatile("v_mul_f32 %0, %1, %1\n\tv_mul_f32 %1, %2, %2\n\tv_add_f32 %0, %0, %1" : "=&v"(s), "+v"(dx) : "v"(dy));
                    dn[q] = __builtin_amdgcn_sqrtf(s);
                    t[q] = (dc[q] + dn[q]) - ec[a + q];
                }
            }
            float rm = fminf(fminf(t[0], t[1]), fminf(t[2], t[3])) - ec[a];
            mm = fminf(mm, rm);
#pragma unroll
            for (int q = 0; q < 4; ++q) dc[q] = dn[q];
            rx += 1e-3f; ry -= 1e-3f;    // new row point each row (stands for the LDS row load)
        }
        best = fminf(best, mm);
#pragma unroll
        for (int i = 0; i < 8; ++i) { cx[i] += 1e-4f; }   // stands for the window slide (4 new points)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = best + dc[0] + dc[1] + dc[2] + dc[3];
}

// two independent tile streams per lane, every value a (streamA, streamB) pair -> v_pk_* throughout
__global__ void k2(float* out, float seed)
{
    v2f cx[8], cy[8], dc[4], ec[8];
    for (int i = 0; i < 8; ++i) {
        cx[i] = (v2f){seed * (i + 1) + threadIdx.x * 1e-3f, seed * (i + 3)};
        cy[i] = (v2f){seed * (i + 2) * 0.7f, seed * (i + 1) * 0.3f};
        ec[i] = (v2f){0.01f * i, 0.02f * i};
    }
    for (int i = 0; i < 4; ++i) dc[i] = (v2f){0.1f * i, 0.2f * i};
    float rx = seed * 0.3f, ry = seed * 0.2f, best = 1e9f;
    for (int it = 0; it < ITERS / 2; ++it) {
        float mm = 1e9f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            v2f t[4], dn[4];
            const v2f rx2 = {rx, rx}, ry2 = {ry, ry};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const v2f dx = rx2 - cx[a + q], dy = ry2 - cy[a + q];
                const v2f s = __builtin_elementwise_fma(dy, dy, dx * dx);
                dn[q] = (v2f){__builtin_amdgcn_sqrtf(s.x), __builtin_amdgcn_sqrtf(s.y)};
                t[q] = (dc[q] + dn[q]) - ec[a + q];
            }
            const float m0 = __builtin_fminf(__builtin_fminf(t[0].x, t[0].y), t[1].x);
            const float m1 = __builtin_fminf(__builtin_fminf(t[1].y, t[2].x), t[2].y);
            const float m2 = __builtin_fminf(__builtin_fminf(t[3].x, t[3].y), m0);
            const float rm = __builtin_fminf(m1, m2) - ec[a].x;
            mm = fminf(mm, rm);
#pragma unroll
            for (int q = 0; q < 4; ++q) dc[q] = dn[q];
            rx += 1e-3f; ry -= 1e-3f;
        }
        best = fminf(best, mm);
#pragma unroll
        for (int i = 0; i < 8; ++i) { cx[i] += (v2f){1e-4f, 2e-4f}; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = best + dc[0].x + dc[1].y + dc[2].x + dc[3].y;
}

void run2(const char* name, float* d)
{
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k2, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k2, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double pairwaves_per_simd = (double)(ITERS / 2) * 32 * wps;
        printf("%-24s waves/SIMD=%d  %.3f ms  %.1f cycles@2.4GHz per 64 pairs per SIMD\n", name, wps, ms,
               ms * 1e6 / pairwaves_per_simd * 2.4);
    }
}

template <int KIND>
void run(const char* name, float* d)
{
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double pairwaves_per_simd = (double)ITERS * 16 * wps;
        printf("%-24s waves/SIMD=%d  %.3f ms  %.1f cycles@2.4GHz per 64 pairs per SIMD\n", name, wps, ms,
               ms * 1e6 / pairwaves_per_simd * 2.4);
    }
}

int main()
{
    float* d; (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    run<0>("chain plain", d); run<1>("chain pk_mul/pk_fma", d); run<2>("chain no sqrt", d); run<3>("chain mul mul add", d); run2("two streams, all pk", d);
    return 0;
}
