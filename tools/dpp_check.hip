// tools/dpp_check.hip -- checks the DPP wave64 reductions used by tsp2opt.hip against a serial min.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int x) { return __builtin_amdgcn_update_dpp(x, x, CTRL, ROW_MASK, 0xf, false); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float min_step(float v) { return fminf(v, __int_as_float(dpp_i<CTRL, ROW_MASK>(__float_as_int(v)))); }
__device__ __forceinline__ float wave_min(float v)
{
    v = min_step<0xB1, 0xf>(v); v = min_step<0x4E, 0xf>(v); v = min_step<0x141, 0xf>(v);
    v = min_step<0x140, 0xf>(v); v = min_step<0x142, 0xa>(v); v = min_step<0x143, 0xc>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__global__ void k(const float* in, float* out) { out[blockIdx.x * 64 + threadIdx.x] = wave_min(in[blockIdx.x * 64 + threadIdx.x]); }
int main()
{
    const int W = 4096; float *h = (float*)malloc(W * 64 * 4), *o = (float*)malloc(W * 64 * 4), *d, *e;
    srand(1); for (int i = 0; i < W * 64; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    for (int w = 0; w < 64; ++w) h[w * 64 + w] = -10.f - w;    // min at every lane position once
    hipMalloc(&d, W * 64 * 4); hipMalloc(&e, W * 64 * 4); hipMemcpy(d, h, W * 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(W), dim3(64), 0, 0, d, e); hipMemcpy(o, e, W * 64 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < W; ++w) { float m = h[w * 64]; for (int l = 1; l < 64; ++l) m = fminf(m, h[w * 64 + l]);
        for (int l = 0; l < 64; ++l) if (o[w * 64 + l] != m) { if (bad < 5) printf("wave %d lane %d got %f want %f\n", w, l, o[w*64+l], m); ++bad; } }
    printf("dpp wave_min mismatches: %d of %d\n", bad, W * 64);
    return bad != 0;
}
