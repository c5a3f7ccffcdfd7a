// ml4tspbench_amd/csrc/wave_reduce.hpp -- wave64 reductions on DPP for gfx950 (shared by the kernels)
#pragma once
#include <hip/hip_runtime.h>

namespace t2o {

__device__ __forceinline__ bool better(double e, unsigned idx, double be, unsigned bidx)
{
    return e < be || (e == be && idx < bidx);
}

// ---- wave64 reductions on DPP (no LDS round trips): quad swaps, half-row and row mirrors, then the
// row broadcasts; the full result lands in lane 63 and is broadcast with v_readlane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int x)
{
    return __builtin_amdgcn_update_dpp(x, x, CTRL, ROW_MASK, 0xf, false);   // masked-off lanes keep x
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void best_step(double& e, unsigned& idx)
{
    const int olo = dpp_i<CTRL, ROW_MASK>(__double2loint(e));
    const int ohi = dpp_i<CTRL, ROW_MASK>(__double2hiint(e));
    const unsigned oi = (unsigned)dpp_i<CTRL, ROW_MASK>((int)idx);
    const double oe = __hiloint2double(ohi, olo);
    if (better(oe, oi, e, idx)) { e = oe; idx = oi; }
}

// lexicographic min of (e, idx) over the wave; result in every lane
__device__ __forceinline__ void wave_best(double& e, unsigned& idx)
{
    best_step<0xB1, 0xf>(e, idx);       // quad_perm [1,0,3,2]
    best_step<0x4E, 0xf>(e, idx);       // quad_perm [2,3,0,1]
    best_step<0x141, 0xf>(e, idx);      // row_half_mirror
    best_step<0x140, 0xf>(e, idx);      // row_mirror
    best_step<0x142, 0xa>(e, idx);      // row_bcast15 into rows 1 and 3
    best_step<0x143, 0xc>(e, idx);      // row_bcast31 into rows 2 and 3
    const int lo = __builtin_amdgcn_readlane(__double2loint(e), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(e), 63);
    idx = (unsigned)__builtin_amdgcn_readlane((int)idx, 63);
    e = __hiloint2double(hi, lo);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float min_step(float v)
{
    return fminf(v, __int_as_float(dpp_i<CTRL, ROW_MASK>(__float_as_int(v))));
}

__device__ __forceinline__ float wave_min(float v)
{
    v = min_step<0xB1, 0xf>(v);
    v = min_step<0x4E, 0xf>(v);
    v = min_step<0x141, 0xf>(v);
    v = min_step<0x140, 0xf>(v);
    v = min_step<0x142, 0xa>(v);
    v = min_step<0x143, 0xc>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// unsigned minimum over the wave.  Masked-off lanes read the identity, which lets the compiler fold the move
// into v_min_u32_dpp: one instruction a step.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned umin_step(unsigned v)
{
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, CTRL, ROW_MASK, 0xf, false);
    return o < v ? o : v;
}

__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    v = umin_step<0xB1, 0xf>(v);
    v = umin_step<0x4E, 0xf>(v);
    v = umin_step<0x141, 0xf>(v);
    v = umin_step<0x140, 0xf>(v);
    v = umin_step<0x142, 0xa>(v);
    v = umin_step<0x143, 0xc>(v);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// wave_min for values that are never NaN, on their order-preserving integer images (fminf quiets both operands
// first: four instructions a step)
__device__ __forceinline__ float wave_min_ordered(float v)
{
    unsigned u = (unsigned)__float_as_int(v);
    u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;
    u = wave_min_u32(u);
    u ^= (u >> 31) ? 0x80000000u : 0xffffffffu;
    return __int_as_float((int)u);
}

// inclusive prefix sum over the wave (Hillis-Steele inside the rows of 16 lanes on row_shr, then the
// row totals on row_bcast15 / row_bcast31); lanes without a source add 0
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned scan_step(unsigned x)
{
    return x + (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xf, false);
}

__device__ __forceinline__ unsigned wave_scan_add(unsigned x)
{
    x = scan_step<0x111, 0xf>(x);       // row_shr:1
    x = scan_step<0x112, 0xf>(x);       // row_shr:2
    x = scan_step<0x114, 0xf>(x);       // row_shr:4
    x = scan_step<0x118, 0xf>(x);       // row_shr:8
    x = scan_step<0x142, 0xa>(x);       // row_bcast15 into rows 1 and 3
    x = scan_step<0x143, 0xc>(x);       // row_bcast31 into rows 2 and 3
    return x;
}

}  // namespace t2o
