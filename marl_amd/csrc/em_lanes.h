// em_lanes.h - exchanges between the four 16-lane rows of a wavefront (the MLP kernels keep batch row c on lanes c, c + 16,
// c + 32, c + 48) with gfx950's row-swap instructions instead of ds_bpermute: v_permlane16_swap / v_permlane32_swap are
// vector-ALU instructions, a shuffle is an LDS round trip in the middle of a dependent chain.
#pragma once
#include <hip/hip_runtime.h>

namespace {

// sum over the four lanes that share lane & 15, the same value on all four: (r0 + r1) + (r2 + r3) in every row, which is
// also what  v += __shfl_xor(v, 16); v += __shfl_xor(v, 32)  gives (float addition commutes), bit for bit.
// v_permlane16_swap D, S exchanges D's odd rows with S's even rows; with D = S = v: D' = {r0, r0, r2, r2}, S' = {r1, r1, r3, r3}.
// v_permlane32_swap D, S exchanges D's upper half with S's lower half; with D = S = s: D' = {lo, lo}, S' = {hi, hi}.
__device__ __forceinline__ float em_rows_sum(float v)
{
    const unsigned b = __float_as_uint(v);
    const auto p = __builtin_amdgcn_permlane16_swap(b, b, false, false);
    const float s1 = __uint_as_float(p[0]) + __uint_as_float(p[1]);
    const unsigned c = __float_as_uint(s1);
    const auto r = __builtin_amdgcn_permlane32_swap(c, c, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// the value that lane (lane ^ 16) / (lane ^ 32) holds, i.e. __shfl_xor(v, 16) / __shfl_xor(v, 32); q = lane >> 4
__device__ __forceinline__ unsigned em_xor16(unsigned v, int q)
{
    const auto p = __builtin_amdgcn_permlane16_swap(v, v, false, false);   // p[0] = {r0, r0, r2, r2}, p[1] = {r1, r1, r3, r3}
    return (q & 1) ? p[0] : p[1];
}
__device__ __forceinline__ unsigned em_xor32(unsigned v, int q)
{
    const auto p = __builtin_amdgcn_permlane32_swap(v, v, false, false);   // p[0] = {lo, lo}, p[1] = {hi, hi}
    return (q & 2) ? p[0] : p[1];
}

}  // namespace
