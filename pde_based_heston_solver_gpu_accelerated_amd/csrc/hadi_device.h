// hadi_device.h -- one include that makes the kernel sources compile either with hipcc for
// gfx950 (the product) or with g++ against the wave emulator in tests/emu (a test-only tool that
// runs the SAME kernel source on host threads so indexing bugs are found without a GPU box).
#pragma once

#if defined(HADI_EMU)
#include "wave_emu.h"  // tests/emu/wave_emu.h: __global__, threadIdx, __shfl_*, __syncthreads ...
#define HADI_HD
#define HADI_DEV
#define HADI_FORCEINLINE inline
#define HADI_DYN_SMEM(T, name) T *name = reinterpret_cast<T *>(emu::t_block->dyn_smem)
#define HADI_UNIFORM(x) (x)
#else
#include <hip/hip_runtime.h>
#define HADI_HD __host__ __device__
#define HADI_DEV __device__
#define HADI_FORCEINLINE __forceinline__
#define HADI_DYN_SMEM(T, name) extern __shared__ __attribute__((aligned(16))) unsigned char name##_raw_[]; \
    T *name = reinterpret_cast<T *>(name##_raw_)
// value is the same in every lane of the wavefront: tell the compiler (SGPR, scalar branches)
#define HADI_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif

#include <math.h>
#include <stddef.h>
#include <stdint.h>

// 1/x without the ~14-instruction IEEE division expansion; relative error up to 10 ulp (2e-15; measured over the pivots'
// range, tests/test_gpu_pins.py), a backward error no larger than the rounding of the pivot's own operands.
HADI_DEV HADI_FORCEINLINE double hadi_rcp(double x) {
#if defined(HADI_EMU)
    return 1.0 / x;
#else
    // v_rcp_f64 delivers 2^-24.5 .. 2^-26; ONE Newton step r (1 + e), e = 1 - x r, leaves e^2: within 10 ulp, mostly 1-2
    // (two FMAs; the third-order step r (1 + e + e^2) used before cost a third FMA on each of the 15 reciprocals of a
    // row -- 3 % of the row pass -- for digits far below the 1e-10 parity tolerance)
#if defined(HADI_RCP_THIRD_ORDER)
    const double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    const double t = fma(e, e, e);
    return fma(r, t, r);
#else
    const double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    return fma(r, e, r);
#endif
#endif
}
