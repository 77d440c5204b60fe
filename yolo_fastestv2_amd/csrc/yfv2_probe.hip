// yfv2_probe.hip - the box's effective shader clock, measured by the shader itself (measurement helper, not on the hot path).
//
// Boxes of one pool run the latency-bound launches of this path (one workgroup per image from stage 3 on) at visibly
// different speeds while the memory-bound ones agree; a bench line without the clock it ran at cannot be compared with
// another box's.  s_memtime counts SHADER cycles (MI355X_MICROARCH.md: "s_memtime tick = shader cycle"), s_memrealtime
// counts a constant reference clock (hipDeviceAttributeWallClockRate, 100 MHz on gfx9): the quotient of two deltas taken
// by the same wave is the clock that wave's engine ran at over the interval - DVFS, power cap and perf level included.
//
// Two forms of one kernel:
//   busy  = 1: every lane runs dependent FMAs between the stamps (the clock the chip sustains when all CUs issue VALU work)
//   busy  = 0: the wave sleeps (s_sleep) between looks at the reference clock - one wave per workgroup, a handful of
//              workgroups: it takes no issue slots worth naming, so it can run on a side stream WHILE the forward runs on
//              the main one and reports the clock the forward's kernels actually saw.
#include <hip/hip_runtime.h>

#include "yfv2_internal.h"

namespace {

__global__ __launch_bounds__(64) void clock_probe_kernel(ClockProbeArgs a) {
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long c0 = __builtin_readcyclecounter();
  float v = (float)threadIdx.x * 1e-3f;
  unsigned long long r1 = r0;
  while (r1 - r0 < a.ref_ticks) {
    if (a.busy) {
#pragma unroll
      for (int i = 0; i < 256; ++i) v = __builtin_fmaf(v, 0.999999f, 1e-7f);
    } else {
      __builtin_amdgcn_s_sleep(127);
    }
    r1 = __builtin_amdgcn_s_memrealtime();
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    a.out[4 * blockIdx.x + 0] = c1 - c0;
    a.out[4 * blockIdx.x + 1] = r1 - r0;
    a.out[4 * blockIdx.x + 2] = xcc & 0xf;
    a.out[4 * blockIdx.x + 3] = (unsigned long long)(v != 12345.f);   // keeps the FMA chain alive
  }
}

}  // namespace

bool yfv2_launch_clock_probe(const ClockProbeArgs& a, int workgroups, hipStream_t s) {
  hipLaunchKernelGGL(clock_probe_kernel, dim3(workgroups), dim3(64), 0, s, a);
  return hipGetLastError() == hipSuccess;
}
