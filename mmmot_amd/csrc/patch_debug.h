// Phase timers of the trunk kernel (conv3x3_hl16_patch.hip) - included by -DMMMOT_DEBUG builds ONLY
// (libmmmot_hip_debug.so, tools/patch_phase_timers*.py, tools/fused1_phase_timers.py); the product translation unit
// never sees this file, its kernel has no TIMED template parameter and its PT_STAMP / PT_COUNT_ITEM are empty.
//
// TIMED instantiations accumulate s_memtime deltas of thread 0 of every workgroup, summed over the items it walks:
//   pt_dbg[0] decode .. loads issued, [1] prologue wait, [2] K loop, [3] accumulators -> LDS, [4] encode + stores
//   issued, [5] closing barrier, [7] items.  mmmot_set_patch_variant(9) selects them (0: the product kernels; the
// numbering is what is left of the round-1..4 timing experiments, which were removed from the kernel in round 5 -
// their results are in profiles/HISTORY.md); mmmot_debug_read_patch_timers reads / resets the counters.
#pragma once

__device__ unsigned long long pt_dbg[8];

#define PT_TIMED_TPARAM , bool TIMED = false
#define PT_TIMED_TARG(v) , v
#define PT_STAMP_DECL() unsigned long long tprev_ = 0;
#define PT_STAMP(i)                                                                          \
  if constexpr (TIMED) {                                                                     \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();                            \
    if (threadIdx.x == 0 && (i) >= 0) atomicAdd(&pt_dbg[(i) < 0 ? 0 : (i)], now_ - tprev_);  \
    tprev_ = now_;                                                                           \
  }
#define PT_COUNT_ITEM()                                 \
  if constexpr (TIMED) {                                \
    if (threadIdx.x == 0) atomicAdd(&pt_dbg[7], 1ull);  \
  }

static int g_patch_timed = 0;
extern "C" int mmmot_set_patch_variant(int v) {
  if (v != 0 && v != 9) return MMMOT_EINVAL;
  g_patch_timed = (v == 9);
  return MMMOT_OK;
}

extern "C" int mmmot_debug_read_patch_timers(unsigned long long* out8, int reset) {
  hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(pt_dbg), 8 * sizeof(unsigned long long));
  if (e != hipSuccess) return (int)e;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(pt_dbg), z, sizeof(z));
  }
  return mm_check(e);
}
