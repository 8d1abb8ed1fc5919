// tools_timeline.h -- host entry points of the -DAMDAT_FQ_TIMELINE tools build (tools/fit_timeline_one.py): the quad fit's
// per-cluster wall-clock log.  Included by detector.hip inside its extern "C" section or after it; the product build compiles none of it.
#pragma once
#ifdef AMDAT_FQ_TIMELINE
// tools-only: returns and clears the quad fit's per-cluster wall-clock log (start tick, duration << 32 | threads << 20 | points)
extern "C" int amdAprilTagsDebugTimelinePhases(unsigned int* out, unsigned int n) {   // call BEFORE amdAprilTagsDebugTimeline (which clears the log)
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (n > (1u << 16)) n = 1u << 16;
  if (n && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fq_ph), (size_t)n * 32) != hipSuccess) return -1;
  return (int)n;
}
// earliest prefilter block start, latest prefilter block end, (unused), earliest k_quad_finish block start; resets them
extern "C" int amdAprilTagsDebugTimelineSpan(unsigned long long* out4) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_pf_span), 32) != hipSuccess) return -1;
  const unsigned long long init[4] = {~0ull, 0ull, 0ull, ~0ull};
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_pf_span), init, 32) != hipSuccess) return -1;
  return 0;
}
extern "C" int amdAprilTagsDebugTimeline(unsigned long long* out, unsigned int cap) {
  unsigned int n = 0;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_fq_tl_n), 4) != hipSuccess) return -1;
  if (n > (1u << 16)) n = 1u << 16;
  if (n > cap) n = cap;
  if (n && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fq_tl), (size_t)n * 16) != hipSuccess) return -1;
  const unsigned int zero = 0;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_fq_tl_n), &zero, 4) != hipSuccess) return -1;
  return (int)n;
}
#endif
