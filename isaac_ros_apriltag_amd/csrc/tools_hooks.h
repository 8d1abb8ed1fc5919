// tools_hooks.h -- every measurement hook of the kernels, in one place.  The PRODUCT build defines none of the AMDAT_*
// switches below, so every macro here expands to nothing and the kernel sources carry no conditional code of their own.
// Tools builds (isaac_ros_apriltag_amd.build.build_amd_variant, tools/*.sh) switch single hooks on:
//   -DAMDAT_FQ_PROFILE      shader-cycle counters per phase of k_fit_quads / k_fit_prefilter / k_points (prof[] slots)
//   -DAMDAT_FQ_TIMELINE     wall-clock log of every cluster k_fit_quads processes (tools/fit_timeline_one.py)
//   -DAMDAT_FQ_STOP=n       k_fit_quads drops every cluster after phase n   (instruction counts per phase, tools/fq_phase_insts.sh)
//   -DAMDAT_PT_STOP=n       k_points returns after phase n                  (tools/pt_phase_insts.sh)
//   -DAMDAT_CC_STOP=n       k_cc_local returns after phase n
//   -DAMDAT_FQ_SKIP=mask    the launch sequence leaves out k_fit_quads classes (bits 0..4) or the prefilter (bit 5)
//   -DAMDAT_FQ_NO_*         one of the quad fit's sound early exits compiled out (see below)
//   -DAMDAT_MUTATE=n        a deliberately WRONG build for tools/mutation_check.sh, which shows that the GPU suite fails on it:
//                           1 = the launch sequence leaves out k_fit_small (the throughput set's small-cluster fit);
//                           2 = k_cc_local<4> writes the wrong row as its tile's last perimeter row (one row constant off in the 4-wave instance only)
//                           3 = k_fit_prefilter<64>'s 64-sector test counts one sector too many at either end of every forward arc
//                               (the cut sectors themselves, which hold the corners): it then "proves" real quads above 2048 points
//                               impossible and drops them
//                           the GPU suite ships all three (build.py: build_mutants) and asserts that its stage tests FAIL on each
//                           (tests/test_gpu_parity.py::test_the_suite_fails_on_wrong_builds)
// The stop builds key on P.max_nmaxima == 10 (always true) so that the compiler cannot fold the early exit at compile time
// into dead-code elimination of the phases before it.
#pragma once

// -DAMDAT_ASM_MARKS: phase markers as assembler comments (static instruction counts per phase from hipcc -S)
#ifdef AMDAT_ASM_MARKS
#define AT_MARK(name) asm volatile("; ==MARK " name);
#else
#define AT_MARK(name)
#endif

// ---- k_cc_local ---------------------------------------------------------------------------------------------------------
#ifdef AMDAT_CC_STOP
#define CC_STOP_AT(n) if (AMDAT_CC_STOP == (n) && P.max_nmaxima == 10) return;
#else
#define CC_STOP_AT(n)
#endif

#if defined(AMDAT_MUTATE) && AMDAT_MUTATE == 2
#define CC_LAST_ROW_OFFSET(NW) ((NW) == 4 ? 2 : 1)
#else
#define CC_LAST_ROW_OFFSET(NW) 1
#endif

// ---- k_points -----------------------------------------------------------------------------------------------------------
#ifdef AMDAT_FQ_PROFILE   // prof[0..5]: tile load, emission tests + scan, list, block table, stores, frame table
#define PT_HOOKS_DECL unsigned long long t_prev_ = prof ? __builtin_readcyclecounter() : 0ull;
#define PT_TICK(slot) if (prof && threadIdx.x == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); atomicAdd(&prof[slot], now_ - t_prev_); t_prev_ = now_; }
#else
#define PT_HOOKS_DECL (void)prof;
#define PT_TICK(slot)
#endif
#ifdef AMDAT_PT_STOP      // `keep_live` is a store that keeps the results of the phases so far alive
#define PT_STOP_AT(n, keep_live) if (AMDAT_PT_STOP == (n) && P.max_nmaxima == 10) { keep_live; return; }
#else
#define PT_STOP_AT(n, keep_live)
#endif

// ---- k_fit_quads --------------------------------------------------------------------------------------------------------
#ifdef AMDAT_FQ_TIMELINE
#define FQ_TIMELINE_GLOBALS                                                                                             \
  __device__ unsigned long long g_pf_span[4];   /* prefilter: earliest block start, latest block end; k_scatter's latest end; k_quad_finish's earliest start */ \
  __device__ unsigned long long g_fq_tl[1 << 16][2];                                                                    \
  __device__ unsigned int g_fq_ph[1 << 16][8];   /* wall-clock ticks (10 ns) per phase of the same cluster */           \
  __device__ unsigned int g_fq_tl_n;
#else
#define FQ_TIMELINE_GLOBALS
#endif

#if defined(AMDAT_FQ_PROFILE)
// points of the clusters that reach the pre-sort test (0), that it rejects (1), that the test after the first walk rejects
// (2): prof[40 + 4 * class + k] (the launch passes prof + 8 * class)
#define FQ_HOOKS_DECL unsigned long long t_prev_ = prof ? __builtin_readcyclecounter() : 0ull;
#define FQ_TICK(slot)                                                                  \
  if (prof && tid == 0) {                                                              \
    const unsigned long long now_ = __builtin_readcyclecounter();                      \
    atomicAdd(&prof[slot], now_ - t_prev_);                                            \
    t_prev_ = now_;                                                                    \
  }
#define FQ_COUNT(k, n) if (prof && tid == 0) atomicAdd(&prof[40 - 4 * ((NT == 64) ? 0 : (NT == 128) ? 1 : (NT == 256) ? 2 : (NT == 512) ? 3 : 4) + (k)], (unsigned long long)(n));
#define FQ_TL_NEXT_CLUSTER()
#define FQ_TL_SIZE(sz)
#elif defined(AMDAT_FQ_TIMELINE)
#define FQ_HOOKS_DECL (void)prof; unsigned long long tl_t0_ = 0, tl_prev_ = 0; unsigned int tl_sz_ = 0; unsigned int tl_ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define FQ_TICK(slot) if (tid == 0) { const unsigned long long now_ = wall_clock64(); tl_ph_[slot] += (unsigned int)(now_ - tl_prev_); tl_prev_ = now_; }
#define FQ_COUNT(k, n)
#define FQ_TL_NEXT_CLUSTER()                                                                                            \
  if (tid == 0) {                                                                                                       \
    const unsigned long long now_ = wall_clock64();                                                                     \
    if (tl_sz_) {                                                                                                       \
      const unsigned int k_ = atomicAdd(&g_fq_tl_n, 1u);                                                                \
      if (k_ < (1u << 16)) {                                                                                            \
        g_fq_tl[k_][0] = tl_t0_; g_fq_tl[k_][1] = ((now_ - tl_t0_) << 32) | ((unsigned long long)NT << 20) | (unsigned long long)tl_sz_; \
        for (int j_ = 0; j_ < 8; j_++) g_fq_ph[k_][j_] = tl_ph_[j_];                                                    \
      }                                                                                                                 \
    }                                                                                                                   \
    tl_t0_ = now_; tl_prev_ = now_; tl_sz_ = 0;                                                                         \
    for (int j_ = 0; j_ < 8; j_++) tl_ph_[j_] = 0;                                                                      \
  }
#define FQ_TL_SIZE(sz) tl_sz_ = (unsigned int)(sz);
#else
#define FQ_HOOKS_DECL (void)prof;
#define FQ_TICK(slot)
#define FQ_COUNT(k, n)
#define FQ_TL_NEXT_CLUSTER()
#define FQ_TL_SIZE(sz)
#endif
#ifdef AMDAT_FQ_STOP
#define FQ_STOP_AT(n) if (AMDAT_FQ_STOP == (n) && P.max_nmaxima == 10) continue;
#else
#define FQ_STOP_AT(n)
#endif

#ifdef AMDAT_FQ_TIMELINE
#define TL_MARK_MIN(slot) if (threadIdx.x == 0) atomicMin(&g_pf_span[slot], wall_clock64());
#define TL_MARK_MAX(slot) if (threadIdx.x == 0) atomicMax(&g_pf_span[slot], wall_clock64());
#else
#define TL_MARK_MIN(slot)
#define TL_MARK_MAX(slot)
#endif
// ---- k_fit_prefilter: prof[60..63] = box + dot, sector sums, scan + 32-sector test, 64-sector test --------------------
#ifdef AMDAT_FQ_PROFILE
#define PF_HOOKS_DECL unsigned long long t_prev_ = prof ? __builtin_readcyclecounter() : 0ull;
#define PF_TICK(slot) if (prof && tid == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); atomicAdd(&prof[slot], now_ - t_prev_); t_prev_ = now_; }
#else
#define PF_HOOKS_DECL (void)prof;
#define PF_TICK(slot)
#endif

// ---- the SOUND early exits of the quad fit can be compiled out, one by one, to show that no result depends on them
// (-DAMDAT_FQ_NO_PRESORT_EXIT, -DAMDAT_FQ_NO_EARLY_EXIT, -DAMDAT_FQ_NO_PREFILTER: the A/B builds give the same bytes) ---------
#if defined(AMDAT_MUTATE) && AMDAT_MUTATE == 3
#define PF_MUT_EXTRA_SECTOR(nt) ((nt) == 64 ? 1 : 0)
#else
#define PF_MUT_EXTRA_SECTOR(nt) 0
#endif
#ifdef AMDAT_FQ_NO_PRESORT_EXIT
#define FQ_SOUND_EXIT_PRESORT 0
#else
#define FQ_SOUND_EXIT_PRESORT 1
#endif
#ifdef AMDAT_FQ_NO_EARLY_EXIT
#define FQ_SOUND_EXIT_AFTER_WALK1 0
#else
#define FQ_SOUND_EXIT_AFTER_WALK1 1
#endif
#ifdef AMDAT_FQ_NO_PREFILTER
#define FQ_SOUND_EXIT_PREFILTER 0
#else
#define FQ_SOUND_EXIT_PREFILTER 1
#endif

// ---- launch sequence (detector.hip) ---------------------------------------------------------------------------------------
#if defined(AMDAT_MUTATE) && AMDAT_MUTATE == 1
#define FQ_SKIP_PREFILTER() 0
#define FQ_SKIP_CLASS(c) ((c) < FQ_C0)
#elif defined(AMDAT_FQ_SKIP)
#define FQ_SKIP_PREFILTER() (((AMDAT_FQ_SKIP) >> 5) & 1)
#define FQ_SKIP_CLASS(c) ((c) >= FQ_C0 && (((AMDAT_FQ_SKIP) >> ((c) - FQ_C0)) & 1))
#else
#define FQ_SKIP_PREFILTER() 0
#define FQ_SKIP_CLASS(c) 0
#endif
