/* LD_PRELOAD helper of the crash hunt (tools/_run_crash*.sh): prints the native backtrace of a SIGSEGV / SIGABRT to stderr
 * before the process dies, so that a crash inside the HIP runtime names its frames.
 *   gcc -O1 -g -shared -fPIC tools/segv_backtrace.c -o tools/libsegv_backtrace.so */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

static void on_fault(int sig, siginfo_t* si, void* ctx) {
  (void)ctx;
  void* frames[64];
  char msg[96];
  int n = snprintf(msg, sizeof(msg), "\n== signal %d at address %p: native backtrace\n", sig, si ? si->si_addr : 0);
  if (write(2, msg, (size_t)n) < 0) {}
  const int k = backtrace(frames, 64);
  backtrace_symbols_fd(frames, k, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}

__attribute__((constructor)) static void install(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = on_fault;
  sa.sa_flags = SA_SIGINFO | SA_RESETHAND;
  sigaction(SIGSEGV, &sa, 0);
  sigaction(SIGBUS, &sa, 0);
}
