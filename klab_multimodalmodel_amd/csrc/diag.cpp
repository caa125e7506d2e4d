// Crash diagnostics of the C-ABI library (test / bring-up aid, opt-in; no effect on the product path unless installed).
//
// klab_segv_trace_install(fd): a SIGSEGV / SIGBUS / SIGABRT / SIGFPE / SIGILL handler that writes, with async-signal-safe
// calls only, the caller-supplied context string (the test harness sets it to the running pytest node id), the native
// backtrace of the faulting thread (backtrace_symbols_fd) and the context string again to `fd`, then restores the default
// action and re-raises: the exit status stays that of the fault.  The harness installs it BEFORE Python's faulthandler, which
// chains to the previously installed handler after its own dump: the Python stacks come first, this block is the last thing
// in the log, so a truncated log still names the test in its tail.
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

#include "klab_mm.h"

namespace {

char g_ctx[512] = "(no context set)";
bool g_installed = false;
int g_fd = 2;

void put(const char* s) { (void)!write(g_fd, s, strlen(s)); }

void on_fatal(int sig, siginfo_t* info, void*) {
  static volatile sig_atomic_t entered = 0;
  if (entered) _exit(128 + sig);
  entered = 1;
  put("\n==== klab: fatal signal ");
  char num[16]; int n = 0, v = sig; char tmp[8];
  do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v && n < 7);
  for (int i = 0; i < n; ++i) num[i] = tmp[n - 1 - i];
  num[n] = 0;
  put(num);
  put(" while running: ");
  put(g_ctx);
  put("\n---- native backtrace of the faulting thread ----\n");
  void* frames[64];
  const int nf = backtrace(frames, 64);
  backtrace_symbols_fd(frames, nf, g_fd);
  put("==== klab: fatal signal while running: ");
  put(g_ctx);
  put("\n");
  (void)info;
  signal(sig, SIG_DFL);
  raise(sig);
}

}  // namespace

extern "C" int klab_segv_set_context(const char* text) {
  if (!text) return KLAB_ERR_BADARG;
  strncpy(g_ctx, text, sizeof(g_ctx) - 1);
  g_ctx[sizeof(g_ctx) - 1] = 0;
  return KLAB_OK;
}

extern "C" int klab_segv_trace_install(int fd) {
  g_fd = fd >= 0 ? fd : 2;
  if (g_installed) return KLAB_OK;
  void* warm[4];
  backtrace(warm, 4);  // first call loads libgcc: not something to do inside a signal handler
  static char altstack[1 << 16];
  stack_t ss;
  ss.ss_sp = altstack; ss.ss_size = sizeof(altstack); ss.ss_flags = 0;
  sigaltstack(&ss, nullptr);
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = on_fatal;
  sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_NODEFER;
  sigemptyset(&sa.sa_mask);
  const int sigs[] = {SIGSEGV, SIGBUS, SIGABRT, SIGFPE, SIGILL};
  for (int sg : sigs)
    if (sigaction(sg, &sa, nullptr) != 0) return KLAB_ERR_BADARG;
  g_installed = true;
  return KLAB_OK;
}
