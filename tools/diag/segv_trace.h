// Diagnostic only (tools/diag): print a backtrace on SIGSEGV/SIGABRT/SIGBUS, stdout unbuffered.
#pragma once
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
namespace {
void diag_handler(int sig, siginfo_t* info, void*) {
  char head[128];
  int n = snprintf(head, sizeof(head), "\n[diag] signal %d, fault address %p\n", sig, info ? info->si_addr : nullptr);
  (void)!write(2, head, (size_t)n);
  void* frames[64];
  int depth = backtrace(frames, 64);
  backtrace_symbols_fd(frames, depth, 2);
  _exit(128 + sig);
}
struct DiagInstall {
  DiagInstall() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = diag_handler;
    sa.sa_flags = SA_SIGINFO;
    sigaction(SIGSEGV, &sa, nullptr);
    sigaction(SIGABRT, &sa, nullptr);
    sigaction(SIGBUS, &sa, nullptr);
  }
} diag_install;
}  // namespace
