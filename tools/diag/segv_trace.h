// Diagnostic only (tools/diag): print a backtrace on SIGSEGV/SIGABRT/SIGBUS, stdout unbuffered.
// Async-signal-safe as far as glibc allows: no stdio in the handler (the heap may be the thing that
// is corrupt), backtrace() warmed up at install time so its first call does not load libgcc (and
// allocate) inside the handler, the default disposition restored on entry (SA_RESETHAND) so a second
// fault inside the handler ends the process with the signal's own status instead of re-entering,
// and _exit(128 + sig) at the end.  (Round 2's version formatted with snprintf and was entered twice
// on a corrupted heap; the run it served ended in the watchdog's timeout, rc 124.)
#pragma once
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
namespace {
inline void diag_write(const char* s) { (void)!write(2, s, strlen(s)); }
inline void diag_write_hex(unsigned long v) {
  char buf[2 + 16 + 1];
  buf[0] = '0'; buf[1] = 'x';
  for (int i = 0; i < 16; ++i) buf[2 + i] = "0123456789abcdef"[(v >> (60 - 4 * i)) & 15];
  buf[18] = 0;
  diag_write(buf);
}
void diag_handler(int sig, siginfo_t* info, void*) {
  diag_write("\n[diag] signal ");
  char d[4] = {(char)('0' + sig / 10), (char)('0' + sig % 10), 0, 0};
  diag_write(d);
  diag_write(", fault address ");
  diag_write_hex((unsigned long)(info ? info->si_addr : nullptr));
  diag_write("\n");
  void* frames[64];
  const int depth = backtrace(frames, 64);
  backtrace_symbols_fd(frames, depth, 2);
  _exit(128 + sig);
}
struct DiagInstall {
  DiagInstall() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    void* warm[4];
    (void)backtrace(warm, 4);  // loads the unwinder now, not inside the handler
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = diag_handler;
    sa.sa_flags = SA_SIGINFO | SA_RESETHAND;
    sigaction(SIGSEGV, &sa, nullptr);
    sigaction(SIGABRT, &sa, nullptr);
    sigaction(SIGBUS, &sa, nullptr);
  }
} diag_install;
}  // namespace
