// Native backtrace on SIGSEGV / SIGABRT (debugging HIP runtime crashes on the GPU box; the one abort of round 5's eleven suite runs came
// from a thread without a Python frame: faulthandler alone cannot name it).  Works as LD_PRELOAD and when dlopen'ed (tests/conftest.py
// loads it with ctypes): the constructor installs the handlers, the handler prints the native frames of the faulting thread and then
// hands the signal to whoever was installed before (faulthandler's Python dump, or the default action).
// Runs on an alternate stack so a stack overflow is reported too.
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
static struct sigaction old_segv, old_abrt;
static void on_sig(int sig, siginfo_t* si, void* uc) {
    static const char msg[] = "== SIGSEGV, native frames of the faulting thread:\n", msga[] = "== SIGABRT, native frames of the aborting thread:\n";
    if (sig == SIGABRT) write(2, msga, sizeof(msga) - 1); else write(2, msg, sizeof(msg) - 1);
    void* fr[64];
    int n = backtrace(fr, 64);
    backtrace_symbols_fd(fr, n, 2);
    sigaction(sig, sig == SIGABRT ? &old_abrt : &old_segv, 0);      // the previous handler (or the default action) takes it from here
    if (sig == SIGABRT) raise(sig);                                  // (a SIGSEGV re-faults by itself when this handler returns)
}
__attribute__((constructor)) static void init(void) {
    stack_t ss;
    ss.ss_sp = malloc(1 << 16);
    ss.ss_size = 1 << 16;
    ss.ss_flags = 0;
    sigaltstack(&ss, 0);
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = on_sig;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
    sigaction(SIGSEGV, &sa, &old_segv);
    sigaction(SIGABRT, &sa, &old_abrt);
}
