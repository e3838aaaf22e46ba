// LD_PRELOAD aid: print a native backtrace on SIGSEGV (debugging HIP runtime crashes on the GPU box).
// Runs on an alternate stack so a stack overflow is reported too.
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
static void on_segv(int sig, siginfo_t* si, void* uc) {
    static const char msg[] = "== SIGSEGV, native frames:\n";
    write(2, msg, sizeof(msg) - 1);
    void* fr[48];
    int n = backtrace(fr, 48);
    backtrace_symbols_fd(fr, n, 2);
    _exit(139);
}
__attribute__((constructor)) static void init(void) {
    stack_t ss;
    ss.ss_sp = malloc(1 << 16);
    ss.ss_size = 1 << 16;
    ss.ss_flags = 0;
    sigaltstack(&ss, 0);
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = on_segv;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
    sigaction(SIGSEGV, &sa, 0);
}
