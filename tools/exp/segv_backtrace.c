// development: LD_PRELOAD this (gcc -shared -fPIC -o /tmp/segv.so tools/exp/segv_backtrace.c) to get a backtrace on stderr when a process dies of SIGSEGV / SIGABRT
#define _GNU_SOURCE
#include <signal.h>
#include <execinfo.h>
#include <unistd.h>
#include <stdlib.h>
static void h(int s, siginfo_t* i, void* c) { void* b[64]; int n = backtrace(b, 64); backtrace_symbols_fd(b, n, 2); _exit(139); }
__attribute__((constructor)) static void init(void) { struct sigaction a = {0}; a.sa_sigaction = h; a.sa_flags = SA_SIGINFO | SA_ONSTACK; static char st[1<<16]; stack_t ss = {st, 0, sizeof st}; sigaltstack(&ss, 0); sigaction(SIGSEGV, &a, 0); sigaction(SIGABRT, &a, 0); sigaction(SIGBUS, &a, 0); }
