// Debug helper: print a native backtrace on SIGSEGV (own signal stack, so a stack overflow is caught too).
//   gcc -shared -fPIC -O1 segv_bt.c -o segv_bt.so ; ctypes.CDLL(...).segv_bt_install()
#include <execinfo.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
static void on_segv(int sig, siginfo_t* si, void* ctx) {
    (void)ctx;
    void* frames[96];
    char msg[64] = "SIGSEGV at address ";
    unsigned long a = (unsigned long)si->si_addr;
    char hex[20]; int n = 0;
    do { hex[n++] = "0123456789abcdef"[a & 15]; a >>= 4; } while (a);
    size_t L = strlen(msg);
    while (n) msg[L++] = hex[--n];
    msg[L++] = '\n';
    (void)!write(2, msg, L);
    int k = backtrace(frames, 96);
    backtrace_symbols_fd(frames, k, 2);
    _exit(128 + sig);
}
void segv_bt_install(void) {
    static char* stack;
    if (!stack) stack = malloc(1 << 18);
    stack_t ss = {.ss_sp = stack, .ss_size = 1 << 18, .ss_flags = 0};
    sigaltstack(&ss, 0);
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_segv;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
    sigaction(SIGSEGV, &sa, 0);
}
