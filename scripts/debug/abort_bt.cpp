// Debug aid: print the NATIVE backtrace of the thread that raises SIGABRT / SIGSEGV (python's faulthandler only shows
// python frames). Build: g++ -O1 -g -shared -fPIC -o abort_bt.so abort_bt.cpp ; load with ctypes.CDLL before the run.
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <cstring>
#include <cstdlib>
#include <fcntl.h>

static int out_fd = 2;      // DRL_ABORT_BT_FILE: write there instead of stderr (pytest captures fd 2 and loses it on abort)

static void handler(int sig) {
    void* frames[64];
    const char* msg = sig == SIGABRT ? "\n=== SIGABRT native backtrace ===\n" : "\n=== SIGSEGV native backtrace ===\n";
    (void)!write(out_fd, msg, strlen(msg));
    int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, out_fd);
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install() {
    if (const char* p = getenv("DRL_ABORT_BT_FILE")) {
        int fd = open(p, O_WRONLY | O_CREAT | O_APPEND, 0644);
        if (fd >= 0) out_fd = fd;
    }
    signal(SIGABRT, handler);
    signal(SIGSEGV, handler);
}
