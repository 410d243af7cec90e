/*
 * last_gasp.c -- async-signal-safe "write what is in hand and leave" for bench.py (libnb_lastgasp.so; tooling, not part
 * of the drop-in boundary).
 *
 * The library's error convention is the reference's: print and abort() (src/lib/util.h:17-29).  When an optional bench
 * leg dies that way -- or by a fault -- inside a C call, rank 0 must still deliver the JSON line it had prepared.  A
 * Python-level handler cannot run then, and a ctypes callback would take the GIL and allocate inside signal context.
 * This handler does only what POSIX allows there: write() of bytes registered beforehand, then _exit(6).
 * nb_last_gasp_set copies the line into one of two static buffers and publishes it with one atomic store, so a signal
 * that lands during an update sees either the old or the new line, never a torn one.
 */
#define _GNU_SOURCE
#include <signal.h>
#include <stdatomic.h>
#include <string.h>
#include <unistd.h>

#define NB_GASP_MAX (1u << 20)

static char g_buf[2][NB_GASP_MAX];
static size_t g_len[2];
static atomic_int g_live = -1; /* which buffer the handler writes; -1: nothing registered */
static atomic_int g_fd = 1;
static atomic_int g_done;

static void on_fatal(int sig) {
    if (!atomic_exchange(&g_done, 1)) {
        const int live = atomic_load(&g_live);
        if (live >= 0) {
            const char *p = g_buf[live];
            size_t left = g_len[live];
            while (left > 0) {
                const ssize_t w = write(atomic_load(&g_fd), p, left);
                if (w <= 0) break;
                p += w;
                left -= (size_t)w;
            }
        }
        static const char head[] = "[bench] fatal signal ";
        static const char tail[] = " inside an optional leg; the line written holds what was in hand (exit 6)\n";
        char num[2] = {(char)('0' + (sig / 10) % 10), (char)('0' + sig % 10)};
        (void)!write(2, head, sizeof head - 1);
        (void)!write(2, sig >= 10 ? num : num + 1, sig >= 10 ? 2 : 1);
        (void)!write(2, tail, sizeof tail - 1);
    }
    _exit(6);
}

/* Register (or replace) the line; the first call installs the handler for SIGABRT, SIGSEGV, SIGBUS and SIGFPE.
 * Returns 0, or -1 when the line does not fit. */
int nb_last_gasp_set(int fd, const void *line, unsigned long len) {
    if (len > NB_GASP_MAX) return -1;
    const int live = atomic_load(&g_live);
    const int next = live == 0 ? 1 : 0;
    memcpy(g_buf[next], line, len);
    g_len[next] = len;
    atomic_store(&g_fd, fd);
    atomic_store(&g_live, next);
    if (live < 0) {
        struct sigaction sa;
        memset(&sa, 0, sizeof sa);
        sa.sa_handler = on_fatal;
        sigemptyset(&sa.sa_mask);
        const int sigs[] = {SIGABRT, SIGSEGV, SIGBUS, SIGFPE};
        for (unsigned i = 0; i < sizeof sigs / sizeof sigs[0]; i++) sigaction(sigs[i], &sa, NULL);
    }
    return 0;
}

/* Nothing is written from now on; a fatal signal still ends the process with status 6. */
void nb_last_gasp_disarm(void) { atomic_store(&g_done, 1); }
