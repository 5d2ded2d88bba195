/* snn_guard -- a protected arena for host buffers and a fault handler that names whoever touches them.  TEST INFRASTRUCTURE
 * (tests/guard_arena.py drives it; nothing under spiking-neural-networks_amd/ knows about it).
 *
 * Why: rounds 3 - 5 saw five executions in ~800 000 in which host memory of the test process -- an oracle array, a downloaded
 * history -- held a word nobody should have written.  Comparing arrays after a call names the call, never the writer.  Here every
 * buffer lives in one big PROT_NONE reservation:
 *   - a buffer ends exactly at an inaccessible page (an overrun faults at the instruction that does it),
 *   - a buffer that is retired becomes PROT_NONE and its address range is never handed out again (a LATE write or read --
 *     a staging thread of the runtime finishing after the call that owned the buffer returned -- faults too),
 *   - regions can be made read-only for a while (the oracle's arrays during a call into the device library).
 * The SIGSEGV / SIGBUS handler writes, with async-signal-safe calls only, the faulting address, the region and its tag, read or
 * write, the thread id and name, the program counter and a backtrace (module + offset) of the FAULTING thread to the log, makes
 * the page accessible and returns, so that the access completes and the process carries on; a fault outside the arena goes to
 * the handler that was installed before (Python's faulthandler). */
#define _GNU_SOURCE
#include <errno.h>
#include <execinfo.h>
#include <fcntl.h>
#include <pthread.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <time.h>
#include <ucontext.h>
#include <unistd.h>

#define PAGE 4096u
#define TAG_BYTES 88

enum { ST_LIVE = 1, ST_READONLY = 2, ST_QUARANTINE = 3, ST_CANARY = 4 };

typedef struct {
    uintptr_t first_page;      /* first data page */
    uint64_t pages;            /* data pages; the page after them is the guard page */
    uintptr_t start;           /* first byte of the buffer (its last byte is the last byte of the data pages) */
    uint64_t bytes;
    uint32_t state, serial;
    char tag[TAG_BYTES];
} region_t;

static uintptr_t g_base, g_end, g_next;
static region_t *g_regions;
static uint64_t g_capacity, g_count;
static int g_log = -1;
static volatile uint64_t g_faults;
static struct sigaction g_old_segv, g_old_bus;
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static int g_dump_signal;      /* raised after a report: faulthandler.register()'s dump of the Python stacks */

static void put(const char *s) { if (g_log >= 0) { ssize_t r = write(g_log, s, strlen(s)); (void)r; } }

static void put_hex(uint64_t v)
{
    char b[19] = "0x"; int n = 2, started = 0;
    for (int s = 60; s >= 0; s -= 4) { int d = (int)((v >> s) & 15); if (d || started || s == 0) { b[n++] = "0123456789abcdef"[d]; started = 1; } }
    b[n] = 0; put(b);
}

static void put_dec(uint64_t v)
{
    char b[24]; int n = 23; b[n] = 0;
    do { b[--n] = (char)('0' + v % 10); v /= 10; } while (v);
    put(b + n);
}

/* the regions are handed out at ascending addresses: binary search for the last region whose first page is <= a */
static region_t *find_region(uintptr_t a)
{
    uint64_t lo = 0, hi = g_count;
    while (lo < hi) { uint64_t mid = (lo + hi) / 2; if (g_regions[mid].first_page <= a) lo = mid + 1; else hi = mid; }
    if (lo == 0) return NULL;
    region_t *r = &g_regions[lo - 1];
    return a < r->first_page + (r->pages + 1) * PAGE ? r : NULL;
}

static void on_fault(int sig, siginfo_t *si, void *uctx)
{
    uintptr_t a = (uintptr_t)si->si_addr;
    if (!g_base || a < g_base || a >= g_end) {
        struct sigaction *old = sig == SIGBUS ? &g_old_bus : &g_old_segv;
        if (old->sa_flags & SA_SIGINFO) { if (old->sa_sigaction) { old->sa_sigaction(sig, si, uctx); return; } }
        else if (old->sa_handler != SIG_DFL && old->sa_handler != SIG_IGN) { old->sa_handler(sig); return; }
        signal(sig, SIG_DFL);
        raise(sig);
        return;
    }
    int saved = errno;
    uint64_t n = __atomic_add_fetch(&g_faults, 1, __ATOMIC_RELAXED);
    ucontext_t *uc = (ucontext_t *)uctx;
    region_t *r = find_region(a);
    put("=== snn_guard fault "); put_dec(n); put(" ===\n");
    struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts);
    put("time "); put_dec((uint64_t)ts.tv_sec); put("."); put_dec((uint64_t)ts.tv_nsec / 1000000); put("\n");
    put("signal "); put_dec((uint64_t)sig); put(" code "); put_dec((uint64_t)si->si_code); put(" address "); put_hex(a);
#if defined(__x86_64__)
    put((uc->uc_mcontext.gregs[REG_ERR] & 2) ? " access WRITE" : " access READ");
    put(" pc "); put_hex((uint64_t)uc->uc_mcontext.gregs[REG_RIP]);
#endif
    put("\n");
    if (r) {
        const char *what = r->state == ST_LIVE ? "live" : r->state == ST_READONLY ? "read-only" : r->state == ST_QUARANTINE ? "retired" : "canary";
        put("region serial "); put_dec(r->serial); put(" state "); put(what); put(" tag \""); put(r->tag); put("\" buffer "); put_hex(r->start);
        put(" bytes "); put_dec(r->bytes);
        if (a >= r->first_page + r->pages * PAGE) { put(" -- GUARD PAGE, "); put_dec(a - (r->start + r->bytes)); put(" bytes past the end"); }
        else if (a < r->start) { put(" -- slack before the buffer, "); put_dec(r->start - a); put(" bytes before it"); }
        else { put(" -- offset "); put_dec(a - r->start); }
        put("\n");
    } else put("region: none (arena space not handed out yet)\n");
    long tid = syscall(SYS_gettid);
    put("thread "); put_dec((uint64_t)tid);
    {
        char path[64] = "/proc/self/task/", num[24]; int k = 23; long t = tid; num[k] = 0;
        do { num[--k] = (char)('0' + t % 10); t /= 10; } while (t);
        strcat(path, num + k); strcat(path, "/comm");
        int fd = open(path, O_RDONLY);
        if (fd >= 0) { char name[32]; ssize_t m = read(fd, name, sizeof(name) - 1); close(fd); if (m > 0) { name[m] = 0; put(" name "); put(name); } }
        put("\n");
    }
    put("backtrace of the faulting thread:\n");
    void *frames[64];
    int depth = backtrace(frames, 64);
    if (g_log >= 0) backtrace_symbols_fd(frames, depth, g_log);
    put("=== end of fault "); put_dec(n); put(" ===\n");
    /* let the access complete: this page (not the whole region: every further page reports again) becomes ordinary memory */
    mprotect((void *)(a & ~(uintptr_t)(PAGE - 1)), PAGE, PROT_READ | PROT_WRITE);
    if (g_dump_signal) raise(g_dump_signal);
    errno = saved;
}

/* reserve `reserve_bytes` of address space, open the log, install the handlers; 0 on success */
int snn_guard_init(uint64_t reserve_bytes, uint64_t max_regions, const char *log_path, int dump_signal)
{
    if (g_base) return 0;
    void *p = mmap(NULL, reserve_bytes, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) return errno ? errno : -1;
    void *t = mmap(NULL, max_regions * sizeof(region_t), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (t == MAP_FAILED) { munmap(p, reserve_bytes); return errno ? errno : -1; }
    g_regions = (region_t *)t; g_capacity = max_regions; g_count = 0;
    g_next = (uintptr_t)p + PAGE;                /* the first page stays inaccessible: an underrun of the first buffer faults too */
    g_end = (uintptr_t)p + reserve_bytes;
    g_log = open(log_path, O_WRONLY | O_CREAT | O_APPEND, 0644);
    g_dump_signal = dump_signal;
    void *warm[4]; backtrace(warm, 4);           /* loads libgcc's unwinder now, not inside the handler */
    struct sigaction sa; memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_fault; sa.sa_flags = SA_SIGINFO | SA_NODEFER; sigemptyset(&sa.sa_mask);
    sigaction(SIGSEGV, &sa, &g_old_segv);
    sigaction(SIGBUS, &sa, &g_old_bus);
    __atomic_store_n(&g_base, (uintptr_t)p, __ATOMIC_RELEASE);
    return 0;
}

/* a buffer of `bytes` bytes whose last byte is the last byte before an inaccessible page; the slack in front of it (to the start
 * of its first page) is filled with 0xC7 -- snn_guard_slack_intact checks it.  NULL when the arena or the table is used up. */
void *snn_guard_alloc(uint64_t bytes, const char *tag)
{
    if (!g_base || bytes == 0) return NULL;
    uint64_t pages = (bytes + PAGE - 1) / PAGE;
    pthread_mutex_lock(&g_lock);
    if (g_count >= g_capacity || g_next + (pages + 1) * PAGE > g_end) { pthread_mutex_unlock(&g_lock); return NULL; }
    region_t *r = &g_regions[g_count];
    r->first_page = g_next; r->pages = pages; r->bytes = bytes;
    r->start = g_next + pages * PAGE - bytes; r->state = ST_LIVE; r->serial = (uint32_t)g_count;
    strncpy(r->tag, tag ? tag : "", TAG_BYTES - 1); r->tag[TAG_BYTES - 1] = 0;
    g_next += (pages + 1) * PAGE;
    if (mprotect((void *)r->first_page, pages * PAGE, PROT_READ | PROT_WRITE) != 0) { pthread_mutex_unlock(&g_lock); return NULL; }
    memset((void *)r->first_page, 0xC7, r->start - r->first_page);
    __atomic_store_n(&g_count, g_count + 1, __ATOMIC_RELEASE);
    pthread_mutex_unlock(&g_lock);
    return (void *)r->start;
}

static region_t *region_of_buffer(void *p)
{
    region_t *r = find_region((uintptr_t)p);
    return r && r->start == (uintptr_t)p ? r : NULL;
}

/* 1 when the bytes between the start of the buffer's first page and the buffer still hold the fill */
int snn_guard_slack_intact(void *p)
{
    region_t *r = region_of_buffer(p);
    if (!r) return -1;
    for (const unsigned char *q = (const unsigned char *)r->first_page; q < (const unsigned char *)r->start; ++q) if (*q != 0xC7) return 0;
    return 1;
}

/* state: ST_LIVE (read + write), ST_READONLY, ST_QUARANTINE (no access, physical pages dropped), ST_CANARY (stays read + write:
 * the caller has filled it with a pattern and will look at it again later -- a writer that does not go through the CPU's page
 * tables, a DMA engine, leaves no fault but a changed pattern) */
int snn_guard_set_state(void *p, int state)
{
    region_t *r = region_of_buffer(p);
    if (!r) return -1;
    int prot = state == ST_LIVE || state == ST_CANARY ? PROT_READ | PROT_WRITE : state == ST_READONLY ? PROT_READ : PROT_NONE;
    r->state = (uint32_t)state;
    if (mprotect((void *)r->first_page, r->pages * PAGE, prot) != 0) return errno;
    /* (the physical pages of small retired buffers stay: dropping them is a second system call -- and a second round of
     * inter-processor interrupts -- per buffer, which quartered the campaign's execution rate; 4 KiB x a few million buffers) */
    if (state == ST_QUARANTINE && r->pages > 16) madvise((void *)r->first_page, r->pages * PAGE, MADV_DONTNEED);
    return 0;
}

int snn_guard_retag(void *p, const char *tag)
{
    region_t *r = region_of_buffer(p);
    if (!r) return -1;
    strncpy(r->tag, tag, TAG_BYTES - 1); r->tag[TAG_BYTES - 1] = 0;
    return 0;
}

/* the two functions libsnn_amd.so takes through snn_debug_set_host_allocator: its host tables live in the arena */
static volatile uint64_t g_host_tables;
void *snn_guard_host_alloc(size_t bytes, const char *tag)
{
    __atomic_add_fetch(&g_host_tables, 1, __ATOMIC_RELAXED);
    return snn_guard_alloc(bytes ? bytes : 1, tag);
}
int snn_guard_host_release(void *p, size_t bytes)
{
    (void)bytes;
    if (!p || (uintptr_t)p < g_base || (uintptr_t)p >= g_end) return 0;
    snn_guard_set_state(p, ST_QUARANTINE);
    return 1;
}
uint64_t snn_guard_host_table_count(void) { return __atomic_load_n(&g_host_tables, __ATOMIC_RELAXED); }

uint64_t snn_guard_fault_count(void) { return __atomic_load_n(&g_faults, __ATOMIC_RELAXED); }
uint64_t snn_guard_region_count(void) { return __atomic_load_n(&g_count, __ATOMIC_ACQUIRE); }
uint64_t snn_guard_bytes_reserved(void) { return g_base ? g_next - g_base : 0; }
void snn_guard_note(const char *line) { put(line); put("\n"); }

/* ---- helpers of tests/test_guard_arena.py: "a library" that misbehaves in the three ways the trap is there for ---- */
typedef struct { float *p; uint64_t index; float value; int delay_ms; } late_t;

__attribute__((noinline)) void snn_guard_test_late_store(float *p, uint64_t index, float value) { p[index] = value; __asm__ volatile("" ::: "memory"); }

static void *late_writer_thread(void *arg)
{
    late_t *l = (late_t *)arg;
    struct timespec ts = { l->delay_ms / 1000, (long)(l->delay_ms % 1000) * 1000000L };
    nanosleep(&ts, NULL);
    snn_guard_test_late_store(l->p, l->index, l->value);
    free(l);
    return NULL;
}

/* fills out[0..n) with `value` now and stores once more into out[index] from another thread `delay_ms` later: what a getter looks
 * like whose host-side staging is not finished when the call returns */
int snn_guard_test_getter(float *out, uint64_t n, float value, uint64_t index, int delay_ms, pthread_t *thread)
{
    for (uint64_t i = 0; i < n; ++i) out[i] = value;
    late_t *l = (late_t *)malloc(sizeof *l);
    l->p = out; l->index = index; l->value = value + 1.0f; l->delay_ms = delay_ms;
    pthread_t t;
    if (pthread_create(&t, NULL, late_writer_thread, l) != 0) return -1;
    pthread_setname_np(t, "late-writer");
    if (thread) *thread = t; else pthread_detach(t);
    return 0;
}

void snn_guard_test_join(pthread_t t) { pthread_join(t, NULL); }

/* writes n + extra floats into a buffer of n: a getter that overruns */
int snn_guard_test_overrun(float *out, uint64_t n, uint64_t extra) { for (uint64_t i = 0; i < n + extra; ++i) snn_guard_test_late_store(out, i, 3.0f); return 0; }

/* "the library" scribbling over memory it was never given */
int snn_guard_test_stray(uint32_t *somewhere, uint32_t word) { *(volatile uint32_t *)somewhere = word; return 0; }

/* a setter: sums what it was given (reads only) */
float snn_guard_test_setter(const float *in, uint64_t n) { float s = 0.0f; for (uint64_t i = 0; i < n; ++i) s += in[i]; return s; }
