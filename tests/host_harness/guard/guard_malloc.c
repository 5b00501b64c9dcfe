/* TEST INFRASTRUCTURE ONLY: LD_PRELOAD interposer that puts every posix_memalign allocation of 4 KB or more (that is: every
 * CPU tensor storage of torch) between two inaccessible pages, END-aligned (FCN_GUARD_MODE unset / "end": a read or write past
 * the last element faults at once) or START-aligned ("front": an access before the first element faults).  With the kernels
 * running under the host emulation (tests/host_harness), an out-of-bounds access of ANY kernel on ANY buffer of an emulated GPU
 * test becomes a segmentation fault with a Python traceback instead of a silent read of a neighbour -- or, on the GPU, a
 * "Memory access fault" once the stars align.   gcc -O2 -shared -fPIC guard_malloc.c -o libguard_malloc.so -ldl -lpthread
 *   LD_PRELOAD=.../libguard_malloc.so FCN_EMULATE=1 python -X faulthandler -m pytest tests -m gpu -k ... */
#define _GNU_SOURCE
#include <errno.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>

extern void __libc_free(void *);
extern void *__libc_memalign(size_t, size_t);

#define PAGE 4096ul
#define NSLOT (1u << 16)
static struct { void *user; void *base; size_t len; } tab[NSLOT];
static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
static int front = -1;

static unsigned slot_of(const void *p) { return (unsigned)(((uintptr_t)p >> 6) * 2654435761u) & (NSLOT - 1); }

int posix_memalign(void **out, size_t align, size_t size)
{
    if (size < PAGE || align > PAGE || (PAGE % align) != 0) {
        void *p = __libc_memalign(align, size);
        if (!p) return ENOMEM;
        *out = p;
        return 0;
    }
    if (front < 0) {
        const char *m = getenv("FCN_GUARD_MODE");
        front = (m && strcmp(m, "front") == 0) ? 1 : 0;
    }
    const size_t need = (size + align - 1) / align * align;
    const size_t body = (need + PAGE - 1) / PAGE * PAGE;
    const size_t total = body + 2 * PAGE;
    char *base = mmap(NULL, total, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (base == MAP_FAILED) return ENOMEM;
    mprotect(base, PAGE, PROT_NONE);
    mprotect(base + total - PAGE, PAGE, PROT_NONE);
    char *user = front ? base + PAGE : base + total - PAGE - need;
    pthread_mutex_lock(&mu);
    unsigned s = slot_of(user);
    for (unsigned i = 0; i < NSLOT; ++i, s = (s + 1) & (NSLOT - 1))
        if (!tab[s].user || tab[s].user == (void *)1) { tab[s].user = user; tab[s].base = base; tab[s].len = total; break; }
    pthread_mutex_unlock(&mu);
    *out = user;
    return 0;
}

void free(void *p)
{
    if (!p) return;
    if (((uintptr_t)p & 63) == 0) {              /* ours are at least 64-byte aligned (torch asks for 64) */
        pthread_mutex_lock(&mu);
        unsigned s = slot_of(p);
        for (unsigned i = 0; i < NSLOT && tab[s].user; ++i, s = (s + 1) & (NSLOT - 1))
            if (tab[s].user == p) {
                void *b = tab[s].base;
                size_t l = tab[s].len;
                tab[s].user = (void *)1;         /* tombstone */
                pthread_mutex_unlock(&mu);
                munmap(b, l);
                return;
            }
        pthread_mutex_unlock(&mu);
    }
    __libc_free(p);
}
