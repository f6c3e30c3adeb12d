/* util.c -- errors, memory, string table, chromosome table. */
#define _GNU_SOURCE
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include "l2r_host.h"

void (*h_before_exit)(void) = 0;

void h_fatal(const char *where, const char *fmt, ...)
{
    /* reference src/utils.c:91-100 err_fatal: message on stderr, exit status 1 */
    va_list ap;
    fprintf(stderr, "[%s] ", where);
    va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap);
    fprintf(stderr, "\n");
    /* A helper thread may be inside the HIP runtime's start-up (cmds.c early_engine_start: the engine's context is made beside the
     * readers); exit() would run the runtime's teardown against it.  The registered hook waits for that thread first. */
    if (h_before_exit) h_before_exit();
    exit(EXIT_FAILURE);
}

void h_fatal_core(const char *where, const char *fmt, ...)
{
    /* reference src/utils.c:102-111 err_fatal_core: abort() (SIGABRT) */
    va_list ap;
    fprintf(stderr, "[%s] ", where);
    va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap);
    fprintf(stderr, " Abort!\n");
    abort();
}

/* A 10 M-read run touches ~10 GB of fresh memory (inflated BAM, record arrays, results, 4 GB of output text): with 4 KB pages
 * that is millions of page faults in one address space, and they -- not the parsing or formatting -- bound the host stages.
 * Large blocks are therefore offered to the kernel as transparent huge pages (this box: THP mode "madvise"): 3.9 -> 2.7 s end to
 * end for 10 M reads.  L2R_NO_THP=1 turns it off. */
static void advise_huge(void *p, size_t n)
{
    static int off = -1;
    if (off < 0) off = getenv("L2R_NO_THP") != NULL;
    if (off || n < ((size_t)4 << 20)) return;
    const size_t pg = 4096;
    char *a = (char *)(((uintptr_t)p + pg - 1) & ~(uintptr_t)(pg - 1)), *e = (char *)(((uintptr_t)p + n) & ~(uintptr_t)(pg - 1));
    if (e > a) (void)madvise(a, (size_t)(e - a), MADV_HUGEPAGE);
}

void *h_malloc(size_t n)
{
    void *p = malloc(n ? n : 1);
    if (!p) h_fatal_core("h_malloc", "Malloc fail!\nSize: %lld\n", (long long)n);
    advise_huge(p, n);
    return p;
}

void *h_realloc(void *q, size_t n)
{
    void *p = realloc(q, n ? n : 1);
    if (!p) h_fatal_core("h_realloc", "Realloc fail!\nSize: %lld\n", (long long)n);
    advise_huge(p, n);
    return p;
}

/* A FILE that appends to a growing memory block made by h_realloc (open_memstream with the allocator above): *buf / *len are
 * valid after fclose(). */
typedef struct { char **buf; size_t *len; size_t cap; } growbuf;
static ssize_t growbuf_write(void *ck, const char *data, size_t n)
{
    growbuf *g = (growbuf *)ck;
    if (*g->len + n > g->cap) {
        size_t c = g->cap ? g->cap : (size_t)1 << 20;
        while (c < *g->len + n) c *= 2;
        *g->buf = (char *)h_realloc(*g->buf, c); g->cap = c;
    }
    memcpy(*g->buf + *g->len, data, n);
    *g->len += n;
    return (ssize_t)n;
}
static int growbuf_close(void *ck) { free(ck); return 0; }
FILE *h_open_growbuf(char **buf, size_t *len)
{
    growbuf *g = (growbuf *)h_malloc(sizeof *g);
    *buf = NULL; *len = 0;
    g->buf = buf; g->len = len; g->cap = 0;
    cookie_io_functions_t io = { NULL, growbuf_write, NULL, growbuf_close };
    FILE *f = fopencookie(g, "w", io);
    if (f) setvbuf(f, NULL, _IOFBF, 1 << 16);
    return f;
}

uint32_t h_str_add(h_strtab *t, const char *s)
{
    size_t n = strlen(s) + 1;
    if (t->len + n > t->cap) {
        size_t c = t->cap ? t->cap * 2 : 1 << 16;
        while (c < t->len + n) c *= 2;
        if (c >= 0xffffffffu) h_fatal("h_str_add", "string table exceeds 4 GiB");
        t->buf = (char *)h_realloc(t->buf, c); t->cap = c;
    }
    memcpy(t->buf + t->len, s, n);
    uint32_t id = (uint32_t)t->len;
    t->len += n;
    return id;
}

int h_chrom_find(const h_chroms *c, const char *s, int limit)
{
    /* records and GTF lines come grouped by chromosome: the last hit of this thread is tried first (a hint only: it is checked
       against the table it is used on), so an assembly with thousands of contigs does not cost a scan per record */
    static __thread int last = 0;
    if (last < limit && strcmp(c->name[last], s) == 0) return last;
    for (int i = 0; i < limit; ++i) if (strcmp(c->name[i], s) == 0) { last = i; return i; }
    return -1;
}

int h_chrom_intern(h_chroms *c, const char *s)
{
    int i = h_chrom_find(c, s, c->n);
    if (i >= 0) return i;
    if (strlen(s) >= H_NAME_MAX) h_fatal("get_chr_id", "chromosome name \"%s\" has 100 or more characters", s);
    if (c->n == c->cap) { c->cap = c->cap ? c->cap * 2 : 32; c->name = (char **)h_realloc(c->name, (size_t)c->cap * sizeof(char *)); }
    c->name[c->n] = strdup(s);
    return c->n++;
}

void h_chroms_free(h_chroms *c)
{
    for (int i = 0; i < c->n; ++i) free(c->name[i]);
    free(c->name); memset(c, 0, sizeof *c);
}

void h_result_alloc(h_result *r, int64_t n, int64_t cap)
{
    r->n = n; r->n_ex = cap;
    r->ex_off = (int64_t *)h_malloc((size_t)(n + 1) * 8);
    r->ex_start = (int32_t *)h_malloc((size_t)cap * 4);
    r->ex_end = (int32_t *)h_malloc((size_t)cap * 4);
    r->ex_flag = (uint8_t *)h_malloc((size_t)cap);
    r->info = (uint32_t *)h_malloc((size_t)n * 4);
    r->ref_tx = (int32_t *)h_malloc((size_t)n * 4);
}

void h_result_free(h_result *r)
{
    free(r->ex_off); free(r->ex_start); free(r->ex_end); free(r->ex_flag); free(r->info); free(r->ref_tx);
    memset(r, 0, sizeof *r);
}
