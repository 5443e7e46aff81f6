/* ORACLE (test infrastructure only; see kzo.h).
 * Per-thread cache of large scratch buffers.  The CPU baseline runs one block per thread on up to hundreds of threads and every
 * block wants the same multi-megabyte arrays (suffix array 4 n bytes, link array, stage buffers).  glibc serves such sizes with
 * mmap / munmap per call whatever M_MMAP_THRESHOLD says above its 32 MiB cap: on a 256-CPU host the page faults and the kernel's
 * address-space lock then cost several times the coding itself.  kzo.h maps malloc / calloc / realloc / free of the oracle's
 * sources onto these functions; buffers of 64 KiB and more are parked per thread on free and handed out again.
 * (The reference keeps its arrays per codec instance and the JVM's allocator recycles them; this is the equivalent.) */
#define KZO_NO_ALLOC_MACROS
#include "kzo.h"
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#define KZO_BIG ((size_t)64 << 10)
#define KZO_SLOTS 12
typedef struct { size_t cap; uint64_t magic; } kzo_hdr;            /* 16 bytes: the payload keeps malloc's alignment */
#define KZO_MAGIC 0x6b7a6f616c6c6f63ULL

static __thread kzo_hdr* tl_cache[KZO_SLOTS];
static pthread_key_t tl_key;
static pthread_once_t tl_once = PTHREAD_ONCE_INIT;
static void tl_drop(void* unused) {
  (void)unused;
  for (int i = 0; i < KZO_SLOTS; i++) { free(tl_cache[i]); tl_cache[i] = NULL; }
}
static void tl_make_key(void) { pthread_key_create(&tl_key, tl_drop); }
void kzo_alloc_thread_cleanup(void) { tl_drop(NULL); }

void* kzo_malloc(size_t n) {
  if (n >= KZO_BIG) {
    int best = -1;
    for (int i = 0; i < KZO_SLOTS; i++)
      if (tl_cache[i] && tl_cache[i]->cap >= n && tl_cache[i]->cap <= 2 * n + (1 << 20) && (best < 0 || tl_cache[i]->cap < tl_cache[best]->cap)) best = i;
    if (best >= 0) { kzo_hdr* h = tl_cache[best]; tl_cache[best] = NULL; return h + 1; }
  }
  kzo_hdr* h = (kzo_hdr*)malloc(n + sizeof(kzo_hdr));
  if (!h) return NULL;
  h->cap = n; h->magic = KZO_MAGIC;
  return h + 1;
}
void kzo_free(void* p) {
  if (!p) return;
  kzo_hdr* h = (kzo_hdr*)p - 1;
  if (h->magic != KZO_MAGIC) abort();                              /* not one of ours: a bug, fail loudly */
  if (h->cap >= KZO_BIG) {
    pthread_once(&tl_once, tl_make_key);
    pthread_setspecific(tl_key, (void*)1);                         /* arms the thread-exit destructor */
    int slot = -1, smallest = -1;
    for (int i = 0; i < KZO_SLOTS; i++) {
      if (!tl_cache[i]) { slot = i; break; }
      if (smallest < 0 || tl_cache[i]->cap < tl_cache[smallest]->cap) smallest = i;
    }
    if (slot < 0) {                                                /* full: the smallest parked buffer goes */
      if (tl_cache[smallest]->cap >= h->cap) { h->magic = 0; free(h); return; }
      tl_cache[smallest]->magic = 0; free(tl_cache[smallest]); slot = smallest;
    }
    tl_cache[slot] = h;
    return;
  }
  h->magic = 0;
  free(h);
}
void* kzo_calloc(size_t a, size_t b) {
  void* p = kzo_malloc(a * b);
  if (p) memset(p, 0, a * b);
  return p;
}
void* kzo_realloc(void* p, size_t n) {
  if (!p) return kzo_malloc(n);
  kzo_hdr* h = (kzo_hdr*)p - 1;
  if (h->cap >= n) return p;
  void* q = kzo_malloc(n);
  if (!q) return NULL;
  memcpy(q, p, h->cap);
  kzo_free(p);
  return q;
}
