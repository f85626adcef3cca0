// diagnostic: who calls rand()/srand()/srandom()/initstate() in this process (LD_PRELOAD)
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <sys/syscall.h>
static void who(const char *what, void *ra, long v)
{
  Dl_info i; const char *n = "?";
  if (dladdr(ra, &i) && i.dli_fname) n = i.dli_fname;
  fprintf(stderr, "[randshim] %s -> %ld from %s tid %ld\n", what, v, n, (long)syscall(SYS_gettid));
}
int rand(void) { static int (*real)(void); if (!real) real = dlsym(RTLD_NEXT, "rand"); int v = real(); who("rand", __builtin_return_address(0), v % 500); return v; }
void srand(unsigned s) { static void (*real)(unsigned); if (!real) real = dlsym(RTLD_NEXT, "srand"); who("srand", __builtin_return_address(0), s); real(s); }
void srandom(unsigned s) { static void (*real)(unsigned); if (!real) real = dlsym(RTLD_NEXT, "srandom"); who("srandom", __builtin_return_address(0), s); real(s); }
long random(void) { static long (*real)(void); if (!real) real = dlsym(RTLD_NEXT, "random"); long v = real(); who("random", __builtin_return_address(0), v % 500); return v; }
