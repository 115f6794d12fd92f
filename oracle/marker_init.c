/*
 * marker_init.c -- TEST INFRASTRUCTURE.  One allocator used ONLY to build
 * oracle/_ref/libsparsework_m1.so: the reference's own src/sparsework.cpp, unedited, compiled
 * and linked with -Wl,--wrap=calloc so that its marker array (the only calloc in that file,
 * sparsework.cpp:45 / :190) starts at -1 instead of 0.
 *
 * Why: at HEAD the marker test `workArray[col] >= row_start` with a zero-filled marker treats
 * every product as "already present" (SURVEY F2a), so HEAD's sparse path cannot run as is; the
 * survey's disassembly of the working revision shows the marker initialised to -1.  This build
 * is therefore NOT "the reference as shipped" -- it is the reference's loop (sparsework.cpp:
 * 56-129 / :201-280) executed with that one initial value, and it is used for one thing: to
 * show by execution that the loop's append order (first-touch order) is what oracle/smm_oracle.c
 * restates.  The dense / triple / limits pins use the unmodified build (libsparse_ref.so).
 */
#include <stdlib.h>
#include <string.h>

void *__wrap_calloc(size_t n, size_t size)
{
    void *p = malloc(n * size ? n * size : 1);
    if (p) memset(p, 0xFF, n * size);
    return p;
}
