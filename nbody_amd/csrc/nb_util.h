/*
 * nb_util.h -- host-side helpers of the C layer.
 *
 * Error convention kept from the reference (src/lib/util.h:17-29): a failed
 * check prints "file:line [func] errno..., message" to stderr and abort()s;
 * nothing returns an error code.
 */
#ifndef NB_UTIL_H
#define NB_UTIL_H

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NB_NEW(COUNT, TYPE) ((TYPE *)malloc((size_t)(COUNT) * sizeof(TYPE)))

#define NB_CHECK(COND, ...)                                                                   \
    do {                                                                                      \
        if (!(COND)) {                                                                        \
            int nb_errno_ = errno;                                                            \
            fprintf(stderr, "%s:%d [%s] errno = %d, str = %s\n", __FILE__, __LINE__, __func__, \
                    nb_errno_, strerror(nb_errno_));                                          \
            fprintf(stderr, "%s:%d [%s] ", __FILE__, __LINE__, __func__);                     \
            fprintf(stderr, __VA_ARGS__);                                                     \
            fputc('\n', stderr);                                                              \
            abort();                                                                          \
        }                                                                                     \
    } while (0)

#endif /* NB_UTIL_H */
