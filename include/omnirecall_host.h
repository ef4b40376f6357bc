/*
 * omnirecall_host.h -- C API of libomnirecall_host.so: the host side of the
 * recall-search path written in C++ because the reference's own host language
 * (C#, net10.0) has no toolchain in this image.  It mirrors, name for name, the
 * pieces of the reference that stay on the host around the one native call:
 *
 *   query tokenisation      RecallSearchService.cs:92-108  (KeywordScore prologue)
 *   ToLowerInvariant        RecallSearchService.cs:96,110
 *   IsNullOrWhiteSpace      RecallSearchService.cs:22,92-93
 *   BuildSnippet            TextSnippetHelper.cs:5-11
 *   Math.Round(score, 4)    RecallSearchService.cs:51
 *
 * A .NET host does not need this library: it calls the BCL for these and binds
 * include/omnirecall_hip.h directly.  Pure host code, no GPU.
 */
#ifndef OMNIRECALL_HOST_H
#define OMNIRECALL_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* string.IsNullOrWhiteSpace on UTF-8 */
int32_t orrh_is_blank(const uint8_t *s, int64_t len);

/* string.ToLowerInvariant on UTF-8; returns bytes written, or -1 if out_cap is
 * too small (4*len+4 always suffices). */
int64_t orrh_lower_invariant(const uint8_t *s, int64_t len, uint8_t *out, int64_t out_cap);

/* The queryTerms array of RecallSearchService.cs:95-108: Split on whitespace,
 * ToLowerInvariant, Distinct, drop the 28 stop words unless that leaves nothing.
 * Terms are written back to back into `terms`, boundaries into term_off[0..T].
 * Returns T >= 0, or -1 when a buffer is too small. */
int32_t orrh_query_terms(const uint8_t *query, int64_t query_len, uint8_t *terms, int64_t terms_cap,
                         uint32_t *term_off, int32_t term_off_cap);

/* TextSnippetHelper.BuildSnippet(content, max_chars); returns bytes written or -1. */
int64_t orrh_build_snippet(const uint8_t *content, int64_t content_len, int32_t max_chars,
                           uint8_t *out, int64_t out_cap);

/* Math.Round(x, 4): banker's rounding of x*1e4 */
double orrh_round4(double x);

#ifdef __cplusplus
}
#endif
#endif
