/*
 * error.cpp — last-error text for the C ABI.  The reference prints and calls
 * exit(1) (checkCUDAError, src/pathtrace.cu:27-48); the library returns a code
 * and keeps the message per thread instead.
 */
#include "../../../include/mi_pathtrace.h"
#include <cstdarg>
#include <cstdio>

static thread_local char g_err[512] = "";

void pt_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* pt_last_error(void) { return g_err; }
extern "C" const char* pt_version(void) { return "mi_pathtrace 0.1 (gfx950)"; }
