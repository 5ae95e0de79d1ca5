// Error plumbing of the C ABI.
#include "common.h"

static thread_local std::string g_last_error;

void jb_set_error(const std::string& msg) { g_last_error = msg; }

extern "C" const char* jb_last_error(void) { return g_last_error.c_str(); }

extern "C" int jb_version(void) { return 1; }
