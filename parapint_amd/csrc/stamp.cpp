// Build stamp: the SHA-1 of the kernel sources this library was compiled from (passed by __graft_entry__.build()).
// parapint_amd/_native.py compares it with the sources next to the library and refuses a stale build.
#include "../../include/parapint_hip.h"

#ifndef PP_SOURCE_SHA1
#define PP_SOURCE_SHA1 "unstamped"
#endif

extern "C" const char* pp_source_sha1(void) { return PP_SOURCE_SHA1; }
