// gsss_digest.hip -- which sources this library was built from.
//
// geosss_amd/build.py passes -DGSSS_SOURCE_DIGEST="<sha256 over csrc/*.h, *.hip, *.inc and include/gsss.h>" to this translation
// unit alone and recompiles it whenever that digest moves, so the string names the sources of EVERY object linked beside it.
// bench.py quotes the committed rocprofv3 counters (profiles/traffic.json) only when they were measured on the sources of the
// LOADED library -- not of whatever files lie on disk next to a prebuilt .so.
#include "../../include/gsss.h"

#ifndef GSSS_SOURCE_DIGEST
#define GSSS_SOURCE_DIGEST "unknown"
#endif

extern "C" const char *gsss_source_digest(void) { return GSSS_SOURCE_DIGEST; }
