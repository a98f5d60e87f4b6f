/* lthip_build_id(): identity of the sources this library was built from (tools/build_id.py: sha256 over csrc/ and
 * include/, first 16 hex digits), baked in by the Makefile through build/gen/build_id.h.  Tests recompute it from the
 * tree that travels with the library: a stale liblongtail_hip.so cannot pass for HEAD. */
#include "longtail_hip.h"
#include "build_id.h"

LTHIP_EXPORT const char* lthip_build_id(void) { return LTHIP_BUILD_ID; }

LTHIP_EXPORT int lthip_abi_version(void) { return LTHIP_ABI_VERSION; }
