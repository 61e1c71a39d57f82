// Build identity: a hash of every source the library was built from (happy_simulator_amd/_native.py `_sources_hash`) is compiled
// INTO the library, so "is this .so older than its sources?" is answered by the .so itself -- not by a side file that a checkout
// can change without changing the binary next to it.  `_native.is_stale()` finds the marker in the file's bytes;
// hs_build_sources_hash() returns it to a loaded process.
#include "../../include/hs_engine.h"

#ifndef HS_SOURCES_HASH
#define HS_SOURCES_HASH "unstamped"
#endif

// what `_native.built_from()` looks for in the file's bytes
extern "C" __attribute__((used, visibility("default"))) const char hs_build_marker[] = "HS_SRC_HASH=" HS_SOURCES_HASH "=HS_SRC_HASH_END";

extern "C" const char *hs_build_sources_hash(void) { return HS_SOURCES_HASH; }
