// ref_mc.cpp — the REFERENCE'S OWN marching-cubes case tables, compiled where they lie
// (/root/reference/src/mc_constants.h + its cl_types.h / vendored CL/cl_platform.h; nothing is copied): accessors only.
// Built into oracle/_ref/libref_mc.so by oracle/Makefile when /root/reference is present.  TEST INFRASTRUCTURE:
// tests/test_mc_tables.py pins our generated tables (tools/gen_mc_tables.py) against these.
#include <cstdint>

#include "mc_constants.h"  // -I/root/reference/src -I/root/reference/src/ocl -I/root/reference/include

extern "C" {
uint32_t ref_mc_edge(int ci) { return EdgeTable[ci & 255]; }
uint32_t ref_mc_numverts(int ci) { return NumVertsTable[ci & 255]; }
uint32_t ref_mc_tri(int ci, int j) { return TriTable[ci & 255][j & 15]; }
}
