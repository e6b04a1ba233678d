#include "common.h"
#include "lc2is_hip.h"
extern "C" const char* lc2is_version(void) { return "lc2is_hip 1 gfx950"; }
