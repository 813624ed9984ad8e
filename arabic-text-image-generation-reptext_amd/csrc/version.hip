#include "rt_common.h"
extern "C" const char* rt_version(void) { return "reptext_hip abi8 gfx950"; }
extern "C" int rt_abi_version(void) { return 8; }
