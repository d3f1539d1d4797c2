// rf_build_id(): digest of the sources this library was compiled from, passed in by
// rag_fin_amd/build.py as -DRF_BUILD_ID="..." (this file is recompiled whenever it changes).
#include "../../include/ragfin.h"
#ifndef RF_BUILD_ID
#define RF_BUILD_ID "unstamped"
#endif
extern "C" const char* rf_build_id(void) { return RF_BUILD_ID; }
