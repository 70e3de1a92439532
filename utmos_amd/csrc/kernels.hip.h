// Device code of libutmos_hip.so -- hand-written for gfx950 (CDNA4, wave64).  No portability layer.
// One translation unit; the kernels live in the headers below (see DESIGN.md section 4 for the map).
#pragma once
#include "common.hip.h"
#include "pick.hip.h"
#include "score_int.hip.h"
#include "score_af.hip.h"
#include "covered.hip.h"
#include "decremental.hip.h"
#include "af_verify.hip.h"
#include "loop_int.hip.h"
#include "af_defer.hip.h"
#include "ingest.hip.h"
