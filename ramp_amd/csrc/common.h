// Every declaration of the RAMP sampler HIP library (gfx950 / MI355X only): what engine.hip / ops.hip / bench.hip include.  A kernel file includes
// core.h and the argument header of ITS family only, so that an edit to one family's arguments recompiles that family and the three files above
// (ramp_amd/build.py follows the #include lines).
#pragma once
#include "core.h"
#include "args_gemm.h"
#include "args_token.h"
#include "args_attention.h"
#include "args_conv.h"
#include "args_rows.h"
#include "args_sampler.h"
#include "args_scene.h"
