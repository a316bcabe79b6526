// muavta_diag.h — switches of DIAGNOSTIC builds only (tools/ablate_probe.py, tools/phase_profile.py ... build them into tools/_build/).
// Some of them produce WRONG RESULTS by design (a phase of the step compiled out, parts of the observation not written): they exist to time
// what a phase costs on a workload where it has nothing to do.  The shipped library (muavta_amd/native.py: HIPCC_FLAGS) never defines
// MUAVTA_DIAGNOSTIC_BUILD and therefore cannot include this file; naming one of the switches without it is a compile error (muavta_device.h).
#pragma once
#ifndef MUAVTA_DIAGNOSTIC_BUILD
#error "muavta_diag.h belongs to diagnostic builds: compile with -DMUAVTA_DIAGNOSTIC_BUILD (never the shipped libmuavta.so)"
#endif
// -DMUAVTA_ABLATE=<bit mask>: compiles a phase of the step out, so that the launch time without it is its true cost (results are wrong unless
// the ablated phase has nothing to do, as in tools/quiet_probe.py's quiet workload).
#ifndef MUAVTA_ABLATE
#define MUAVTA_ABLATE 0
#endif
#define ABL(bit) ((MUAVTA_ABLATE >> (bit)) & 1)
// -DMUAVTA_OBS_SKIP=<bit mask>: parts of the observation write left out (results are wrong): bit 0 task rows, 1 legal mask, 2 agent rows +
// flags, 3 pad flags, 4 everything
#ifndef MUAVTA_OBS_SKIP
#define MUAVTA_OBS_SKIP 0
#endif
#define OBS_SKIP(mask) ((MUAVTA_OBS_SKIP & (mask)) != 0)
