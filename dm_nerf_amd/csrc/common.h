// common.h -- error plumbing shared by the host-side translation units.
#pragma once
#include <cstdarg>
#include <cstdio>

// Records a printf-style message for dmnerf_last_error() and returns `code`.
int dmn_fail(int code, const char* fmt, ...);
// After a kernel launch: 0, or DMNERF_E_LAUNCH with hipGetErrorString recorded.
int dmn_check_launch(const char* what);
