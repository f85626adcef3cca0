// gpsx_compat_internal.hpp -- shared by the reference-named layers (gpsx_compat.cpp, gpsx_steps.cpp)
#pragma once
#include "../../include/gpsx.h"

// the process-wide default context of the reference-named interface; aborts loudly if no GPU can be opened
gpsx_ctx *gpsx_compat_ctx();
[[noreturn]] void gpsx_compat_die(const char *what, int rc);
// the reference-named capture interface (gpsx_steps.cpp) drops its ring handles; called before the context goes away
void gpsx_compat_capture_forget();
// the parts of gpsx_compat_receiver_reset in the files that own the state
void gpsx_nav_master_reset();
void gpsx_pvt_reset();
