// engine_handle.h -- process-wide access to the C-ABI engine for the host mirror.
#pragma once

#include "../../include/tpamd.h"

namespace tpamd {

// Returns the lazily created engine of `device` (default 0, or the value of the
// TPAMD_DEVICE environment variable). Aborts the calling operation (returns nullptr and
// prints to stderr) when no GPU is available: the mirror has no CPU fallback.
// The engine is not thread-safe; callers serialise through engine_mutex_lock/unlock.
tpamd_engine *shared_engine();
void engine_lock();
void engine_unlock();

struct EngineGuard {
  EngineGuard() { engine_lock(); }
  ~EngineGuard() { engine_unlock(); }
};

}  // namespace tpamd
