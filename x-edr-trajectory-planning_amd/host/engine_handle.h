// engine_handle.h -- the host mirror's access to C-ABI engines: a pool of engines per device.
//
// A tpamd_engine is not thread-safe and is bound to one HIP device (include/tpamd.h). The mirror
// classes are used like the reference's: distinct objects on distinct threads
// (SURVEY.md 8b "Threading"). Every engine call of the mirror therefore LEASES an engine for its
// duration: the pool keeps the engines of each device, creates one when all of a device's engines
// are out (so N threads working at once hold N engines, each with its own workspace and streams)
// and hands an idle one out again later. Nothing is global any more except the pool's own lock,
// which is held only while an engine changes hands -- never while it runs.
#pragma once

#include <vector>

#include "../../include/tpamd.h"

namespace tpamd {

// HIP devices visible to the engine library (0: none -- the mirror has no CPU fallback).
int device_count();
// The device the mirror uses when none is asked for: TPAMD_DEVICE, else 0.
int default_device();

class EngineLease {
 public:
  EngineLease() = default;
  EngineLease(tpamd_engine *engine, int device) : engine_(engine), device_(device) {}
  EngineLease(EngineLease &&o) noexcept : engine_(o.engine_), device_(o.device_) { o.engine_ = nullptr; }
  EngineLease &operator=(EngineLease &&o) noexcept;
  EngineLease(const EngineLease &) = delete;
  EngineLease &operator=(const EngineLease &) = delete;
  ~EngineLease();   // hands the engine back to the pool
  tpamd_engine *get() const { return engine_; }
  explicit operator bool() const { return engine_ != nullptr; }
  int device() const { return device_; }

 private:
  tpamd_engine *engine_ = nullptr;
  int device_ = -1;
};

// Exclusive use of one engine on `device` (-1: default_device()) until the lease goes away.
// An empty lease (and a line on stderr) if the device has no engine to give: no GPU, bad ordinal.
EngineLease acquire_engine(int device = -1);

// Pool statistics for tests: engines created so far on `device`, and how many are idle now.
int engines_created(int device);
int engines_idle(int device);
// Destroys the idle engines of every device (leased ones are destroyed when they come back).
void release_idle_engines();

}  // namespace tpamd
