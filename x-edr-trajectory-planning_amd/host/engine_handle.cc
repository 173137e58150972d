#include "engine_handle.h"

#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>

namespace tpamd {
namespace {
struct DevicePool {
  std::vector<tpamd_engine *> idle;
  int created = 0;
  bool draining = false;   // release_idle_engines() ran while engines were out
};
std::mutex g_mutex;                    // guards the maps below; never held across an engine call
std::map<int, DevicePool> g_pools;
}  // namespace

int device_count() { return tpamd_device_count(); }

int default_device() {
  if (const char *env = std::getenv("TPAMD_DEVICE")) return std::atoi(env);
  return 0;
}

EngineLease acquire_engine(int device) {
  if (device < 0) device = default_device();
  {
    std::lock_guard<std::mutex> lock(g_mutex);
    DevicePool &p = g_pools[device];
    if (!p.idle.empty()) {
      tpamd_engine *e = p.idle.back();
      p.idle.pop_back();
      return EngineLease(e, device);
    }
  }
  // created outside the lock: engine creation initialises the device (slow the first time)
  tpamd_engine *e = nullptr;
  const int rc = tpamd_engine_create(device, &e);
  if (rc != 0 || !e) {
    std::fprintf(stderr, "[tpamd host] cannot create a GPU engine on device %d: %s\n", device,
                 tpamd_error_string(rc));
    return EngineLease();
  }
  std::lock_guard<std::mutex> lock(g_mutex);
  g_pools[device].created++;
  return EngineLease(e, device);
}

EngineLease &EngineLease::operator=(EngineLease &&o) noexcept {
  if (this != &o) {
    this->~EngineLease();
    engine_ = o.engine_;
    device_ = o.device_;
    o.engine_ = nullptr;
  }
  return *this;
}

EngineLease::~EngineLease() {
  if (!engine_) return;
  std::lock_guard<std::mutex> lock(g_mutex);
  g_pools[device_].idle.push_back(engine_);
  engine_ = nullptr;
}

int engines_created(int device) {
  std::lock_guard<std::mutex> lock(g_mutex);
  return g_pools[device < 0 ? default_device() : device].created;
}

int engines_idle(int device) {
  std::lock_guard<std::mutex> lock(g_mutex);
  return (int)g_pools[device < 0 ? default_device() : device].idle.size();
}

void release_idle_engines() {
  std::vector<tpamd_engine *> victims;
  {
    std::lock_guard<std::mutex> lock(g_mutex);
    for (auto &kv : g_pools) {
      victims.insert(victims.end(), kv.second.idle.begin(), kv.second.idle.end());
      kv.second.created -= (int)kv.second.idle.size();
      kv.second.idle.clear();
    }
  }
  for (tpamd_engine *e : victims) tpamd_engine_destroy(e);
}

}  // namespace tpamd
