// lib.hip — library-level entry points (version, status text, device probe).
#include "hd_common.h"

extern "C" int hd_abi_version(void) { return 34; }

extern "C" const char* hd_status_str(hd_status s) {
  switch (s) {
    case HD_OK: return "ok";
    case HD_ERR_NULL: return "required pointer is NULL";
    case HD_ERR_SHAPE: return "size/shape argument out of range";
    case HD_ERR_ALIGN: return "pointer not aligned as documented";
    case HD_ERR_WORKSPACE: return "workspace missing or too small";
    case HD_ERR_LAUNCH: return "HIP launch failed";
    case HD_ERR_NO_DEVICE: return "no HIP device visible";
    default: return "unknown status";
  }
}

extern "C" int hd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

// Arms the calling thread: the next kernel launched through hd_launch() (the fused BN kernels, the dense loss
// gradient, RoIAlign, pairwise IoU) stamps `start` / `stop` (hipEvent_t created with timing enabled) with that
// kernel's own begin and end.  NULL, NULL disarms.  Measurement aid of bench.py; no effect on results.
extern "C" hd_status hd_profile_time_next_launch(void* start_event, void* stop_event) {
  if ((start_event == nullptr) != (stop_event == nullptr)) return HD_ERR_NULL;
  HdLaunchTiming& t = hd_launch_timing();
  t.start = reinterpret_cast<hipEvent_t>(start_event);
  t.stop = reinterpret_cast<hipEvent_t>(stop_event);
  return HD_OK;
}
