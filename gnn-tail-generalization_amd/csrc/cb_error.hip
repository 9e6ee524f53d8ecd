#include <mutex>

#include "cb_common.h"

namespace cb {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Device-side errors.  Kernels that can give up a bounded wait (the LDS tile hand-over of cb_agg_gemm.hip) record the reason in four ints of
// fine-grained, device-mapped HOST memory: the host reads them without a synchronisation (lazily, at the next launch of such a kernel, and
// in cb_device_status()).  Allocated once per process on first use — the library's only state besides the thread-local message above; it
// is written by the device alone and cleared by cb_device_status().
static int* g_dev_err = nullptr;
static std::once_flag g_dev_err_once;
int* device_error_word() {
  std::call_once(g_dev_err_once, [] {
    void* p = nullptr;
    if (hipHostMalloc(&p, 4 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
      (void)hipGetLastError();
      set_error("hipHostMalloc of the device error word failed");
      return;
    }
    int* w = (int*)p;
    w[0] = w[1] = w[2] = w[3] = 0;
    g_dev_err = w;
  });
  return g_dev_err;
}

const char* device_error_text() {
  static thread_local char buf[160];
  const volatile int* w = g_dev_err;
  if (!w || w[0] == 0) return "none";
  if (w[0] == CB_DEVERR_HANDOVER)
    snprintf(buf, sizeof(buf), "LDS tile hand-over timed out in the aggregation + GEMM kernel (block %d, counter target %d)", w[1], w[2]);
  else if (w[0] == CB_DEVERR_GRADROWS)
    snprintf(buf, sizeof(buf), "a gradient row outside the loss rows is not zero (row %lld): the row-sparse backward does not apply to this objective",
             (long long)w[1] + ((long long)w[2] << 31));
  else
    snprintf(buf, sizeof(buf), "device error code %d (%d, %d)", w[0], w[1], w[2]);
  return buf;
}
}  // namespace cb

extern "C" int cb_version(void) { return 2; }
extern "C" const char* cb_last_error(void) { return cb::g_err; }

// CB_OK, or CB_E_DEVICE (message through cb_last_error()) if a kernel launched by this process recorded a device-side error since the
// last call; the word is cleared, so a caller that can recover (re-run the step) may continue.  Does not synchronise: call it after the
// synchronisation that ends a step (the trainer does, where it reads the loss).
extern "C" int cb_device_status(void) {
  int* w = cb::device_error_word();
  if (!w) return CB_E_HIP;
  if (*(volatile int*)w == 0) return CB_OK;
  cb::set_error("device-side error: %s; results since that launch are invalid", cb::device_error_text());
  w[1] = w[2] = w[3] = 0;
  __atomic_store_n(w, 0, __ATOMIC_RELEASE);
  return CB_E_DEVICE;
}
