#include "common.h"

namespace nabu {
char *err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace nabu

extern "C" int nabu_version(void) { return NABU_ABI_VERSION; }
extern "C" const char *nabu_last_error(void) { return nabu::err_buf(); }
