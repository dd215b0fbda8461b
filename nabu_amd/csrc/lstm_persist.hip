#include "lstm_persist.h"

namespace nabu {
bool lstm_persist_supported(int, int, int) { return false; }
size_t lstm_persist_ws_bytes(int, int, int) { return 0; }
int lstm_persist_fwd(int, int, int, int, int, const int32_t *, const float *const[2], float *const[2],
                     float *const[2], float *, void *, size_t, hipStream_t) {
  return fail(NABU_EUNSUP, "persistent LSTM kernel not built");
}
int lstm_persist_bwd(int, int, int, int, int, const int32_t *, const float *const[2], float *const[2],
                     float *const[2], const float *, void *, size_t, hipStream_t) {
  return fail(NABU_EUNSUP, "persistent LSTM kernel not built");
}
}  // namespace nabu
