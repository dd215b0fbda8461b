// host_io.hip — host-side helpers of the on-disk data path (no device code).
//
// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78) of the TFRecord framing the reference's
// per-utterance files use (nabu/processing/tfwriters/tfwriter.py:30-45 -> tf.python_io.TFRecordWriter;
// read back by tf.TFRecordReader in nabu/processing/tfreaders/tfreader.py:71-92).  Slicing-by-8:
// eight table look-ups per 8 input bytes, ~1.5 GB/s on one core, against a few MB/s of a Python
// byte loop — the record check must not be slower than the training step it feeds.
#include "../../include/nabu_hip.h"

#include <cstring>
#include <mutex>

namespace {

uint32_t g_tab[8][256];
std::once_flag g_once;

void build_tables() {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
    g_tab[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i)
    for (int t = 1; t < 8; ++t) g_tab[t][i] = (g_tab[t - 1][i] >> 8) ^ g_tab[0][g_tab[t - 1][i] & 0xFF];
}

}  // namespace

extern "C" uint32_t nabu_crc32c_host(const void *data_host, size_t n, uint32_t crc) {
  std::call_once(g_once, build_tables);
  const unsigned char *p = static_cast<const unsigned char *>(data_host);
  uint32_t c = ~crc;
  while (n && (reinterpret_cast<uintptr_t>(p) & 7)) {
    c = g_tab[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    --n;
  }
  while (n >= 8) {
    uint64_t w;
    std::memcpy(&w, p, 8);
    w ^= c;
    c = g_tab[7][w & 0xFF] ^ g_tab[6][(w >> 8) & 0xFF] ^ g_tab[5][(w >> 16) & 0xFF] ^
        g_tab[4][(w >> 24) & 0xFF] ^ g_tab[3][(w >> 32) & 0xFF] ^ g_tab[2][(w >> 40) & 0xFF] ^
        g_tab[1][(w >> 48) & 0xFF] ^ g_tab[0][(w >> 56) & 0xFF];
    p += 8;
    n -= 8;
  }
  while (n--) c = g_tab[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
  return ~c;
}
