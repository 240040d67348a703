// Host-side check of the "support legs first" bookkeeping of csrc/balance_coop.hpp (no GPU: only the constexpr host halves of
// the functions run): the slot order of every support mask, and a working set carried to slot order and back.
// Prints one line per support mask: mask, the four legs behind slots 0..3, and a working set (5 and 11 rows per leg) mapped to
// slots and back; tests/test_slot_legs_cpu.py builds it with hipcc and checks the lines against a Python restatement.
#include <cstdint>
#include <cstdio>

#include "balance_coop.hpp"

namespace {
template <int kKinds, class mask_t>
mask_t to_slots_host(mask_t by_leg, unsigned perm) {
  const mask_t rows = ((mask_t)1 << kKinds) - 1;
  mask_t out = 0;
  for (int sl = 0; sl < 4; sl++) out |= ((by_leg >> (kKinds * ((perm >> (2 * sl)) & 3u))) & rows) << (kKinds * sl);
  return out;
}
template <int kKinds, class mask_t>
mask_t to_legs_host(mask_t by_slot, unsigned perm) {
  const mask_t rows = ((mask_t)1 << kKinds) - 1;
  mask_t out = 0;
  for (int sl = 0; sl < 4; sl++) out |= ((by_slot >> (kKinds * sl)) & rows) << (kKinds * ((perm >> (2 * sl)) & 3u));
  return out;
}
} // namespace

int main() {
  using qlamd::coop::slot_legs_of;
  using qlamd::coop::slot_legs_table;
  const unsigned long long lo = slot_legs_table(0), hi = slot_legs_table(1);
  for (unsigned m = 0; m < 16; m++) {
    const unsigned perm = slot_legs_of(m);
    const unsigned from_table = (unsigned)(((m & 8u) ? hi : lo) >> (8 * (m & 7u))) & 0xFFu; // what slot_legs() reads on the device
    const uint32_t ws5 = 0x9A3C5u ^ (m * 0x11111u);
    const uint64_t ws11 = 0x5A5A5A5A5A5ull ^ ((uint64_t)m * 0x123456789ull);
    const uint32_t s5 = to_slots_host<5, uint32_t>(ws5 & 0xFFFFFu, perm);
    const uint64_t s11 = to_slots_host<11, uint64_t>(ws11 & 0xFFFFFFFFFFFull, perm);
    std::printf("%u %u %u %u %u %u %u %llu %u %llu\n", m, perm & 3u, (perm >> 2) & 3u, (perm >> 4) & 3u, (perm >> 6) & 3u,
                from_table == perm ? 1u : 0u, s5, (unsigned long long)s11,
                to_legs_host<5, uint32_t>(s5, perm) == (ws5 & 0xFFFFFu) ? 1u : 0u,
                (unsigned long long)(to_legs_host<11, uint64_t>(s11, perm) == (ws11 & 0xFFFFFFFFFFFull) ? 1 : 0));
  }
  return 0;
}
